"""Scratch microbenchmark: the bf16 conv layers of one DIR forward at B=64 (shape, residual, count as in the engine's plan),
each timed alone with rotating buffers (so L2/MALL cannot hold the activations between repeats).
Prints per-layer time, TFLOP/s, GB/s of algorithmic traffic and the count-weighted total."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F

# name, H(in), Cin, Cout, k, stride, residual, count, pre
L = [
    ('l1.c1a 1x1 64->64', 64, 64, 64, 1, 1, 0, 1, 0),
    ('l1.c1 1x1 256->64', 64, 256, 64, 1, 1, 0, 2, 0),
    ('l1.c2 3x3 64->64', 64, 64, 64, 3, 1, 0, 3, 0),
    ('l1.c3 1x1 64->256 +res', 64, 64, 256, 1, 1, 1, 3, 0),
    ('l1.ds 1x1 64->256', 64, 64, 256, 1, 1, 0, 1, 0),
    ('l2.c1a 1x1 256->128 @64', 64, 256, 128, 1, 1, 0, 1, 0),
    ('l2.c2a 3x3s2 128->128', 64, 128, 128, 3, 2, 0, 1, 0),
    ('l2.ds 1x1s2 256->512', 64, 256, 512, 1, 2, 0, 1, 0),
    ('l2.c1 1x1 512->128', 32, 512, 128, 1, 1, 0, 3, 0),
    ('l2.c2 3x3 128->128', 32, 128, 128, 3, 1, 0, 3, 0),
    ('l2.c3 1x1 128->512 +res', 32, 128, 512, 1, 1, 1, 4, 0),
    ('l3.c1a 1x1 512->256 @32', 32, 512, 256, 1, 1, 0, 1, 0),
    ('l3.c2a 3x3s2 256->256', 32, 256, 256, 3, 2, 0, 1, 0),
    ('l3.ds 1x1s2 512->1024', 32, 512, 1024, 1, 2, 0, 1, 0),
    ('l3.c1 1x1 1024->256', 16, 1024, 256, 1, 1, 0, 5, 0),
    ('l3.c2 3x3 256->256', 16, 256, 256, 3, 1, 0, 5, 0),
    ('l3.c3 1x1 256->1024 +res', 16, 256, 1024, 1, 1, 1, 6, 0),
    ('l4.c1a 1x1 1024->512 @16', 16, 1024, 512, 1, 1, 0, 1, 0),
    ('l4.c2a 3x3s2 512->512', 16, 512, 512, 3, 2, 0, 1, 0),
    ('l4.ds 1x1s2 1024->2048', 16, 1024, 2048, 1, 2, 0, 1, 0),
    ('l4.c1 1x1 2048->512', 8, 2048, 512, 1, 1, 0, 2, 0),
    ('l4.c2 3x3 512->512', 8, 512, 512, 3, 1, 0, 2, 0),
    ('l4.c3 1x1 512->2048 +res', 8, 512, 2048, 1, 1, 1, 3, 0),
    ('attn 3x3 2048->2048', 8, 2048, 2048, 3, 1, 0, 1, 0),
    ('dec 3x3 128->128 @32', 32, 128, 128, 3, 1, 0, 3, 0),
    ('dec 1x1 512->128 @32', 32, 512, 128, 1, 1, 0, 3, 0),
    ('dec 1x1 128->256 @32 +res', 32, 128, 256, 1, 1, 1, 3, 0),
    ('dec 1x1 512->256 @32', 32, 512, 256, 1, 1, 0, 4, 0),
    ('dec 3x3 256->256 @32', 32, 256, 256, 3, 1, 0, 1, 0),
    ('dec 3x3 256->128 @32', 32, 256, 128, 3, 1, 0, 2, 0),
    ('dec 1x1 2304->128 @16', 16, 2304, 128, 1, 1, 0, 1, 0),
    ('dec 3x3 128->128 @16', 16, 128, 128, 3, 1, 0, 3, 0),
    ('fusion 3x3 2560->256 @32 dense', 32, 2560, 256, 3, 1, 0, 0, 0),
]
B = int(os.environ.get('B', 64))
only = os.environ.get('ONLY')
dt = torch.bfloat16
NROT = 3
tot = 0.0
for name, H, Ci, Co, k, s, res, cnt, pre in L:
    if only and only not in name:
        continue
    Ho = H // s
    xs = [torch.randn(B, H, H, Ci, device='cuda').to(dt) for _ in range(NROT)]
    rs = [torch.randn(B, Ho, Ho, Co, device='cuda').to(dt) for _ in range(NROT)] if res else [None] * NROT
    ys = [torch.empty(B, Ho, Ho, Co, device='cuda', dtype=dt) for _ in range(NROT)]
    w = (torch.randn(Co, k, k, Ci, device='cuda') * 0.02).to(dt)
    sc, sh = torch.ones(Co, device='cuda'), torch.zeros(Co, device='cuda')
    p = k // 2
    for i in range(3):
        F.conv2d_nhwc(xs[i % NROT], w, s, p, scale=sc, shift=sh, relu=True, residual=rs[i % NROT], out=ys[i % NROT], variant=int(os.environ.get("VARIANT", 0)))
    torch.cuda.synchronize()
    n = 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        F.conv2d_nhwc(xs[i % NROT], w, s, p, scale=sc, shift=sh, relu=True, residual=rs[i % NROT], out=ys[i % NROT], variant=int(os.environ.get("VARIANT", 0)))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    M = B * Ho * Ho
    fl = 2.0 * M * Co * k * k * Ci
    by = 2.0 * (B * H * H * Ci / (s * s if k == 1 else 1) + Co * k * k * Ci + M * Co * (2 if res else 1))
    tot += ms * cnt
    print('%-32s x%d %7.1f us %7.1f TF %6.2f TB/s  (M=%d N=%d K=%d)' % (name, cnt, ms * 1e3, fl / ms / 1e9, by / ms / 1e9,
                                                                         M, Co, k * k * Ci), flush=True)
print('count-weighted total %.3f ms' % tot)
