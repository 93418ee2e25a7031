"""Per-layer A/B of the activation-stationary kernel (conv_as.hip) against every other variant, on the small-map layers of the B = 64 forward.
usage: python tools/bench_as.py [f16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import engine as E

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'f16') else torch.bfloat16
LAYERS = [  # B, H, W, Cin, Cout, k, residual
    (64, 16, 16, 256, 256, 3, False), (64, 8, 8, 512, 512, 3, False), (64, 16, 16, 1024, 256, 1, False), (64, 16, 16, 256, 1024, 1, True),
    (64, 8, 8, 512, 2048, 1, True), (64, 8, 8, 2048, 512, 1, False), (64, 16, 16, 128, 128, 3, False), (64, 32, 32, 128, 128, 3, False),
    (64, 32, 32, 256, 256, 3, False), (64, 16, 16, 512, 128, 1, False), (64, 16, 16, 1024, 512, 1, False), (64, 8, 8, 2048, 2048, 3, False),
]
g = torch.Generator(device='cuda').manual_seed(1)
for (B, H, W, Ci, Co, k, with_res) in LAYERS:
    w = torch.randn(Co, Ci, k, k, device='cuda', generator=g) * (2.0 / (k * k * Ci)) ** 0.5
    op = E.ConvOp(w, dt, stride=1, pad=k // 2, scale=torch.ones(Co, device='cuda'), shift=torch.zeros(Co, device='cuda'), relu=True)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g).to(dt)
    res = torch.randn(B, H, W, Co, device='cuda', generator=g).to(dt) if with_res else None
    out = torch.empty(B, H, W, Co, device='cuda', dtype=dt)
    flops = 2.0 * B * H * W * Co * k * k * Ci
    row, base = [], None
    for v in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14, 15, 21, 22, 23, 25, 26, 27, 28):
        E._TLS.variant = v
        for _ in range(3):
            op(x, out=out, residual=res)
        torch.cuda.synchronize()
        if base is None:
            base = out.clone()
        same = torch.equal(out, base)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            op(x, out=out, residual=res)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append((us, v, same))
    E._TLS.variant = None
    best_old = min(r for r in row if r[1] < 25)
    print('M=%6d N=%4d K=%5d k%d%s | best tiled: v%-2d %6.1f us %6.0f TF | AS: %s' % (
        B * H * W, Co, k * k * Ci, k, ' +res' if with_res else '     ', best_old[1], best_old[0], flops / best_old[0] / 1e6,
        '  '.join('v%d %6.1f us %5.0f TF%s' % (v, us, flops / us / 1e6, '' if same else ' DIFF') for us, v, same in row if v >= 25)), flush=True)
