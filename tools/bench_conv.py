"""Scratch microbenchmark: MFMA conv kernel on the B=64 layer shapes of DIR."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F

SHAPES = [  # name, H, Cin, Cout, k, s
    ('fusion32 3x3 2560->256', 32, 2560, 256, 3, 1),
    ('fusion16 3x3 2560->256', 16, 2560, 256, 3, 1),
    ('attn 3x3 2048->1024 @8', 8, 2048, 1024, 3, 1),
    ('l1 3x3 64->64 @64', 64, 64, 64, 3, 1),
    ('l1 1x1 64->256 @64', 64, 64, 256, 1, 1),
    ('l1 1x1 256->64 @64', 64, 256, 64, 1, 1),
    ('l2 3x3 128->128 @32', 32, 128, 128, 3, 1),
    ('l2 1x1 512->128 @32', 32, 512, 128, 1, 1),
    ('l3 3x3 256->256 @16', 16, 256, 256, 3, 1),
    ('l3 1x1 256->1024 @16', 16, 256, 1024, 1, 1),
    ('l3 1x1 1024->256 @16', 16, 1024, 256, 1, 1),
    ('l4 3x3 512->512 @8', 8, 512, 512, 3, 1),
    ('l4 1x1 512->2048 @8', 8, 512, 2048, 1, 1),
    ('l4 1x1 2048->512 @8', 8, 2048, 512, 1, 1),
    ('dec 1x1 2304->128 @16', 16, 2304, 128, 1, 1),
    ('final 3x3 256->256 @32', 32, 256, 256, 3, 1),
]
B = int(os.environ.get('B', 64))
for dt in (torch.bfloat16, torch.float32):
    tot_t = tot_f = 0
    for name, H, Ci, Co, k, s in SHAPES:
        x = torch.randn(B, H, H, Ci, device='cuda').to(dt)
        w = (torch.randn(Co, k, k, Ci, device='cuda') * 0.02).to(dt)
        p = k // 2
        for _ in range(3):
            y = F.conv2d_nhwc(x, w, s, p, relu=True)
        torch.cuda.synchronize()
        n = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            y = F.conv2d_nhwc(x, w, s, p, relu=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * (H // s) ** 2 * Co * k * k * Ci
        tot_t += ms; tot_f += fl
        print('%-6s %-26s %8.3f ms  %8.1f TFLOP/s  (M=%d N=%d K=%d tiles=%d)' % (
            str(dt).split('.')[-1], name, ms, fl / ms / 1e9, B * (H // s) ** 2, Co, k * k * Ci,
            ((B * (H // s) ** 2 + 127) // 128) * ((Co + 127) // 128)))
    print('   total %.2f ms, %.1f TFLOP/s avg' % (tot_t, tot_f / tot_t / 1e9))
