"""Split-K factor sweep on the layers whose 128x128-tile grid leaves CUs idle at B = 64 (dir_conv2d_splitk_forward vs the best tiled
variant)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd import engine as E  # noqa: E402

LAYERS = [  # H, Cin, Cout, k, stride, pre
    (8, 512, 512, 3, 1, False), (16, 512, 512, 3, 2, False), (8, 2048, 512, 1, 1, False), (16, 1024, 512, 1, 1, False),
    (16, 128, 128, 3, 1, False), (16, 1024, 128, 1, 1, True), (16, 2304, 128, 1, 1, True), (16, 512, 128, 1, 1, True),
    (16, 256, 256, 3, 1, False), (16, 1024, 256, 1, 1, False), (8, 512, 2048, 1, 1, False), (8, 2048, 2048, 3, 1, False),
]
B = 64


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for H, Cin, Cout, k, stride, pre in LAYERS:
    w = torch.randn(Cout, Cin, k, k) * (Cin * k * k) ** -0.5
    prep = (torch.rand(Cin).cuda() + 0.5, torch.randn(Cin).cuda()) if pre else None
    op = E.ConvOp(w.cuda(), torch.bfloat16, stride=stride, pad=k // 2, scale=torch.ones(Cout).cuda(), shift=torch.zeros(Cout).cuda(), relu=True,
                  pre=prep, pre_relu=pre)
    x = torch.randn(B, H, H, Cin, device='cuda').to(torch.bfloat16)
    res = {}
    op.split[B] = 1
    best = None
    for v in E.DirEngine.CONV_VARIANTS:
        E._TLS.variant = v
        t = timeit(lambda: op(x))
        if best is None or t < best[0]:
            best = (t, v)
    E._TLS.variant = None
    nk = k * k * Cin // 64
    line = ''
    for S in (2, 3, 4, 6, 8, 12, 16):
        if S > nk:
            continue
        op.split[B] = S
        line += '  S=%d %5.1f' % (S, timeit(lambda: op(x)))
    op.split[B] = None
    del op.split[B]
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    auto = op.splits(B, Ho, Ho)
    gf = 2.0 * B * Ho * Ho * Cout * k * k * Cin / 1e9
    print('M=%5d N=%4d K=%5d %s: best tiled %5.1f us (variant %2d, %4.0f TF) |%s | heuristic S=%d' % (B * Ho * Ho, Cout, k * k * Cin, 'pre' if pre else '   ', best[0], best[1],
                                                                                                    gf / best[0] / 1e3, line, auto))
