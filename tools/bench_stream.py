"""dir_conv1x1_stream_forward against the best tiled variant of dir_conv2d_forward / dir_conv2d_dual_forward on the path's HBM-bound
1x1 layers (B = 64, bf16): HIP-event time per launch, algorithmic bytes / time.  python tools/bench_stream.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dir_amd import engine as E  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(os.environ.get('B', '64'))
gen = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=gen)  # noqa: E731
# (name, HW, Cin, Cout, pre, Cin2, stride2)
LAYERS = [('Residual.conv1 @32 512->128 pre', 32, 512, 128, True, 0, 1), ('Residual.conv3+skip @32 128+512->256', 32, 128, 256, False, 512, 1),
          ('fusion4.conv1 @16 2304->128 pre', 16, 2304, 128, True, 0, 1), ('fusion4.conv3+skip @16 128+2304->256', 16, 128, 256, False, 2304, 1),
          ('skip4.conv1 @16 1024->128 pre', 16, 1024, 128, True, 0, 1), ('layer2.0 conv3+ds @32 128+256(s2)->512', 32, 128, 512, False, 256, 2),
          ('layer3.0 conv3+ds @16 256+512(s2)->1024', 16, 256, 1024, False, 512, 2), ('layer4.0.conv1 @16 1024->512', 16, 1024, 512, False, 0, 1),
          ('layer4.1.conv1 @8 2048->512', 8, 2048, 512, False, 0, 1), ('final3 @32 256->256', 32, 256, 256, False, 0, 1)]
for name, HW, Cin, Cout, pre, Cin2, s2 in LAYERS:
    x = rnd(B, HW, HW, Cin).to(BF)
    M = B * HW * HW
    if Cin2:
        x2 = rnd(B, HW * s2, HW * s2, Cin2).to(BF)
        op = E.DualConvOp(rnd(Cout, Cin, 1, 1) * 0.05, torch.ones(Cout, device='cuda'), rnd(Cout), rnd(Cout, Cin2, 1, 1) * 0.05,
                          torch.ones(Cout, device='cuda'), rnd(Cout), s2, BF, relu=True)
        call = lambda: op(x, x2)  # noqa: E731
        nbytes = (M * (Cin + Cout) + x2.numel() // (s2 * s2) + op.w.numel()) * 2
    else:
        op = E.ConvOp(rnd(Cout, Cin, 1, 1) * 0.05, BF, scale=torch.ones(Cout, device='cuda'), shift=rnd(Cout), relu=True,
                      pre=(torch.ones(Cin, device='cuda'), rnd(Cin)) if pre else None, pre_relu=pre)
        call = lambda: op(x)  # noqa: E731
        nbytes = (M * (Cin + Cout) + op.w.numel()) * 2
    best = None
    ts64 = None
    for v in tuple(v for v in E.DirEngine.CONV_VARIANTS if v not in E.STREAM_VARIANTS) + E.STREAM_VARIANTS:
        E._TLS.variant = v
        t = timeit(call)
        if v == E.STREAM_VARIANT:
            ts = t
        elif v == E.STREAM64_VARIANT:
            ts64 = t
        elif v == E.STREAM32_VARIANT:
            ts32 = t
        elif v == E.STREAMP_VARIANT:
            tsp = t
        elif best is None or t < best[0]:
            best = (t, v)
    E._TLS.variant = None
    print('%-44s M=%6d  stream %6.1f us (%.2f TB/s)   pipelined (24) %6.1f us (%.2f TB/s)   64 px %6.1f us (%.2f TB/s)   32 px %6.1f us   best tiled %6.1f us (variant %2d, %.2f TB/s)   %.1f MB'
          % (name, M, ts, nbytes / ts / 1e6, tsp, nbytes / tsp / 1e6, ts64, nbytes / ts64 / 1e6, ts32, best[0], best[1], nbytes / best[0] / 1e6, nbytes / 1e6))
