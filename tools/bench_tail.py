"""dir_bottleneck_tail_forward against the unfused pair it replaces (conv3 + residual + ReLU, next conv1), B = 64, bf16:
HIP-event time per launch over 50 graph-less repetitions, algorithmic bytes / time.  python tools/bench_tail.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dir_amd import functional as F  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=50):
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


for P, N2, HW in ((128, 128, 32), (128, 256, 32), (256, 256, 16)):
    B, C4 = int(os.environ.get('B', '64')), 4 * P
    g = torch.Generator(device='cuda').manual_seed(1)
    y2 = torch.randn(B, HW, HW, P, device='cuda', generator=g).relu().to(BF)
    res = torch.randn(B, HW, HW, C4, device='cuda', generator=g).relu().to(BF)
    w3 = (torch.randn(C4, P, device='cuda', generator=g) * (2.0 / P) ** 0.5)
    w1 = (torch.randn(N2, C4, device='cuda', generator=g) * (2.0 / C4) ** 0.5)
    s3, h3, s1, h1 = (torch.rand(C4, device='cuda') + 0.5, torch.randn(C4, device='cuda') * 0.3, torch.rand(N2, device='cuda') + 0.5,
                      torch.randn(N2, device='cuda') * 0.3)
    w3p, w1p = F.pack_conv_weight(w3.reshape(C4, P, 1, 1), BF), F.pack_conv_weight(w1.reshape(N2, C4, 1, 1), BF)
    import ctypes as C
    from dir_amd import _capi
    from dir_amd.engine import pack_tail_stream
    streams = {n: pack_tail_stream(w3, w1, n) for n in (8, 4)}
    out = torch.empty(B, HW, HW, C4, device='cuda', dtype=BF)
    y1n = torch.empty(B, HW, HW, N2, device='cuda', dtype=BF)
    ps = {n: _capi.BneckTailParams(_capi.ptr(streams[n]), _capi.ptr(s3), _capi.ptr(h3), _capi.ptr(s1), _capi.ptr(h1), P, N2, n) for n in (8, 4)}

    def fused(n=8):
        _capi.check(_capi.lib().dir_bottleneck_tail_forward(C.byref(ps[n]), _capi.ptr(y2), _capi.ptr(res), _capi.ptr(out), _capi.ptr(y1n),
                                                            B * HW * HW, _capi.stream_ptr()), 'tail')

    def unfused():
        o = F.conv2d_nhwc(y2, w3p, 1, 0, s3, h3, relu=True, residual=res)
        F.conv2d_nhwc(o, w1p, 1, 0, s1, h1, relu=True)
    M = B * HW * HW
    tf, tu, t4 = timeit(fused), timeit(unfused), timeit(lambda: fused(4))
    bf = (M * (P + 2 * C4 + N2) + C4 * P + N2 * C4) * 2
    bu = bf + M * C4 * 2
    print('P=%d N2=%d M=%d: fused 8-wave %.1f us (%.2f TB/s of %.1f MB)   4-wave thin %.1f us (%.2f TB/s)   unfused pair %.1f us (%.2f TB/s of %.1f MB)'
          % (P, N2, M, tf, bf / tf / 1e6, bf / 1e6, t4, bf / t4 / 1e6, tu, bu / tu / 1e6, bu / 1e6))
