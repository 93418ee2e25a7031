"""Scratch: time dir_stem_pool_forward (fp32 / uint8 input) against the staged stem path at B = 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F, _capi, engine as E
B = int(os.environ.get('B', 64))
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.1
sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda') * 0.3
pw = F.pack_stem_weight(w)
xf = torch.randn(B, 3, 256, 256, device='cuda')
x8 = torch.randint(0, 256, (B, 256, 256, 3), device='cuda', dtype=torch.uint8)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('fused f32 in : %.1f us' % timeit(lambda: F.stem_pool(xf, pw, sc, sh)))
print('fused u8 in  : %.1f us' % timeit(lambda: F.stem_pool(x8, pw, sc, sh)))
L = _capi.lib()
stem = E.stem_conv_op(w, sc, sh, torch.bfloat16)
xp = torch.empty(B, 131, 132, 16, device='cuda', dtype=torch.bfloat16)
z = torch.empty(B, 64, 64, 64, device='cuda', dtype=torch.bfloat16)


def staged():
    L.dir_stem_prep_s2d(_capi.ptr(xf), _capi.ptr(xp), B, 256, 256, 131, 132, 1, _capi.stream_ptr())
    s1 = stem(xp)
    L.dir_maxpool3x3s2(_capi.ptr(s1), _capi.ptr(z), B, 128, 128, 64, 1, _capi.stream_ptr())


print('staged f32 in: %.1f us' % timeit(staged))
