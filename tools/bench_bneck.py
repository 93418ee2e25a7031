"""Scratch: time dir_bottleneck_chain_forward at the layer1 shape (B = 64, 64x64) against the unfused conv sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F
B = int(os.environ.get('B', 64))
BF = torch.bfloat16
d = 'cuda'
y1 = torch.randn(B, 64, 64, 64, device=d).relu().to(BF)
res = torch.randn(B, 64, 64, 256, device=d).to(BF)
w2 = F.pack_conv_weight(torch.randn(64, 64, 3, 3, device=d) * 0.06, BF)
w3o = torch.randn(256, 64, 1, 1, device=d) * 0.17
w1o = torch.randn(64, 256, 1, 1, device=d) * 0.09
w3, w1 = w3o.reshape(256, 64).to(BF).contiguous(), w1o.reshape(64, 256).to(BF).contiguous()
w3p, w1p = F.pack_conv_weight(w3o, BF), F.pack_conv_weight(w1o, BF)
s2, h2, s3, h3, s1, h1 = [torch.rand(c, device=d) + 0.5 for c in (64, 64, 256, 256, 64, 64)]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused():
    y2 = F.conv2d_nhwc(y1, w2, 1, 1, s2, h2, relu=True)
    o = F.conv2d_nhwc(y2, w3p, 1, 0, s3, h3, relu=True, residual=res)
    return F.conv2d_nhwc(o, w1p, 1, 0, s1, h1, relu=True)


print('chain res+next : %.1f us' % timeit(lambda: F.bottleneck_chain(y1, w2, s2, h2, w3, s3, h3, residual=res, nxt=(w1, s1, h1))))
print('chain res      : %.1f us' % timeit(lambda: F.bottleneck_chain(y1, w2, s2, h2, w3, s3, h3, residual=res)))
print('chain next     : %.1f us' % timeit(lambda: F.bottleneck_chain(y1, w2, s2, h2, w3, s3, h3, nxt=(w1, s1, h1))))
print('unfused c2,c3,c1: %.1f us' % timeit(unfused))
