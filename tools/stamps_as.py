"""DIR_STAMPS=conv_as python tools/stamps_as.py : phase stamps of workgroup 0 of the activation-stationary kernel on a few layers"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import engine as E
dt = torch.float16
g = torch.Generator(device='cuda').manual_seed(1)
for (B, H, W, Ci, Co, k) in [(64, 16, 16, 256, 256, 3), (64, 8, 8, 512, 512, 3), (64, 16, 16, 1024, 256, 1), (64, 8, 8, 2048, 2048, 3), (64, 8, 8, 2048, 512, 1)]:
    w = torch.randn(Co, Ci, k, k, device='cuda', generator=g) * (2.0 / (k * k * Ci)) ** 0.5
    op = E.ConvOp(w, dt, stride=1, pad=k // 2, scale=torch.ones(Co, device='cuda'), shift=torch.zeros(Co, device='cuda'), relu=True)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g).to(dt)
    out = torch.empty(B, H, W, Co, device='cuda', dtype=dt)
    for v in (25, 26, 27, 28):
        E._TLS.variant = v
        sys.stderr.write('--- M=%d N=%d K=%d variant %d (A, PB) = %s\n' % (B * H * W, Co, k * k * Ci, v, E.AS_VARIANTS[v])); sys.stderr.flush()
        for _ in range(3):
            op(x, out=out)
        torch.cuda.synchronize()
