"""Scratch experiment: does padding the NHWC channel stride (pixel pitch no longer a multiple of 4 KiB) change conv time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F
B = 64
for name, H, Ci, Co, k in (('attn 3x3 2048->2048 @8', 8, 2048, 2048, 3), ('l4.c2 3x3 512->512 @8', 8, 512, 512, 3), ('l3.c2 3x3 256->256 @16', 16, 256, 256, 3),
                           ('dec 3x3 256->256 @32', 32, 256, 256, 3), ('l3.c1 1x1 1024->256 @16', 16, 1024, 256, 1)):
    w = (torch.randn(Co, k, k, Ci, device='cuda') * 0.02).to(torch.bfloat16)
    for pad in (0, 64, 192):
        xs = [torch.randn(B, H, H, Ci + pad, device='cuda').to(torch.bfloat16) for _ in range(3)]
        ys = [torch.empty(B, H, H, Co, device='cuda', dtype=torch.bfloat16) for _ in range(3)]
        for i in range(3):
            F.conv2d_nhwc(xs[i], w, 1, k // 2, relu=True, out=ys[i], cin=Ci)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            F.conv2d_nhwc(xs[i % 3], w, 1, k // 2, relu=True, out=ys[i % 3], cin=Ci)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 12 * 1e3
        print('%-28s channel pad %3d: %7.1f us  %6.1f TF' % (name, pad, us, 2.0 * B * H * H * Co * k * k * Ci / us / 1e6))
