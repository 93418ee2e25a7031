"""Scratch: run one conv shape repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F
B, H, Ci, Co, k = [int(v) for v in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
x = torch.randn(B, H, H, Ci, device='cuda').bfloat16()
w = (torch.randn(Co, k, k, Ci, device='cuda') * 0.02).bfloat16()
for _ in range(n):
    y = F.conv2d_nhwc(x, w, 1, k // 2, relu=True)
torch.cuda.synchronize()
