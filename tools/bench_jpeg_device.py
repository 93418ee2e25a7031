"""Kernel time of dir_jpeg_decode_records at B = 256 (HIP events): python tools/bench_jpeg_device.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd import _capi  # noqa: E402
from dir_amd.apps import jpeg as AJ  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g22_jpeg.npz'))
data = g['full_256.jpg'].tobytes()
stride = AJ.record_bytes(256)
rec = np.zeros(stride, np.uint8)
assert AJ.host_lib().dir_jpeg_decode_coefficients(data, len(data), rec.ctypes.data, rec.size) == 0
B = 256
recs = torch.from_numpy(np.tile(rec, (B, 1))).cuda()
out = torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device='cuda')
dec = AJ.RecordDecoder(B, stride, 256)
for _ in range(3):
    dec(recs, out)
torch.cuda.synchronize()
_capi.PROFILE = []
for _ in range(10):
    dec(recs, out)
torch.cuda.synchronize()
ms = [r['e0'].elapsed_time(r['e1']) for r in _capi.PROFILE]
_capi.PROFILE = None
print('dir_jpeg_decode_records B=%d: %.3f ms per call (min %.3f) = %.0f images/s; %.1f MB in + %.1f MB out -> %.0f GB/s of algorithmic traffic'
      % (B, np.mean(ms), min(ms), B / (min(ms) * 1e-3), B * stride / 1e6, out.numel() / 1e6, (B * stride + out.numel()) / (min(ms) * 1e-3) / 1e9))
