"""dir_adamw_step at the model's size (92.7 M fp32 parameters): 4 reads + 3 writes x 4 B per element."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import _capi
L = _capi.lib()
for n in (92_730_000, 23_000_000):
    bufs = [torch.randn(n, device='cuda') * 0.01 for _ in range(2)] + [torch.zeros(n, device='cuda') for _ in range(2)]
    def step(k):
        _capi.check(L.dir_adamw_step(*[_capi.ptr(b) for b in bufs], n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, k, _capi.stream_ptr()), 'adamw')
    for k in range(1, 4): step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(4, 24): step(k)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('n = %d: %.1f us per step, %.0f GB/s (%.1f %% of 8 TB/s)' % (n, us, n * 28 / us / 1e3, n * 28 / us / 1e3 / 80))
