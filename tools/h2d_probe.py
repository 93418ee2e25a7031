"""H2D rate of a 50 MB batch from torch-pinned memory vs shared memory registered with cudaHostRegister (what DecodeRing's slots are): python tools/h2d_probe.py"""
import time
import torch
n = 256 * 197120
dev = torch.zeros(n, dtype=torch.uint8, device='cuda')
a = torch.zeros(n, dtype=torch.uint8).pin_memory()
b = torch.zeros(n, dtype=torch.uint8).share_memory_()
rt = torch.cuda.cudart()
print('cudaHostRegister rc', int(rt.cudaHostRegister(b.data_ptr(), n, 0)))
c = torch.zeros(n, dtype=torch.uint8)
for name, t in (('torch pinned', a), ('registered shm', b), ('pageable', c)):
    for _ in range(2):
        dev.copy_(t, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dev.copy_(t, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-15s issue %.2f ms per copy, complete %.2f ms per copy = %.1f GB/s' % (name, (t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3, n * 10 / (t2 - t0) / 1e9))
