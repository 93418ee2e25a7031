"""HBM copy probe shapes (dir_probe_launch mode 2): loads in flight per lane x workgroups per CU x (non-temporal | plain), TB/s of bytes read + written."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import _capi
L = _capi.lib()
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); big.fill_(1)
sp = torch.cuda.current_stream().cuda_stream
def rate(mode, nbytes, iters, seconds=0.6):
    per = L.dir_probe_launch(mode, _capi.ptr(big), nbytes, iters, sp); assert per > 0, per
    torch.cuda.synchronize(); t0 = time.perf_counter(); work = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4): work += L.dir_probe_launch(mode, _capi.ptr(big), nbytes, iters, sp)
        torch.cuda.synchronize()
    return work / (time.perf_counter() - t0) / 1e12
print('read-only loop: %.3f TB/s' % rate(1, 1 << 30, 0))
for nt in (0, 1):
    for U in (4, 8, 16):
        for wpc in (2, 4, 8, 12, 16):
            print('copy %s U=%2d wg/CU=%2d : %.3f TB/s' % ('plain' if nt else 'nt   ', U, wpc, rate(2, 1 << 30, U | (wpc << 4) | (nt << 8))))
