export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${TAG:-r06_i}
mkdir -p $out
cd $R
python - > $out/probes2.txt 2>&1 <<'PY'
import time, torch
from dir_amd import _capi
L = _capi.lib()
big = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); big.fill_(1)
sp = torch.cuda.current_stream().cuda_stream
def rate(mode, nbytes, iters, seconds=0.5):
    per = L.dir_probe_launch(mode, _capi.ptr(big), nbytes, iters, sp); assert per > 0, per
    torch.cuda.synchronize(); t0 = time.perf_counter(); work = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4): work += L.dir_probe_launch(mode, _capi.ptr(big), nbytes, iters, sp)
        torch.cuda.synchronize()
    return work / (time.perf_counter() - t0) / 1e12
for mode in (4, 6):
    for kb in (512, 1152, 4096, 16384):
        print('mode %d (%d workgroup(s) per CU): every workgroup reads the same %6d KB: %.2f TB/s aggregate' % (mode, 2 if mode == 4 else 1, kb, rate(mode, kb * 1024, max(1, 65536 // kb))))
PY
