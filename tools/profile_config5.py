"""Kernel trace target: BASELINE config 5's per-GPU forward (HRNet-W48 + init + 4 refinement stages, 32 images, bf16), eager, after autotune.
( cd /tmp && rocprofv3 --kernel-trace -d <out> -o r -- python tools/profile_config5.py ); python tools/prof_summary.py <db> 40"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import synth
from dir_amd.engine import DirEngine
from dir_amd.models.dir import DIR

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = DIR(21, 'x', 0, backbone='hrnet_w48', extra_stages=2)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
del net
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=True).items()}
g = torch.Generator(device='cuda').manual_seed(5)
img = torch.randn(B, 3, 256, 256, device='cuda', generator=g)
eng = DirEngine(sd, dtype={'bf16': torch.bfloat16, 'f16': torch.float16}[os.environ.get('DTYPE', 'f16')])
eng.calibrate(img)
eng.forward(img); torch.cuda.synchronize()
if os.environ.get('AUTOTUNE', '1') == '1':          # AUTOTUNE=0: the trace then holds the forwards only (heuristic kernel choice), not every candidate variant
    eng.autotune(img, reps=1)
torch.cuda.synchronize()
print('PROFILE_BEGIN', flush=True)
for _ in range(10):
    eng.forward(img)
torch.cuda.synchronize()
