"""Scratch: phase durations inside ste_kernel (DIR_STE_STAMPS=1: s_memtime at every barrier of workgroup 0) at B = 64."""
import json, os, sys
os.environ['DIR_STAMPS'] = 'ste'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctypes as C
from dir_amd import _capi, engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in synth.synth_state_dict(shapes, 1234).items()}
keep = []
st = E.StageOp(sd, 'decoder.projecter_3', 32, 2, torch.bfloat16, 0, keep)
L = _capi.lib()
B = 64
xs = torch.randn(B, 42, 128, device='cuda'); ys = torch.empty(B, 42, 64, device='cuda')
for _ in range(4):
    L.dir_ste_forward(C.byref(st.ste), _capi.ptr(xs), None, _capi.ptr(ys), B, _capi.stream_ptr())
torch.cuda.synchronize()
