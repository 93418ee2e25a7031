"""Scratch: phase stamps (DIR_STAMPS=<kernel>) of a token-path kernel at B = 64.  usage: kernel_stamps.py ste|grid_tokens"""
import json, os, sys
which = sys.argv[1] if len(sys.argv) > 1 else 'ste'
os.environ['DIR_STAMPS'] = which
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
eng.overlap = False
img = torch.randn(64, 3, 256, 256, device='cuda')
for _ in range(3):
    eng.forward(img)
torch.cuda.synchronize()
