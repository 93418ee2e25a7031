import os, sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dir_amd import engine as E, synth
shapes = {k: tuple(v) for k, v in json.load(open('/root/repo/tests/golden/manifest_dir.json')).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
img = torch.randn(64, 3, 256, 256, device='cuda')
eng.forward(img); torch.cuda.synchronize()
for name in ('regress', 'grid_tokens', 'mano', 'init_head', 'pgcn'):
    os.environ['DIR_STAMPS'] = name
    sys.stderr.write('== %s\n' % name); sys.stderr.flush()
    eng.forward(img); torch.cuda.synchronize()
os.environ.pop('DIR_STAMPS')
