"""For each conv kernel variant: force it on every layer it applies to (B = 64, golden images in rows 5 and 63) and report how far the
refined stage outputs move from the default-kernel run -- variants are meant to be interchangeable to bf16 rounding."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd import engine as E  # noqa: E402
from dir_amd import synth  # noqa: E402

with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
    shapes = {k: tuple(v) for k, v in json.load(f).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
img = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), 1234)).cuda()
gen = torch.Generator(device='cuda').manual_seed(0)
big = torch.randn(B, 3, 256, 256, device='cuda', generator=gen)
big[5], big[B - 1] = img[0], img[1]
eng = E.DirEngine(sd, dtype=torch.bfloat16)
keys = ('pd_mesh_xyz_left', 'pd_joint_uv_right', 'pd_offset')


def run():
    o = eng.forward(big)
    torch.cuda.synchronize()
    rows = [5, B - 1]
    return [o[s][k][rows].clone() for s in range(3) for k in keys] + [o[3]['seg'][rows].clone()]


ref = run()
names = ['s%d.%s' % (s, k) for s in range(3) for k in keys] + ['seg']
for v in (0, 21):
    E._TLS.variant = v
    try:
        got = run()
    finally:
        E._TLS.variant = None
    d = [float((a.float() - b.float()).abs().max()) for a, b in zip(got, ref)]
    w = int(np.argmax(d[:-1]))
    print('variant %2d: worst stage diff %.3e (%s)   seg %.3e (max |seg| %.3e)  ' % (v, d[w], names[w], d[-1], float(ref[-1].abs().max())) + ' '.join('%.1e' % x for x in d[:-1]))
