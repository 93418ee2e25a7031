"""VERDICT r3 item 4d probe: where does the bf16 mode's init-stage error (0.036 / 0.050 mm on the trained-like golden G7c) come from?  The bf16 backbone up to
c3, then layer4 + attention conv + init head in exact fp32 on that c3 -- against the all-bf16 and all-fp32 engines.  python tools/init_stage_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=True).items()}
img = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), 1234)).cuda()
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g7c_dir.npz'))


def mpjpe(o):
    out = []
    for side in ('left', 'right'):
        d = o['pd_joint_xyz_' + side].cpu().numpy() - g['s0.pd_joint_xyz_' + side]
        out.append(float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3)
    return out


e16 = E.DirEngine(sd, dtype=torch.bfloat16)
e16.bb.decimate_c1 = False
e32 = E.DirEngine(sd, dtype=torch.float32)
ef = E.DirEngine(sd, dtype=torch.float32, arith='f16')
ef.calibrate(img)
print('init stage mean per-joint error vs the reference golden (mm), left / right')
print('  all bf16                         ', mpjpe(e16.forward(img)[0]))
print('  all fp32                         ', mpjpe(e32.forward(img)[0]))
print('  all f16 arithmetic (fp32 maps)   ', mpjpe(ef.forward(img)[0]))
for upto in (3, 2, 1):
    feats = e16.bb(img)                                   # bf16 c1..c4
    x = feats[upto - 1].float().contiguous()             # the bf16 map handed to fp32 layers upto+1 .. 4
    c = e32.bb._layers(x, start=upto)                    # fp32 layers
    init = e32.init_regressor(c[-1])
    print('  bf16 up to c%d, fp32 from layer%d on ' % (upto, upto + 1), mpjpe(init))
    init = ef.init_regressor(ef.bb._layers(x, start=upto)[-1])
    print('  bf16 up to c%d, f16 arithmetic after ' % upto, mpjpe(init))
