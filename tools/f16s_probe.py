import os, sys, json, numpy as np, torch
sys.path.insert(0, '.')
from dir_amd import synth
from dir_amd.engine import DirEngine
GOLDEN='tests/golden'
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, 'manifest_dir.json'))).items()}
for cond in (True, False):
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=cond).items()}
    img = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), 1234)).cuda()
    g = np.load(os.path.join(GOLDEN, 'g7c_dir.npz' if cond else 'g7_dir.npz'))
    for name, dt in (('bf16', torch.bfloat16), ('f16s', torch.float16)):
        eng = DirEngine(sd, dtype=dt)
        outs = eng.forward(img); torch.cuda.synchronize()
        mp = []
        for i in range(3):
            for side in ('left', 'right'):
                d = outs[i]['pd_joint_xyz_' + side].cpu().numpy() - g['s%d.pd_joint_xyz_%s' % (i, side)]
                mp.append(float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3)
        print('cond' if cond else 'rand', name, 'MPJPE per stage/hand (mm):', np.round(mp, 5), 'finite', all(bool(torch.isfinite(outs[i]['pd_mesh_xyz_left']).all()) for i in range(3)))
