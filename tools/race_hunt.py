"""Scratch: which tensor of the init stage diverges first when the skip branch runs on the side stream (DIR_OVERLAP semantics)?
Part of the packed-FP32 hunt; with the library built as it is now this prints an empty histogram."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = 64
img = torch.randn(B, 3, 256, 256, device='cuda', generator=torch.Generator(device='cuda').manual_seed(64))
eng = E.DirEngine(sd, dtype=torch.bfloat16)
stash = {}
orig_init = eng.init_regressor
orig_attn = eng.attn


def attn(c4):
    hh = orig_attn(c4)
    stash['c4'] = c4.clone(); stash['hh'] = hh.clone()
    return hh


def init(c4):
    r = orig_init(c4)
    again = E.run_mano_pair(eng.init_mano, r['pd_mano_para_left'], r['pd_mano_para_right'], B)
    stash['again_same_as_first'] = (again[0][0] == r['pd_mesh_xyz_left']).all().reshape(1).float()
    torch.cuda.synchronize()
    third = E.run_mano_pair(eng.init_mano, r['pd_mano_para_left'], r['pd_mano_para_right'], B)
    stash['third_after_sync'] = third[0][0].clone()
    stash['second'] = again[0][0].clone()
    for k in ('pd_mano_para_left', 'pd_mano_para_right', 'pd_offset', 'pd_mesh_xyz_left', 'pd_joint_uv_right'):
        stash[k] = r[k].clone()
    return r


eng.attn = attn
eng.init_regressor = init
eng.overlap = False
eng.forward(img); torch.cuda.synchronize()
base = dict(stash)
eng.overlap = True
counts = {}
for rep in range(60):
    taps = {}
    eng.forward(img, taps=taps); torch.cuda.synchronize()
    bad = [k for k in base if not torch.equal(stash[k], base[k])]
    if rep < 8 and not torch.equal(stash['pd_mesh_xyz_left'], base['pd_mesh_xyz_left']):
        d = (stash['pd_mesh_xyz_left'] != base['pd_mesh_xyz_left']).any(dim=2)          # [B, 778]
        bs = d.any(dim=1).nonzero().flatten().tolist()
        for b_ in bs[:4]:
            vs = d[b_].nonzero().flatten()
            print('   sample %d: %d vertices differ, range %d..%d; max |diff| %.2e' % (b_, len(vs), int(vs[0]), int(vs[-1]),
                  float((stash['pd_mesh_xyz_left'][b_] - base['pd_mesh_xyz_left'][b_]).abs().max())))
        print('   samples', bs)
    if rep < 0: print('first==base', torch.equal(stash['pd_mesh_xyz_left'], base['pd_mesh_xyz_left']), 'second==base', torch.equal(stash['second'], base['pd_mesh_xyz_left']), 'third==base', torch.equal(stash['third_after_sync'], base['pd_mesh_xyz_left']))
    if bad:
        counts[tuple(bad)] = counts.get(tuple(bad), 0) + 1
        if len(counts) <= 3 and counts[tuple(bad)] == 1:
            for k in bad:
                d = (stash[k].float() - base[k].float()).abs()
                idx = d.flatten().argmax().item()
                print('  ', k, tuple(stash[k].shape), 'n_diff', int((d > 0).sum()), 'max', float(d.max()), 'first idx', int((d.flatten() > 0).nonzero()[0]))
print(counts)
