"""Scratch: which DIR_CONV_VARIANT the engine's autotune picks per conv layer at B=64 (bf16), with the per-variant times."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
B = int(os.environ.get('B', 64))
img = torch.randn(B, 3, 256, 256, device='cuda')
eng.forward(img); torch.cuda.synchronize()
eng.overlap = False
times = {}
for v in eng.CONV_VARIANTS:
    E.ConvOp.default_variant = v
    E.PROFILE = []; eng.forward(img); torch.cuda.synchronize()
    E.PROFILE = []
    for _ in range(3): eng.forward(img)
    torch.cuda.synchronize()
    for r in E.PROFILE:
        times.setdefault((r[6], r[4]), {}).setdefault(v, []).append(r[2].elapsed_time(r[3]) * 1e3)
E.ConvOp.default_variant = None; E.PROFILE = None
tot_auto = tot_best = 0
names = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x128', 4: '64x64', 17: '128x128r', 18: '128x64r', 19: '64x128r', 20: '64x64r', 8: 'P256x128', 9: 'P128x128', 10: 'P256x64', 12: 'H256x128', 13: 'H128x128', 14: 'H256x64'}
for (op, shp), d in times.items():
    m = {v: min(ts) for v, ts in d.items()}
    bv = min(m, key=m.get)
    tot_auto += m[0]; tot_best += m[bv]
    print('%-38s pre=%d auto %6.1f best %-9s %6.1f   ' % (shp, getattr(op, 'pre_scale', None) is not None, m[0], names[bv], m[bv]) + ' '.join('%s:%.0f' % (names[v], m[v]) for v in sorted(m)))
print('sum auto %.1f us, sum best %.1f us' % (tot_auto, tot_best))
