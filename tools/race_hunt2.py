"""Scratch: the MANO launch alone, beside an unrelated convolution on another stream -- do its outputs vary?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = 64
eng = E.DirEngine(sd, dtype=torch.bfloat16)
g = torch.Generator(device='cuda').manual_seed(3)
para_l = torch.randn(B, 64, device='cuda', generator=g) * 0.3
para_r = torch.randn(B, 64, device='cuda', generator=g) * 0.3
c3 = torch.randn(B, 16, 16, 1024, device='cuda', generator=g).to(torch.bfloat16)
cat4 = torch.empty(B, 16, 16, 2304, device='cuda', dtype=torch.bfloat16)
side = torch.cuda.Stream()
base = E.run_mano_pair(eng.init_mano, para_l, para_r, B)
torch.cuda.synchronize()
base = [[t.clone() for t in h] for h in base]
c4 = torch.randn(B, 8, 8, 2048, device='cuda', generator=g).to(torch.bfloat16)
res = eng.res['skip_layer4']
y1 = res.c1(c3); y2 = res.c2(y1)
torch.cuda.synchronize()
modes = {'alone': None, 'skip_layer4': lambda: res(c3, out=cat4, out_coff=2048), 'c1 (pre-activation 1x1)': lambda: res.c1(c3),
         'c2 (3x3)': lambda: res.c2(y1), 'dual (conv3 + skip)': lambda: res.dual(y2, c3, out=cat4, out_coff=2048),
         'upsample': lambda: eng.upsample_into(c4, cat4, 0)}
for mode, fn in modes.items():
    bad = 0
    for rep in range(40):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        eng.attn(c4)                                  # 250 us on the main stream: the CPU gets ahead, as in the forward
        if fn is not None:
            with torch.cuda.stream(side):
                for _ in range(3): fn()
        out = E.run_mano_pair(eng.init_mano, para_l, para_r, B)
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for ha, hb in zip(out, base) for a, b in zip(ha, hb)):
            bad += 1
    print('%-26s: %d of 40 runs differ' % (mode, bad))
