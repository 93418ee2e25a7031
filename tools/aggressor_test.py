"""Scratch: victim / aggressor harness.  Stream B replays a graph of N MANO launches (distinct outputs); stream A replays a graph
of one kernel type back to back.  Which aggressor makes the victim's results vary?"""
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
rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
para_l, para_r = rn(B, 64) * 0.3, rn(B, 64) * 0.3
NV = 30
sv, sa = torch.cuda.Stream(), torch.cuda.Stream()


def victim_graph():
    outs = []
    with torch.cuda.stream(sv):
        E.run_mano_pair(eng.init_mano, para_l, para_r, B); sv.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=sv):
            for _ in range(NV):
                outs.append(E.run_mano_pair(eng.init_mano, para_l, para_r, B))
    return gr, outs


vg, vouts = victim_graph()
with torch.cuda.stream(sv):
    vg.replay()
torch.cuda.synchronize()
ref = [t.clone() for t in vouts[0][0]] + [t.clone() for t in vouts[0][1]]
assert all(torch.equal(a, b) for o in vouts for a, b in zip(o[0] + o[1], ref)), 'victim not deterministic alone'

c4 = rn(B, 8, 8, 2048).to(torch.bfloat16)
c3 = rn(B, 16, 16, 1024).to(torch.bfloat16)
c2 = rn(B, 32, 32, 512).to(torch.bfloat16)
x32 = rn(B, 32, 32, 256).to(torch.bfloat16)
img = rn(B, 3, 256, 256)
res4, res3 = eng.res['skip_layer4'], eng.res['skip_layer3']
y1 = res4.c1(c3); y2 = res4.c2(y1)
z1 = res3.c1(c2)
torch.cuda.synchronize()


def with_variant(op, v, fn):
    def run():
        op.variant[B] = v
        return fn()
    return run


aggr = {}
if os.environ.get('ONLY19'):
    aggr['skip4.dual variant 19'] = with_variant(res4.dual, 19, lambda: res4.dual(y2, c3))
elif os.environ.get('DUAL_SWEEP'):
    for v in (0, 1, 2, 3, 4, 17, 18, 19, 20):
        aggr['skip4.dual variant %d' % v] = with_variant(res4.dual, v, lambda: res4.dual(y2, c3))
    for v in (0, 2, 18, 4, 20):
        aggr['skip4.c2 (3x3) variant %d' % v] = with_variant(res4.c2, v, lambda: res4.c2(y1))
    z2 = res3.c2(z1)
    for v in (0, 2, 18):
        aggr['skip3.dual variant %d' % v] = with_variant(res3.dual, v, lambda: res3.dual(z2, c2))
aggr_all = {
    'attention conv (conv_pipe 256x128)': lambda: eng.attn(c4),
    'skip4.c1 (pre-act 1x1)': lambda: res4.c1(c3),
    'skip4.c2 (3x3 @16)': lambda: res4.c2(y1),
    'skip4.dual': lambda: res4.dual(y2, c3),
    'skip3.c1 (pre-act 1x1 @32)': lambda: res3.c1(c2),
    'skip3.c2 (3x3 @32)': lambda: res3.c2(z1),
    'conv_final (3x3 256->256 @32)': lambda: eng.final0(x32),
    'backbone (stem + chains + layers)': lambda: eng.bb(img),
    'upsample': lambda: eng.upsample_into(c4, torch.empty(B, 16, 16, 2304, device='cuda', dtype=torch.bfloat16), 0),
}
if not aggr: aggr = aggr_all
for name, fn in aggr.items():
    with torch.cuda.stream(sa):
        fn(); sa.synchronize()
        ag = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ag, stream=sa):
            for _ in range(4 if 'backbone' in name else 24):
                fn()
    torch.cuda.synchronize()
    bad = 0
    for rep in range(20):
        with torch.cuda.stream(sa):
            ag.replay()
        with torch.cuda.stream(sv):
            vg.replay()
        torch.cuda.synchronize()
        nb = sum(0 if all(torch.equal(a, b) for a, b in zip(o[0] + o[1], ref)) else 1 for o in vouts)
        if nb and not bad and os.environ.get('DETAIL'):
            names = ['L.verts', 'L.joints', 'L.uv', 'R.verts', 'R.joints', 'R.uv']
            for li, o in enumerate(vouts):
                for nm, a, b in zip(names, o[0] + o[1], ref):
                    if not torch.equal(a, b):
                        d = (a != b)
                        fl = d.flatten().nonzero().flatten()
                        per = d.reshape(B, -1).any(1).nonzero().flatten().tolist()
                        print('   launch %d %s: %d elements, samples %s, flat idx within sample: %s, max |d| %.2e' % (
                            li, nm, len(fl), per[:6], sorted(set((fl % (a.numel() // B)).tolist()))[:8], float((a - b).abs().max())))
                if li > 3: break
        bad += nb
    print('%-40s: %d of %d victim launches differ' % (name, bad, 20 * NV))
