"""the high-resolution half (stem -> layer1 -> layer2) depth-first over sub-batches, so that its maps stay in the Infinity
Cache.  Same lease A/B: whole batch vs sub-batches of 32 / 16 / 8 images, 1 / 2 / 4 forwards in flight, ms and joules per 64-image step.
python tools/ab_subbatch.py"""
import os, sys, time, json, statistics
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth, power as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
    shapes = {k: tuple(v) for k, v in json.load(f).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
B = 64
g = torch.Generator(device='cuda').manual_seed(1)
imgs = [torch.randn(B, 3, 256, 256, device='cuda', generator=g) for _ in range(4)]
eng.forward(imgs[0]); torch.cuda.synchronize()
eng.autotune(imgs[0])
table = os.environ.get('TABLE', 'throughput')
if table == 'throughput':
    print('throughput table:', eng.load_tuning_table(imgs[0], 'gfx950_bf16_b64_throughput') is not None)
ref = [t.clone() for t in (eng.forward(imgs[0])[2]['pd_mesh_xyz_left'], eng.forward(imgs[0])[3]['seg'])]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
for rnd in range(2):
    for nb in (0, 32, 16, 8):
        eng.bb.subbatch = nb
        o = eng.forward(imgs[0])
        torch.cuda.synchronize()
        same = torch.equal(o[2]['pd_mesh_xyz_left'], ref[0]) and torch.equal(o[3]['seg'], ref[1])
        row = 'sub-batch %2d (bit-identical %s):' % (nb, same)
        for nfl in (1, 2, 4):
            pipe = E.ForwardPipeline(eng, imgs[:nfl], streams=streams[:nfl])
            k = [0]

            def step():
                pipe.launch(k[0] % nfl); k[0] += 1
            for _ in range(3 * nfl):
                step()
            torch.cuda.synchronize()
            e0 = P.energy_joules(); t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 1.5:
                for _ in range(40):
                    step()
                torch.cuda.synchronize(); n += 40
            dt = time.perf_counter() - t0; e1 = P.energy_joules()
            j = (e1[0] - e0[0]) / n if e0 and e1 else float('nan')
            row += '   %d in flight %.3f ms %.3f J' % (nfl, dt / n * 1e3, j)
            del pipe
        print(row, flush=True)
