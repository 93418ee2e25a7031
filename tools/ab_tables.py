"""Same-process, same-streams A/B of throughput tables: python tools/ab_tables.py tableA.json tableB.json [...]  (f16 storage, B = 64, four forwards in flight;
alternating rounds; tables are the JSON files tools/energy_tune.py writes / dir_amd/tuning/ ships)"""
import json, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=True).items()}
eng = E.DirEngine(sd, dtype=torch.float16)
g = torch.Generator(device='cuda').manual_seed(0)
imgs = [torch.randn(64, 3, 256, 256, device='cuda', generator=g) for _ in range(4)]
eng.forward(imgs[0]); eng.autotune(imgs[0])
tables = {'time-tuned (live)': eng.export_tuning(64)}
for p in sys.argv[1:]:
    tables[os.path.basename(os.path.dirname(p)) + '/' + os.path.basename(p)] = json.load(open(p))['table']
streams = [torch.cuda.Stream() for _ in range(4)]
pipes = {}
for name, t in tables.items():
    eng.import_tuning(imgs[0], t)
    pipes[name] = E.ForwardPipeline(eng, imgs, streams=streams)
res = {n: [] for n in pipes}
for rnd in range(4):
    for name, pipe in pipes.items():
        k = [0]
        def step():
            pipe.launch(k[0] % 4); k[0] += 1
        for _ in range(12): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 200 * 1e3)
for name, r in res.items():
    print('%-60s median %.3f ms/step  (%s)' % (name, statistics.median(r), ' '.join('%.3f' % x for x in r)))
