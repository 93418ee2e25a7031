"""Scratch: N engine instances, each with its own captured HIP graph and stream, replayed round-robin -- does a second
forward in flight (its low-occupancy token kernels / kernel tails under the other forward's convolutions) raise throughput?"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--inflight', type=int, default=2)
ap.add_argument('--steps', type=int, default=40)
ap.add_argument('--autotune-cache', default=None)
ap.add_argument('--share', action='store_true', help='one engine (shared weights), several captured graphs')
ap.add_argument('--stagger-us', type=float, nargs='*', default=[0.0])
args = ap.parse_args()
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
dev = torch.device('cuda', 0)
B = args.batch
engs, imgs, graphs, streams = [], [], [], []
table = None
for i in range(args.inflight):
    eng = engs[0] if (args.share and engs) else E.DirEngine(sd, dtype=torch.bfloat16, device=dev)
    img = torch.randn(B, 3, 256, 256, device=dev)
    eng.forward(img); torch.cuda.synchronize()
    if table is None:
        if args.autotune_cache and os.path.exists(args.autotune_cache):
            table = json.load(open(args.autotune_cache))
        else:
            eng.autotune(img)
            table = eng.export_tuning(B)
            if args.autotune_cache:
                json.dump(table, open(args.autotune_cache, 'w'))
    eng.import_tuning(img, table)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.forward(img); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.forward(img)
    torch.cuda.synchronize()
    engs.append(eng); imgs.append(img); graphs.append(g); streams.append(s)


def spin(us):                                   # busy kernel of ~us microseconds on the current stream
    torch.cuda._sleep(int(us * 100))            # wall_clock64 ticks at 100 MHz


def run(n_inflight, steps, stagger_us=0.0):
    for w in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if stagger_us and n_inflight > 1:
            for i in range(1, n_inflight):
                with torch.cuda.stream(streams[i]):
                    spin(stagger_us * i)
        for k in range(steps):
            i = k % n_inflight
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3


for n in range(1, args.inflight + 1):
    for st in (args.stagger_us if n > 1 else [0.0]):
        ms = run(n, args.steps, st)
        print('in flight %d stagger %6.0f us : %.3f ms per forward, %.0f images/s' % (n, st, ms, B / ms * 1e3))
