"""Where a forward's joules go, launch by launch: every convolution-family call of one bf16 forward at B = 64 (the tile convolutions with the
variant the shipped throughput table gives them, the fused layer1 chain and layer2 / layer3 tail launches) replayed back to back on its real
tensors for 0.2 s while the socket's energy accumulator is read (dir_amd/power.py).  Prints microseconds, watts and joules above idle per
launch, largest first, and the sum -- to be held against the forward's total (bench.py: power.energy_counter.joules_per_step).
usage (GPU box): python tools/energy_profile.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth, power
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
img = torch.randn(64, 3, 256, 256, device='cuda')
eng.forward(img); eng.autotune(img)
which = 'throughput table' if eng.load_tuning_table(img, 'gfx950_bf16_b64_throughput') is not None else 'time-tuned'
if power.energy_joules() is None:
    sys.exit('no amdsmi energy counter on this machine')
E._TLS.capture, E._TLS.capture_fused = [], []
eng.forward(img); torch.cuda.synchronize()
calls = [('conv', c) for c in E._TLS.capture] + [('fused', c) for c in E._TLS.capture_fused]
E._TLS.capture, E._TLS.capture_fused = None, None
rows = []
for kind, (op, args, kw) in calls:
    for _ in range(5):
        op(*args, **kw)
    torch.cuda.synchronize()
    e0 = power.energy_joules(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.2:
        for _ in range(20):
            op(*args, **kw)
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    e1 = power.energy_joules()
    w = (e1[0] - e0[0]) / dt
    name = type(op).__name__ + ' cout %d cin %s k%d' % (op.cout, getattr(op, 'cin', '?'), getattr(op, 'kh', 1))
    rows.append((dt / n * 1e6, w, (w - power.IDLE_W) * dt / n, name, op.variant.get(64, 0) if kind == 'conv' else '-'))
rows.sort(key=lambda r: -r[2])
print('%d launches of the conv family (%s), replayed alone:' % (len(rows), which))
for us, w, j, name, v in rows:
    print('  %-44s variant %-3s %7.1f us  %5.0f W  %.4f J above idle' % (name, v, us, w, j))
print('sum %.3f J above idle, %.1f us' % (sum(r[2] for r in rows), sum(r[0] for r in rows)))
