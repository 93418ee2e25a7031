"""Makes dir_amd/tuning/gfx950_bf16_b64_throughput.json: the per-layer DIR_CONV_VARIANT table for THROUGHPUT (several forwards in flight at
the socket power cap: DESIGN.md 9), by DirEngine.autotune_energy -- every convolution call of a bf16 forward at B = 64 replayed per variant
while rocm-smi is sampled, the choice minimising time x (power - idle power).  Then the four-in-flight step with the time-tuned and the
energy-tuned tables, alternating on the same four streams.
usage (GPU box): python tools/energy_tune.py [seconds per (layer, variant); 0 = default: 0.2 with the energy counter, 0.8 with rocm-smi] [rounds = 3]      -> gpurun_out/tuning/*.json (copy into dir_amd/tuning/)"""
import json, os, subprocess, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from dir_amd import engine as E, synth, power
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECS = float(sys.argv[1]) if len(sys.argv) > 1 and float(sys.argv[1]) > 0 else None
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODE = os.environ.get('MODE', 'bf16')          # bf16 | f16x3 | f16 (the fp32-tensor modes: DirEngine(dtype=float32, arith=MODE))
NAME = 'gfx950_%s_b64_throughput' % MODE
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16) if MODE == 'bf16' else E.DirEngine(sd, dtype=torch.float32, arith=MODE)
B, NSLOT = 64, 4
g = torch.Generator(device='cuda').manual_seed(0)
imgs = [torch.randn(B, 3, 256, 256, device='cuda', generator=g) for _ in range(NSLOT)]
img = imgs[0]
if MODE != 'bf16':
    eng.calibrate(img)
ref = eng.forward(img)
ref = [ref[i]['pd_mesh_xyz_left'].clone() for i in range(3)] + [ref[3]['seg'].clone()]
eng.autotune(img)
t_time = eng.export_tuning(B)
t0 = time.perf_counter()
rep = eng.autotune_energy(img, seconds=SECS, log=lambda r: print(r, flush=True), min_saving=float(os.environ.get('MIN_SAVING', '0.08')))
print('autotune_energy: %.0f s, idle %.0f W' % (time.perf_counter() - t0, rep['idle_w']), flush=True)
t_energy = eng.export_tuning(B)
out = eng.forward(img)
same = all(torch.equal(a, b) for a, b in zip(ref, [out[i]['pd_mesh_xyz_left'] for i in range(3)] + [out[3]['seg']]))
print('outputs bit-identical to the untuned forward:', same)
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=ROOT).stdout.strip() or os.environ.get('DIR_HEAD', '')
meta = {'made_by': 'tools/energy_tune.py', 'min_saving': float(os.environ.get('MIN_SAVING', '0.08')), 'instrument': rep['instrument'], 'objective': 'time x (socket power - idle power) per launch, launches replayed back to back',
        'device': torch.cuda.get_device_name(0), 'idle_w': rep['idle_w'], 'head': head, 'weights': 'dir_amd.synth seed 1234 (the choice depends on shapes only)',
        'changed_vs_time_tuned': sum(1 for a, b in zip(t_time, t_energy) if a[5] != b[5]), 'layers': rep['layers']}
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'tuning'), exist_ok=True)
json.dump({'batch': B, 'dtype': MODE, 'meta': meta, 'table': t_energy, 'time_tuned_table': t_time},
          open(os.path.join(ROOT, 'gpurun_out', 'tuning', NAME + '.json'), 'w'), indent=0)

STREAMS = [torch.cuda.Stream() for _ in range(NSLOT)]


def run(tag, table):
    eng.import_tuning(img, table)
    pipe = E.ForwardPipeline(eng, imgs, streams=STREAMS)
    k = [0]

    def step():
        pipe.launch(k[0] % NSLOT); k[0] += 1
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(60):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 60 * 1e3)
    smp = power.Sampler(skip=1.0, period=0.2).start()
    e0 = power.energy_joules()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50):
            step()
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    e1 = power.energy_joules()
    wc = (e1[0] - e0[0]) / dt if e0 and e1 else float('nan')
    s = smp.stop()
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            pipe.launch(0)
        torch.cuda.synchronize()
        one.append((time.perf_counter() - t0) / 10 * 1e3)
    print('%-14s %d in flight %.3f ms (regions %s), 3 s run %.3f ms at %4.0f W (energy counter %4.0f W = %.3f J per step) %4.0f MHz; one in flight %.3f ms' % (
        tag, NSLOT, statistics.median(res), ' '.join('%.3f' % r for r in res), dt / n * 1e3, power.median(s, 'w'), wc, wc * dt / n, power.median(s, 'sclk'),
        statistics.median(one)), flush=True)


for _ in range(ROUNDS):
    run('time-tuned', t_time)
    run('energy-tuned', t_energy)
