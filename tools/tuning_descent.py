"""Scratch: coordinate descent on the shipped throughput table, measured where it counts -- the four-in-flight step itself.  For the dozen
launches that cost the most joules, try a handful of other variants one at a time (graphs re-captured on the same four streams), keep a change
only if it wins by > 0.4 % twice.  Prints the resulting table diff."""
import json, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.bfloat16)
B, NSLOT = 64, 4
g = torch.Generator(device='cuda').manual_seed(0)
imgs = [torch.randn(B, 3, 256, 256, device='cuda', generator=g) for _ in range(NSLOT)]
img = imgs[0]
eng.forward(img); eng.autotune(img)
T = json.load(open(os.path.join(ROOT, 'dir_amd', 'tuning', 'gfx950_bf16_b64_throughput.json')))
table = [list(r) for r in T['table']]
STREAMS = [torch.cuda.Stream() for _ in range(NSLOT)]


def measure(tab, reps=3, steps=40):
    eng.import_tuning(img, tab)
    pipe = E.ForwardPipeline(eng, imgs, streams=STREAMS)
    k = [0]

    def step():
        pipe.launch(k[0] % NSLOT); k[0] += 1
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    res = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    del pipe
    return statistics.median(res)


# rank table rows by the joules of the rated launches (meta.layers is in conv-call order; match rows by shape, in order)
layers = T['meta']['layers']
rows_for_layer, j = [], 0
for L in layers:
    while j < len(table) and not (table[j][0] == L['cout'] and table[j][1] == L['cin'] and table[j][2] == L['kh'] and table[j][5] == L['chosen']):
        j += 1
    if j == len(table):
        break
    rows_for_layer.append((j, L)); j += 1
print('matched %d of %d rated launches to table rows' % (len(rows_for_layer), len(layers)))
rows_for_layer.sort(key=lambda t: -t[1]['us'] * (t[1]['w'] - 243))
base = measure(table); base = min(base, measure(table))
print('baseline %.3f ms' % base, flush=True)
changes = []
for idx, L in rows_for_layer[:int(os.environ.get('TOP', 12))]:
    cur = table[idx][5]
    best_v, best_t = cur, base
    for v in [L['fastest'], 0, 8, 11, 12, 15, 21, 17]:
        if v == cur or v not in eng.CONV_VARIANTS:
            continue
        tab = [list(r) for r in table]; tab[idx][5] = v
        t = measure(tab)
        if t < best_t * 0.996:
            t2 = measure(tab)                       # confirm
            if t2 < base * 0.996:
                best_v, best_t = v, max(t, t2)
    print('row %2d cout %4d cin %4d k%d: %2d -> %2d  %.3f ms (base %.3f)' % (idx, L['cout'], L['cin'], L['kh'], cur, best_v, best_t, base), flush=True)
    if best_v != cur:
        table[idx][5] = best_v
        changes.append((idx, cur, best_v))
        base = measure(table)
print('final %.3f ms, changes %s' % (measure(table), changes))
json.dump(table, open(os.path.join(ROOT, 'gpurun_out', 'descent_table.json'), 'w'))
