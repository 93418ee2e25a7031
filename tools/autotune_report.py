"""Scratch: per conv layer at B=64 (bf16), the time of every DIR_CONV_VARIANT the engine's autotune may pick (eager, one forward in
flight, HIP events around each library call), with the layer's algorithmic TFLOP/s and GB/s at the best one."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth, _capi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = E.DirEngine(sd, dtype=torch.float16 if os.environ.get('DT', 'f16') == 'f16' else torch.bfloat16)
B = int(os.environ.get('B', 64))
img = torch.randn(B, 3, 256, 256, device='cuda')
eng.forward(img); torch.cuda.synchronize()
eng.overlap = False
times, info, order = {}, {}, []
for v in eng.CONV_VARIANTS:
    E._TLS.variant = v
    eng._profiled_forwards(img, 1)
    for r in eng._profiled_forwards(img, 3):
        k = id(r['op'])
        if k not in info:
            order.append(k)
        info[k] = r if v == 0 else info.get(k, r)
        times.setdefault(k, {}).setdefault(v, []).append(r['e0'].elapsed_time(r['e1']) * 1e3)
E._TLS.variant = None
names = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x128', 4: '64x64', 17: '128x128r', 18: '128x64r', 19: '64x128r', 20: '64x64r', 8: 'P256x128', 9: 'P128x128',
         10: 'P256x64', 11: 'B256x256', 12: 'H256x128', 13: 'H128x128', 14: 'H256x64', 15: 'P128x64', E.STREAM_VARIANT: 'stream', 22: 'stream64', 23: 'stream32', 25: 'AS2x2', 26: 'AS2x4', 27: 'AS4x2', 28: 'AS1x2'}
tot_auto = tot_best = tot_noas = 0
for k in order:
    m = {v: min(ts) for v, ts in times[k].items()}
    n = len(times[k][0]) // 3
    bv = min(m, key=m.get)
    tot_auto += m[0] * n; tot_best += m[bv] * n
    tot_noas += min(t for v, t in m.items() if v not in E.AS_VARIANTS) * n
    r = info[k]
    print('%-52s x%d auto %6.1f best %-9s %6.1f us %7.1f TF %7.1f GB/s | ' % (r.get('shape', r['api'])[:52], n, m[0], names[bv], m[bv], r.get('flops', 0) / m[bv] / 1e6,
          r.get('bytes', 0) / m[bv] / 1e3) + ' '.join('%s:%.0f' % (names[v], m[v]) for v in sorted(m) if v))
print('sum auto %.1f us, sum best %.1f us, sum best without the activation-stationary kernel %.1f us' % (tot_auto, tot_best, tot_noas))
