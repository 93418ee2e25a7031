"""dir_train_pack_conv_weights on the network's real weight shapes, whole and by class.  python tools/bench_weight_pack.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd.train import conv as TC
shapes = [tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()
          if k.endswith('.weight') and len(v) == 4 and v[1] % 4 == 0]
print('%d convolution weights, %.1f M parameters' % (len(shapes), sum(a * b * c * d for a, b, c, d in shapes) / 1e6))


def timeit(ws, n=10):
    pk = TC.WeightPack(ws)
    for _ in range(3):
        pk.refresh()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pk.refresh()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, pk.total_rows


classes = {'all': shapes, '1x1': [s for s in shapes if s[2] == 1], '3x3': [s for s in shapes if s[2] == 3],
           '3x3 Cin >= 2048': [s for s in shapes if s[2] == 3 and s[1] >= 2048], '3x3 Cin < 2048': [s for s in shapes if s[2] == 3 and s[1] < 2048],
           '1x1 Cout >= 1024': [s for s in shapes if s[2] == 1 and s[0] >= 1024], '1x1 small': [s for s in shapes if s[2] == 1 and s[0] < 1024]}
for name, ss in classes.items():
    ws = [torch.randn(*s, device='cuda') for s in ss]
    t, rows = timeit(ws)
    nb = sum(w.numel() for w in ws) * 4
    print('%-18s %3d weights %6d rows %7.1f MB  %8.1f us  (%.2f TB/s of read + write)' % (name, len(ws), rows, nb / 1e6, t, nb * 3 / t / 1e6))
