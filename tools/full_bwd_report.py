"""Per-parameter report of the whole training step's gradient (dir_amd/train/net.py) against G20: our error and the reference's own fp32
evaluation noise, both relative to the float64 gradient of the reference graph."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dir_amd import synth
from dir_amd.train import net as TN
from conftest import loss_case
G8 = dict(np.load('/root/repo/tests/golden/g8_loss.npz')); g20 = dict(np.load('/root/repo/tests/golden/g20_full_grad.npz'))
shapes = {k: tuple(v) for k, v in json.load(open('/root/repo/tests/golden/manifest_dir.json')).items()}
sd = synth.synth_state_dict(shapes, 1234)
P = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items() if 'num_batches' not in k}
img = torch.from_numpy(synth.synth_input('loss.img', (2, 3, 256, 256), 1234)).cuda()
preds, gt, faces, _, _, gt_seg, gt_dense = loss_case(G8)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
target = {k: dv(v) for k, v in gt.items() if 'center' not in k}; target.update(seg=dv(gt_seg), dense=dv(gt_dense))
meta = {k: dv(v) for k, v in gt.items() if 'center' in k}
fc = tuple(dv(f.astype(np.int64)) for f in faces)
outs, ctx = TN.forward(P, img)
G = TN.backward(P, ctx, outs, target, meta, fc)
order = list(shapes)
res = []
for k in order:
    if k not in G: continue
    a = G[k].cpu().numpy().astype(np.float64)
    while a.ndim > 2 and a.shape[-1] == 1: a = a[..., 0]
    if k.endswith('gconv.W'): a = a.reshape(-1, 128)
    if 'grad.' + k in g20:
        ref = g20['grad.' + k]; e = np.abs(a.reshape(ref.shape) - ref).max() / (np.abs(ref).max() + 1e-30)
    else:
        a2 = a.reshape(a.shape[0], -1) if (a.ndim == 4 and a.shape[-1] <= 7) else a.reshape(-1, a.shape[-1])
        ck = [q for q in g20 if q.startswith('grad.' + k + '.cols')][0]; step = int(ck.rsplit('.cols', 1)[1])
        e = np.abs(a2[:, ::step] - g20[ck]).max() / (np.abs(g20[ck]).max() + 1e-30)
    res.append((k, e, float(g20['ref32_err.' + k])))
for k, e, r in res:
    if e > 3 * r + 1e-4: print('%-70s ours %.2e   reference fp32 %.2e' % (k, e, r))
print('worse than 3x the reference fp32 noise:', sum(e > 3 * r + 1e-4 for _, e, r in res), 'of', len(res))
import statistics
print('median ours %.2e   median reference fp32 %.2e' % (statistics.median(e for _, e, _ in res), statistics.median(r for _, _, r in res)))

