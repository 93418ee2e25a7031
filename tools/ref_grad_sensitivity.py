#!/usr/bin/env python3
"""AUTHORING CONTAINER ONLY (imports /root/reference through oracle/gen_golden.py's harness): how reproducible is the REFERENCE's own
fp32 training gradient under a change of summation order?  The same DIR graph (train mode, trained-like synthetic parameters, G8c's
targets), the same fp32 kernels, evaluated with 8 and with 1 BLAS / oneDNN threads; optionally with the BatchNorm layers in eval mode,
at another batch size, or for a subset of the 42 loss terms.

    python tools/ref_grad_sensitivity.py                 # B = 2, all terms, BatchNorm in training mode
    python tools/ref_grad_sensitivity.py --evalbn        # the BatchNorm layers frozen
    python tools/ref_grad_sensitivity.py --batch 8 --terms _0

Measured (round 3): training-mode BatchNorm: median 2e-2 .. 4e-2 of each gradient's maximum (B = 2 and B = 8, every subset of terms, two
runs at the same thread count bit-identical); eval-mode BatchNorm: 4e-5.  That is the floor any whole-step gradient parity test can be
held to (tests/test_gpu_full_bwd.py uses the per-parameter figure stored in tests/golden/g20c_full_grad.npz)."""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--evalbn', action='store_true')
ap.add_argument('--terms', default='', help='only the loss terms whose key ends with this suffix')
args = ap.parse_args()
sys.argv = sys.argv[:1]
spec = importlib.util.spec_from_file_location('gen_golden', os.path.join(ROOT, 'oracle', 'gen_golden.py'))
gg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gg)
from dir_amd import synth  # noqa: E402

gg.import_reference()
from models.dir import DIR  # noqa: E402

SEED = 1234
g8 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g8c_loss.npz'))


def run(nthreads):
    torch.set_num_threads(nthreads)
    B = args.batch
    net = DIR(21, 'unused', 0)
    gg.load_synth(net, cond=True)
    net.train()
    if args.evalbn:
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.eval()
    for side in ('left', 'right'):
        fc = torch.from_numpy(synth.loss_faces(side, SEED))
        getattr(net, 'normal_loss_' + side).face = fc
        getattr(net, 'edge_loss_' + side).face = fc
    img = torch.from_numpy(synth.synth_input('loss.img', (B, 3, 256, 256), SEED))
    rep = lambda t: t.repeat((B // 2,) + (1,) * (t.dim() - 1))  # noqa: E731
    target = {k[3:]: rep(torch.from_numpy(g8[k])) for k in g8.files if k.startswith('gt_') and not k.endswith('_u8') and 'center' not in k}
    target['seg'] = rep(torch.from_numpy(g8['gt_seg_u8'].astype(np.float32)))
    target['dense'] = rep(torch.from_numpy(g8['gt_dense_u8'].astype(np.float32) / np.float32(255.0)))
    meta = {k[3:]: rep(torch.from_numpy(g8[k])) for k in g8.files if k.startswith('gt_center')}
    _, loss = net({'img': img}, target, meta)
    sum(v for k, v in loss.items() if k.endswith(args.terms)).backward()
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


a, a2, b = run(8), run(8), run(1)
same = all(torch.equal(a[k], a2[k]) for k in a)
errs = sorted(float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)) for k in a if float(a[k].abs().max()) > 0)
print('two 8-thread runs bit-identical: %s;  8 threads vs 1 thread: median %.2e, 90th percentile %.2e of each gradient maximum (%d tensors)'
      % (same, errs[len(errs) // 2], errs[int(0.9 * len(errs))], len(errs)))
