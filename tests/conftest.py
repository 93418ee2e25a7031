import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        return cache[name]
    return load


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def loss_case(g):
    """inputs of the G8 fixture in the shape oracle.losses / dir_amd.models.loss take them: (per-stage prediction dicts, ground
    truth dict, (faces_left, faces_right), seg logits, dense prediction, gt seg [B,1,256,256] f32, gt dense [B,3,256,256] f32)"""
    from dir_amd import synth
    preds = []
    for i in range(3):
        d = {k.split('.', 1)[1]: g[k] for k in g if k.startswith('s%d.' % i)}
        preds.append(d)
    gt = {k[3:]: g[k] for k in g if k.startswith('gt_') and not k.endswith('_u8')}
    faces = tuple(synth.loss_faces(side, 1234) for side in ('left', 'right'))
    gt_seg = g['gt_seg_u8'].astype(np.float32)
    gt_dense = g['gt_dense_u8'].astype(np.float32) / np.float32(255.0)
    return preds, gt, faces, g['seg'], g['dense'], gt_seg, gt_dense


def check_compact_grads(got, g, tol, prefix='grad.', zero_suffixes=()):
    """compare {key: gradient array} with a fixture written by oracle/gen_golden.py::compact_grads (small tensors whole; matrices as every
    n-th column + float64 row / column sums).  Returns the worst error relative to each gradient's maximum; asserts it is < tol."""
    worst = 0.0
    keys = set()
    for k in g:
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        for suf in ('.rowsum', '.colsum'):
            if name.endswith(suf):
                name = name[:-len(suf)]
        if '.cols' in name and name.rsplit('.cols', 1)[1].isdigit():
            name = name.rsplit('.cols', 1)[0]
        keys.add(name)
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith(prefix))
    for name in sorted(keys):
        if name not in got:
            continue
        a = np.asarray(got[name], np.float64)
        while a.ndim > 2 and a.shape[-1] == 1:              # Conv1d k=1 weights [out, in, 1]
            a = a[..., 0]
        if any(name.endswith(z) for z in zero_suffixes):
            # analytically zero (e.g. a bias in front of a training-mode BatchNorm, which subtracts the batch mean): both sides hold
            # rounding noise only
            assert np.abs(a).max() < 1e-4 * gmax and np.abs(g[prefix + name]).max() < 1e-4 * gmax, name
            continue
        if prefix + name in g:
            ref = g[prefix + name]
            e = np.abs(a.reshape(ref.shape) - ref).max() / (np.abs(ref).max() + 1e-30)
        else:
            a2 = a.reshape(a.shape[0], -1) if (a.ndim == 4 and a.shape[-1] <= 7) else a.reshape(-1, a.shape[-1])       # conv OIHW -> [O, I*kh*kw]
            ck = [k for k in g if k.startswith(prefix + name + '.cols')][0]
            step = int(ck.rsplit('.cols', 1)[1])
            # the sums are compared on the scale of the L1 norms of the rows / columns they sum: a gradient that flows out of a
            # LayerNorm has (exactly) zero channel sum, so some of these sums are pure cancellation
            e = max(np.abs(a2[:, ::step] - g[ck]).max() / (np.abs(g[ck]).max() + 1e-30),
                    np.abs(a2.sum(1) - g[prefix + name + '.rowsum']).max() / (np.abs(a2).sum(1).max() + 1e-30),
                    np.abs(a2.sum(0) - g[prefix + name + '.colsum']).max() / (np.abs(a2).sum(0).max() + 1e-30))
        worst = max(worst, float(e))
        assert e < tol, (name, float(e))
    return worst
