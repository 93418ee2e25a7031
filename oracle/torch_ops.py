"""ORACLE (test infrastructure, never shipped or measured as product): the dense operators of oracle/nnops.py re-stated on stock
PyTorch CPU kernels, for bench.py's `cpu_baseline` leg ("the same graph built from stock torch ops on the host cores").

`stock_torch_dense_ops()` is a context manager that swaps oracle.nnops' conv2d / batchnorm / relu / maxpool / upsample / linear /
conv1d for torch.nn.functional calls on zero-copy views of the same numpy arrays, so that oracle.dir_forward.dir_forward -- the
numpy restatement of DIR.forward (models/dir.py:513-540) -- runs 99.9 % of its FLOPs through ATen's own CPU kernels (oneDNN /
MKL), the way the reference itself runs on a CPU; the token path (P-GCN, STE, MANO, bone rasterisation: < 0.3 % of the FLOPs)
stays numpy.  tests/test_oracle_golden.py holds the swapped forward to the numpy one.

Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import contextlib

import numpy as np

from . import nnops as N


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def conv2d(x, w, b=None, stride=1, pad=0):
    import torch.nn.functional as F
    return F.conv2d(_t(x), _t(w), None if b is None else _t(b), stride, pad).numpy()


def conv1d_k1(x, w, b):
    import torch.nn.functional as F
    return F.conv1d(_t(x), _t(w), _t(b)).numpy()


def linear(x, w, b=None):
    import torch.nn.functional as F
    return F.linear(_t(x), _t(w), None if b is None else _t(b)).numpy()


def batchnorm(x, P, eps=1e-5):
    import torch.nn.functional as F
    return F.batch_norm(_t(x), _t(P['running_mean']), _t(P['running_var']), _t(P['weight']), _t(P['bias']), False, 0.0, eps).numpy()


def relu(x):
    import torch
    return torch.relu(_t(x)).numpy()


def maxpool3x3s2p1(x):
    import torch.nn.functional as F
    return F.max_pool2d(_t(x), 3, 2, 1).numpy()


def upsample_bilinear2x(x):
    import torch.nn.functional as F
    return F.interpolate(_t(x), scale_factor=2, mode='bilinear').numpy()


_SWAP = dict(conv2d=conv2d, conv1d_k1=conv1d_k1, linear=linear, batchnorm=batchnorm, relu=relu, maxpool3x3s2p1=maxpool3x3s2p1,
             upsample_bilinear2x=upsample_bilinear2x)


@contextlib.contextmanager
def stock_torch_dense_ops(num_threads=None):
    import torch
    saved = {k: getattr(N, k) for k in _SWAP}
    old_threads = torch.get_num_threads()
    try:
        if num_threads:
            torch.set_num_threads(int(num_threads))
        for k, f in _SWAP.items():
            setattr(N, k, f)
        with torch.no_grad():
            yield
    finally:
        for k, f in saved.items():
            setattr(N, k, f)
        torch.set_num_threads(old_threads)
