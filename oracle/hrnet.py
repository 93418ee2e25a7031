"""ORACLE (test infrastructure): numpy forward of the HRNet-W48 backbone of dir_amd/models/backbone/hrnet.py.  The
reference has NO HRNet: this restates the published architecture (Sun et al., CVPR 2019, the official pose / classification code's module
structure: HighResolutionModule branches + fuse_layers with nearest upsampling) exactly as the mirror module's docstring words it, on the
oracle's own conv / batchnorm / relu (oracle/nnops.py, held to the reference goldens elsewhere).  Parity of this row is pinned to this file
only ("parity unpinned" against any third party)."""
import numpy as np

from . import nnops as N
from .dir_forward import bottleneck

WIDTHS = (48, 96, 192, 384)
MODULES = ((2, 1), (3, 4), (4, 3))


def _cbr(x, P, stride, pad, relu=True):
    y = N.batchnorm(N.conv2d(x, P['0.weight'], None, stride, pad), P.sub('1'))
    return N.relu(y) if relu else y


def _basic(x, P):
    y = N.relu(N.batchnorm(N.conv2d(x, P['conv1.weight'], None, 1, 1), P.sub('bn1')))
    y = N.batchnorm(N.conv2d(y, P['conv2.weight'], None, 1, 1), P.sub('bn2'))
    return N.relu(y + x)


def _up(x, f):
    return np.repeat(np.repeat(x, f, axis=2), f, axis=3)


def _module(xs, P):
    nb = len(xs)
    ys = []
    for b in range(nb):
        y = xs[b]
        for k in range(4):
            y = _basic(y, P.sub('branches.%d.%d' % (b, k)))
        ys.append(y)
    outs = []
    for i in range(nb):
        acc = None
        for j in range(nb):
            if j == i:
                t = ys[j]
            elif j > i:
                t = _up(_cbr(ys[j], P.sub('fuse_layers.%d.%d' % (i, j)), 1, 0, relu=False), 2 ** (j - i))
            else:
                t = ys[j]
                for s in range(i - j):
                    t = _cbr(t, P.sub('fuse_layers.%d.%d.%d' % (i, j, s)), 2, 1, relu=(s < i - j - 1))
            acc = t if acc is None else acc + t
        outs.append(N.relu(acc))
    return outs


def hrnet_w48(x, P):
    """x NCHW [B,3,H,W] -> [c1, c2, c3, c4] NCHW (256@H/4, 512@H/8, 1024@H/16, 2048@H/32)"""
    x = N.relu(N.batchnorm(N.conv2d(x, P['conv1.weight'], None, 2, 1), P.sub('bn1')))
    x = N.relu(N.batchnorm(N.conv2d(x, P['conv2.weight'], None, 2, 1), P.sub('bn2')))
    for b in range(4):
        x = bottleneck(x, P.sub('layer1.%d' % b), 1)
    xs = [_cbr(x, P.sub('transition1.0'), 1, 1), _cbr(x, P.sub('transition1.1'), 2, 1)]
    for st, n in MODULES:
        if st > 2:
            xs = xs + [_cbr(xs[-1], P.sub('transition%d' % (st - 1)), 2, 1)]
        for m in range(n):
            xs = _module(xs, P.sub('stage%d.%d' % (st, m)))
    return [_cbr(xs[b], P.sub('incre.%d' % b), 1, 0) for b in range(4)]
