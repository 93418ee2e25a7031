"""ORACLE (test infrastructure): the two residual blocks of the image half in TRAINING mode, forward and analytic backward in numpy
float64 -- Bottleneck (models/backbone/resnet.py:86-142) and Residual (models/backbone/hourglass.py:33-70) as torch autograd
differentiates nn.Conv2d / nn.BatchNorm2d (batch statistics) / nn.ReLU.  Pinned against torch autograd through the reference's own
classes (tests/golden/g18_block_grad_*.npz, oracle/gen_golden.py::gen_block_grad).  Layout NCHW, weights OIHW (the reference's)."""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


def conv_fwd(x, w, b, stride, pad):
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = sliding_window_view(xp, w.shape[2:], axis=(2, 3))[:, :, ::stride, ::stride]          # [B,C,Ho,Wo,kh,kw]
    y = np.einsum('bchwij,ocij->bohw', win, w, optimize=True)
    return y if b is None else y + b[None, :, None, None]


def conv_bwd(x, w, gy, stride, pad):
    """-> (g x, g w, g b)"""
    B, C, H, W = x.shape
    kh, kw = w.shape[2:]
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    win = sliding_window_view(xp, (kh, kw), axis=(2, 3))[:, :, ::stride, ::stride]
    gw = np.einsum('bchwij,bohw->ocij', win, gy, optimize=True)
    gxp = np.zeros_like(xp)
    Ho, Wo = gy.shape[2:]
    for i in range(kh):
        for j in range(kw):
            gxp[:, :, i:i + stride * Ho:stride, j:j + stride * Wo:stride] += np.einsum('bohw,oc->bchw', gy, w[:, :, i, j], optimize=True)
    return gxp[:, :, pad:pad + H, pad:pad + W], gw, gy.sum((0, 2, 3))


def bn_fwd(P, pre, x, running, momentum=0.1, eps=1e-5):
    mu, var = x.mean((0, 2, 3)), x.var((0, 2, 3))
    rs = 1 / np.sqrt(var + eps)
    xh = (x - mu[None, :, None, None]) * rs[None, :, None, None]
    n = x.size // x.shape[1]
    running[pre + 'running_mean'] = (1 - momentum) * P[pre + 'running_mean'] + momentum * mu
    running[pre + 'running_var'] = (1 - momentum) * P[pre + 'running_var'] + momentum * var * n / (n - 1)
    return xh * P[pre + 'weight'][None, :, None, None] + P[pre + 'bias'][None, :, None, None], (xh, rs)


def bn_bwd(P, pre, saved, gy, G):
    xh, rs = saved
    G[pre + 'weight'], G[pre + 'bias'] = (gy * xh).sum((0, 2, 3)), gy.sum((0, 2, 3))
    g = gy * P[pre + 'weight'][None, :, None, None]
    m1, m2 = g.mean((0, 2, 3), keepdims=True), (g * xh).mean((0, 2, 3), keepdims=True)
    return rs[None, :, None, None] * (g - m1 - xh * m2)


def _f8(P):
    return {k: np.asarray(v, np.float64) for k, v in P.items()}


def bottleneck(P, x, gy, stride):
    """-> (y, g x, {key: gradient}, {running statistic after the step})"""
    P, x, gy = _f8(P), np.asarray(x, np.float64), np.asarray(gy, np.float64)
    R, G = {}, {}
    h1 = conv_fwd(x, P['conv1.weight'], None, 1, 0)
    n1, s1 = bn_fwd(P, 'bn1.', h1, R)
    a1 = np.maximum(n1, 0)
    h2 = conv_fwd(a1, P['conv2.weight'], None, stride, 1)
    n2, s2 = bn_fwd(P, 'bn2.', h2, R)
    a2 = np.maximum(n2, 0)
    h3 = conv_fwd(a2, P['conv3.weight'], None, 1, 0)
    n3, s3 = bn_fwd(P, 'bn3.', h3, R)
    down = 'downsample.0.weight' in P
    if down:
        hd = conv_fwd(x, P['downsample.0.weight'], None, stride, 0)
        idn, sd = bn_fwd(P, 'downsample.1.', hd, R)
    else:
        idn = x
    y = np.maximum(n3 + idn, 0)
    g = gy * (y > 0)
    g3 = bn_bwd(P, 'bn3.', s3, g, G)
    g2, G['conv3.weight'], _ = conv_bwd(a2, P['conv3.weight'], g3, 1, 0)
    g2 = bn_bwd(P, 'bn2.', s2, g2 * (a2 > 0), G)
    g1, G['conv2.weight'], _ = conv_bwd(a1, P['conv2.weight'], g2, stride, 1)
    g1 = bn_bwd(P, 'bn1.', s1, g1 * (a1 > 0), G)
    gx, G['conv1.weight'], _ = conv_bwd(x, P['conv1.weight'], g1, 1, 0)
    if down:
        gd = bn_bwd(P, 'downsample.1.', sd, g, G)
        gxd, G['downsample.0.weight'], _ = conv_bwd(x, P['downsample.0.weight'], gd, stride, 0)
        gx = gx + gxd
    else:
        gx = gx + g
    return y, gx, G, R


def residual(P, x, gy):
    P, x, gy = _f8(P), np.asarray(x, np.float64), np.asarray(gy, np.float64)
    R, G = {}, {}
    n0, s0 = bn_fwd(P, 'bn1.', x, R)
    a0 = np.maximum(n0, 0)
    h1 = conv_fwd(a0, P['conv1.conv.weight'], P['conv1.conv.bias'], 1, 0)
    n1, s1 = bn_fwd(P, 'bn2.', h1, R)
    a1 = np.maximum(n1, 0)
    h2 = conv_fwd(a1, P['conv2.conv.weight'], P['conv2.conv.bias'], 1, 1)
    n2, s2 = bn_fwd(P, 'bn3.', h2, R)
    a2 = np.maximum(n2, 0)
    y = conv_fwd(a2, P['conv3.conv.weight'], P['conv3.conv.bias'], 1, 0)
    skip = P['skip_layer.conv.weight'].shape[0] != P['skip_layer.conv.weight'].shape[1]
    y = y + (conv_fwd(x, P['skip_layer.conv.weight'], P['skip_layer.conv.bias'], 1, 0) if skip else x)
    g, G['conv3.conv.weight'], G['conv3.conv.bias'] = conv_bwd(a2, P['conv3.conv.weight'], gy, 1, 0)
    g = bn_bwd(P, 'bn3.', s2, g * (a2 > 0), G)
    g, G['conv2.conv.weight'], G['conv2.conv.bias'] = conv_bwd(a1, P['conv2.conv.weight'], g, 1, 1)
    g = bn_bwd(P, 'bn2.', s1, g * (a1 > 0), G)
    g, G['conv1.conv.weight'], G['conv1.conv.bias'] = conv_bwd(a0, P['conv1.conv.weight'], g, 1, 0)
    gx = bn_bwd(P, 'bn1.', s0, g * (a0 > 0), G)
    if skip:
        gs, G['skip_layer.conv.weight'], G['skip_layer.conv.bias'] = conv_bwd(x, P['skip_layer.conv.weight'], gy, 1, 0)
        gx = gx + gs
    else:
        gx = gx + gy
    return y, gx, G, R
