"""ORACLE (test infrastructure, never shipped or measured as product): numpy restatement of the
ATen operators the DIR hot path uses, with the exact semantics the reference relies on implicitly
(SURVEY.md 3.6).  Tensors are NCHW / [B,C,L] like the reference; dtype follows the input (float32
mirrors the reference, float64 gives an arbitration-grade answer).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import numpy as np
from scipy.special import erf as _erf


class Params(object):
    """state-dict view with a key prefix (mirrors nn.Module nesting)."""

    def __init__(self, sd, prefix='', dtype=None):
        self.sd, self.prefix, self.dtype = sd, prefix, dtype

    def sub(self, name):
        return Params(self.sd, self.prefix + name + '.', self.dtype)

    def __contains__(self, k):
        return (self.prefix + k) in self.sd

    def __getitem__(self, k):
        v = np.asarray(self.sd[self.prefix + k])
        if self.dtype is not None and v.dtype.kind == 'f':
            v = v.astype(self.dtype, copy=False)
        return v


def conv2d(x, w, b=None, stride=1, pad=0):
    """nn.Conv2d forward (cross-correlation), zero padding.  Accumulates tap by tap as
    W[:,:,ky,kx] @ x_shifted so no im2col buffer is materialised."""
    B, C, H, W = x.shape
    Co, Ci, kh, kw = w.shape
    assert Ci == C
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    out = np.zeros((B, Co, Ho * Wo), x.dtype)
    wt = np.ascontiguousarray(w.transpose(2, 3, 0, 1))          # [kh,kw,Co,Ci]: contiguous taps -> BLAS
    for ky in range(kh):
        for kx in range(kw):
            xs = xp[:, :, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride]
            out += np.matmul(wt[ky, kx], np.ascontiguousarray(xs).reshape(B, C, Ho * Wo))
    out = out.reshape(B, Co, Ho, Wo)
    if b is not None:
        out = out + b.reshape(1, -1, 1, 1)
    return out


def conv1d_k1(x, w, b):
    """nn.Conv1d(kernel_size=1) on [B,C,L]; w [Co,Ci,1]."""
    return np.matmul(w[:, :, 0], x) + b.reshape(1, -1, 1)


def linear(x, w, b=None):
    y = np.matmul(x, w.T)
    return y if b is None else y + b


def batchnorm(x, P, eps=1e-5):
    """Eval-mode BatchNorm{1,2}d on [B,C,...]: y = x*alpha + beta with alpha = w/sqrt(var+eps),
    beta = b - mean*alpha (the form ATen's CPU kernel evaluates)."""
    alpha = P['weight'] / np.sqrt(P['running_var'] + x.dtype.type(eps))
    beta = P['bias'] - P['running_mean'] * alpha
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return x * alpha.reshape(shp).astype(x.dtype) + beta.reshape(shp).astype(x.dtype)


def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    with np.errstate(over='ignore'):
        return 1 / (1 + np.exp(-x))


def gelu(x):
    """exact erf form (nn.GELU() default; transformer/mixSTE.py:12,27)."""
    return (0.5 * x * (1 + _erf(x / np.sqrt(2.0)))).astype(x.dtype)


def layernorm(x, w, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * w + b


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def maxpool3x3s2p1(x):
    B, C, H, W = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    out = np.full((B, C, Ho, Wo), -np.inf, x.dtype)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, :, ky:ky + 2 * (Ho - 1) + 1:2, kx:kx + 2 * (Wo - 1) + 1:2])
    return out


def upsample_bilinear2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear') => align_corners=False (models/dir.py:392,398):
    src = (dst+0.5)/2 - 0.5 clamped at 0; neighbour index clamped at the border."""
    B, C, H, W = x.shape
    t = x.dtype.type

    def idx(n):
        src = np.maximum((np.arange(2 * n, dtype=x.dtype) + t(0.5)) * t(0.5) - t(0.5), 0)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n - 1)
        lam = (src - i0).astype(x.dtype)
        return i0, i1, lam
    y0, y1, ly = idx(H)
    x0, x1, lx = idx(W)
    ly = ly.reshape(1, 1, -1, 1)
    lx = lx.reshape(1, 1, 1, -1)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def grid_sample_points(feat, uv):
    """F.grid_sample(feat[B,C,H,W], uv[B,1,N,2]) -> [B,C,N]: bilinear, padding_mode='zeros',
    align_corners=False (the defaults used at models/dir.py:198).  uv[...,0] indexes W, uv[...,1] H."""
    B, C, H, W = feat.shape
    t = feat.dtype.type
    u = uv[..., 0].astype(feat.dtype)
    v = uv[..., 1].astype(feat.dtype)
    ix = ((u + 1) * t(W) - 1) / 2
    iy = ((v + 1) * t(H) - 1) / 2
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1, y1 = x0 + 1, y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    out = np.zeros((B, C, uv.shape[1]), feat.dtype)
    bidx = np.arange(B)[:, None]
    for xx, yy, ww in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = np.clip(xx, 0, W - 1).astype(np.int64)
        yi = np.clip(yy, 0, H - 1).astype(np.int64)
        val = feat[bidx, :, yi, xi]                      # [B,N,C]
        out += (val * (ww * ok)[..., None]).transpose(0, 2, 1).astype(feat.dtype)
    return out
