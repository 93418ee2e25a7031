"""ORACLE (test infrastructure): ResSimplePGCN (SemGCN/p_gcn.py:64-73) in TRAINING mode -- batch-statistics BatchNorm1d -- forward and
analytic backward in numpy float64, as torch autograd differentiates SemGCN/p_graph_conv.py:39-59 + nn.BatchNorm1d + ReLU.  Pinned against
torch autograd through the reference's own modules (tests/golden/g16_pgcn_grad.npz)."""
import numpy as np

from .tokens import adjacency_mask


def _softmax_rows(e1, mask):
    A = np.full(mask.shape, -np.inf)
    A[mask] = e1.reshape(-1)
    A = np.exp(A - A.max(1, keepdims=True))
    return A / A.sum(1, keepdims=True)


def pgcn_train_forward_backward(P, x, gy, num_layers=4, eps=1e-5, dtype=np.float64):
    """P: {key: array}; x, gy [B,21,128] -> (y, g x, {key: gradient}, {running stats after the step})"""
    f = lambda k: np.asarray(P[k], dtype)  # noqa: E731
    mask = adjacency_mask(21)
    x = np.asarray(x, dtype)
    B = x.shape[0]
    saved, running = [], {}
    for l in range(num_layers):
        p = 'gconv_layers.%d.' % l
        W, A1 = f(p + 'gconv.W'), _softmax_rows(f(p + 'gconv.e_1'), mask)
        h0, h1 = np.einsum('bjc,jcd->bjd', x, W[0]), np.einsum('bjc,jcd->bjd', x, W[1])
        z = h0 + np.einsum('jk,bkd->bjd', A1, h1) + f(p + 'gconv.bias')
        mu, var = z.mean((0, 1)), z.var((0, 1))
        rs = 1.0 / np.sqrt(var + eps)
        zh = (z - mu) * rs
        y = np.maximum(zh * f(p + 'bn.weight') + f(p + 'bn.bias'), 0)
        n = B * 21
        running[p + 'bn.running_mean'] = 0.9 * f(p + 'bn.running_mean') + 0.1 * mu
        running[p + 'bn.running_var'] = 0.9 * f(p + 'bn.running_var') + 0.1 * var * n / (n - 1)
        saved.append(dict(x=x, h1=h1, A1=A1, zh=zh, rs=rs, y=y, W=W))
        x = y
    out = x
    g = np.asarray(gy, dtype)
    G = {}
    for l in range(num_layers - 1, -1, -1):
        p, s = 'gconv_layers.%d.' % l, saved[l]
        g = g * (s['y'] > 0)
        G[p + 'bn.weight'], G[p + 'bn.bias'] = (g * s['zh']).sum((0, 1)), g.sum((0, 1))
        gh = g * f(p + 'bn.weight')
        gz = s['rs'] * (gh - gh.mean((0, 1)) - s['zh'] * (gh * s['zh']).mean((0, 1)))
        G[p + 'gconv.bias'] = gz.sum((0, 1))
        gA = np.einsum('bjd,bkd->jk', gz, s['h1'])
        gE = s['A1'] * (gA - (gA * s['A1']).sum(1, keepdims=True))
        G[p + 'gconv.e_1'] = gE[mask].reshape(1, -1)
        G[p + 'gconv.e_0'] = np.zeros((1, 21))
        gh1 = np.einsum('jk,bjd->bkd', s['A1'], gz)
        G[p + 'gconv.W'] = np.stack([np.einsum('bjc,bjd->jcd', s['x'], gz), np.einsum('bjc,bjd->jcd', s['x'], gh1)])
        g = np.einsum('bjd,jcd->bjc', gz, s['W'][0]) + np.einsum('bjd,jcd->bjc', gh1, s['W'][1])
    return out, g, G, running
