"""ORACLE (test infrastructure): STE (transformer/mixSTE.py:194-205) forward + analytic backward in numpy (float64 by default), operator by
operator as torch autograd differentiates the reference modules: nn.LayerNorm, nn.Linear, softmax attention (:76-97), exact-erf GELU.
Pinned against torch autograd through the reference's own STE (tests/golden/g15_ste_grad.npz, oracle/gen_golden.py::gen_ste_grad)."""
import numpy as np
from scipy.special import erf

HEADS = 4


def _ln_fwd(x, w, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rs = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * rs
    return xh * w + b, (xh, rs)


def _ln_bwd(gy, w, st):
    xh, rs = st
    gh = gy * w
    gx = rs * (gh - gh.mean(-1, keepdims=True) - xh * (gh * xh).mean(-1, keepdims=True))
    return gx, (gy * xh).reshape(-1, xh.shape[-1]).sum(0), gy.reshape(-1, xh.shape[-1]).sum(0)


def _lin_bwd(gy, x, W):
    g2, x2 = gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1])
    return gy @ W, g2.T @ x2, g2.sum(0)


def ste_forward_backward(P, x, gy, depth=4, dtype=np.float64):
    """P: {key: array}; x [B,T,C]; gy [B,T,out] -> (y, g x, {key: gradient})"""
    f = lambda k: np.asarray(P[k], dtype)  # noqa: E731
    x = np.asarray(x, dtype) + f('spatial_pos_embed')
    gy = np.asarray(gy, dtype)
    B, T, C = x.shape
    D = C // HEADS
    scale = D ** -0.5
    saved = []
    for i in range(1, depth):
        p = 'STEblocks.%d.' % i
        a, st1 = _ln_fwd(x, f(p + 'norm1.weight'), f(p + 'norm1.bias'), 1e-6)
        qkv = a @ f(p + 'attn.qkv.weight').T + f(p + 'attn.qkv.bias')
        q, k, v = (qkv.reshape(B, T, 3, HEADS, D).transpose(2, 0, 3, 1, 4)[j] for j in range(3))        # [B,H,T,D]
        S = q @ k.transpose(0, 1, 3, 2) * scale
        Pm = np.exp(S - S.max(-1, keepdims=True))
        Pm = Pm / Pm.sum(-1, keepdims=True)
        o = (Pm @ v).transpose(0, 2, 1, 3).reshape(B, T, C)
        x1 = x + o @ f(p + 'attn.proj.weight').T + f(p + 'attn.proj.bias')
        m, st2 = _ln_fwd(x1, f(p + 'norm2.weight'), f(p + 'norm2.bias'), 1e-6)
        h = m @ f(p + 'mlp.fc1.weight').T + f(p + 'mlp.fc1.bias')
        g = 0.5 * h * (1 + erf(h / np.sqrt(2.0)))
        x2 = x1 + g @ f(p + 'mlp.fc2.weight').T + f(p + 'mlp.fc2.bias')
        x3, st3 = _ln_fwd(x2, f('spatial_norm.weight'), f('spatial_norm.bias'), 1e-6)
        saved.append(dict(a=a, st1=st1, q=q, k=k, v=v, Pm=Pm, o=o, m=m, st2=st2, h=h, g=g, st3=st3))
        x = x3
    hn, sth = _ln_fwd(x, f('head.0.weight'), f('head.0.bias'), 1e-5)
    y = hn @ f('head.1.weight').T + f('head.1.bias')
    G = {}
    ghn, G['head.1.weight'], G['head.1.bias'] = _lin_bwd(gy, hn, f('head.1.weight'))
    gx, G['head.0.weight'], G['head.0.bias'] = _ln_bwd(ghn, f('head.0.weight'), sth)
    G['spatial_norm.weight'], G['spatial_norm.bias'] = 0.0, 0.0
    for i in range(depth - 1, 0, -1):
        p, s = 'STEblocks.%d.' % i, saved[i - 1]
        gx2, gw, gb = _ln_bwd(gx, f('spatial_norm.weight'), s['st3'])
        G['spatial_norm.weight'] = G['spatial_norm.weight'] + gw
        G['spatial_norm.bias'] = G['spatial_norm.bias'] + gb
        gg, G[p + 'mlp.fc2.weight'], G[p + 'mlp.fc2.bias'] = _lin_bwd(gx2, s['g'], f(p + 'mlp.fc2.weight'))
        h = s['h']
        gh = gg * (0.5 * (1 + erf(h / np.sqrt(2.0))) + h * np.exp(-0.5 * h * h) / np.sqrt(2 * np.pi))
        gm, G[p + 'mlp.fc1.weight'], G[p + 'mlp.fc1.bias'] = _lin_bwd(gh, s['m'], f(p + 'mlp.fc1.weight'))
        g1, G[p + 'norm2.weight'], G[p + 'norm2.bias'] = _ln_bwd(gm, f(p + 'norm2.weight'), s['st2'])
        gx1 = gx2 + g1
        go, G[p + 'attn.proj.weight'], G[p + 'attn.proj.bias'] = _lin_bwd(gx1, s['o'], f(p + 'attn.proj.weight'))
        go = go.reshape(B, T, HEADS, D).transpose(0, 2, 1, 3)                                            # [B,H,T,D]
        gv = s['Pm'].transpose(0, 1, 3, 2) @ go
        gP = go @ s['v'].transpose(0, 1, 3, 2)
        gS = s['Pm'] * (gP - (gP * s['Pm']).sum(-1, keepdims=True)) * scale
        gq, gk = gS @ s['k'], gS.transpose(0, 1, 3, 2) @ s['q']
        gqkv = np.stack([gq, gk, gv]).transpose(1, 3, 0, 2, 4).reshape(B, T, 3 * C)                      # [3,B,H,T,D] -> [B,T,3,H,D]
        ga, G[p + 'attn.qkv.weight'], G[p + 'attn.qkv.bias'] = _lin_bwd(gqkv, s['a'], f(p + 'attn.qkv.weight'))
        g0, G[p + 'norm1.weight'], G[p + 'norm1.bias'] = _ln_bwd(ga, f(p + 'norm1.weight'), s['st1'])
        gx = gx1 + g0
    G['spatial_pos_embed'] = gx.sum(0, keepdims=True)
    return y, gx, G
