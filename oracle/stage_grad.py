"""ORACLE (test infrastructure): the token half of one refinement stage (Joint2BoneFeature.forward, models/dir.py:86-116) in TRAINING mode
-- batch-statistics BatchNorm in the token MLPs and the P-GCNs -- forward and backward in numpy float64: the analytic chain rule of
every module (this file: grid_sample, Conv1d-BN-ReLU-Conv1d; oracle/pgcn_grad.py; oracle/ste_grad.py), and RegressorOffset + MANO through
oracle/grad.py (central differences on the float64 MANO forward).  Pinned against torch autograd through the reference's own
Joint2BoneFeature (tests/golden/g17_stage_grad.npz, oracle/gen_golden.py::gen_stage_grad)."""
import numpy as np

from . import grad as OG
from .pgcn_grad import pgcn_train_forward_backward
from .ste_grad import ste_forward_backward

SIDES = ('left', 'right')


def _sub(P, pre):
    n = len(pre)
    return {k[n:]: np.asarray(v, np.float64) for k, v in P.items() if k.startswith(pre)}


def _taps(uv, S):
    """bilinear taps of F.grid_sample (zeros padding, align_corners False): -> list of (iy, ix, w) arrays [B,21]"""
    fx, fy = ((uv[..., 0] + 1) * S - 1) / 2, ((uv[..., 1] + 1) * S - 1) / 2
    x0, y0 = np.floor(fx), np.floor(fy)
    out = []
    for xx, yy, w in ((x0, y0, (x0 + 1 - fx) * (y0 + 1 - fy)), (x0 + 1, y0, (fx - x0) * (y0 + 1 - fy)),
                      (x0, y0 + 1, (x0 + 1 - fx) * (fy - y0)), (x0 + 1, y0 + 1, (fx - x0) * (fy - y0))):
        ok = (xx >= 0) & (xx <= S - 1) & (yy >= 0) & (yy <= S - 1)
        out.append((np.clip(yy, 0, S - 1).astype(int), np.clip(xx, 0, S - 1).astype(int), w * ok))
    return out


def mlp_train(P, x, momentum=0.1, eps=1e-5):
    """Sequential(Conv1d k=1, BatchNorm1d (batch statistics), ReLU, Conv1d k=1) on rows [R,Cin] -> (y, saved, running stats after)"""
    W0, W3 = P['0.weight'].reshape(P['0.weight'].shape[0], -1), P['3.weight'].reshape(P['3.weight'].shape[0], -1)
    h = x @ W0.T + P['0.bias']
    mu, var = h.mean(0), h.var(0)
    rs = 1 / np.sqrt(var + eps)
    hh = (h - mu) * rs
    a = np.maximum(hh * P['1.weight'] + P['1.bias'], 0)
    n = x.shape[0]
    run = {'1.running_mean': (1 - momentum) * P['1.running_mean'] + momentum * mu,
           '1.running_var': (1 - momentum) * P['1.running_var'] + momentum * var * n / (n - 1)}
    return a @ W3.T + P['3.bias'], dict(x=x, hh=hh, rs=rs, a=a, W0=W0, W3=W3), run


def mlp_train_bwd(P, s, gy):
    G = {'3.weight': (gy.T @ s['a']).reshape(P['3.weight'].shape), '3.bias': gy.sum(0)}
    ga = (gy @ s['W3']) * (s['a'] > 0)
    G['1.weight'], G['1.bias'] = (ga * s['hh']).sum(0), ga.sum(0)
    gh = ga * P['1.weight']
    gh = s['rs'] * (gh - gh.mean(0) - s['hh'] * (gh * s['hh']).mean(0))
    G['0.weight'], G['0.bias'] = (gh.T @ s['x']).reshape(P['0.weight'].shape), gh.sum(0)
    return gh @ s['W0'], G


def stage_token_grads(P, mano_l, mano_r, img_feat, xyz_l, xyz_r, uv_l, uv_r, para_l, para_r, offset, cot, root_joint=0):
    """P: {key relative to the projecter -> array}; img_feat NCHW; cot: cotangents of the regressor outputs (pd_offset, pd_mano_para_*,
    pd_mesh_xyz_*, pd_joint_xyz_*, pd_joint_uv_*, pd_mesh_uv_*) and 'joint_feat' (STE output).
    -> (ste_out [B,42,64], g img_feat NCHW, {key: gradient}, {running statistic after the step})"""
    f8 = lambda a: np.asarray(a, np.float64)  # noqa: E731
    feat = f8(img_feat)
    B, C, S, _ = feat.shape
    xyz, uv, off = [f8(xyz_l), f8(xyz_r)], [f8(uv_l), f8(uv_r)], f8(offset).reshape(B, 1, 3)
    running, hand, toks = {}, [], []
    glob_state = _sub(P, 'global_pos_emb.')
    for h, s in enumerate(SIDES):
        taps = _taps(uv[h], S)
        bi = np.arange(B)[:, None]
        rows = sum(feat[bi, :, iy, ix] * w[..., None] for iy, ix, w in taps).reshape(B * 21, C)
        Pi, Pp = _sub(P, 'img2joint_%s.filters.' % s), _sub(P, 'pos_emb_%s.' % s)
        yi, si, ri = mlp_train(Pi, rows)
        yp, sp, rp = mlp_train(Pp, (xyz[h] / 0.15).reshape(B * 21, 3))
        running.update({'img2joint_%s.filters.%s' % (s, k): v for k, v in ri.items()})
        running.update({'pos_emb_%s.%s' % (s, k): v for k, v in rp.items()})
        x_gcn = (yi + yp).reshape(B, 21, 128)
        Pg = _sub(P, 'gcn_%s.' % s)
        y_gcn, _, _, rg = pgcn_train_forward_backward(Pg, x_gcn, np.zeros_like(x_gcn))
        running.update({'gcn_%s.%s' % (s, k): v for k, v in rg.items()})
        gin = (xyz[h] / 0.15 + (off / 2 if s == 'right' else -off / 2)).reshape(B * 21, 3)
        yg, sg, rgl = mlp_train(glob_state, gin)                         # shared module: the running statistics chain left -> right
        glob_state = dict(glob_state, **rgl)
        toks.append(y_gcn + yg.reshape(B, 21, 128))
        hand.append(dict(taps=taps, Pi=Pi, si=si, Pp=Pp, sp=sp, Pg=Pg, x_gcn=x_gcn, sg=sg))
    running.update({'global_pos_emb.' + k: glob_state[k] for k in ('1.running_mean', '1.running_var')})
    cat = np.concatenate(toks, 1)
    Ps = _sub(P, 'interaction.')
    tok, _, _ = ste_forward_backward(Ps, cat, np.zeros((B, 42, 64)))
    Pr = _sub(P, 'regressor.')
    r = OG.regressor_vjp(Pr, mano_l, mano_r, tok[:, :21], tok[:, 21:], para_l, para_r, offset, cot, root_joint)
    G = {'regressor.' + k: r[k] for k in ('mano_left.weight', 'mano_left.bias', 'mano_right.weight', 'mano_right.bias', 'offset.weight', 'offset.bias')}
    g_tok = np.concatenate([r['feat_l'], r['feat_r']], 1)
    if cot.get('joint_feat') is not None:
        g_tok = g_tok + f8(cot['joint_feat'])
    _, g_cat, g_ste = ste_forward_backward(Ps, cat, g_tok)
    G.update({'interaction.' + k: v for k, v in g_ste.items()})
    g_feat = np.zeros_like(feat)
    Pglob = _sub(P, 'global_pos_emb.')
    for h, s in enumerate(SIDES):
        hc = hand[h]
        g_h = g_cat[:, 21 * h:21 * (h + 1)]
        _, gg = mlp_train_bwd(Pglob, hc['sg'], g_h.reshape(B * 21, 128))
        for k, v in gg.items():
            G['global_pos_emb.' + k] = G.get('global_pos_emb.' + k, 0) + v
        _, g_x, g_gcn, _ = pgcn_train_forward_backward(hc['Pg'], hc['x_gcn'], g_h)
        G.update({'gcn_%s.%s' % (s, k): v for k, v in g_gcn.items()})
        g_x = g_x.reshape(B * 21, 128)
        _, gp = mlp_train_bwd(hc['Pp'], hc['sp'], g_x)
        G.update({'pos_emb_%s.%s' % (s, k): v for k, v in gp.items()})
        g_rows, gi = mlp_train_bwd(hc['Pi'], hc['si'], g_x)
        G.update({'img2joint_%s.filters.%s' % (s, k): v for k, v in gi.items()})
        g_rows = g_rows.reshape(B, 21, C)
        for iy, ix, w in hc['taps']:
            for b in range(B):
                for j in range(21):
                    g_feat[b, :, iy[b, j], ix[b, j]] += w[b, j] * g_rows[b, j]
    return tok, g_feat, G, running
