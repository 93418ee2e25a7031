"""ORACLE (test infrastructure, never shipped or measured as product): gradients by DEFINITION -- central differences in float64
on the numpy restatements of the forward operators.  This is the CPU reference of the backward kernels: it
shares no derivation with them, and it is itself pinned against torch autograd through the reference's own modules
(tests/golden/g13_*.npz, oracle/gen_golden.py::gen_mano_grad).

    numeric_vjp(f, x, cot)      d/dx sum_i <cot_i, f(x)_i>      f: batched, sample b of the outputs depends on row b of x only
"""
import numpy as np

from . import mano as OM


def numeric_vjp(f, x, cots, h=1e-6):
    """x [B,P] float64; f(x) -> tuple of arrays with leading dimension B; cots: matching tuple (None = no contribution).
    Every sample is independent, so all 2P perturbations of all B samples are evaluated in ONE batched call."""
    x = np.asarray(x, np.float64)
    B, P = x.shape
    xs = np.repeat(x[:, None, :], 2 * P, 1)                 # [B, 2P, P]
    idx = np.arange(P)
    xs[:, idx, idx] += h
    xs[:, P + idx, idx] -= h
    outs = f(xs.reshape(B * 2 * P, P))
    val = np.zeros(B * 2 * P)
    for o, c in zip(outs, cots):
        if c is None:
            continue
        c = np.asarray(c, np.float64)
        o = o.reshape((B, 2 * P) + c.shape[1:])
        val += (o * c[:, None]).reshape(B * 2 * P, -1).sum(1)
    val = val.reshape(B, 2 * P)
    return (val[:, :P] - val[:, P:]) / (2 * h)


def mano_outputs(buf, para, side, center_idx):
    """the four tensors a stage derives from a hand's 64-vector (models/dir.py:352-363): pose 51 | betas 10 | cam (s, tx, ty)"""
    verts, joints = OM.mano_forward(buf, para[:, :51], para[:, 51:61], side, center_idx)
    s, t = para[:, 61], para[:, 62:64]
    return verts, joints, OM.projection_batch_xy(s, t, joints), OM.projection_batch_xy(s, t, verts)


def mano_vjp(buf, para, side, center_idx, g_verts=None, g_joints=None, g_joint_uv=None, g_mesh_uv=None):
    """gradient of <g_verts, verts> + <g_joints, joints> + <g_joint_uv, joint_uv> + <g_mesh_uv, mesh_uv> w.r.t. the 64-vector"""
    buf64 = {k: np.asarray(v, np.float64) for k, v in buf.items() if getattr(np.asarray(v), 'dtype', None) is not None and np.asarray(v).dtype.kind == 'f'}
    return numeric_vjp(lambda p: mano_outputs(buf64, p, side, center_idx), para, (g_verts, g_joints, g_joint_uv, g_mesh_uv))


def regressor_vjp(P, mano_l, mano_r, feat_l, feat_r, para_l, para_r, offset, cot, root_joint=0):
    """RegressorOffset (models/dir.py:339-381) backward: cot = cotangents of pd_offset and of the MANO outputs per hand
    (pd_joint_uv_*, pd_mesh_uv_*, pd_joint_xyz_*, pd_mesh_xyz_*; missing keys = no contribution).  P: Params view with
    mano_left / mano_right / offset (weight, bias).  The previous stage's mano_para / offset are detached inputs (:344-345).
    Returns {'feat_l', 'feat_r', 'mano_left.weight', ..., 'offset.bias'} in float64; the MANO part by central differences."""
    f8 = lambda a: np.asarray(a, np.float64)  # noqa: E731
    B = feat_l.shape[0]
    fl, fr = f8(feat_l).reshape(B, -1), f8(feat_r).reshape(B, -1)
    gl, gr = np.concatenate([fl, f8(para_l)], 1), np.concatenate([fr, f8(para_r)], 1)
    gf = np.concatenate([fl, fr, f8(offset).reshape(B, 3)], 1)
    Wl, Wr, Wo = f8(P['mano_left.weight']), f8(P['mano_right.weight']), f8(P['offset.weight'])
    pl = gl @ Wl.T + f8(P['mano_left.bias'])
    pr = gr @ Wr.T + f8(P['mano_right.bias'])
    g_para = {}
    for side, p, buf in (('left', pl, mano_l), ('right', pr, mano_r)):
        g_para[side] = mano_vjp(buf, p, side, root_joint, cot.get('pd_mesh_xyz_' + side), cot.get('pd_joint_xyz_' + side),
                                cot.get('pd_joint_uv_' + side), cot.get('pd_mesh_uv_' + side))
        if cot.get('pd_mano_para_' + side) is not None:                 # the 64-vector is itself an output (the next stage's input, :366-367)
            g_para[side] = g_para[side] + f8(cot['pd_mano_para_' + side])
    g_off = f8(cot['pd_offset']) if cot.get('pd_offset') is not None else np.zeros((B, 3))
    n = fl.shape[1]
    return {'feat_l': (g_para['left'] @ Wl[:, :n] + g_off @ Wo[:, :n]).reshape(feat_l.shape),
            'feat_r': (g_para['right'] @ Wr[:, :n] + g_off @ Wo[:, n:2 * n]).reshape(feat_r.shape),
            'mano_left.weight': g_para['left'].T @ gl, 'mano_left.bias': g_para['left'].sum(0),
            'mano_right.weight': g_para['right'].T @ gr, 'mano_right.bias': g_para['right'].sum(0),
            'offset.weight': g_off.T @ gf, 'offset.bias': g_off.sum(0),
            'g_para_left': g_para['left'], 'g_para_right': g_para['right']}
