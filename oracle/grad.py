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
