"""ORACLE (test infrastructure): analytic float64 backward of Joint2BoneFeature.bone_proj (models/dir.py:146-174) w.r.t. the joint uv and the
re-embedded joint features.  The capsule mask comes from the float32 forward restatement (oracle/tokens.py::bone_proj: the mask is a
comparison, it carries no gradient and must be the reference's, bit for bit).  Pinned against torch autograd through the reference's own
method (tests/golden/g19_bone_grad.npz)."""
import numpy as np

from .tokens import CHILD, PARENT, bone_proj


def bone_proj_backward(joint_uv, joint_feat, g_img, S, distance):
    """g_img [B, 20*C, S, S] -> (g joint_uv [B,21,2], g joint_feat [B,21,C]) in float64"""
    _, mask = bone_proj(np.asarray(joint_uv, np.float32), np.asarray(joint_feat, np.float32), S, distance, return_mask=True)     # [B,S,S,20]
    uv, feat = np.asarray(joint_uv, np.float64), np.asarray(joint_feat, np.float64)
    B, J, C = feat.shape
    g = np.asarray(g_img, np.float64).reshape(B, 20, C, S, S).transpose(0, 3, 4, 1, 2)         # [B,y,x,bone,C]
    g = g * mask[..., None]
    c = np.arange(S) + 0.5
    py, px = np.meshgrid(c, c, indexing='ij')                                                  # pixel (x + .5, y + .5): grid = (gy, gx) of dir.py:66-70
    P = np.stack((px, py), -1)[None, :, :, None, :]                                            # [1,y,x,1,2]
    U = (uv + 1) / 2 * S
    A, Bp = U[:, PARENT][:, None, None], U[:, CHILD][:, None, None]                            # [B,1,1,20,2]
    ea, eb = P - A + 1e-6, P - Bp + 1e-6
    da, db = np.sqrt((ea ** 2).sum(-1)), np.sqrt((eb ** 2).sum(-1))                             # [B,y,x,20]
    sm = da + db
    wa, wb = 1 - da / sm, 1 - db / sm
    fa, fb = feat[:, PARENT][:, None, None], feat[:, CHILD][:, None, None]                     # [B,1,1,20,C]
    g_feat = np.zeros((B, J, C))
    gfa, gfb = (g * wa[..., None]).sum((1, 2)), (g * wb[..., None]).sum((1, 2))                # [B,20,C]
    np.add.at(g_feat, (slice(None), PARENT), gfa)
    np.add.at(g_feat, (slice(None), CHILD), gfb)
    gwa, gwb = (g * fa).sum(-1), (g * fb).sum(-1)                                              # [B,y,x,20]
    gda, gdb = (gwb - gwa) * db / sm ** 2, (gwa - gwb) * da / sm ** 2
    gA = -(gda / da)[..., None] * ea * mask[..., None]
    gB = -(gdb / db)[..., None] * eb * mask[..., None]
    g_uv = np.zeros((B, J, 2))
    np.add.at(g_uv, (slice(None), PARENT), gA.sum((1, 2)))
    np.add.at(g_uv, (slice(None), CHILD), gB.sum((1, 2)))
    return g_uv * S / 2, g_feat
