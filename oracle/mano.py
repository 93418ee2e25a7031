"""ORACLE (test infrastructure): numpy restatement of the manopth ManoLayer forward in the one
configuration the network uses (root_rot_mode='6D', joint_rot_mode='axisang', use_pca=True, ncomps=45,
robust_rot=True; models/dir.py:221-224,315-318).  Follows manopth/manopth/manolayer.py:110-270 step
by step; helper citations inline.
"""
import numpy as np

LEV1, LEV2, LEV3 = [1, 4, 7, 10, 13], [2, 5, 8, 11, 14], [3, 6, 9, 12, 15]     # manolayer.py:196-198
REORDER_T = [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15]             # manolayer.py:228
REORDER_J = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # manolayer.py:259
TIPS = {'right': [745, 317, 444, 556, 673], 'left': [745, 317, 445, 556, 673]}  # manolayer.py:249-252


def normalize_vector(v):
    """rot6d.py:54-60: v / max(||v||, 1e-8)."""
    mag = np.sqrt((v * v).sum(1))
    mag = np.maximum(mag, v.dtype.type(1e-8))
    return v / mag[:, None]


def cross(u, v):
    """rot6d.py:63-71"""
    return np.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1],
                     u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2],
                     u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], 1)


def robust_rot6d(p6):
    """rot6d.robust_compute_rotation_matrix_from_ortho6d (rot6d.py:26-51); columns (x', y', z).
    The reference additionally asserts det >= 0 per sample (rot6d.py:50); reported here as a flag."""
    x = normalize_vector(p6[:, 0:3])
    y = normalize_vector(p6[:, 3:6])
    middle = normalize_vector(x + y)
    orthmid = normalize_vector(x - y)
    x = normalize_vector(middle + orthmid)
    y = normalize_vector(middle - orthmid)
    z = normalize_vector(cross(x, y))
    return np.stack([x, y, z], 2)


def quat2mat(q):
    """rodrigues_layer.py:15-40 (re-normalises the quaternion)."""
    q = q / np.sqrt((q * q).sum(1, keepdims=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                     2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                     2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1)


def batch_rodrigues(aa):
    """rodrigues_layer.py:43-54: angle = ||aa + 1e-8|| (eps added per component), axis = aa/angle,
    quaternion (cos(a/2), sin(a/2)*axis) -> quat2mat.  [N,3] -> [N,9]."""
    t = aa.dtype.type
    angle = np.sqrt(((aa + t(1e-8)) ** 2).sum(1))[:, None]
    axis = aa / angle
    half = angle * t(0.5)
    quat = np.concatenate([np.cos(half), np.sin(half) * axis], 1)
    return quat2mat(quat)


def with_zeros(m34):
    """tensutils.py:15-22: append the row (0,0,0,1)."""
    pad = np.zeros(m34.shape[:-2] + (1, 4), m34.dtype)
    pad[..., 0, 3] = 1
    return np.concatenate([m34, pad], -2)


def mano_forward(buf, pose, betas, side, center_idx=0, return_aux=False):
    """buf: dict of th_* buffers (float arrays, manolayer.py:71-98).  pose [B,51], betas [B,10].
    Returns verts [B,778,3], joints [B,21,3] in metres."""
    dt = pose.dtype
    g = lambda k: np.asarray(buf[k]).astype(dt)  # noqa: E731
    B = pose.shape[0]
    comps, mean = g('th_selected_comps'), g('th_hands_mean')
    full_hand = pose[:, 6:51] @ comps                                     # :136
    full_pose = np.concatenate([pose[:, :6], mean + full_hand], 1)         # :141-144
    rot_map = batch_rodrigues(full_pose[:, 6:].reshape(-1, 3)).reshape(B, 135)   # :153, tensutils.py:6-12
    pose_map = rot_map - np.tile(np.eye(3, dtype=dt).reshape(1, 9), (B, 15))      # tensutils.py:34-42
    root_rot = robust_rot6d(full_pose[:, :6])                              # :155
    shapedirs, posedirs, vt = g('th_shapedirs'), g('th_posedirs'), g('th_v_template')
    v_shaped = np.einsum('vck,bk->bvc', shapedirs, betas) + vt             # :180-182
    th_j = np.matmul(g('th_J_regressor'), v_shaped)                        # :183  [B,16,3]
    v_posed = v_shaped + np.einsum('vck,bk->bvc', posedirs, pose_map)      # :186-187
    root_j = th_j[:, 0].reshape(B, 3, 1)
    root_trans = with_zeros(np.concatenate([root_rot, root_j], 2))         # :192-193
    all_rots = rot_map.reshape(B, 15, 3, 3)
    l1r, l2r, l3r = (all_rots[:, [i - 1 for i in L]] for L in (LEV1, LEV2, LEV3))
    l1j, l2j, l3j = th_j[:, LEV1], th_j[:, LEV2], th_j[:, LEV3]
    rel1 = with_zeros(np.concatenate([l1r, (l1j - root_j.transpose(0, 2, 1))[..., None]], 3))
    lev1 = np.matmul(root_trans[:, None], rel1)                            # :210-214
    rel2 = with_zeros(np.concatenate([l2r, (l2j - l1j)[..., None]], 3))
    lev2 = np.matmul(lev1, rel2)                                           # :217-220
    rel3 = with_zeros(np.concatenate([l3r, (l3j - l2j)[..., None]], 3))
    lev3 = np.matmul(lev2, rel3)                                           # :223-226
    results = np.concatenate([root_trans[:, None], lev1, lev2, lev3], 1)[:, REORDER_T]   # :228-229
    joint_js = np.concatenate([th_j, np.zeros((B, 16, 1), dt)], 2)
    tmp2 = np.matmul(results, joint_js[..., None])                         # [B,16,4,1]
    results2 = results - np.concatenate([np.zeros((B, 16, 4, 3), dt), tmp2], 3)     # :232-234
    results2 = results2.transpose(0, 2, 3, 1)                              # [B,4,4,16]
    T = np.matmul(results2, g('th_weights').T)                             # :236  [B,4,4,778]
    rest_h = np.concatenate([v_posed.transpose(0, 2, 1), np.ones((B, 1, 778), dt)], 1)   # [B,4,778]
    verts = (T * rest_h[:, None]).sum(2).transpose(0, 2, 1)[:, :, :3]      # :245-246
    jtr = results[:, :, :3, 3]                                             # :247
    jtr = np.concatenate([jtr, verts[:, TIPS[side]]], 1)[:, REORDER_J]     # :249-259
    if center_idx is not None:                                             # :261-265
        c = jtr[:, center_idx][:, None]
        jtr = jtr - c
        verts = verts - c
    if return_aux:
        return verts, jtr, dict(root_rot=root_rot, rot_map=rot_map, th_j=th_j, results=results)
    return verts, jtr


def projection_batch_xy(scale, trans2d, xyz):
    """utils/utils.py:47-63: uv = s * xyz[..., :2] + t."""
    return scale.reshape(-1, 1, 1) * xyz[..., :2] + trans2d[:, None, :]
