"""TEST INFRASTRUCTURE (CPU oracle, numpy) -- never imported by the product path.

Restatement of the reference's second MANO layer, models/manolayer.py: the formulation the dataset
uses to synthesise ground truth (dataset/interhand.py:130-149) -- rotation-matrix root, PCA pose through classic
Rodrigues, serial 16-joint SE(3) chain.
  rodrigues_batch   models/manolayer.py:32-48
  gt_mano_forward   models/manolayer.py:251-323 (ManoLayer.forward; pca2axis / axis2Rmat :161-174)
Pinned by tests/golden/g10_gtmano.npz, produced by oracle/gen_golden.py::gen_gtmano from the reference class itself
(constructed from a synthetic pickle, the licensed MANO pkl being absent).
"""
import numpy as np

NEW_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)      # models/manolayer.py:111-116
TIP_VERTS = (745, 317, 444, 556, 673)                                                      # models/manolayer.py:300
PARENT = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)                              # kintree_table[0]


def rodrigues_batch(axis):
    """models/manolayer.py:32-48: [n,3] -> [n,3,3]"""
    dt = axis.dtype
    n = axis.shape[0]
    angle = np.sqrt((axis * axis).sum(1, keepdims=True, dtype=dt)) + dt.type(1e-8)
    axes = axis / angle
    sin, cos = np.sin(angle)[:, :, None], np.cos(angle)[:, :, None]
    L = np.zeros((n, 3, 3), dt)
    L[:, 2, 1] = axes[:, 0]; L[:, 1, 2] = -axes[:, 0]
    L[:, 0, 2] = axes[:, 1]; L[:, 2, 0] = -axes[:, 1]
    L[:, 1, 0] = axes[:, 2]; L[:, 0, 1] = -axes[:, 2]
    return np.eye(3, dtype=dt)[None] + sin * L + (1 - cos) * np.matmul(L, L)


def _se3(R, t):
    B = R.shape[0]
    M = np.zeros((B, 4, 4), R.dtype)
    M[:, :3, :3], M[:, :3, 3], M[:, 3, 3] = R, t, 1
    return M


def gt_mano_forward(T, root_rotation, pose, shape, trans=None, scale=None, center_idx=None, use_pca=True, new_skel=False):
    """T: dict with hands_components [45,45], hands_mean [45], J_regressor [16,778], weights [778,16], posedirs [778,3,135],
    v_template [778,3], shapedirs [778,3,10] (the buffers of models/manolayer.py:118-147).  Returns verts [B,778,3], joints [B,21,3]."""
    dt = root_rotation.dtype
    B = root_rotation.shape[0]
    if use_pca:                                                            # :161-174
        axis = pose @ T['hands_components'][:pose.shape[1]].astype(dt) + T['hands_mean'].astype(dt)
        rot = rodrigues_batch(axis.reshape(-1, 3)).reshape(B, 15, 3, 3)
    else:
        rot = pose
    v_shaped = T['v_template'].astype(dt) + np.matmul(T['shapedirs'].astype(dt), shape.T).transpose(2, 0, 1)      # :265-266
    j_tpose = np.matmul(T['J_regressor'].astype(dt), v_shaped)                                                   # :268
    pose_shape = rot.reshape(B, -1) - np.tile(np.eye(3, dtype=dt).reshape(-1), 15)[None]                          # :270-271
    v_tpose = v_shaped + np.matmul(T['posedirs'].astype(dt), pose_shape.T).transpose(2, 0, 1)                     # :272-273
    eye = np.eye(3, dtype=dt)[None]
    R0 = root_rotation.reshape(B, 3, 3)
    se3 = [_se3(R0, np.matmul(eye - R0, j_tpose[:, 0, :, None])[:, :, 0])]                                        # :275-278
    for i in range(1, 16):                                                                                       # :279-283
        R = rot[:, i - 1]
        local = _se3(R, np.matmul(eye - R, j_tpose[:, i, :, None])[:, :, 0])
        se3.append(np.matmul(se3[PARENT[i]], local))
    se3 = np.stack(se3, 1)                                                                                       # [B,16,4,4]
    jl = [j_tpose[:, 0]]                                                                                         # :286-289
    for i in range(1, 16):
        M = se3[:, PARENT[i]]
        jl.append(np.matmul(M[:, :3, :3], j_tpose[:, i, :, None])[:, :, 0] + M[:, :3, 3])
    se3_v = np.matmul(T['weights'].astype(dt), se3.reshape(B, 16, 16)).reshape(B, 778, 4, 4)                      # :292
    v_out = np.matmul(se3_v[:, :, :3, :3], v_tpose[..., None])[..., 0] + se3_v[:, :, :3, 3]                      # :294-295
    jl += [v_out[:, t] for t in TIP_VERTS]                                                                       # :297
    j_out = np.stack(jl, 1)[:, list(NEW_ORDER)]                                                                  # :299-300
    if center_idx is not None:                                                                                   # :302-305
        c = j_out[:, center_idx:center_idx + 1]
        v_out, j_out = v_out - c, j_out - c
    if scale is not None:                                                                                        # :307-310
        v_out, j_out = v_out * scale.reshape(B, 1, 1), j_out * scale.reshape(B, 1, 1)
    if trans is not None:                                                                                        # :312-315
        v_out, j_out = v_out + trans.reshape(B, 1, 3), j_out + trans.reshape(B, 1, 3)
    if new_skel:                                                                                                 # :317-321
        j_out = j_out.copy()
        j_out[:, 5] = (v_out[:, 63] + v_out[:, 144]) / 2
        j_out[:, 9] = (v_out[:, 271] + v_out[:, 220]) / 2
        j_out[:, 13] = (v_out[:, 148] + v_out[:, 290]) / 2
        j_out[:, 17] = (v_out[:, 770] + v_out[:, 83]) / 2
    return v_out.astype(dt), j_out.astype(dt)


def tables(side, seed=1234):
    """the synthetic MANO tables in the layout of the buffers models/manolayer.py registers (float32)"""
    from dir_amd import synth
    t = synth.synthetic_mano_tables(side, seed)
    f = np.float32
    return {'hands_components': t['hands_components'].astype(f), 'hands_mean': t['hands_mean'].astype(f),
            'J_regressor': t['J_regressor'].astype(f), 'weights': t['weights'].astype(f), 'posedirs': t['posedirs'].astype(f),
            'v_template': t['v_template'].astype(f), 'shapedirs': t['shapedirs'].astype(f)}
