"""ORACLE (test infrastructure): numpy restatement of the joint-token operators of one DIR refinement
stage: P-GCN (SemGCN/p_graph_conv.py:39-59, SemGCN/p_gcn.py:20-27,71-73), the STE transformer
(transformer/mixSTE.py:76-97,129-131,194-205), ImgFeature2JointFeature (models/dir.py:197-200),
bone_proj / lineseg_dists (models/dir.py:132-174), RegressorOffset (models/dir.py:339-381) and
Joint2BoneFeature.forward (models/dir.py:86-130).
"""
import numpy as np

from . import nnops as N
from .mano import mano_forward, projection_batch_xy

EDGES = [[0, 1], [1, 2], [2, 3], [3, 4], [0, 5], [5, 6], [6, 7], [7, 8], [0, 9], [9, 10], [10, 11], [11, 12],
         [0, 13], [13, 14], [14, 15], [15, 16], [0, 17], [17, 18], [18, 19], [19, 20]]   # SemGCN/utils.py:66-71
PARENT = [0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]           # models/dir.py:25
CHILD = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]          # models/dir.py:26


def adjacency_mask(J=21):
    """adj_mx_from_edges(..., eye=False) > 0 (SemGCN/utils.py:27-43): symmetric, no self loops."""
    m = np.zeros((J, J), bool)
    for a, b in EDGES:
        m[a, b] = m[b, a] = True
    return m


def edge_softmax(e1, mask):
    """A_1 = softmax over each row of (-9e15 everywhere, e_1 scattered in row-major nonzero order)
    (SemGCN/p_graph_conv.py:43-50)."""
    A = np.full(mask.shape, -9e15, e1.dtype)
    A[mask] = e1.reshape(-1)
    return N.softmax(A, axis=1)


def pgraphconv(x, P):
    """PGraphConv.forward.  A_0 = softmax of a diagonal-only mask == identity for any e_0
    (verified bit-exact), so out = x.W0[j] + A_1 (x.W1[j]) + b."""
    W = P['W']
    h0 = np.einsum('bjc,jcd->bjd', x, W[0])
    h1 = np.einsum('bjc,jcd->bjd', x, W[1])
    mask = adjacency_mask(x.shape[1])
    A0 = edge_softmax(P['e_0'], np.eye(x.shape[1], dtype=bool))      # == identity exactly
    A1 = edge_softmax(P['e_1'], mask)
    out = np.matmul(A0, h0) + np.matmul(A1, h1)
    return out + P['bias'].reshape(1, 1, -1)


def graphconv_layer(x, P):
    """_GraphConv.forward: gconv -> BN1d over channels -> ReLU (SemGCN/p_gcn.py:20-27)."""
    y = pgraphconv(x, P.sub('gconv')).transpose(0, 2, 1)
    y = N.batchnorm(y, P.sub('bn')).transpose(0, 2, 1)
    return N.relu(y)


def pgcn_stack(x, P, num_layers=4, collect=None):
    for i in range(num_layers):
        x = graphconv_layer(x, P.sub('gconv_layers.%d' % i))
        if collect is not None:
            collect.append(x)
    return x


def ste_attention(x, P, heads=4, collect=None):
    B, Nt, C = x.shape
    hd = C // heads
    qkv = N.linear(x, P['qkv.weight'], P['qkv.bias']).reshape(B, Nt, 3, heads, hd).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = np.matmul(q, k.transpose(0, 1, 3, 2)) * x.dtype.type(hd ** -0.5)
    attn = N.softmax(attn, axis=-1)
    if collect is not None:
        collect['probs'] = attn
    o = np.matmul(attn, v).transpose(0, 2, 1, 3).reshape(B, Nt, C)
    return N.linear(o, P['proj.weight'], P['proj.bias'])


def ste_block(x, P, collect=None):
    h = N.layernorm(x, P['norm1.weight'], P['norm1.bias'], 1e-6)
    x = x + ste_attention(h, P.sub('attn'), collect=collect)
    h = N.layernorm(x, P['norm2.weight'], P['norm2.bias'], 1e-6)
    h = N.gelu(N.linear(h, P['mlp.fc1.weight'], P['mlp.fc1.bias']))
    return x + N.linear(h, P['mlp.fc2.weight'], P['mlp.fc2.bias'])


def ste_forward(x, P, depth=4, collect=None):
    """STE.forward: block 0 is never executed; spatial_norm after every executed block
    (transformer/mixSTE.py:196-200)."""
    x = x + P['spatial_pos_embed']
    for i in range(1, depth):
        c = {} if (collect is not None and i == 1) else None
        x = ste_block(x, P.sub('STEblocks.%d' % i), collect=c)
        x = N.layernorm(x, P['spatial_norm.weight'], P['spatial_norm.bias'], 1e-6)
        if collect is not None:
            collect.setdefault('after_norm', []).append(x)
            if c:
                collect['probs'] = c['probs']
    h = N.layernorm(x, P['head.0.weight'], P['head.0.bias'], 1e-5)
    return N.linear(h, P['head.1.weight'], P['head.1.bias'])


def token_mlp(x_cl, P):
    """Conv1d(k=1) -> BN1d -> ReLU -> Conv1d(k=1) on [B,C,L] (models/dir.py:31-36,180-185)."""
    h = N.conv1d_k1(x_cl, P['0.weight'], P['0.bias'])
    h = N.relu(N.batchnorm(h, P.sub('1')))
    return N.conv1d_k1(h, P['3.weight'], P['3.bias'])


def img2joint(feat, uv, P):
    """ImgFeature2JointFeature.forward -> [B,21,128] (token-major view taken at models/dir.py:94)."""
    sampled = N.grid_sample_points(feat, uv)
    return token_mlp(sampled, P.sub('filters')).transpose(0, 2, 1)


def lineseg_dists(p, a, b):
    """models/dir.py:132-144 (point-to-segment distance, fp ops in the reference's order)."""
    d_ba = b - a
    d = d_ba / np.hypot(d_ba[:, 0], d_ba[:, 1]).reshape(-1, 1)
    s = ((a - p) * d).sum(1)
    t = ((p - b) * d).sum(1)
    h = np.maximum(np.maximum(s, t), 0)
    d_pa = p - a
    c = d_pa[:, 0] * d[:, 1] - d_pa[:, 1] * d[:, 0]
    return np.hypot(h, c)


def bone_proj(joint_uv, joint_feat, S, distance, return_mask=False):
    """Joint2BoneFeature.bone_proj (models/dir.py:146-174) -> [B, 20*C, S, S]."""
    dt = joint_feat.dtype
    B, J, C = joint_feat.shape
    uv = ((joint_uv.astype(dt) + 1) / 2 * S).astype(dt)
    c = np.arange(S, dtype=dt) + dt.type(0.5)
    gx, gy = np.meshgrid(c, c, indexing='ij')
    grid = np.stack((gy, gx), -1).reshape(S * S, 2)                        # models/dir.py:66-70
    a = np.broadcast_to(uv[:, PARENT].reshape(B, 1, 20, 2), (B, S * S, 20, 2)).reshape(-1, 2)
    b = np.broadcast_to(uv[:, CHILD].reshape(B, 1, 20, 2), (B, S * S, 20, 2)).reshape(-1, 2)
    p = np.broadcast_to(grid.reshape(1, S * S, 1, 2), (B, S * S, 20, 2)).reshape(-1, 2)
    with np.errstate(all='ignore'):
        dist = lineseg_dists(p, a, b).reshape(B, S * S, 20)
        mask = dist < dt.type(distance)
        eps = dt.type(1e-6)                                                # F.pairwise_distance eps
        da = np.sqrt(((p - a + eps) ** 2).sum(1))
        db = np.sqrt(((p - b + eps) ** 2).sum(1))
        wa = (1 - da / (da + db)).reshape(B, S * S, 20, 1)
        wb = (1 - db / (da + db)).reshape(B, S * S, 20, 1)
        fa = joint_feat[:, PARENT].reshape(B, 1, 20, C)
        fb = joint_feat[:, CHILD].reshape(B, 1, 20, C)
        img = fa * wa + fb * wb
    img = np.where(mask[..., None], img, 0).astype(dt)
    img = img.reshape(B, S, S, 20 * C).transpose(0, 3, 1, 2)
    if return_mask:
        return img, mask.reshape(B, S, S, 20)
    return img


def regressor_offset(fl, fr, para_l, para_r, offset, P, mano_l, mano_r, root_joint=0):
    """RegressorOffset.forward (models/dir.py:339-381).  mano_l/r: th_* buffer dicts."""
    B = fl.shape[0]
    fl2, fr2 = fl.reshape(B, -1), fr.reshape(B, -1)
    gl = np.concatenate([fl2, para_l], -1)
    gr = np.concatenate([fr2, para_r], -1)
    gf = np.concatenate([fl2, fr2, offset.reshape(B, 3)], -1)
    pd_offset = N.linear(gf, P['offset.weight'], P['offset.bias'])
    pl = N.linear(gl, P['mano_left.weight'], P['mano_left.bias'])
    pr = N.linear(gr, P['mano_right.weight'], P['mano_right.bias'])
    return mano_outputs(pl, pr, pd_offset, mano_l, mano_r, root_joint)


def mano_outputs(pl, pr, pd_offset, mano_l, mano_r, root_joint=0):
    """Split the 64-vector (51 pose | 10 beta | 3 weak-persp) and run MANO + projection
    (models/dir.py:272-304 / 353-381)."""
    out = {'pd_offset': pd_offset, 'pd_mano_para_left': pl, 'pd_mano_para_right': pr}
    for side, p, buf in (('left', pl, mano_l), ('right', pr, mano_r)):
        pose, beta, cam = p[:, :51], p[:, 51:61], p[:, 61:64]
        verts, joints = mano_forward(buf, pose, beta, side, root_joint)
        out['pd_mano_pose_' + side] = pose
        out['pd_mano_beta_' + side] = beta
        out['pd_proj_' + side] = cam
        out['pd_mesh_xyz_' + side] = verts
        out['pd_joint_xyz_' + side] = joints
        out['pd_joint_uv_' + side] = projection_batch_xy(cam[:, 0], cam[:, 1:], joints)
        out['pd_mesh_uv_' + side] = projection_batch_xy(cam[:, 0], cam[:, 1:], verts)
    return out


def mano_bufs(P, side):
    sub = P.sub('mano_layer_' + side)
    return {k: sub[k] for k in ('th_selected_comps', 'th_hands_mean', 'th_shapedirs', 'th_posedirs',
                                'th_v_template', 'th_J_regressor', 'th_weights')}


def stage_forward(P, S, distance, img_feat, xyz_l, xyz_r, uv_l, uv_r, para_l, para_r, offset, root_joint=0):
    """Joint2BoneFeature.forward (models/dir.py:86-130).  Returns (result dict, feat dict)."""
    dt = img_feat.dtype
    s015 = dt.type(0.15)
    jl = img2joint(img_feat, uv_l, P.sub('img2joint_left'))
    jr = img2joint(img_feat, uv_r, P.sub('img2joint_right'))
    pl = token_mlp((xyz_l.transpose(0, 2, 1) / s015), P.sub('pos_emb_left')).transpose(0, 2, 1)
    pr = token_mlp((xyz_r.transpose(0, 2, 1) / s015), P.sub('pos_emb_right')).transpose(0, 2, 1)
    fl = pgcn_stack(pl + jl, P.sub('gcn_left'))
    fr = pgcn_stack(pr + jr, P.sub('gcn_right'))
    gl = token_mlp((xyz_l / s015 - offset / 2).transpose(0, 2, 1), P.sub('global_pos_emb')).transpose(0, 2, 1)
    gr = token_mlp((xyz_r / s015 + offset / 2).transpose(0, 2, 1), P.sub('global_pos_emb')).transpose(0, 2, 1)
    tok = np.concatenate([fl + gl, fr + gr], 1)
    tok = ste_forward(tok, P.sub('interaction'))
    tl, tr = tok[:, :21], tok[:, 21:]
    R = P.sub('regressor')
    result = regressor_offset(tl, tr, para_l, para_r, offset, R, mano_bufs(R, 'left'), mano_bufs(R, 'right'),
                              root_joint)
    el = token_mlp(tl.transpose(0, 2, 1), P.sub('proj_feat_emb')).transpose(0, 2, 1)
    er = token_mlp(tr.transpose(0, 2, 1), P.sub('proj_feat_emb')).transpose(0, 2, 1)
    il = bone_proj(result['pd_joint_uv_left'], el, S, distance)
    ir = bone_proj(result['pd_joint_uv_right'], er, S, distance)
    F = P.sub('fusion')
    h = N.conv2d(np.concatenate([il, ir], 1), F['0.weight'], F['0.bias'], 1, 1)
    h = N.relu(N.batchnorm(h, F.sub('1')))
    img_out = N.conv2d(h, F['3.weight'], F['3.bias'], 1, 0)
    feats = {'img_feat': img_out, 'joint_feat_left': el, 'joint_feat_right': er, 'vis_img_feat': il + ir}
    return result, feats
