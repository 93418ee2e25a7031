"""Seeded input builders shared by oracle/gen_golden.py (which feeds them to the reference) and the tests
(which feed them to the oracle / the HIP path).  Test infrastructure; numpy only."""
import numpy as np

from dir_amd import synth

SEED = 1234


def mano_inputs(case, B=4):
    pose = synth.synth_input('mano.pose.' + case, (B, 51), SEED) * 0.6
    pose[:, :6] = synth.synth_input('mano.rot6d.' + case, (B, 6), SEED)
    betas = synth.synth_input('mano.betas.' + case, (B, 10), SEED) * 0.8
    if case == 'zero_pose':
        pose[:, 6:] = 0
    elif case == 'large':
        pose[:, 6:] *= 5.0
        pose[:, :6] *= 30.0
    elif case == 'tiny6d':
        pose[:, :6] *= 1e-5
    elif case == 'near_parallel':
        pose[:, 3:6] = pose[:, 0:3] * 1.5 + 1e-3 * synth.synth_input('mano.np.' + case, (B, 3), SEED)
    return pose, betas


MANO_CASES = ['normal', 'zero_pose', 'large', 'tiny6d', 'near_parallel']


def edge_uv(name, B, S):
    uv = synth.synth_input(name, (B, 21, 2), SEED, kind='uniform', lo=-0.9, hi=0.9)
    uv[0, 0] = [-1.0, -1.0]; uv[0, 1] = [1.0, 1.0]; uv[0, 2] = [-1.3, 0.2]; uv[0, 3] = [0.4, 1.7]
    uv[0, 4] = [(3 + 0.5) / S * 2 - 1, (5 + 0.5) / S * 2 - 1]      # exactly a pixel centre
    uv[0, 5] = [-1.0 + 1.0 / S, 1.0 - 1.0 / S]                       # centre of corner pixels
    uv[0, 6] = [0.0, 0.0]
    uv[1, 0] = [1.0 + 1.0 / S, 0.0]                                  # half a pixel outside
    uv[1, 1] = [5.0, -7.0]
    return uv


def bone_uv(name, B, S):
    uv = synth.synth_input(name, (B, 21, 2), SEED, kind='uniform', lo=-0.85, hi=0.85)
    # hand-like: children near parents so bones are a few pixels long
    par = [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]
    step = synth.synth_input(name + '.step', (B, 21, 2), SEED, kind='uniform', lo=-0.25, hi=0.25)
    for j in range(1, 21):
        uv[:, j] = uv[:, par[j]] + step[:, j]
    uv[0, 2] = uv[0, 1]                                              # coincident parent/child -> NaN distance
    uv[0, 5] = [(4 + 0.5) / S * 2 - 1, (6 + 0.5) / S * 2 - 1]      # endpoint exactly on a pixel centre
    uv[0, 6] = [(7 + 0.5) / S * 2 - 1, (6 + 0.5) / S * 2 - 1]      # axis-aligned bone through centres
    uv[1, 9] = [1.4, -1.2]                                           # bone leaving the image
    return uv


def stage_inputs(S, B=2):
    p = 'stage%d.' % S
    img_feat = synth.synth_input(p + 'img_feat', (B, 256, S, S), SEED)
    xyz_l = synth.synth_input(p + 'xyz_l', (B, 21, 3), SEED) * 0.05
    xyz_r = synth.synth_input(p + 'xyz_r', (B, 21, 3), SEED) * 0.05
    uv_l = bone_uv(p + 'uv_l', B, S)
    uv_r = bone_uv(p + 'uv_r', B, S)
    para_l = synth.synth_input(p + 'para_l', (B, 64), SEED)
    para_r = synth.synth_input(p + 'para_r', (B, 64), SEED)
    offset = synth.synth_input(p + 'offset', (B, 1, 3), SEED)
    return img_feat, xyz_l, xyz_r, uv_l, uv_r, para_l, para_r, offset


def eval_inputs(B=6):
    """synthetic batch with the layout of apps/eval.py's dataloader tuple (data[0..10]) and of network(...)[2]"""
    from oracle import mano as OM
    out = {}
    for side in ('left', 'right'):
        pose, betas = mano_inputs('normal', B)
        buf = synth.mano_buffers(side, SEED)
        v, j = OM.mano_forward(buf, pose + synth.synth_input('eval.pose.' + side, (B, 51), SEED) * 0.2, betas, side, None)
        trans = np.array([[-0.08 if side == 'left' else 0.08, 0.0, 0.75]], np.float32) + \
            synth.synth_input('eval.trans.' + side, (B, 3), SEED) * np.float32(0.03)
        out['verts_gt_' + side] = (v + trans[:, None]).astype(np.float32)
        noise = synth.synth_input('eval.noise.' + side, (B, 778, 3), SEED) * np.float32(0.004)
        scl = 1.0 + synth.synth_input('eval.scl.' + side, (B, 1, 1), SEED) * np.float32(0.05)
        out['verts_pd_' + side] = ((v + noise) * scl).astype(np.float32)            # root-relative, wrong scale
        out['jreg_' + side] = buf['th_J_regressor']
    cam = np.tile(np.array([[1500., 0, 128.], [0, 1500., 128.], [0, 0, 1.]], np.float32), (B, 1, 1))
    cam[:, 0, 0] += synth.synth_input('eval.f', (B,), SEED) * 20
    out['cam'] = cam
    for side in ('left', 'right'):
        p = out['verts_gt_' + side] @ cam.transpose(0, 2, 1)
        out['verts2d_gt_' + side] = (p[..., :2] / p[..., 2:]).astype(np.float32) + \
            synth.synth_input('eval.v2d.' + side, (B, 778, 2), SEED) * np.float32(0.5)
    out['pd_offset'] = synth.synth_input('eval.off', (B, 3), SEED)
    return out


GTMANO_CASES = [('normal', 45, None, False, False, False), ('pca12', 12, None, False, False, False),
                ('center9_scale', 45, 9, True, True, False), ('center0_notrans', 45, 0, False, False, False),
                ('rotmat_newskel', 0, None, True, False, True), ('zero_pose', 45, None, False, False, False)]


def gtmano_inputs(case, B=5):
    """inputs of models/manolayer.py::ManoLayer.forward for the case tuple (name, ncomps (0 = rotation matrices), center_idx,
    with_trans(ignored for 'center0_notrans'), with_scale, new_skel)"""
    name, ncomps = case[0], case[1]
    from oracle.gt_mano import rodrigues_batch
    ax = synth.synth_input('gtmano.root.' + name, (B, 3), SEED) * np.float32(1.2)
    R = rodrigues_batch(ax.astype(np.float64)).astype(np.float32)
    if ncomps > 0:
        pose = synth.synth_input('gtmano.pose.' + name, (B, ncomps), SEED) * np.float32(0.0 if name == 'zero_pose' else 0.8)
    else:
        a = synth.synth_input('gtmano.rotpose.' + name, (B * 15, 3), SEED) * np.float32(0.5)
        pose = rodrigues_batch(a.astype(np.float64)).astype(np.float32).reshape(B, 15, 3, 3)
    shape = synth.synth_input('gtmano.shape.' + name, (B, 10), SEED)
    trans = None if name == 'center0_notrans' else (synth.synth_input('gtmano.trans.' + name, (B, 3), SEED) * np.float32(0.1) +
                                                   np.array([0, 0, 0.7], np.float32))
    scale = (1 + 0.1 * synth.synth_input('gtmano.scale.' + name, (B,), SEED)).astype(np.float32) if case[4] else None
    return R, pose, shape, trans, scale


MANO_GRAD_CASES = [('normal', 0), ('normal', 9), ('normal', -1), ('large', 0), ('zero_pose', 0)]


def mano_grad_inputs(case, side, B=3):
    """64-vectors (pose 51 | betas 10 | cam 3) and cotangents of the four outputs a regressor derives from them (models/dir.py:352-363)"""
    pose, betas = mano_inputs(case, B)
    cam = synth.synth_input('manograd.cam.%s.%s' % (case, side), (B, 3), SEED) * np.float32(0.3) + np.array([1.2, 0.0, 0.0], np.float32)
    para = np.concatenate([pose, betas, cam], 1).astype(np.float32)
    cot = {'verts': synth.synth_input('manograd.gv.%s.%s' % (case, side), (B, 778, 3), SEED),
           'joints': synth.synth_input('manograd.gj.%s.%s' % (case, side), (B, 21, 3), SEED),
           'joint_uv': synth.synth_input('manograd.gju.%s.%s' % (case, side), (B, 21, 2), SEED),
           'mesh_uv': synth.synth_input('manograd.gmu.%s.%s' % (case, side), (B, 778, 2), SEED)}
    return para, cot


REGRESS_OUT_KEYS = ('pd_offset', 'pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_uv_left', 'pd_mesh_uv_right', 'pd_joint_xyz_left',
                    'pd_joint_xyz_right', 'pd_mesh_xyz_left', 'pd_mesh_xyz_right')


def regress_grad_inputs(B=4):
    """inputs of RegressorOffset.forward (models/dir.py:339) and cotangents of the outputs the training objective reads (:571-592)"""
    g = lambda n, shp: synth.synth_input('regressgrad.' + n, shp, SEED)  # noqa: E731
    ins = dict(feat_l=g('fl', (B, 21, 64)), feat_r=g('fr', (B, 21, 64)), para_l=g('pl', (B, 64)) * np.float32(0.3), para_r=g('pr', (B, 64)) * np.float32(0.3),
               offset=g('off', (B, 1, 3)))
    shp = {'pd_offset': (B, 3)}
    for s_ in ('left', 'right'):
        shp.update({'pd_joint_uv_' + s_: (B, 21, 2), 'pd_mesh_uv_' + s_: (B, 778, 2), 'pd_joint_xyz_' + s_: (B, 21, 3), 'pd_mesh_xyz_' + s_: (B, 778, 3)})
    cot = {k: g('cot.' + k, shp[k]) for k in REGRESS_OUT_KEYS}
    return ins, cot


def stage_grad_inputs(S=16, B=4):
    """G17: stage inputs (as stage_inputs) + the cotangents of the stage's outputs"""
    ins = stage_inputs(S, B)
    shapes = {'pd_offset': (B, 3), 'joint_feat': (B, 42, 64)}
    for s in ('left', 'right'):
        shapes.update({'pd_mano_para_' + s: (B, 64), 'pd_mesh_xyz_' + s: (B, 778, 3), 'pd_joint_xyz_' + s: (B, 21, 3),
                       'pd_joint_uv_' + s: (B, 21, 2), 'pd_mesh_uv_' + s: (B, 778, 2)})
    cot = {k: synth.synth_input('stagegrad.' + k, shp, SEED) for k, shp in shapes.items()}
    cot['joint_feat'] = cot['joint_feat'] * np.float32(0.05)
    return ins, cot


BLOCK_GRAD_CASES = {  # name: (kind, stride, input shape NCHW, output shape NCHW)       (oracle/gen_golden.py::BLOCK_CASES)
    'bneck_plain': ('bottleneck', 1, (3, 256, 16, 16), (3, 256, 16, 16)),
    'bneck_down': ('bottleneck', 2, (3, 256, 16, 16), (3, 512, 8, 8)),
    'res_skip': ('residual', 1, (3, 512, 16, 16), (3, 256, 16, 16)),
    'res_same': ('residual', 1, (3, 256, 16, 16), (3, 256, 16, 16)),
}


def block_grad_inputs(name, batch=None):
    kind, stride, xs, ys = BLOCK_GRAD_CASES[name]
    if batch is not None:
        xs, ys = (batch,) + xs[1:], (batch,) + ys[1:]
    return synth.synth_input('blockgrad.%s.x' % name, xs, SEED), synth.synth_input('blockgrad.%s.gy' % name, ys, SEED)


def bone_grad_inputs(S, B=3):
    """G19: one hand's joint uv [B,21,2], re-embedded joint features [B,21,64], cotangent of the rasterised map [B,1280,S,S]"""
    p = 'bonegrad%d.' % S
    return bone_uv(p + 'uv', B, S), synth.synth_input(p + 'feat', (B, 21, 64), SEED), synth.synth_input(p + 'g', (B, 1280, S, S), SEED)


def extra_stage_shapes(shapes, n):
    """parameter / buffer shapes of n extra refinement stages (decoder.projecter_x.<i>.* = the shapes of decoder.projecter_3.*,
    decoder.enhance_layer_x.<i>.* = those of decoder.enhance_layer3.*: the same classes with the same arguments, models/dir.py:401-402)"""
    out = {}
    for i in range(n):
        for k, v in shapes.items():
            if k.startswith('decoder.projecter_3.'):
                out['decoder.projecter_x.%d.' % i + k[len('decoder.projecter_3.'):]] = v
            elif k.startswith('decoder.enhance_layer3.'):
                out['decoder.enhance_layer_x.%d.' % i + k[len('decoder.enhance_layer3.'):]] = v
    return out
