#!/usr/bin/env python3
"""Golden-vector generator: imports the REFERENCE (read-only, /root/reference) on CPU and dumps
small input/output fixtures under tests/golden/.

Runs ONLY in the authoring container (the reference never travels to the GPU box).  Everything here
is test infrastructure.  Parameters are never stored: both this script and the tests regenerate them
by name with dir_amd.synth (counter-based RNG), so the fixtures hold inputs + expected outputs only.

Stubs (none carries hot-path arithmetic):
  timm.models.layers (DropPath never instantiated: transformer/mixSTE.py:118,133),
  torchvision.models (ImageNet weight download: models/dir.py:490-498),
  cv2 / imgaug / yacs (top-level imports of utils/utils.py),
  mano.webuser.smpl_handpca_wrapper_HAND_only.ready_arguments (licensed pkl loader:
  manopth/manopth/manolayer.py:65) -> synthetic MANO tables of the real shapes.
  torch.Tensor.cuda / nn.Module.cuda -> identity (models/dir.py:514).

usage: python oracle/gen_golden.py [--only NAME ...]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dir_amd import synth  # noqa: E402
from oracle.golden_inputs import MANO_CASES, bone_uv, edge_uv, mano_inputs, stage_inputs  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
SEED = 1234


# ----------------------------------------------------------------------------- reference import
class _R(object):
    """mimics a chumpy array: the ndarray lives in `.r`"""

    def __init__(self, a):
        self.r = a


def _fake_ready_arguments(path):
    side = 'left' if 'LEFT' in os.path.basename(path) else 'right'
    t = synth.synthetic_mano_tables(side, SEED)
    return {
        'hands_components': t['hands_components'], 'hands_mean': t['hands_mean'],
        'betas': _R(t['betas']), 'shapedirs': _R(t['shapedirs']), 'posedirs': _R(t['posedirs']),
        'v_template': _R(t['v_template']), 'weights': _R(t['weights']),
        'J_regressor': sp.csc_matrix(t['J_regressor']), 'f': t['f'], 'kintree_table': t['kintree_table'],
    }


def import_reference():
    sys.dont_write_bytecode = True
    sys.path[:0] = [REF, os.path.join(REF, 'manopth')]

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    mod('timm'); mod('timm.models')
    mod('timm.models.layers', DropPath=DropPath, to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None)

    class _W:
        IMAGENET1K_V2 = None
    mod('torchvision')
    mod('torchvision.models', resnet50=lambda weights=None: nn.Module(), ResNet50_Weights=_W)
    mod('cv2')
    ia = mod('imgaug')
    ia.augmenters = mod('imgaug.augmenters')
    mod('yacs')
    mod('yacs.config', CfgNode=dict)
    mod('mano'); mod('mano.webuser')
    mod('mano.webuser.smpl_handpca_wrapper_HAND_only', ready_arguments=_fake_ready_arguments)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def load_synth(module, seed=SEED, cond=False):
    """cond: the trained-like flavour of the synthetic parameters (dir_amd.synth.synth_tensor): activations O(1) in every layer"""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    vals = synth.synth_state_dict(shapes, seed, cond=cond)
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in vals.items()}, strict=True)
    return shapes


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('  wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- G1 MANO


def gen_mano():
    from manopth.manolayer import ManoLayer
    from manopth import rot6d
    from manopth.tensutils import th_posemap_axisang
    out = {}
    for side in ('left', 'right'):
        for center in (0, 9, -1):
            for flat in (False, True):
                if flat and center != 0:
                    continue
                layer = ManoLayer(root_rot_mode='6D', joint_rot_mode='axisang', use_pca=True, mano_root='unused',
                                  side=side, ncomps=45, center_idx=(None if center < 0 else center),
                                  flat_hand_mean=flat, robust_rot=True)
                for case in MANO_CASES:
                    if (center != 0 or flat) and case not in ('normal', 'zero_pose'):
                        continue
                    pose, betas = mano_inputs(case)
                    tp, tb = torch.from_numpy(pose), torch.from_numpy(betas)
                    verts, joints = layer(tp, tb)
                    tag = '%s_c%d_f%d_%s' % (side, center, int(flat), case)
                    out[tag + '.pose'] = pose
                    out[tag + '.betas'] = betas
                    out[tag + '.verts'] = verts
                    out[tag + '.joints'] = joints
                    if center == 0 and not flat:
                        full = torch.cat([tp[:, :6], layer.th_hands_mean + tp[:, 6:].mm(layer.th_selected_comps)], 1)
                        out[tag + '.root_rot'] = rot6d.robust_compute_rotation_matrix_from_ortho6d(full[:, :6])
                        out[tag + '.rot_map'] = th_posemap_axisang(full[:, 6:])[1]
    save('g1_mano', **out)


# ----------------------------------------------------------------------------- G2 PGCN
def gen_pgcn():
    from SemGCN.utils import adj_mx_from_edges, get_sketch_setting
    from SemGCN.p_gcn import ResSimplePGCN
    from SemGCN.p_graph_conv import PGraphConv
    import torch.nn.functional as F
    adj = adj_mx_from_edges(21, get_sketch_setting(), sparse=False, eye=False)
    net = ResSimplePGCN(adj, 128, num_layers=4).eval()
    load_synth(net)
    x = torch.from_numpy(synth.synth_input('pgcn.x', (2, 21, 128), SEED))
    acts = {}
    for i, l in enumerate(net.gconv_layers):
        l.gconv.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__('gconv%d' % i, o.clone()))
        l.register_forward_hook(lambda m, a, o, i=i: acts.__setitem__('layer%d' % i, o.clone()))
    with torch.no_grad():
        y = net(x.clone())
        g0 = net.gconv_layers[0].gconv
        A1 = -9e15 * torch.ones_like(g0.adj_1)
        A1[g0.m_1] = g0.e_1
        A1 = F.softmax(A1, dim=1)
        single = PGraphConv(128, 128, adj).eval()
        load_synth(single)
        ys = single(x.clone())
    nz = np.stack(np.nonzero(adj.numpy() > 0), 1)
    save('g2_pgcn', x=x, y=y, A1=A1, adj=adj, edge_order=nz, single_y=ys, **acts)


# ----------------------------------------------------------------------------- G3 STE
def gen_ste():
    from transformer.mixSTE import STE
    net = STE(num_joints=42, in_chans=128, out_dim=64, depth=4).eval()
    load_synth(net)
    x = torch.from_numpy(synth.synth_input('ste.x', (2, 42, 128), SEED))
    acts = {}
    cnt = [0]

    def hook(m, a, o):
        acts['after_norm%d' % cnt[0]] = o.clone()
        cnt[0] += 1
    net.spatial_norm.register_forward_hook(hook)
    probs = {}
    net.STEblocks[1].attn.attn_drop.register_forward_hook(lambda m, a, o: probs.__setitem__('p', o.clone()))
    with torch.no_grad():
        y = net(x.clone())
    save('g3_ste', x=x, y=y, attn_probs_block1=probs['p'], **acts)


# ----------------------------------------------------------------------------- G4 grid tokens
def gen_grid():
    from models.dir import ImgFeature2JointFeature
    out = {}
    net = ImgFeature2JointFeature(256, 128).eval()
    load_synth(net)
    for S in (16, 32):
        feat = torch.from_numpy(synth.synth_input('grid.feat%d' % S, (2, 256, S, S), SEED))
        uv = torch.from_numpy(edge_uv('grid.uv%d' % S, 2, S))
        with torch.no_grad():
            y = net(feat, uv)
            sampled = torch.nn.functional.grid_sample(feat, uv.unsqueeze(1)).squeeze(-2)
        out['S%d.uv' % S] = uv
        out['S%d.y' % S] = y            # [2, 128*21] channel-major flatten (models/dir.py:200)
        out['S%d.sampled' % S] = sampled  # [2,256,21]
    save('g4_grid', **out)


# ----------------------------------------------------------------------------- G5 bone_proj
def gen_bone():
    from models.dir import Joint2BoneFeature
    out = {}
    for S, dist in ((16, 1), (32, 2)):
        net = Joint2BoneFeature(256, 128, 64, 21, S, 'unused', 0, distance=dist).eval()
        uv = torch.from_numpy(bone_uv('bone.uv%d' % S, 2, S))
        feat = torch.from_numpy(synth.synth_input('bone.feat%d' % S, (2, 21, 64), SEED))
        with torch.no_grad():
            y = net.bone_proj(uv, feat)
        out['S%d.uv' % S] = uv
        out['S%d.y' % S] = y            # [2,1280,S,S]
    save('g5_bone', **out)


# ----------------------------------------------------------------------------- G6 refinement stage
def gen_stage():
    from models.dir import Joint2BoneFeature
    for S, dist in ((16, 1), (32, 2)):
        net = Joint2BoneFeature(256, 128, 64, 21, S, 'unused', 0, distance=dist).eval()
        shapes = load_synth(net)
        ins = [torch.from_numpy(a) for a in stage_inputs(S)]
        with torch.no_grad():
            result, feats = net(*ins)
        out = {k: v for k, v in result.items() if v is not None}
        out['img_feat'] = feats['img_feat']
        out['joint_feat_left'] = feats['joint_feat_left']
        out['joint_feat_right'] = feats['joint_feat_right']
        vis = feats['vis_img_feat']
        out['vis_sum'] = vis.double().sum(dim=(2, 3))
        out['vis_slice'] = vis[:, 0:1280:97]
        save('g6_stage%d' % S, **out)
        with open(os.path.join(OUT, 'manifest_stage%d.json' % S), 'w') as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)


# ----------------------------------------------------------------------------- G7 full DIR
def gen_full(cond=False):
    """cond: G7c, the same pass on the trained-like parameters (activations O(1): the fixture the bf16 mode's < 0.01 mm gate is held to)"""
    from models.dir import DIR
    net = DIR(21, 'unused', 0).eval()
    shapes = load_synth(net, cond=cond)
    if not cond:
        with open(os.path.join(OUT, 'manifest_dir.json'), 'w') as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    B = 2
    img = torch.from_numpy(synth.synth_input('dir.img', (B, 3, 256, 256), SEED))
    taps = {}

    def tap(name):
        def h(m, a, o):
            t = o[1]['img_feat'] if isinstance(o, tuple) else o
            taps[name + '.sum'] = t.double().sum(dim=(2, 3))
            taps[name + '.abssum'] = t.double().abs().sum(dim=(2, 3))
            taps[name + '.slice'] = t[:, :4].clone()
        return h
    for name, m in [('c1', net.backbone.layer1), ('c2', net.backbone.layer2), ('c3', net.backbone.layer3),
                    ('c4', net.backbone.layer4), ('skip4', net.decoder.skip_layer4),
                    ('fusion4', net.decoder.fusion_layer4), ('proj4', net.decoder.projecter_4),
                    ('enh4', net.decoder.enhance_layer4), ('fusion3', net.decoder.fusion_layer3),
                    ('proj3', net.decoder.projecter_3), ('enh3', net.decoder.enhance_layer3),
                    ('final', net.decoder.conv_final), ('stem', net.backbone.maxpool)]:
        m.register_forward_hook(tap(name))
    with torch.no_grad():
        outs, loss = net({'img': img}, None, None)
    assert loss == {}
    out = dict(taps)
    for i in range(3):
        for k, v in outs[i].items():
            if v is not None:
                out['s%d.%s' % (i, k)] = v
    out['seg'] = outs[3]['seg']
    out['dense'] = outs[3]['dense']
    pf = outs[3]['proj_feat']
    out['proj_feat.sum'] = pf.double().sum(dim=(2, 3))
    out['proj_feat.slice'] = pf[:, 0:1280:97]
    for i in range(3):
        uv = outs[i]['pd_joint_uv_left']
        print('   stage %d uv range [%.3f, %.3f]  verts absmax %.4f' % (
            i, float(uv.min()), float(uv.max()), float(outs[i]['pd_mesh_xyz_left'].abs().max())))
    print('   c4 absmean %.3f' % float(taps['c4.abssum'].sum() / (B * 2048 * 64)))
    save('g7c_dir' if cond else 'g7_dir', **out)


# ----------------------------------------------------------------------------- G9 eval metric maths
def gen_eval():
    """The metric maths is inside apps/eval.py's `if __name__ == '__main__':` block, so it cannot be imported: the
    loop body (from `with torch.no_grad():` to the first np.concatenate) is extracted from the reference file AT
    GENERATION TIME and executed on synthetic batches, with the reference's own Jr and xyz2uvd definitions."""
    import textwrap
    src = open(os.path.join(REF, 'apps', 'eval.py')).read().split('\n')
    a = next(i for i, l in enumerate(src) if l.strip() == 'with torch.no_grad():')
    b = next(i for i, l in enumerate(src) if l.strip().startswith("joints_loss['left'] = np.concatenate"))
    body = textwrap.dedent('\n'.join(src[a:b]))
    c0 = next(i for i, l in enumerate(src) if l.startswith('class Jr'))
    c1 = next(i for i, l in enumerate(src) if l.startswith('class handDataset'))
    d0 = next(i for i, l in enumerate(src) if l.startswith('def xyz2uvd'))
    d1 = next(i for i, l in enumerate(src) if l.startswith("if __name__ == '__main__':"))
    defs = '\n'.join(src[c0:c1] + src[d0:d1])
    from oracle.golden_inputs import eval_inputs
    ins = eval_inputs()
    T = lambda k: torch.from_numpy(ins[k])  # noqa: E731
    out = {}
    for root_joint in (0, 9):
        for scale in (True, False):
            ns = {'torch': torch, 'np': np, 'tqdm': (lambda x: x)}
            exec(defs, ns)
            data = [torch.zeros(ins['cam'].shape[0], 3, 8, 8), None, None, T('verts_gt_left'), None, T('verts_gt_right'), None,
                    T('verts2d_gt_left'), None, T('verts2d_gt_right'), T('cam')]
            for i in (2, 4, 6, 8):
                data[i] = torch.zeros(1)
            result = [None, None, {'pd_offset': T('pd_offset'), 'pd_mesh_xyz_left': T('verts_pd_left'),
                                   'pd_mesh_xyz_right': T('verts_pd_right')}]
            ns.update(dataloader=[tuple(data)], network=lambda *a, **k: (result, {}),
                      J_regressor={'left': ns['Jr'](T('jreg_left'), device='cpu'), 'right': ns['Jr'](T('jreg_right'), device='cpu')},
                      opt=types.SimpleNamespace(root_joint=root_joint, scale=scale), stage_num=3, idx=0,
                      joints_loss={'left': [], 'right': []}, verts_loss={'left': [], 'right': []},
                      joints_xyz_list={'left': [], 'right': []}, joints_xyz_gt_list={'left': [], 'right': []},
                      joints_2d_loss={'left': [], 'right': []}, verts_2d_loss={'left': [], 'right': []},
                      root_loss_list=[], inter_volume_list=[], val_root_list=[])
            exec(body, ns)
            tag = 'r%d_s%d.' % (root_joint, int(scale))
            for side in ('left', 'right'):
                out[tag + 'joint_err_' + side] = ns['joints_loss'][side][0]
                out[tag + 'vert_err_' + side] = ns['verts_loss'][side][0]
                out[tag + 'joint2d_err_' + side] = ns['joints_2d_loss'][side][0]
                out[tag + 'vert2d_err_' + side] = ns['verts_2d_loss'][side][0]
                out[tag + 'joints_pd_' + side] = ns['joints_xyz_list'][side][0]
                out[tag + 'joints_gt_' + side] = ns['joints_xyz_gt_list'][side][0]
            out[tag + 'root_err'] = ns['root_loss_list'][0]
    save('g9_eval', **out)


# ----------------------------------------------------------------------------- G10 GT MANO layer (models/manolayer.py)
def gen_gtmano():
    """models/manolayer.py::ManoLayer built from a synthetic pickle (written to a temp dir at generation time) with the keys
    its constructor reads (:106-158); forward on the seeded inputs of golden_inputs.gtmano_inputs."""
    import pickle
    import tempfile
    import scipy.sparse as sp
    from models.manolayer import ManoLayer as GTManoLayer
    from oracle.golden_inputs import GTMANO_CASES, gtmano_inputs
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for side in ('left', 'right'):
            t = synth.synthetic_mano_tables(side, SEED)
            d = {'hands_components': t['hands_components'], 'J_regressor': sp.csc_matrix(t['J_regressor']),
                 'J': t['J_regressor'] @ t['v_template'], 'weights': t['weights'], 'posedirs': t['posedirs'],
                 'v_template': t['v_template'], 'shapedirs': t['shapedirs'], 'hands_mean': t['hands_mean'], 'f': t['f'],
                 'kintree_table': t['kintree_table']}
            path = os.path.join(tmp, side + '.pkl')
            with open(path, 'wb') as f:
                pickle.dump(d, f)
            for case in GTMANO_CASES:
                name, ncomps, center, _, _, new_skel = case
                layer = GTManoLayer(path, center_idx=center, use_pca=ncomps > 0, new_skel=new_skel)
                R, pose, shape, trans, scale = gtmano_inputs(case)
                T = lambda a: None if a is None else torch.from_numpy(a)  # noqa: E731
                v, j = layer(T(R), T(pose), T(shape), trans=T(trans), scale=T(scale))
                out['%s.%s.verts' % (side, name)], out['%s.%s.joints' % (side, name)] = v.numpy(), j.numpy()
    save('g10_gtmano', **out)


# ----------------------------------------------------------------------------- G11 input tensor preparation
def gen_imgprep():
    """apps/eval.py:59-61 (handDataset.__getitem__), the three statements after cv.resize, extracted from the reference file and
    executed on a seeded uint8 BGR image batch; cv.cvtColor(BGR2RGB) and torchvision's Normalize are the only stubs (channel
    flip; (t - mean) / std as torchvision.transforms.functional.normalize computes it)."""
    import textwrap
    src = open(os.path.join(REF, 'apps', 'eval.py')).read().split('\n')
    a = next(i for i, l in enumerate(src) if 'imgTensor = torch.tensor(cv.cvtColor' in l)
    body = textwrap.dedent('\n'.join(src[a:a + 3]))
    assert 'normalize_img' in body and 'permute(2, 0, 1)' in body

    class _Norm:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std
    cvstub = types.SimpleNamespace(COLOR_BGR2RGB=4, cvtColor=lambda im, code: np.ascontiguousarray(im[..., ::-1]))
    g = np.random.default_rng(SEED)
    imgs = g.integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)
    imgs[0, :4, :4] = 0
    imgs[0, 4:8, :4] = 255
    outs = []
    for im in imgs:
        ns = {'torch': torch, 'cv': cvstub, 'img': im,
              'self': types.SimpleNamespace(normalize_img=_Norm([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]))}
        exec(body, ns)
        outs.append(ns['imgTensor'].numpy())
    save('g11_imgprep', img=imgs, y=np.stack(outs))


# ----------------------------------------------------------------------------- G8 training objective (forward)
def gen_loss(cond=False):
    """(cond: G8c, on the trained-like parameters -- the forward pass G20c differentiates.)  The loss block is inline in DIR.forward (models/dir.py:542-594) and runs only in training mode: the reference model is run
    with .train() (batch-statistics BN) on seeded input, once with placeholder targets to obtain its predictions, then with targets
    built AROUND those predictions (small and large residuals: both SmoothL1 branches) -- the fixture holds that second pass's
    predictions, targets and the 42 loss scalars.  Faces: synth.loss_faces (the synthetic MANO triangles with the degenerate rows repaired; regenerated by the tests)."""
    from models.dir import DIR
    net = DIR(21, 'unused', 0)
    load_synth(net, cond=cond)
    net.train()
    for side in ('left', 'right'):      # non-degenerate triangles (synth.loss_faces); the forward pass never reads the faces
        fc = torch.from_numpy(synth.loss_faces(side, SEED))
        getattr(net, 'normal_loss_' + side).face = fc
        getattr(net, 'edge_loss_' + side).face = fc
    B = 2
    img = torch.from_numpy(synth.synth_input('loss.img', (B, 3, 256, 256), SEED))
    rng = np.random.RandomState(77)

    def targets_around(outs):
        t, m = {}, {}
        last = outs[2]
        for side in ('left', 'right'):
            c = rng.normal(0, 0.05, (B, 1, 3)).astype(np.float32)
            m['center_' + side] = torch.from_numpy(c)
            for key, pk, n in (('joint_3d_', 'pd_joint_xyz_', 21), ('mesh_3d_', 'pd_mesh_xyz_', 778)):
                p = last[pk + side].detach().numpy()
                noise = np.where(rng.rand(B, n, 3) < 0.5, rng.normal(0, 0.0005, (B, n, 3)), rng.normal(0, 0.01, (B, n, 3)))
                t[key + side] = torch.from_numpy((p + c + noise).astype(np.float32))
            for key, pk, n in (('joint_2d_', 'pd_joint_uv_', 21), ('mesh_2d_', 'pd_mesh_uv_', 778)):
                p = (last[pk + side] if pk + side in last else None)
                p = p.detach().numpy() if p is not None else np.zeros((B, n, 2), np.float32)
                noise = np.where(rng.rand(B, n, 2) < 0.5, rng.normal(0, 0.004, (B, n, 2)), rng.normal(0, 0.05, (B, n, 2)))
                uvd = np.concatenate([p + noise, rng.normal(0, 1, (B, n, 1))], axis=2)
                t[key + side] = torch.from_numpy(uvd.astype(np.float32))
        seg_lo = rng.randint(0, 3, (B, 1, 16, 16))
        seg = np.repeat(np.repeat(seg_lo, 16, axis=2), 16, axis=3).astype(np.uint8)
        flip = rng.rand(B, 1, 256, 256) < 0.02
        seg = np.where(flip, rng.randint(0, 3, seg.shape), seg).astype(np.uint8)
        dense_lo = rng.randint(0, 256, (B, 3, 64, 64))
        dense = np.repeat(np.repeat(dense_lo, 4, axis=2), 4, axis=3).astype(np.uint8)
        t['seg'] = torch.from_numpy(seg.astype(np.float32))
        t['dense'] = torch.from_numpy(dense.astype(np.float32) / np.float32(255.0))
        return t, m, seg, dense

    captured = {}

    def grab(mod, args, out):
        captured.setdefault(mod, []).append(out)
    h0 = net.init_regressor.register_forward_hook(grab)
    h1 = net.decoder.register_forward_hook(grab)
    zeros_t = {k + s: torch.zeros(B, n, 3) for s in ('left', 'right')
               for k, n in (('joint_2d_', 21), ('mesh_2d_', 778), ('joint_3d_', 21), ('mesh_3d_', 778))}
    zeros_t.update(seg=torch.zeros(B, 1, 256, 256), dense=torch.zeros(B, 3, 256, 256))
    zeros_m = {'center_left': torch.zeros(B, 1, 3), 'center_right': torch.zeros(B, 1, 3)}
    with torch.no_grad():
        net({'img': img}, zeros_t, zeros_m)
    iter0 = [captured[net.init_regressor][0]] + captured[net.decoder][0]['result_list']
    target, meta, seg_u8, dense_u8 = targets_around(iter0)
    captured.clear()
    with torch.no_grad():
        outs, loss = net({'img': img}, target, meta)
    h0.remove(); h1.remove()
    iter_outs = [captured[net.init_regressor][0]] + captured[net.decoder][0]['result_list']
    assert len(loss) == 42, len(loss)
    out = {'gt_seg_u8': seg_u8, 'gt_dense_u8': dense_u8}
    for k, v in target.items():
        if k not in ('seg', 'dense'):
            out['gt_' + k] = v
    for k, v in meta.items():
        out['gt_' + k] = v
    for i, o in enumerate(iter_outs):
        for k in ('pd_joint_uv_', 'pd_mesh_uv_', 'pd_joint_xyz_', 'pd_mesh_xyz_'):
            for side in ('left', 'right'):
                out['s%d.%s%s' % (i, k, side)] = o[k + side]
        out['s%d.pd_offset' % i] = o['pd_offset']
    out['seg'] = outs[3]['seg']
    out['dense'] = outs[3]['dense']
    for k, v in loss.items():
        out['loss.' + k] = np.float64(float(v))
        print('   %-20s %.6f' % (k, float(v)))
    save('g8c_loss' if cond else 'g8_loss', **out)


# ----------------------------------------------------------------------------- G12 gradients of the training objective
def gen_loss_grad():
    """d(sum of the 42 terms) / d(predictions): torch autograd through the reference's own loss modules (models/loss.py,
    models/lovasz_loss.py, nn.CrossEntropyLoss) composed as models/dir.py:562-592 composes them -- the composition is checked here
    against the 42 forward values G8 holds from the reference's DIR.forward -- on the G8 predictions as leaf tensors (pd_mesh_uv is
    its own leaf: the projection's chain rule is the caller's).  Run after `loss`."""
    import torch.nn.functional as F
    from models.loss import EdgeLengthLoss, NormalVectorLoss, SmoothL1Loss
    from models.lovasz_loss import lovasz_softmax
    g = dict(np.load(os.path.join(OUT, 'g8_loss.npz')))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    faces = {s_: synth.loss_faces(s_, SEED) for s_ in ('left', 'right')}
    l1 = SmoothL1Loss()
    edge = {s_: EdgeLengthLoss(faces[s_]) for s_ in faces}
    normal = {s_: NormalVectorLoss(faces[s_]) for s_ in faces}
    ce = nn.CrossEntropyLoss(weight=torch.Tensor([.1, 0.45, 0.45]))
    leaves, loss = {}, {}

    def leaf(name):
        t = T(g[name]).clone().requires_grad_(True)
        leaves[name] = t
        return t
    seg, dense = leaf('seg'), leaf('dense')
    gt_seg = T(g['gt_seg_u8'].astype(np.float32))
    gt_dense = T(g['gt_dense_u8'].astype(np.float32) / np.float32(255.0))
    S = seg.shape[-1]
    lab = F.interpolate(gt_seg, (S, S), mode='nearest').long().squeeze(1)
    dd = F.interpolate(gt_dense, (S, S), mode='bilinear')
    loss['seg'] = ce(seg, lab) * 0.1
    loss['dense'] = l1(dense, dd)
    loss['lovasz'] = lovasz_softmax(seg, lab) * 0.1
    cl, cr = T(g['gt_center_left']), T(g['gt_center_right'])
    cen = {'left': cl, 'right': cr}
    gt_off = (cr - cl) / 0.15
    for i in range(3):
        for s_ in ('left', 'right'):
            gj = (T(g['gt_joint_3d_' + s_]) - cen[s_]) / 0.15
            gm = (T(g['gt_mesh_3d_' + s_]) - cen[s_]) / 0.15
            loss['joint_%s_uv_%d' % (s_, i)] = l1(leaf('s%d.pd_joint_uv_%s' % (i, s_)), T(g['gt_joint_2d_' + s_])[:, :, :2]) * 10
            loss['mesh_%s_uv_%d' % (s_, i)] = l1(leaf('s%d.pd_mesh_uv_%s' % (i, s_)), T(g['gt_mesh_2d_' + s_])[:, :, :2]) * 10
            jp = leaf('s%d.pd_joint_xyz_%s' % (i, s_)) / 0.15
            mp = leaf('s%d.pd_mesh_xyz_%s' % (i, s_)) / 0.15
            loss['joint_%s_xyz_%d' % (s_, i)] = l1(jp, gj) * 10
            loss['mesh_%s_xyz_%d' % (s_, i)] = l1(mp, gm) * 10
            loss['edge_%s_%d' % (s_, i)] = edge[s_](mp, gm).mean()
            loss['normal_%s_%d' % (s_, i)] = normal[s_](mp, gm).mean() * 0.1
        loss['offset_%d' % i] = l1(leaf('s%d.pd_offset' % i), gt_off.squeeze(1)) * 10
    assert len(loss) == 42
    for k, v in loss.items():
        want = float(g['loss.' + k])
        assert abs(float(v) - want) <= 1e-6 * max(1.0, abs(want)), (k, float(v), want)     # same composition as DIR.forward
    sum(loss.values()).backward()
    out = {'grad.' + k: v.grad for k, v in leaves.items()}
    for k, v in out.items():
        assert v is not None and bool(torch.isfinite(v).all()), k
    print('   %d gradient tensors, |g|max %.3e' % (len(out), max(float(v.abs().max()) for v in out.values())))
    save('g12_loss_grad', **out)


# ----------------------------------------------------------------------------- G13 gradients through the MANO layer + projection
def gen_mano_grad():
    """torch autograd through the reference's own manopth ManoLayer (manopth/manopth/manolayer.py:110-270) and projection_batch_xy
    (utils/utils.py:47-63), composed as RegressorOffset.forward does (models/dir.py:352-363): d <cotangents, outputs> / d para[64]"""
    from manopth.manolayer import ManoLayer
    from utils.utils import projection_batch_xy
    from oracle.golden_inputs import MANO_GRAD_CASES, mano_grad_inputs
    out = {}
    for side in ('left', 'right'):
        for case, center in MANO_GRAD_CASES:
            layer = ManoLayer(root_rot_mode='6D', joint_rot_mode='axisang', use_pca=True, mano_root='unused', side=side, ncomps=45,
                              center_idx=(None if center < 0 else center), flat_hand_mean=False, robust_rot=True)
            para_np, cot = mano_grad_inputs(case, side)
            para = torch.from_numpy(para_np).requires_grad_(True)
            pose, beta, cam = torch.split(para, [51, 10, 3], dim=-1)
            verts, joints = layer(pose, beta)
            juv = projection_batch_xy(cam[:, 0], cam[:, 1:], joints)
            muv = projection_batch_xy(cam[:, 0], cam[:, 1:], verts)
            tag = '%s_%s_c%d' % (side, case, center)
            for sel in ('all', 'verts', 'joints', 'joint_uv', 'mesh_uv'):
                terms = {'verts': verts, 'joints': joints, 'joint_uv': juv, 'mesh_uv': muv}
                L = sum((terms[k] * torch.from_numpy(cot[k])).sum() for k in terms if sel in ('all', k))
                g, = torch.autograd.grad(L, para, retain_graph=True)
                out['%s.%s' % (tag, sel)] = g
    save('g13_mano_grad', **out)


# ----------------------------------------------------------------------------- G14 gradients through RegressorOffset
def gen_regress_grad():
    """torch autograd through the reference's RegressorOffset (models/dir.py:312-381: three Linears -> two manopth layers -> four
    projections): d <cotangents, outputs> / d (sampled features, the six Linear parameters)."""
    from models.dir import RegressorOffset
    from oracle.golden_inputs import REGRESS_OUT_KEYS, regress_grad_inputs
    net = RegressorOffset(21 * 64, 'unused', 0)
    load_synth(net)              # (the synthetic Linear weights give O(1) parameters: the MANO layers are driven well away from the rest pose)
    ins, cot = regress_grad_inputs()
    t = {k: torch.from_numpy(v) for k, v in ins.items()}
    fl, fr = t['feat_l'].requires_grad_(True), t['feat_r'].requires_grad_(True)
    out = net(fl, fr, t['para_l'], t['para_r'], t['offset'])
    L = sum((out[k] * torch.from_numpy(cot[k])).sum() for k in REGRESS_OUT_KEYS)
    params = {'mano_left.weight': net.mano_left.weight, 'mano_left.bias': net.mano_left.bias, 'mano_right.weight': net.mano_right.weight,
              'mano_right.bias': net.mano_right.bias, 'offset.weight': net.offset.weight, 'offset.bias': net.offset.bias}
    gs = torch.autograd.grad(L, [fl, fr] + list(params.values()))
    res = {'grad.feat_l': gs[0], 'grad.feat_r': gs[1]}
    for k, g in zip(params, gs[2:]):          # the two [64,1408] weight gradients: every 4th column + row / column sums (fixture size)
        if g.dim() == 1 or g.shape[0] == 3:
            res['grad.' + k] = g
        else:
            res['grad.' + k + '.cols4'] = g[:, ::4].contiguous()
            res['grad.' + k + '.rowsum'] = g.double().sum(1)
            res['grad.' + k + '.colsum'] = g.double().sum(0)
    res.update({'out.pd_mano_para_left': out['pd_mano_para_left'].detach(), 'out.pd_mano_para_right': out['pd_mano_para_right'].detach(),
                'out.pd_offset': out['pd_offset'].detach()})
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(OUT, 'manifest_regressor.json'), 'w') as f:
        json.dump(shapes, f, indent=0)
    save('g14_regress_grad', **res)


# ----------------------------------------------------------------------------- G15 gradients through STE
def compact_grads(named, step=16):
    """fixture-size form of a set of parameter gradients: small tensors whole; matrices as every `step`-th column + row / column sums"""
    res = {}
    for k, g in named.items():
        while g.dim() > 2 and g.shape[-1] == 1:             # Conv1d k=1 weights [out, in, 1]
            g = g.squeeze(-1)
        if g.dim() < 2 or g.numel() <= 8192:
            res['grad.' + k] = g
        else:
            g2 = g.reshape(g.shape[0], -1) if (g.dim() == 4 and g.shape[-1] <= 7) else g.reshape(-1, g.shape[-1])     # conv OIHW -> [O, I*kh*kw]
            res['grad.' + k + '.cols%d' % step] = g2[:, ::step].contiguous()
            res['grad.' + k + '.rowsum'] = g2.double().sum(1)
            res['grad.' + k + '.colsum'] = g2.double().sum(0)
    return res


def gen_ste_grad():
    """torch autograd through the reference's STE (transformer/mixSTE.py:159-205): d <gy, STE(x)> / d (x, every parameter)"""
    from transformer.mixSTE import STE
    net = STE(num_joints=42, in_chans=128, out_dim=64, depth=4)
    load_synth(net)
    x0 = torch.from_numpy(synth.synth_input('stegrad.x', (3, 42, 128), SEED))
    gy = torch.from_numpy(synth.synth_input('stegrad.gy', (3, 42, 64), SEED))
    x = x0.clone().requires_grad_(True)
    y = net(x + 0.0)                      # (+ 0: the forward adds the positional embedding in place)
    params = {k: v for k, v in net.named_parameters()}
    gs = torch.autograd.grad((y * gy).sum(), [x] + list(params.values()), allow_unused=True)
    unused = [k for k, g in zip(params, gs[1:]) if g is None]
    assert all(k.startswith('STEblocks.0.') for k in unused) and len(unused) == 12      # block 0 is never executed (:197)
    res = {'y': y.detach(), 'grad.x': gs[0]}
    res.update(compact_grads({k: g for k, g in zip(params, gs[1:]) if g is not None}))
    save('g15_ste_grad', **res)


# ----------------------------------------------------------------------------- G16 gradients through the P-GCN stack (training mode)
def gen_pgcn_grad():
    """torch autograd through the reference's ResSimplePGCN in TRAINING mode (batch-statistics BatchNorm1d, SemGCN/p_gcn.py:20-27,64-73)"""
    from SemGCN.p_gcn import ResSimplePGCN
    from SemGCN.utils import adj_mx_from_edges, get_sketch_setting
    adj = adj_mx_from_edges(21, get_sketch_setting(), sparse=False, eye=False)
    net = ResSimplePGCN(adj, 128, num_layers=4)
    load_synth(net)
    net.train()
    x0 = torch.from_numpy(synth.synth_input('pgcngrad.x', (5, 21, 128), SEED))
    gy = torch.from_numpy(synth.synth_input('pgcngrad.gy', (5, 21, 128), SEED))
    x = x0.clone().requires_grad_(True)
    y = net(x)
    params = {k: v for k, v in net.named_parameters()}
    gs = torch.autograd.grad((y * gy).sum(), [x] + list(params.values()))
    res = {'y': y.detach(), 'grad.x': gs[0]}
    res.update(compact_grads(dict(zip(params, gs[1:])), step=32))
    for k, v in net.state_dict().items():
        if 'running_' in k:
            res['after.' + k] = v
    save('g16_pgcn_grad', **res)


# ----------------------------------------------------------------------------- G17 gradients through one stage's token half (training mode)
STAGE_GRAD_COT = ('pd_offset', 'pd_mano_para_left', 'pd_mano_para_right', 'pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left',
                  'pd_joint_xyz_right', 'pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_uv_left', 'pd_mesh_uv_right')


def gen_stage_grad():
    """torch autograd through the reference's Joint2BoneFeature.forward (models/dir.py:86-116) in TRAINING mode: L = sum <cot_k, result_k>
    over the regressor's outputs + <cot, STE output>; gradients w.r.t. img_feat and every parameter on the token path"""
    from models.dir import Joint2BoneFeature
    from oracle.golden_inputs import stage_grad_inputs
    S = 16
    net = Joint2BoneFeature(256, 128, 64, 21, S, 'unused', 0, distance=1)
    load_synth(net)
    net.train()
    ins, cot = stage_grad_inputs(S)
    ins = [torch.from_numpy(a) for a in ins]
    feat = ins[0].clone().requires_grad_(True)
    tap = {}
    net.interaction.register_forward_hook(lambda m, a, o: tap.__setitem__('ste', o))
    result, feats = net(feat, *ins[1:])
    L = (tap['ste'] * torch.from_numpy(cot['joint_feat'])).sum()
    for k in STAGE_GRAD_COT:
        L = L + (result[k] * torch.from_numpy(cot[k])).sum()
    params = {k: v for k, v in net.named_parameters() if not (k.startswith('proj_feat_emb') or k.startswith('fusion'))}
    gs = torch.autograd.grad(L, [feat] + list(params.values()), allow_unused=True)
    res = {('out.' + k): result[k].detach() for k in STAGE_GRAD_COT}
    res['out.joint_feat'] = tap['ste'].detach()
    res['gfeat.ch4'] = gs[0][:, ::4].contiguous()               # every 4th channel + the channel sums of the gradient into fusion_feat
    res['gfeat.chsum'] = gs[0].double().sum(1)
    res['gfeat.abssum'] = gs[0].double().abs().sum(1)
    named = {k: g for k, g in zip(params, gs[1:]) if g is not None}
    unused = sorted(k for k, g in zip(params, gs[1:]) if g is None)
    print('   parameters without gradient:', unused)
    res.update(compact_grads(named, step=32))
    for k, v in net.state_dict().items():
        if 'running_' in k and not (k.startswith('proj_feat_emb') or k.startswith('fusion')):
            res['after.' + k] = v
    save('g17_stage_grad', **res)


# ----------------------------------------------------------------------------- G18 / G19 gradients through the image half's residual blocks
BLOCK_CASES = {  # name: (kind, constructor arguments, input shape NCHW)
    'bneck_plain': ('bottleneck', dict(inplanes=256, planes=64, stride=1, downsample=False), (3, 256, 16, 16)),
    'bneck_down': ('bottleneck', dict(inplanes=256, planes=128, stride=2, downsample=True), (3, 256, 16, 16)),
    'res_skip': ('residual', dict(inp_dim=512, out_dim=256), (3, 512, 16, 16)),
    'res_same': ('residual', dict(inp_dim=256, out_dim=256), (3, 256, 16, 16)),
}


def make_block(kind, kw):
    if kind == 'bottleneck':
        from models.backbone.resnet import Bottleneck, conv1x1
        ds = None
        if kw['downsample']:
            ds = torch.nn.Sequential(conv1x1(kw['inplanes'], kw['planes'] * 4, kw['stride']), torch.nn.BatchNorm2d(kw['planes'] * 4))
        return Bottleneck(kw['inplanes'], kw['planes'], kw['stride'], ds)
    from models.backbone.hourglass import Residual
    return Residual(kw['inp_dim'], kw['out_dim'])


def gen_block_grad():
    """torch autograd through the reference's Bottleneck (models/backbone/resnet.py:86-142) and Residual (models/backbone/hourglass.py:33-70)
    in TRAINING mode (batch-statistics BatchNorm2d): d <gy, block(x)> / d (x, every parameter), and the running statistics after"""
    shapes_all = {}
    for name, (kind, kw, xs) in BLOCK_CASES.items():
        net = make_block(kind, kw)
        shapes_all[name] = {k: list(v) for k, v in load_synth(net).items()}
        net.train()
        x0 = torch.from_numpy(synth.synth_input('blockgrad.%s.x' % name, xs, SEED))
        x = x0.clone().requires_grad_(True)
        y = net(x + 0.0)                                   # (+ 0: the blocks end in in-place ops on views of their input chain)
        gy = torch.from_numpy(synth.synth_input('blockgrad.%s.gy' % name, tuple(y.shape), SEED))
        params = {k: v for k, v in net.named_parameters()}
        gs = torch.autograd.grad((y * gy).sum(), [x] + list(params.values()), allow_unused=True)
        res = {'y.ch8': y.detach()[:, ::8].contiguous(), 'y.chsum': y.detach().double().sum(1),
               'gx.ch8': gs[0][:, ::8].contiguous(), 'gx.chsum': gs[0].double().sum(1), 'gx.abssum': gs[0].double().abs().sum(1)}
        named = {k: g for k, g in zip(params, gs[1:]) if g is not None}
        print('   %s: parameters without gradient: %s' % (name, sorted(k for k, g in zip(params, gs[1:]) if g is None)))
        res.update(compact_grads(named, step=16))
        for k, v in net.state_dict().items():
            if 'running_' in k:
                res['after.' + k] = v
        save('g18_block_grad_' + name, **res)
    with open(os.path.join(OUT, 'manifest_blocks.json'), 'w') as f:
        json.dump(shapes_all, f, indent=0)


# ----------------------------------------------------------------------------- G19 gradients through bone_proj
def gen_bone_grad():
    """torch autograd through the reference's Joint2BoneFeature.bone_proj (models/dir.py:146-174): d <g, img> / d (joint_uv, joint_feat)"""
    from models.dir import Joint2BoneFeature
    from oracle.golden_inputs import bone_grad_inputs
    res = {}
    for S, dist in ((16, 1), (32, 2)):
        net = Joint2BoneFeature(256, 128, 64, 21, S, 'unused', 0, distance=dist)
        uv, feat, g = [torch.from_numpy(a) for a in bone_grad_inputs(S)]
        uv, feat = uv.clone().requires_grad_(True), feat.clone().requires_grad_(True)
        img = net.bone_proj(uv, feat)                                      # [B, 1280, S, S]
        gu, gf = torch.autograd.grad((img * g).sum(), [uv, feat])
        res['S%d.g_uv' % S], res['S%d.g_feat' % S] = gu, gf
        res['S%d.img.sum' % S] = img.detach().double().sum((2, 3))
        print('   S=%d: mask covers %.3f of the map, |g uv| max %.3e' % (S, float((img.detach() != 0).float().mean()), float(gu.abs().max())))
    save('g19_bone_grad', **res)


# ----------------------------------------------------------------------------- G20 the whole training step's gradient
def gen_full_grad(cond=False):
    """(cond: G20c, on the trained-like parameters and G8c's targets: there the reference's own fp32 gradient agrees with its float64
    evaluation to ~1e-5, so the whole step can be pinned at 1e-4 of each gradient's maximum.)  sum(loss.values()).backward() through the reference's own DIR in training mode (train.py:66-68) on G8's input, targets and faces.
    With random weights the network is badly conditioned (seg logits ~ 1e5, BatchNorm backward cancels most of its input), so the fp32
    gradient carries visible evaluation noise: the fixture holds the gradient of the SAME graph evaluated in float64 (compact form) and,
    per parameter, how far the reference's own fp32 evaluation is from it ('ref32_err.<key>', relative to the gradient's maximum) --
    the yardstick for any other fp32 implementation."""
    from models.dir import DIR
    g8 = np.load(os.path.join(OUT, 'g8c_loss.npz' if cond else 'g8_loss.npz'))

    def run(dtype):
        net = DIR(21, 'unused', 0)
        load_synth(net, cond=cond)
        net.train()
        for side in ('left', 'right'):
            fc = torch.from_numpy(synth.loss_faces(side, SEED))
            getattr(net, 'normal_loss_' + side).face = fc
            getattr(net, 'edge_loss_' + side).face = fc
        net = net.to(dtype)
        for m in net.modules():                            # plain tensor attributes (PGraphConv's adjacency, the loss modules' tables) follow
            for name, val in list(vars(m).items()):
                if torch.is_tensor(val) and val.dtype == torch.float32:
                    setattr(m, name, val.to(dtype))
        img = torch.from_numpy(synth.synth_input('loss.img', (2, 3, 256, 256), SEED)).to(dtype)
        target = {k[3:]: torch.from_numpy(g8[k]).to(dtype) for k in g8.files if k.startswith('gt_') and not k.endswith('_u8') and 'center' not in k}
        target['seg'] = torch.from_numpy(g8['gt_seg_u8'].astype(np.float32)).to(dtype)
        target['dense'] = torch.from_numpy(g8['gt_dense_u8'].astype(np.float32) / np.float32(255.0)).to(dtype)
        meta = {k[3:]: torch.from_numpy(g8[k]).to(dtype) for k in g8.files if k.startswith('gt_center')}
        keep_float = torch.Tensor.float
        if dtype == torch.float64:                         # lovasz_loss.py:193 builds its foreground mask with .float(): promoted for the float64 pass
            torch.Tensor.float = lambda self, *a, **k: self.double()
        captured = {}
        h0 = net.init_regressor.register_forward_hook(lambda m, a, o: captured.__setitem__('init', o))
        h1 = net.decoder.register_forward_hook(lambda m, a, o: captured.__setitem__('dec', o))
        inter = {}
        try:
            outs, loss = net({'img': img}, target, meta)
            assert len(loss) == 42
            stages = [captured['init']] + captured['dec']['result_list']
            for i, st in enumerate(stages):                # gradients at the stage outputs: localise a mismatch without the whole net
                for k in ('pd_mano_para_left', 'pd_mano_para_right', 'pd_mesh_xyz_left', 'pd_joint_uv_left', 'pd_offset'):
                    st[k].retain_grad()
                    inter['s%d.%s' % (i, k)] = st[k]
            total = sum(loss[k] for k in loss)
            total.backward()
        finally:
            torch.Tensor.float = keep_float
            h0.remove(); h1.remove()
        net.inter = {k: v.grad for k, v in inter.items()}
        return net, loss, total
    net, loss, total = run(torch.float32)
    for k, v in loss.items():                              # the same pass G8 pinned (batch-statistics BN: independent of the running statistics)
        assert abs(float(v) - float(g8['loss.' + k])) <= 2e-5 * max(1.0, abs(float(g8['loss.' + k]))), (k, float(v), float(g8['loss.' + k]))
    named = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    none = sorted(k for k, p in net.named_parameters() if p.grad is None)
    print('   %d parameters with gradient, %d without: %s ...' % (len(named), len(none), none[:6]))
    if cond:
        # Yardstick on the trained-like parameters: the reference's fp32 gradient is itself reproducible only to PERCENTS under a change of
        # summation order -- the same graph, the same fp32 kernels, 1 BLAS / oneDNN thread instead of 8 (tools/ref_grad_sensitivity.py: median
        # 2e-2 .. 4e-2 of each tensor's maximum at B = 2 and B = 8, against 4e-5 with the BatchNorm layers in eval mode; two runs at the
        # same thread count are bit-identical).  The training-mode BatchNorm backward of this 70-layer network amplifies fp32 rounding by
        # ~1e5, whatever the weights' conditioning; the float64 evaluation sits the same distance away.  So the fixture stores the
        # 8-thread fp32 gradient and, per parameter, its distance to the 1-thread one: the only honest tolerance for a third fp32
        # implementation.
        torch.set_num_threads(1)
        net1, _, _ = run(torch.float32)
        torch.set_num_threads(8)
        named1 = {k: p.grad for k, p in net1.named_parameters() if p.grad is not None}
        res = {'total': total.detach(), 'none': np.array(none)}
        res.update({'inter.' + k: v.float() for k, v in net.inter.items()})
        rep = []
        for k in named:
            e = float((named[k] - named1[k]).abs().max() / (named[k].abs().max() + 1e-30))
            res['ref_repro.' + k] = np.float64(e)
            rep.append(e)
        print('   reference fp32, 8 threads vs 1 thread: median %.2e of each gradient maximum' % float(np.median(rep)))
        res.update({'g32.' + k: (v.float() if torch.is_tensor(v) else v) for k, v in compact_grads_sized(named, coarse=2).items()})
        for k, v in net.state_dict().items():
            if 'running_' in k:
                res['after.' + k] = v
        save('g20c_full_grad', **res)
        return
    net64, loss64, total64 = run(torch.float64)
    named64 = {k: p.grad for k, p in net64.named_parameters() if p.grad is not None}
    res = {'total': total.detach(), 'total64': total64.detach()}
    res.update({'inter.' + k: v.float() for k, v in net64.inter.items()})
    errs = []
    for k in named:
        e = float((named[k].double() - named64[k]).abs().max() / (named64[k].abs().max() + 1e-300))
        res['ref32_err.' + k] = np.float64(e)
        errs.append((e, k))
    errs.sort(reverse=True)
    print('   reference fp32 vs float64 evaluation of its own gradient: median %.2e, worst %s' % (errs[len(errs) // 2][0], errs[:5]))
    res.update({k: (v.float() if torch.is_tensor(v) else v) for k, v in compact_grads_sized(named64).items()})       # fp32 storage: the yardstick is ~1e-2
    for k, v in net.state_dict().items():
        if 'running_' in k:
            res['after.' + k] = v
    res['none'] = np.array(none)
    save('g20_full_grad', **res)


def gen_full_grad_frozen():
    """G20e: the reference's `sum(loss.values()).backward()` (train.py:66-68) through its own DIR in training mode with every
    BatchNorm module in .eval() (running statistics, no batch statistics), on the trained-like parameters and G8c's input / targets.  In this form
    the reference's fp32 gradient IS reproducible under a change of summation order (8 BLAS / oneDNN threads vs 1: stored per parameter as
    'ref_repro.<key>', median ~4e-5 of each tensor's maximum -- against 2e-2 .. 4e-2 with training-mode BatchNorm, G20c), so every other fp32
    implementation of everything EXCEPT the batch-statistics BatchNorm backward can be pinned tightly by it."""
    from models.dir import DIR
    g8 = np.load(os.path.join(OUT, 'g8c_loss.npz'))

    def run():
        net = DIR(21, 'unused', 0)
        load_synth(net, cond=True)
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        for side in ('left', 'right'):
            fc = torch.from_numpy(synth.loss_faces(side, SEED))
            getattr(net, 'normal_loss_' + side).face = fc
            getattr(net, 'edge_loss_' + side).face = fc
        img = torch.from_numpy(synth.synth_input('loss.img', (2, 3, 256, 256), SEED))
        target = {k[3:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith('gt_') and not k.endswith('_u8') and 'center' not in k}
        target['seg'] = torch.from_numpy(g8['gt_seg_u8'].astype(np.float32))
        target['dense'] = torch.from_numpy(g8['gt_dense_u8'].astype(np.float32) / np.float32(255.0))
        meta = {k[3:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith('gt_center')}
        outs, loss = net({'img': img}, target, meta)
        assert len(loss) == 42
        total = sum(loss[k] for k in loss)
        total.backward()
        return net, loss, total
    net, loss, total = run()
    named = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    none = sorted(k for k, p in net.named_parameters() if p.grad is None)
    torch.set_num_threads(1)
    net1, _, _ = run()
    torch.set_num_threads(8)
    named1 = {k: p.grad for k, p in net1.named_parameters() if p.grad is not None}
    res = {'total': total.detach(), 'none': np.array(none)}
    res.update({'loss.' + k: v.detach() for k, v in loss.items()})
    rep = []
    for k in named:
        e = float((named[k] - named1[k]).abs().max() / (named[k].abs().max() + 1e-30))
        res['ref_repro.' + k] = np.float64(e)
        rep.append(e)
    print('   frozen BatchNorm: %d parameters with gradient; reference fp32, 8 threads vs 1 thread: median %.2e, worst %.2e of each gradient maximum'
          % (len(named), float(np.median(rep)), max(rep)))
    res.update({'g32.' + k: (v.float() if torch.is_tensor(v) else v) for k, v in compact_grads_sized(named, coarse=2).items()})
    save('g20e_full_grad_frozen_bn', **res)


def compact_grads_sized(named, coarse=1):
    """compact_grads with the column step growing with the tensor: <= ~1024 sampled values per row block"""
    res = {}
    for k, g in named.items():
        n = g.numel()
        step = coarse * (16 if n <= (1 << 16) else 64 if n <= (1 << 20) else 512)
        res.update(compact_grads({k: g}, step=step))
    return res


# ----------------------------------------------------------------------------- G7x / G21: N more refinement stages, from the reference's OWN modules
def reference_with_extra_stages(n):
    """SURVEY.md 8f rank 4 (config 5: "5 refinement iters").  The reference's decoder hard-wires two stages (models/dir.py:395,401), so there is no
    reference NETWORK with more -- but there are the reference's MODULES: this builds its DIR, then gives the decoder n more
    `Joint2BoneFeature(256, 128, 64, joint_num, 32, ..., distance=2)` (the class and arguments of projecter_3, models/dir.py:401) and
    `Residual(512, 256)` (enhance_layer3, :402) under the names this build's mirror uses (decoder.projecter_x.<i> / decoder.enhance_layer_x.<i>),
    and runs them after the reference's own forward exactly as that forward chains its two stages (models/dir.py:449-472: previous stage's
    predictions detached, cat(running map, img_feat) -> Residual).  The first two stages ARE the reference's forward (super().forward); the map
    they end on is taken from enhance_layer3's output by a hook.  DIR.forward and its loss loop (models/dir.py:521-594) take any number of stages."""
    from models import dir as RD

    class DecoderX(RD.FusionJointInterIterDecoder):
        def __init__(self, joint_num, mano_pth, root_joint):
            super().__init__(joint_num, mano_pth, root_joint)
            self.projecter_x = nn.ModuleList(RD.Joint2BoneFeature(256, 128, 64, joint_num, 32, mano_pth, root_joint, distance=2) for _ in range(n))
            self.enhance_layer_x = nn.ModuleList(RD.Residual(512, 256) for _ in range(n))
            self._map = None
            self.enhance_layer3.register_forward_hook(lambda m, a, o: setattr(self, '_map', o))

        def forward(self, x, result_dict):
            out = super().forward(x, result_dict)               # the reference's two stages (its seg / dense of THAT map are discarded below)
            feat_map, outputs = self._map, list(out['result_list'])
            prev = outputs[-1]
            for proj, enh in zip(self.projecter_x, self.enhance_layer_x):
                res, out_feat = proj(feat_map, prev['pd_joint_xyz_left'].detach(), prev['pd_joint_xyz_right'].detach(),
                                     prev['pd_joint_uv_left'].detach(), prev['pd_joint_uv_right'].detach(),
                                     prev['pd_mano_para_left'].detach(), prev['pd_mano_para_right'].detach(), prev['pd_offset'].detach().unsqueeze(1))
                feat_map = enh(torch.cat((feat_map, out_feat['img_feat']), dim=1))
                outputs.append(dict(res, **out_feat))
                prev = res
            feat = self.conv_final(feat_map)
            return {'result_list': outputs, 'seg': self.seg(feat), 'dense': self.dense(feat), 'proj_feat': outputs[-1]['vis_img_feat']}

    net = RD.DIR(21, 'unused', 0)
    net.decoder = DecoderX(21, 'unused', 0)
    return net


def gen_full_extra():
    """G7x: the composed reference (reference_with_extra_stages(2): config 5's five refinement iterations) in eval mode on the trained-like
    parameters and G7's input -- pins this build's N-stage extension (engine, mirror and numpy oracle) to the reference's own classes."""
    net = reference_with_extra_stages(2).eval()
    shapes = load_synth(net, cond=True)
    assert len(shapes) == 963 + 2 * 240, len(shapes)
    img = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), SEED))
    with torch.no_grad():
        outs, loss = net({'img': img}, None, None)
    assert loss == {} and len(outs) == 6
    out = {}
    for i in range(5):
        for k, v in outs[i].items():
            if v is not None:
                out['s%d.%s' % (i, k)] = v
    out['seg'], out['dense'] = outs[5]['seg'], outs[5]['dense']
    pf = outs[5]['proj_feat']
    out['proj_feat.sum'] = pf.double().sum(dim=(2, 3))
    out['proj_feat.slice'] = pf[:, 0:1280:97]
    for i in range(5):
        print('   stage %d verts absmax %.4f, moved by %.3e m from the stage before' % (i, float(outs[i]['pd_mesh_xyz_left'].abs().max()),
              float((outs[i]['pd_mesh_xyz_left'] - outs[max(i - 1, 0)]['pd_mesh_xyz_left']).abs().max())))
    save('g7x_dir_extra2', **out)


def gen_full_grad_frozen_extra():
    """G21: G20e's gradient (the reference's `sum(loss.values()).backward()` with every BatchNorm in .eval(), reproducible to ~4e-5) through the
    composed reference with ONE extra stage: 55 loss terms, gradients of all parameters including decoder.projecter_x.0.* / enhance_layer_x.0.*."""
    g8 = np.load(os.path.join(OUT, 'g8c_loss.npz'))

    def run():
        net = reference_with_extra_stages(1)
        load_synth(net, cond=True)
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        for side in ('left', 'right'):
            fc = torch.from_numpy(synth.loss_faces(side, SEED))
            getattr(net, 'normal_loss_' + side).face = fc
            getattr(net, 'edge_loss_' + side).face = fc
        img = torch.from_numpy(synth.synth_input('loss.img', (2, 3, 256, 256), SEED))
        target = {k[3:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith('gt_') and not k.endswith('_u8') and 'center' not in k}
        target['seg'] = torch.from_numpy(g8['gt_seg_u8'].astype(np.float32))
        target['dense'] = torch.from_numpy(g8['gt_dense_u8'].astype(np.float32) / np.float32(255.0))
        meta = {k[3:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith('gt_center')}
        outs, loss = net({'img': img}, target, meta)
        assert len(loss) == 3 + 13 * 4, len(loss)
        total = sum(loss[k] for k in loss)
        total.backward()
        return net, loss, total
    net, loss, total = run()
    named = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    none = sorted(k for k, p in net.named_parameters() if p.grad is None)
    torch.set_num_threads(1)
    net1, _, _ = run()
    torch.set_num_threads(8)
    named1 = {k: p.grad for k, p in net1.named_parameters() if p.grad is not None}
    res = {'total': total.detach(), 'none': np.array(none)}
    res.update({'loss.' + k: v.detach() for k, v in loss.items()})
    rep = []
    for k in named:
        e = float((named[k] - named1[k]).abs().max() / (named[k].abs().max() + 1e-30))
        res['ref_repro.' + k] = np.float64(e)
        rep.append(e)
    print('   one extra stage, frozen BatchNorm: %d parameters with gradient (%d of the extra stage); reference fp32, 8 threads vs 1: median %.2e, worst %.2e'
          % (len(named), sum('_x.0.' in k for k in named), float(np.median(rep)), max(rep)))
    res.update({'g32.' + k: (v.float() if torch.is_tensor(v) else v) for k, v in compact_grads_sized(named, coarse=2).items()})
    save('g21_full_grad_frozen_bn_extra1', **res)


GENS = {'full_extra': gen_full_extra, 'full_grad_frozen_extra': gen_full_grad_frozen_extra, 'full_grad_frozen': gen_full_grad_frozen, 'full_cond': lambda: gen_full(True), 'loss_cond': lambda: gen_loss(True), 'full_grad_cond': lambda: gen_full_grad(True),
        'full_grad': gen_full_grad, 'bone_grad': gen_bone_grad, 'block_grad': gen_block_grad, 'stage_grad': gen_stage_grad, 'pgcn_grad': gen_pgcn_grad, 'ste_grad': gen_ste_grad, 'regress_grad': gen_regress_grad, 'mano_grad': gen_mano_grad, 'mano': gen_mano, 'pgcn': gen_pgcn, 'ste': gen_ste, 'grid': gen_grid, 'bone': gen_bone,
        'stage': gen_stage, 'full': gen_full, 'eval': gen_eval, 'gtmano': gen_gtmano, 'imgprep': gen_imgprep, 'loss': gen_loss, 'loss_grad': gen_loss_grad}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*', default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    import warnings
    warnings.filterwarnings('ignore')
    for name, fn in GENS.items():
        if args.only and name not in args.only:
            continue
        print('[gen_golden] ' + name)
        fn()
