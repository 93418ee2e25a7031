"""Pins the CPU oracle (oracle/*.py) against golden vectors produced by the reference itself
(oracle/gen_golden.py imports /root/reference; fixtures in tests/golden/).  CPU-only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, maxabs, relerr
from dir_amd import synth
from oracle import mano as OM
from oracle import nnops as N
from oracle import tokens as OT
from oracle.dir_forward import dir_forward

SEED = 1234


def shapes_of(name, key=None):
    with open(os.path.join(GOLDEN, name)) as f:
        d = json.load(f)
    return {k: tuple(v) for k, v in (d if key is None else d[key]).items()}


def sub_shapes(shapes, prefix):
    return {k[len(prefix):]: v for k, v in shapes.items() if k.startswith(prefix)}


# ------------------------------------------------------------------ G1 MANO (a8)
def _mano_tags(g):
    return sorted({k.rsplit('.', 1)[0] for k in g if k.endswith('.pose')})


def test_mano_matches_reference(golden):
    g = golden('g1_mano')
    tags = _mano_tags(g)
    assert len(tags) == 22
    for tag in tags:
        side, c, f, case = tag.split('_', 3)
        center = int(c[1:])
        buf = synth.mano_buffers(side, SEED, flat_hand_mean=bool(int(f[1:])))
        v, j = OM.mano_forward(buf, g[tag + '.pose'], g[tag + '.betas'], side, None if center < 0 else center)
        # metres; 1e-4 mm == 1e-7 m is the north-star tolerance.  'large' multiplies the pose by 5
        # (angles of tens of radians): fp32 sin/cos argument rounding alone is ~1e-6 there.
        tol = 1e-7 if case != 'large' else 2e-6
        assert maxabs(v, g[tag + '.verts']) < tol, tag
        assert maxabs(j, g[tag + '.joints']) < tol, tag


def test_mano_intermediates(golden):
    g = golden('g1_mano')
    for side in ('left', 'right'):
        tag = side + '_c0_f0_normal'
        buf = synth.mano_buffers(side, SEED)
        _, _, aux = OM.mano_forward(buf, g[tag + '.pose'], g[tag + '.betas'], side, 0, return_aux=True)
        assert maxabs(aux['root_rot'], g[tag + '.root_rot']) < 5e-7
        assert maxabs(aux['rot_map'], g[tag + '.rot_map']) < 5e-7


def test_mano_fp64_arbitration(golden):
    """fp64 oracle vs the reference's fp32 output: bounds the reference's own rounding error."""
    g = golden('g1_mano')
    tag = 'right_c0_f0_normal'
    buf = synth.mano_buffers('right', SEED)
    v, j = OM.mano_forward(buf, g[tag + '.pose'].astype(np.float64), g[tag + '.betas'].astype(np.float64), 'right', 0)
    assert maxabs(v, g[tag + '.verts']) < 1e-7


def test_mano_index_tables():
    assert OM.TIPS['right'] == [745, 317, 444, 556, 673] and OM.TIPS['left'] == [745, 317, 445, 556, 673]
    assert sorted(OM.REORDER_J) == list(range(21)) and sorted(OM.REORDER_T) == list(range(16))


# ------------------------------------------------------------------ G2 P-GCN (a5)
def _pgcn_shapes(prefix=''):
    s = {}
    for i in range(4):
        p = '%sgconv_layers.%d.' % (prefix, i)
        s.update({p + 'gconv.W': (2, 21, 128, 128), p + 'gconv.e_0': (1, 21), p + 'gconv.e_1': (1, 40),
                  p + 'gconv.bias': (128,), p + 'bn.weight': (128,), p + 'bn.bias': (128,),
                  p + 'bn.running_mean': (128,), p + 'bn.running_var': (128,), p + 'bn.num_batches_tracked': ()})
    return s


def test_pgcn_matches_reference(golden):
    g = golden('g2_pgcn')
    mask = OT.adjacency_mask()
    assert np.array_equal(mask, g['adj'] > 0)
    assert np.array_equal(np.stack(np.nonzero(mask), 1), g['edge_order'])     # row-major nonzero order
    sd = synth.synth_state_dict(_pgcn_shapes(), SEED)
    P = N.Params(sd)
    assert maxabs(OT.edge_softmax(P['gconv_layers.0.gconv.e_1'], mask), g['A1']) < 1e-7
    acts = []
    y = OT.pgcn_stack(g['x'], P, collect=acts)
    for i in range(4):
        assert relerr(acts[i], g['layer%d' % i]) < 2e-6
    assert relerr(y, g['y']) < 2e-6
    y0 = OT.pgraphconv(g['x'], P.sub('gconv_layers.0.gconv'))
    assert relerr(y0, g['gconv0']) < 2e-6
    single = synth.synth_state_dict({'W': (2, 21, 128, 128), 'e_0': (1, 21), 'e_1': (1, 40), 'bias': (128,)}, SEED)
    assert relerr(OT.pgraphconv(g['x'], N.Params(single)), g['single_y']) < 2e-6


# ------------------------------------------------------------------ G3 STE (a6)
def ste_shapes():
    s = {'spatial_pos_embed': (1, 42, 128), 'spatial_norm.weight': (128,), 'spatial_norm.bias': (128,),
         'head.0.weight': (128,), 'head.0.bias': (128,), 'head.1.weight': (64, 128), 'head.1.bias': (64,)}
    for i in range(4):
        p = 'STEblocks.%d.' % i
        s.update({p + 'norm1.weight': (128,), p + 'norm1.bias': (128,), p + 'norm2.weight': (128,),
                  p + 'norm2.bias': (128,), p + 'attn.qkv.weight': (384, 128), p + 'attn.qkv.bias': (384,),
                  p + 'attn.proj.weight': (128, 128), p + 'attn.proj.bias': (128,),
                  p + 'mlp.fc1.weight': (256, 128), p + 'mlp.fc1.bias': (256,),
                  p + 'mlp.fc2.weight': (128, 256), p + 'mlp.fc2.bias': (128,)})
    return s


def test_ste_matches_reference(golden):
    g = golden('g3_ste')
    sd = synth.synth_state_dict(ste_shapes(), SEED)
    col = {}
    y = OT.ste_forward(g['x'], N.Params(sd), collect=col)
    assert maxabs(col['probs'], g['attn_probs_block1']) < 1e-6
    for i in range(3):
        assert maxabs(col['after_norm'][i], g['after_norm%d' % i]) < 2e-5
    assert maxabs(y, g['y']) < 2e-5


# ------------------------------------------------------------------ G4 grid-sample tokens (a4)
def test_grid_tokens_match_reference(golden):
    g = golden('g4_grid')
    shapes = {'filters.0.weight': (128, 256, 1), 'filters.0.bias': (128,), 'filters.1.weight': (128,),
              'filters.1.bias': (128,), 'filters.1.running_mean': (128,), 'filters.1.running_var': (128,),
              'filters.1.num_batches_tracked': (), 'filters.3.weight': (128, 128, 1), 'filters.3.bias': (128,)}
    P = N.Params(synth.synth_state_dict(shapes, SEED))
    for S in (16, 32):
        feat = synth.synth_input('grid.feat%d' % S, (2, 256, S, S), SEED)
        uv = g['S%d.uv' % S]
        assert maxabs(N.grid_sample_points(feat, uv), g['S%d.sampled' % S]) < 2e-6
        y = OT.img2joint(feat, uv, P)                                   # [B,21,128]
        ref = g['S%d.y' % S].reshape(2, 128, 21).transpose(0, 2, 1)     # models/dir.py:94
        assert maxabs(y, ref) < 1e-5


# ------------------------------------------------------------------ G5 bone_proj (a10)
def test_bone_proj_matches_reference(golden):
    g = golden('g5_bone')
    for S, dist in ((16, 1), (32, 2)):
        feat = synth.synth_input('bone.feat%d' % S, (2, 21, 64), SEED)
        y = OT.bone_proj(g['S%d.uv' % S], feat, S, dist)
        ref = g['S%d.y' % S]
        assert y.shape == ref.shape == (2, 1280, S, S)
        assert np.array_equal(y != 0, ref != 0), 'mask differs (S=%d)' % S      # bit-exact support
        assert not np.isnan(ref).any()
        assert maxabs(y, ref) < 1e-6
        frac = float((ref != 0).mean())
        assert 0.005 < frac < 0.5


# ------------------------------------------------------------------ G6 refinement stage
@pytest.mark.parametrize('S,dist', [(16, 1), (32, 2)])
def test_stage_matches_reference(golden, S, dist):
    g = golden('g6_stage%d' % S)
    shapes = shapes_of('manifest_stage%d.json' % S)
    sd = synth.synth_state_dict(shapes, SEED)
    p = 'stage%d.' % S
    B = 2
    ins = dict(img_feat=synth.synth_input(p + 'img_feat', (B, 256, S, S), SEED),
               xyz_l=synth.synth_input(p + 'xyz_l', (B, 21, 3), SEED) * np.float32(0.05),
               xyz_r=synth.synth_input(p + 'xyz_r', (B, 21, 3), SEED) * np.float32(0.05),
               para_l=synth.synth_input(p + 'para_l', (B, 64), SEED),
               para_r=synth.synth_input(p + 'para_r', (B, 64), SEED),
               offset=synth.synth_input(p + 'offset', (B, 1, 3), SEED))
    from oracle.golden_inputs import bone_uv
    uv_l, uv_r = bone_uv(p + 'uv_l', B, S), bone_uv(p + 'uv_r', B, S)
    res, ft = OT.stage_forward(N.Params(sd), S, dist, ins['img_feat'], ins['xyz_l'], ins['xyz_r'], uv_l, uv_r,
                               ins['para_l'], ins['para_r'], ins['offset'])
    assert maxabs(res['pd_mano_para_left'], g['pd_mano_para_left']) < 2e-5
    assert maxabs(res['pd_offset'], g['pd_offset']) < 2e-5
    for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right'):
        assert maxabs(res[k], g[k]) < 2e-6, k            # metres (fp32 token path upstream of MANO)
    for k in ('pd_joint_uv_left', 'pd_joint_uv_right'):
        assert maxabs(res[k], g[k]) < 2e-5, k
    assert maxabs(ft['joint_feat_left'], g['joint_feat_left']) < 2e-5
    assert relerr(ft['img_feat'], g['img_feat']) < 2e-5
    assert relerr(ft['vis_img_feat'].astype(np.float64).sum((2, 3)), g['vis_sum']) < 1e-4


# ------------------------------------------------------------------ G7 full DIR.forward
@pytest.mark.parametrize('cond', [False, True])
def test_full_dir_matches_reference(golden, cond):
    """cond: G7c -- the reference's forward on the trained-like flavour of the synthetic parameters (activations O(1) in every layer)"""
    g = golden('g7c_dir' if cond else 'g7_dir')
    shapes = shapes_of('manifest_dir.json')
    assert len(shapes) == 963
    sd = synth.synth_state_dict(shapes, SEED, cond=cond)
    img = synth.synth_input('dir.img', (2, 3, 256, 256), SEED)
    taps = {}
    outs = dir_forward(sd, img, taps=taps)
    for name in ('stem', 'c1', 'c2', 'c3', 'c4', 'skip4', 'fusion4', 'proj4', 'enh4', 'fusion3', 'proj3', 'enh3',
                 'final'):
        assert relerr(taps[name][:, :4], g[name + '.slice']) < 2e-4, name
        assert relerr(np.abs(taps[name]).astype(np.float64).sum((2, 3)), g[name + '.abssum']) < 1e-4, name
    worst = 0.0
    for i in range(3):
        for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right'):
            worst = max(worst, maxabs(outs[i][k], g['s%d.%s' % (i, k)]))
        for k in ('pd_joint_uv_left', 'pd_joint_uv_right', 'pd_proj_left', 'pd_proj_right', 'pd_offset'):
            assert maxabs(outs[i][k], g['s%d.%s' % (i, k)]) < 5e-4, (i, k)
        assert outs[i]['pd_rel_joint'] is None
    # end-to-end fp32 with a different (BLAS) summation order than ATen: positions agree to ~1e-3 mm
    assert worst < 5e-6, worst
    assert relerr(outs[3]['seg'], g['seg']) < 5e-4
    assert relerr(outs[3]['dense'], g['dense']) < 5e-4
    assert relerr(outs[3]['proj_feat'][:, 0:1280:97], g['proj_feat.slice']) < 5e-4


def test_extra_stage_extension_matches_the_composed_reference(golden):
    """G7x (round 4): the reference has no network with more than two refinement stages, but it has the MODULES -- oracle/gen_golden.py builds its
    DIR, appends two more `Joint2BoneFeature` + `Residual` pairs of the classes and arguments of projecter_3 / enhance_layer3 and chains them the way
    the reference's forward chains its own two stages.  The numpy oracle's N-stage extension (oracle/dir_forward.py) is held to that: the f4
    extension is pinned to the reference's own classes, not only to this build's restatement of them."""
    g = golden('g7x_dir_extra2')
    shapes = shapes_of('manifest_dir.json')
    from oracle.golden_inputs import extra_stage_shapes
    shapes.update(extra_stage_shapes(shapes, 2))
    assert len(shapes) == 963 + 2 * 240
    sd = synth.synth_state_dict(shapes, SEED, cond=True)
    outs = dir_forward(sd, synth.synth_input('dir.img', (2, 3, 256, 256), SEED))
    assert len(outs) == 6
    worst = 0.0
    for i in range(5):
        for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right'):
            worst = max(worst, maxabs(outs[i][k], g['s%d.%s' % (i, k)]))
        for k in ('pd_joint_uv_left', 'pd_joint_uv_right', 'pd_proj_left', 'pd_proj_right', 'pd_offset'):
            assert maxabs(outs[i][k], g['s%d.%s' % (i, k)]) < 5e-4, (i, k)
    assert worst < 5e-7, worst                               # measured 5e-8 m
    assert relerr(outs[5]['seg'], g['seg']) < 5e-4 and relerr(outs[5]['dense'], g['dense']) < 5e-4
    assert relerr(outs[5]['proj_feat'][:, 0:1280:97], g['proj_feat.slice']) < 5e-4


# ------------------------------------------------------------------ G9 eval metric maths (8f rank 1)
@pytest.mark.parametrize('root_joint', [0, 9])
@pytest.mark.parametrize('scale', [True, False])
def test_eval_metrics_match_reference(golden, root_joint, scale):
    """oracle/eval_metrics.py vs the arrays the reference's own loop body (apps/eval.py:139-241) produced.
    Tolerance: the reference's fp32 noise on camera-space inputs -- 2e-6 m (3-D), 5e-3 px (2-D); see tests/test_gpu_eval.py."""
    from oracle import eval_metrics as EM
    from oracle.golden_inputs import eval_inputs
    g, ins = golden('g9_eval'), eval_inputs()
    ins.update(jr_left=EM.jr_matrix(ins['jreg_left']), jr_right=EM.jr_matrix(ins['jreg_right']))
    out = EM.batch_metrics(ins, root_joint, scale)
    tag = 'r%d_s%d.' % (root_joint, int(scale))
    keys = [k for k in g if k.startswith(tag)]
    assert len(keys) == 13
    for k in keys:
        tol = 5e-3 if '2d' in k else 2e-6
        assert out[k[len(tag):]].shape == g[k].shape
        assert maxabs(out[k[len(tag):]], g[k]) <= tol, k


# ------------------------------------------------------------------ G10 GT MANO layer (8f rank 1)
def test_gt_mano_matches_reference(golden):
    """oracle/gt_mano.py vs the reference's models/manolayer.py::ManoLayer outputs (12 cases: PCA 45 / 12, rotation-matrix
    pose, centring, scale, no translation, new_skel, zero pose; both hands).  1e-7 m = 1e-4 mm."""
    from oracle import gt_mano as G
    from oracle.golden_inputs import GTMANO_CASES, gtmano_inputs
    g = golden('g10_gtmano')
    assert len([k for k in g if k.endswith('.verts')]) == 12
    for side in ('left', 'right'):
        T = G.tables(side)
        for case in GTMANO_CASES:
            R, pose, shape, trans, scale = gtmano_inputs(case)
            v, j = G.gt_mano_forward(T, R, pose, shape, trans, scale, center_idx=case[2], use_pca=case[1] > 0, new_skel=case[5])
            assert maxabs(v, g['%s.%s.verts' % (side, case[0])]) <= 1e-7, (side, case[0])
            assert maxabs(j, g['%s.%s.joints' % (side, case[0])]) <= 1e-7, (side, case[0])


# ------------------------------------------------------------------ G11 input tensor preparation (8f rank 3, tensor side)
def test_image_prep_matches_reference(golden):
    from oracle.image_prep import normalize_u8_bgr
    g = golden('g11_imgprep')
    assert np.array_equal(normalize_u8_bgr(g['img']), g['y'])          # bit-exact: same fp32 operation order


# ------------------------------------------------------------------ G8 training objective, forward (a13)
def test_losses_match_reference(golden):
    """oracle/losses.py against the 42 scalars of the reference's own DIR.forward in training mode (models/dir.py:542-594)"""
    from conftest import loss_case
    from oracle import losses as OL
    g = golden('g8_loss')
    preds, gt, faces, seg, dense, gt_seg, gt_dense = loss_case(g)
    got = dict(OL.dense_losses(seg, dense, gt_seg, gt_dense))
    for i in range(3):
        for k, v in OL.stage_losses(preds[i], gt, faces).items():
            got['%s_%d' % (k, i)] = v
    want = {k[5:]: float(g[k]) for k in g if k.startswith('loss.')}
    assert set(got) == set(want) and len(want) == 42
    for k in sorted(want):
        assert abs(got[k] - want[k]) <= 2e-6 * max(1.0, abs(want[k])) + 1e-5 * abs(want[k]), (k, got[k], want[k])


def test_loss_interpolation_matches_torch():
    """the two F.interpolate modes of models/dir.py:565-566 at sizes where the taps are not the trivial half-half ones"""
    import torch
    import torch.nn.functional as F
    from oracle import losses as OL
    rng = np.random.RandomState(3)
    for H, S in ((256, 32), (96, 32), (100, 16)):
        x = rng.rand(2, 3, H, H).astype(np.float32)
        want = F.interpolate(torch.from_numpy(x), (S, S), mode='bilinear').numpy()
        assert maxabs(OL.interpolate_bilinear(x, S), want) < 2e-7
        lab = rng.randint(0, 3, (2, 1, H, H)).astype(np.float32)
        want = F.interpolate(torch.from_numpy(lab), (S, S), mode='nearest').numpy()
        assert np.array_equal(OL.interpolate_nearest(lab, S), want)


# ------------------------------------------------------------------ optimiser (8f rank 2): pinned against torch itself
def test_adamw_oracle_matches_torch():
    """train.py:227-230: the numpy restatement against torch.optim.AdamW + CosineAnnealingLR (the reference's dependency)"""
    import torch
    from oracle import optim as OO
    rng = np.random.RandomState(0)
    p0 = rng.normal(0, 0.1, (37, 19)).astype(np.float32)
    w = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([{'params': [w], 'initial_lr': 1e-3}], 1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=6, eta_min=0)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 8):
        g = (rng.normal(0, 1, p0.shape) * 10.0 ** rng.randint(-4, 1)).astype(np.float32)
        lr = opt.param_groups[0]['lr']
        assert abs(lr - OO.cosine_lr(1e-3, step - 1, 6)) < 1e-12
        w.grad = torch.from_numpy(g.copy())
        opt.step()
        sch.step()
        p, m, v = OO.adamw_step(p, g, m, v, step, lr)
        assert relerr(p, w.detach().numpy()) < 2e-7, step
        st = opt.state[w]
        assert relerr(m, st['exp_avg'].numpy()) < 2e-7 and relerr(v, st['exp_avg_sq'].numpy()) < 2e-7


def test_loss_gradients_match_autograd(golden):
    """oracle/losses.py gradients against torch autograd through the reference's loss modules (G12, oracle/gen_golden.py gen_loss_grad)"""
    from conftest import loss_case
    from oracle import losses as OL
    g, gg = golden('g8_loss'), golden('g12_loss_grad')
    preds, gt, faces, seg, dense, gt_seg, gt_dense = loss_case(g)
    d = OL.dense_loss_grads(seg, dense, gt_seg, gt_dense)
    # the reference side is fp32 autograd; its Lovasz term differences J_i - J_{i-1} (~1e-5) carry fp32 cancellation noise
    for k, tol in (('seg', 1e-4), ('dense', 2e-5)):
        assert maxabs(d[k], gg['grad.' + k]) <= tol * np.abs(gg['grad.' + k]).max(), k
    for i in range(3):
        for k, v in OL.stage_loss_grads(preds[i], gt, faces).items():
            want = gg['grad.s%d.%s' % (i, k)]
            assert maxabs(v, want) <= 2e-5 * np.abs(want).max(), (i, k, maxabs(v, want), np.abs(want).max())


def test_stock_torch_dense_ops_equal_the_numpy_oracle():
    """bench.py's second CPU baseline swaps the oracle's dense operators for stock torch CPU kernels (oracle/torch_ops.py): the
    swapped forward must be the same function (fp32 summation-order noise only), and the swap must be undone afterwards"""
    from oracle import nnops as N_
    from oracle.torch_ops import stock_torch_dense_ops
    sd = synth.synth_state_dict(shapes_of('manifest_dir.json'), SEED)
    img = synth.synth_input('dir.img', (1, 3, 256, 256), SEED)
    a = dir_forward(sd, img)
    conv_before = N_.conv2d
    with stock_torch_dense_ops(2):
        assert N_.conv2d is not conv_before
        b = dir_forward(sd, img)
    assert N_.conv2d is conv_before
    for i in range(3):
        for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_uv_right'):
            assert maxabs(a[i][k], b[i][k]) < 2e-6, (i, k)
    assert relerr(a[3]['seg'], b[3]['seg']) < 1e-3 and relerr(a[3]['proj_feat'], b[3]['proj_feat']) < 1e-3


# ------------------------------------------------------------------ G13 / G14 gradients (backward pass)
def test_mano_gradient_oracle_matches_reference_autograd(golden):
    """oracle/grad.py (central differences in float64 on the numpy forward) vs torch autograd through the reference's manopth layer"""
    from oracle import grad as OG
    from oracle.golden_inputs import MANO_GRAD_CASES, mano_grad_inputs
    g = golden('g13_mano_grad')
    for side in ('left', 'right'):
        buf = synth.mano_buffers(side, SEED)
        for case, center in MANO_GRAD_CASES[:3]:
            para, cot = mano_grad_inputs(case, side)
            for sel in ('all', 'joint_uv'):
                kw = {('g_' + k): cot[k] for k in cot if sel in ('all', k)}
                got = OG.mano_vjp(buf, para.astype(np.float64), side, None if center < 0 else center, **kw)
                ref = g['%s_%s_c%d.%s' % (side, case, center, sel)]
                assert maxabs(got, ref) < 1e-6 * np.abs(ref).max(), (side, case, center, sel)


def regressor_fixture():
    shapes = shapes_of('manifest_regressor.json')
    return synth.synth_state_dict(shapes, SEED)


def test_regressor_gradient_oracle_matches_reference_autograd(golden):
    from oracle import grad as OG
    from oracle.golden_inputs import regress_grad_inputs
    g = golden('g14_regress_grad')
    sd = regressor_fixture()
    P = N.Params(sd)
    ins, cot = regress_grad_inputs()
    out = OT.regressor_offset(ins['feat_l'], ins['feat_r'], ins['para_l'], ins['para_r'], ins['offset'], P, OT.mano_bufs(P, 'left'), OT.mano_bufs(P, 'right'))
    assert maxabs(out['pd_mano_para_left'], g['out.pd_mano_para_left']) < 1e-5 and maxabs(out['pd_offset'], g['out.pd_offset']) < 1e-5
    got = OG.regressor_vjp(P, OT.mano_bufs(P, 'left'), OT.mano_bufs(P, 'right'), ins['feat_l'], ins['feat_r'], ins['para_l'], ins['para_r'],
                           ins['offset'], cot)
    for k in ('feat_l', 'feat_r', 'mano_left.bias', 'mano_right.bias', 'offset.weight', 'offset.bias'):
        assert maxabs(got[k], g['grad.' + k]) < 2e-6 * np.abs(g['grad.' + k]).max(), k
    for k in ('mano_left.weight', 'mano_right.weight'):
        ref = g['grad.' + k + '.cols4']
        assert maxabs(got[k][:, ::4], ref) < 2e-6 * np.abs(ref).max(), k
        assert maxabs(got[k].sum(1), g['grad.' + k + '.rowsum']) < 2e-6 * np.abs(g['grad.' + k + '.rowsum']).max()
        assert maxabs(got[k].sum(0), g['grad.' + k + '.colsum']) < 2e-6 * np.abs(g['grad.' + k + '.colsum']).max()


def test_ste_gradient_oracle_matches_reference_autograd(golden):
    from conftest import check_compact_grads
    from oracle.ste_grad import ste_forward_backward
    g = golden('g15_ste_grad')
    sd = synth.synth_state_dict(ste_shapes(), SEED)
    x, gy = synth.synth_input('stegrad.x', (3, 42, 128), SEED), synth.synth_input('stegrad.gy', (3, 42, 64), SEED)
    y, gx, G = ste_forward_backward(sd, x, gy)
    assert maxabs(y, g['y']) < 3e-5
    assert maxabs(gx, g['grad.x']) < 2e-5 * np.abs(g['grad.x']).max()
    assert not any(k.startswith('STEblocks.0.') for k in G)                  # never executed: no gradient (mixSTE.py:197)
    worst = check_compact_grads(G, g, 2e-5)
    assert worst < 2e-5 and len(G) == 12 * 3 + 7


def test_pgcn_train_gradient_oracle_matches_reference_autograd(golden):
    from conftest import check_compact_grads
    from oracle.pgcn_grad import pgcn_train_forward_backward
    g = golden('g16_pgcn_grad')
    sd = synth.synth_state_dict(_pgcn_shapes(), SEED)
    x, gy = synth.synth_input('pgcngrad.x', (5, 21, 128), SEED), synth.synth_input('pgcngrad.gy', (5, 21, 128), SEED)
    y, gx, G, running = pgcn_train_forward_backward(sd, x, gy)
    assert maxabs(y, g['y']) < 2e-5 * np.abs(g['y']).max()
    assert maxabs(gx, g['grad.x']) < 3e-5 * np.abs(g['grad.x']).max()
    G = {k: (v.reshape(2 * 21 * 128, 128) if k.endswith('gconv.W') else v) for k, v in G.items()}
    assert check_compact_grads(G, g, 3e-5, zero_suffixes=('gconv.bias', 'gconv.e_0')) < 3e-5
    for k, v in running.items():
        assert maxabs(v, g['after.' + k]) < 1e-5 * max(1.0, np.abs(g['after.' + k]).max()), k


# ------------------------------------------------------------------ G17 gradients through one stage's token half (training mode)
STAGE_ZERO_GRADS = ('gconv.bias', 'gconv.e_0', 'filters.0.bias', 'pos_emb_left.0.bias', 'pos_emb_right.0.bias', 'global_pos_emb.0.bias')
MANO_KEYS = ('th_selected_comps', 'th_hands_mean', 'th_shapedirs', 'th_posedirs', 'th_v_template', 'th_J_regressor', 'th_weights')


def test_stage_token_gradient_oracle_matches_reference_autograd(golden):
    """the float64 chain rule through sampler -> token MLPs -> P-GCN -> STE -> RegressorOffset -> MANO (oracle/stage_grad.py) against torch
    autograd through the reference's Joint2BoneFeature in training mode"""
    from conftest import check_compact_grads
    from oracle.golden_inputs import stage_grad_inputs
    from oracle.stage_grad import stage_token_grads
    g = golden('g17_stage_grad')
    sd = synth.synth_state_dict(shapes_of('manifest_stage16.json'), SEED)
    ins, cot = stage_grad_inputs(16)
    mano = [{k: sd['regressor.mano_layer_%s.%s' % (s, k)] for k in MANO_KEYS} for s in ('left', 'right')]
    tok, g_feat, G, running = stage_token_grads(sd, mano[0], mano[1], *ins, cot)
    assert maxabs(tok, g['out.joint_feat']) < 2e-5 * np.abs(g['out.joint_feat']).max()
    gm = np.abs(g['gfeat.ch4']).max()
    assert maxabs(g_feat[:, ::4], g['gfeat.ch4']) < 3e-5 * gm
    assert maxabs(g_feat.sum(1), g['gfeat.chsum']) < 3e-5 * g['gfeat.abssum'].max()
    G = {k: (v.reshape(2 * 21 * 128, 128) if k.endswith('gconv.W') else v) for k, v in G.items()}
    worst = check_compact_grads(G, g, 5e-5, zero_suffixes=STAGE_ZERO_GRADS)
    n_ref = len({k for k in g if k.startswith('grad.')})
    assert n_ref > 100 and worst < 5e-5
    for k, v in running.items():
        assert maxabs(v, g['after.' + k]) < 1e-5 * max(1.0, np.abs(g['after.' + k]).max()), k
    assert len(running) == 2 * (2 + 2 + 8) + 2


# ------------------------------------------------------------------ G18 gradients through the image half's residual blocks (training mode)
BLOCK_ZERO = ('conv1.conv.bias', 'conv2.conv.bias')        # a conv bias in front of a training-mode BatchNorm: analytically zero gradient


@pytest.mark.parametrize('name', ['bneck_plain', 'bneck_down', 'res_skip', 'res_same'])
def test_block_gradient_oracle_matches_reference_autograd(golden, name):
    from conftest import check_compact_grads
    from oracle import block_grad as OB
    from oracle.golden_inputs import BLOCK_GRAD_CASES, block_grad_inputs
    g = golden('g18_block_grad_' + name)
    kind, stride = BLOCK_GRAD_CASES[name][:2]
    sd = synth.synth_state_dict(shapes_of('manifest_blocks.json', name), SEED)
    x, gy = block_grad_inputs(name)
    y, gx, G, R = OB.bottleneck(sd, x, gy, stride) if kind == 'bottleneck' else OB.residual(sd, x, gy)
    assert maxabs(y[:, ::8], g['y.ch8']) < 2e-5 * np.abs(g['y.ch8']).max()
    assert maxabs(gx[:, ::8], g['gx.ch8']) < 3e-5 * np.abs(g['gx.ch8']).max()
    assert maxabs(gx.sum(1), g['gx.chsum']) < 3e-5 * g['gx.abssum'].max()
    assert check_compact_grads(G, g, 5e-5, zero_suffixes=BLOCK_ZERO) < 5e-5
    if kind == 'residual':
        assert ('skip_layer.conv.weight' in G) == (name != 'res_same')      # an unused skip_layer gets no gradient (hourglass.py:56-59)
    for k, v in R.items():
        assert maxabs(v, g['after.' + k]) < 1e-5 * max(1.0, np.abs(g['after.' + k]).max()), k


# ------------------------------------------------------------------ G19 gradients through bone_proj
@pytest.mark.parametrize('S,dist', [(16, 1), (32, 2)])
def test_bone_proj_gradient_oracle_matches_reference_autograd(golden, S, dist):
    from oracle.golden_inputs import bone_grad_inputs
    from oracle.spatial_grad import bone_proj_backward
    g = golden('g19_bone_grad')
    uv, feat, gi = bone_grad_inputs(S)
    g_uv, g_feat = bone_proj_backward(uv, feat, gi, S, dist)
    assert maxabs(g_feat, g['S%d.g_feat' % S]) < 2e-5 * np.abs(g['S%d.g_feat' % S]).max()
    assert maxabs(g_uv, g['S%d.g_uv' % S]) < 1e-4 * np.abs(g['S%d.g_uv' % S]).max()     # the fp32 side divides by (d_a + d_b)^2 near the joints
