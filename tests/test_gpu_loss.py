"""a13 / SURVEY 8f rank 2 (forward half): the training objective on the GPU (dir_stage_losses_forward, dir_dense_losses_forward,
host mirror dir_amd/models/loss.py) against the reference's own values (G8 = DIR.forward in training mode, models/dir.py:542-594)
and against the oracle on larger / edge-case inputs.  Tolerance: fp32 elementwise arithmetic in the reference's order, fp64
accumulation on both sides -> 1e-5 relative (the reference's own fp32 reductions sit ~1e-7 from the oracle)."""
import numpy as np
import pytest
import torch

from conftest import loss_case
from dir_amd import synth
from dir_amd.models import loss as ML
from oracle import losses as OL

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-5):
    return abs(a - b) <= tol * max(abs(b), 1e-3)


def cuda(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}


def test_losses_match_reference_golden(golden):
    g = golden('g8_loss')
    preds, gt, faces, seg, dense, gt_seg, gt_dense = loss_case(g)
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    target['seg'] = torch.from_numpy(gt_seg).cuda()
    target['dense'] = torch.from_numpy(gt_dense).cuda()
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    crit = ML.DirLoss(torch.from_numpy(faces[0]), torch.from_numpy(faces[1]))
    loss = crit([cuda(p) for p in preds], cuda({'seg': seg, 'dense': dense}), target, meta)
    want = {k[5:]: float(g[k]) for k in g if k.startswith('loss.')}
    assert set(loss) == set(want) and len(want) == 42
    worst = 0.0
    for k in sorted(want):
        got = float(loss[k])
        worst = max(worst, abs(got - want[k]) / max(abs(want[k]), 1e-3))
        assert close(got, want[k]), (k, got, want[k])
    print('42 loss terms vs the reference: worst relative difference %.2e' % worst)


def _random_stage(rng, B, c2=3):
    pred, gt = {}, {}
    for side in ('left', 'right'):
        gt['center_' + side] = rng.normal(0, 0.05, (B, 1, 3)).astype(np.float32)
        for n, tag in ((21, 'joint'), (778, 'mesh')):
            xyz = rng.normal(0, 0.05, (B, n, 3)).astype(np.float32)
            pred['pd_%s_xyz_%s' % (tag, side)] = xyz
            noise = np.where(rng.rand(B, n, 3) < 0.5, rng.normal(0, 0.0005, (B, n, 3)), rng.normal(0, 0.01, (B, n, 3)))
            gt['%s_3d_%s' % (tag, side)] = (xyz + gt['center_' + side] + noise).astype(np.float32)
            uv = rng.uniform(-1, 1, (B, n, 2)).astype(np.float32)
            pred['pd_%s_uv_%s' % (tag, side)] = uv
            noise = np.where(rng.rand(B, n, 2) < 0.5, rng.normal(0, 0.004, (B, n, 2)), rng.normal(0, 0.05, (B, n, 2)))
            gt['%s_2d_%s' % (tag, side)] = np.concatenate([uv + noise, rng.normal(0, 1, (B, n, c2 - 2))], axis=2).astype(np.float32)
    pred['pd_offset'] = rng.normal(0, 0.3, (B, 3)).astype(np.float32)
    return pred, gt


@pytest.mark.parametrize('B,c2', [(1, 2), (7, 3), (64, 3)])
def test_stage_losses_vs_oracle(B, c2):
    rng = np.random.RandomState(100 + B)
    pred, gt = _random_stage(rng, B, c2)
    faces = tuple(synth.loss_faces(s) for s in ('left', 'right'))
    want = OL.stage_losses(pred, gt, faces)
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    got = ML.stage_losses(cuda(pred), target, meta, [torch.from_numpy(f) for f in faces]).cpu().numpy()
    for i, k in enumerate(ML.STAGE_KEYS):
        assert close(float(got[i]), want[k]), (k, float(got[i]), want[k])
    # exact zero residual and an exactly-at-the-knee residual
    same = dict(pred)
    gt0 = {k: v for k, v in gt.items()}
    for side in ('left', 'right'):
        gt0['joint_2d_' + side] = np.concatenate([pred['pd_joint_uv_' + side], np.zeros((B, 21, c2 - 2), np.float32)], axis=2)
        gt0['joint_2d_' + side][:, 0, 0] += np.float32(0.01)
    want = OL.stage_losses(same, gt0, faces)
    target = cuda({k: v for k, v in gt0.items() if not k.startswith('center')})
    got = ML.stage_losses(cuda(same), target, meta, [torch.from_numpy(f) for f in faces]).cpu().numpy()
    for i, k in enumerate(ML.STAGE_KEYS):
        assert close(float(got[i]), want[k]), (k, float(got[i]), want[k])


@pytest.mark.parametrize('B,S,H,labels,quant', [(2, 32, 256, (0, 1, 2), False), (5, 16, 100, (0, 1, 2), False), (3, 32, 96, (0, 2), False),
                                                (4, 32, 256, (0, 1, 2), True), (64, 32, 256, (0, 1, 2), False)])
def test_dense_losses_vs_oracle(B, S, H, labels, quant):
    """sizes with non-trivial interpolation taps, an absent class ('present' skips it), heavily tied errors (the Lovasz value does
    not depend on the order inside a tie), and the bench batch"""
    rng = np.random.RandomState(7 * B + S)
    seg = rng.normal(0, 1.5, (B, 3, S, S)).astype(np.float32)
    if quant:
        seg = (np.round(seg * 2) / 2).astype(np.float32)
    dense = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    gt_seg = rng.choice(labels, size=(B, 1, H, H)).astype(np.float32)
    gt_dense = (rng.randint(0, 256, (B, 3, H, H)) / np.float32(255.0)).astype(np.float32)
    gt_dense[:, :, ::3] = dense.mean()                               # some residuals under the SmoothL1 knee
    want = OL.dense_losses(seg, dense, gt_seg, gt_dense)
    got = ML.dense_losses(*(torch.from_numpy(t).cuda() for t in (seg, dense, gt_seg, gt_dense))).cpu().numpy()
    for i, k in enumerate(('seg', 'dense', 'lovasz')):
        assert close(float(got[i]), want[k]), (k, float(got[i]), want[k])


def test_loss_argument_errors():
    from dir_amd import _capi
    rng = np.random.RandomState(0)
    pred, gt = _random_stage(rng, 2)
    faces = [torch.from_numpy(synth.loss_faces(s)) for s in ('left', 'right')]
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    with pytest.raises(_capi.DirHipError):
        ML.stage_losses(cuda(pred), target, meta, [faces[0], faces[1][:10]])
    with pytest.raises(_capi.DirHipError):
        ML.dense_losses(torch.zeros(2, 4, 32, 32).cuda(), torch.zeros(2, 4, 32, 32).cuda(), torch.zeros(2, 1, 64, 64).cuda(),
                        torch.zeros(2, 3, 64, 64).cuda())
    with pytest.raises(_capi.DirHipError):
        ML.stage_losses({k: torch.from_numpy(v) for k, v in pred.items()}, target, meta, faces)      # CPU tensors: no fallback


def test_projection_inside_the_kernel_equals_mesh_uv_input():
    """pd_mesh_uv_* absent -> the kernel applies utils/utils.py:47-63 to pd_mesh_xyz_* / pd_proj_* itself: same 13 values"""
    rng = np.random.RandomState(5)
    B = 6
    pred, gt = _random_stage(rng, B)
    faces = [torch.from_numpy(synth.loss_faces(s)) for s in ('left', 'right')]
    for side in ('left', 'right'):
        proj = np.concatenate([rng.uniform(2, 6, (B, 1)), rng.normal(0, 0.2, (B, 2))], axis=1).astype(np.float32)
        pred['pd_proj_' + side] = proj
        pred['pd_mesh_uv_' + side] = (proj[:, None, :1] * pred['pd_mesh_xyz_' + side][..., :2] + proj[:, None, 1:]).astype(np.float32)
        gt['mesh_2d_' + side][..., :2] = pred['pd_mesh_uv_' + side] + rng.normal(0, 0.01, (B, 778, 2)).astype(np.float32)
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    a = ML.stage_losses(cuda(pred), target, meta, faces)
    b = ML.stage_losses(cuda({k: v for k, v in pred.items() if 'mesh_uv' not in k}), target, meta, faces)
    assert torch.equal(a, b)


def test_validation_loss_on_engine_outputs():
    """DirEngine.forward -> DirLoss on its output dicts (fp32 engine) == the oracle's objective on the same tensors"""
    import json
    import os
    from conftest import GOLDEN
    from dir_amd.engine import DirEngine
    with open(os.path.join(GOLDEN, 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    B = 3
    img = torch.from_numpy(synth.synth_input('loss.img', (B, 3, 256, 256), 1234)).cuda()
    outs = DirEngine(sd, dtype=torch.float32).forward(img, want_proj_feat=False)
    rng = np.random.RandomState(9)
    _, gt = _random_stage(rng, B)
    last = outs[2]
    for side in ('left', 'right'):
        for tag, n in (('joint', 21), ('mesh', 778)):
            gt['%s_3d_%s' % (tag, side)] = (last['pd_%s_xyz_%s' % (tag, side)].cpu().numpy() + gt['center_' + side]
                                            + rng.normal(0, 0.003, (B, n, 3))).astype(np.float32)
    gt_seg = rng.randint(0, 3, (B, 1, 256, 256)).astype(np.float32)
    gt_dense = rng.uniform(0, 1, (B, 3, 256, 256)).astype(np.float32)
    faces = tuple(synth.loss_faces(s) for s in ('left', 'right'))
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    target.update(seg=torch.from_numpy(gt_seg).cuda(), dense=torch.from_numpy(gt_dense).cuda())
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    loss = ML.DirLoss(*[torch.from_numpy(f) for f in faces])(outs[:3], outs[3], target, meta)
    assert len(loss) == 42
    want = dict(OL.dense_losses(outs[3]['seg'].cpu().numpy(), outs[3]['dense'].cpu().numpy(), gt_seg, gt_dense))
    for i in range(3):
        p = {k: v.cpu().numpy() for k, v in outs[i].items() if v is not None}
        for side in ('left', 'right'):
            pr = p['pd_proj_' + side]
            p['pd_mesh_uv_' + side] = (pr[:, None, :1] * p['pd_mesh_xyz_' + side][..., :2] + pr[:, None, 1:]).astype(np.float32)
        for k, v in OL.stage_losses(p, gt, faces).items():
            want['%s_%d' % (k, i)] = v
    for k in sorted(want):
        assert close(float(loss[k]), want[k]), (k, float(loss[k]), want[k])


# ----------------------------------------------------------------------------------------------- gradients
def test_loss_gradients_match_autograd_golden(golden):
    """dir_stage_losses_backward / dir_dense_losses_backward against torch autograd through the reference's loss modules (G12)"""
    g, gg = golden('g8_loss'), golden('g12_loss_grad')
    preds, gt, faces, seg, dense, gt_seg, gt_dense = loss_case(g)
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    ft = [torch.from_numpy(f) for f in faces]
    gs, gd = ML.dense_loss_grads(*(torch.from_numpy(t).cuda() for t in (seg, dense, gt_seg, gt_dense)))
    for got, k, tol in ((gs, 'seg', 1e-4), (gd, 'dense', 2e-5)):      # fp32 autograd's Lovasz differences carry cancellation noise
        want = gg['grad.' + k]
        assert np.abs(got.cpu().numpy() - want).max() <= tol * np.abs(want).max(), k
    worst = 0.0
    for i in range(3):
        for k, v in ML.stage_loss_grads(cuda(preds[i]), target, meta, ft).items():
            want = gg['grad.s%d.%s' % (i, k)]
            e = np.abs(v.cpu().numpy() - want).max() / np.abs(want).max()
            worst = max(worst, e)
            assert e <= 2e-5, (i, k, e)
    print('gradients vs autograd: worst error %.2e of the tensor maximum' % worst)


@pytest.mark.parametrize('B', [1, 7, 64])
def test_loss_gradients_vs_oracle(B):
    rng = np.random.RandomState(200 + B)
    pred, gt = _random_stage(rng, B)
    for side in ('left', 'right'):
        pred['pd_mesh_uv_' + side] = pred['pd_mesh_uv_' + side]
    faces = tuple(synth.loss_faces(s) for s in ('left', 'right'))
    want = OL.stage_loss_grads(pred, gt, faces)
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    got = ML.stage_loss_grads(cuda(pred), target, meta, [torch.from_numpy(f) for f in faces])
    for k, w in want.items():
        assert np.abs(got[k].cpu().numpy() - w).max() <= 2e-5 * np.abs(w).max(), k
    # upstream gradients: linear in grad_out
    go = torch.from_numpy(rng.uniform(0.5, 2.0, 13).astype(np.float32)).cuda()
    got2 = ML.stage_loss_grads(cuda(pred), target, meta, [torch.from_numpy(f) for f in faces], grad_out=go)
    k = 'pd_joint_uv_left'
    assert torch.allclose(got2[k], got[k] * go[0], rtol=1e-6, atol=0)
    S, H = 32, 96
    seg = rng.normal(0, 1.5, (B, 3, S, S)).astype(np.float32)
    dense = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    gt_seg = rng.choice((0, 2) if B == 7 else (0, 1, 2), size=(B, 1, H, H)).astype(np.float32)      # B = 7: class 1 absent
    gt_dense = rng.uniform(0, 1, (B, 3, H, H)).astype(np.float32)
    want = OL.dense_loss_grads(seg, dense, gt_seg, gt_dense)
    gs, gd = ML.dense_loss_grads(*(torch.from_numpy(t).cuda() for t in (seg, dense, gt_seg, gt_dense)))
    assert np.abs(gs.cpu().numpy() - want['seg']).max() <= 2e-5 * np.abs(want['seg']).max()
    assert np.abs(gd.cpu().numpy() - want['dense']).max() <= 2e-5 * np.abs(want['dense']).max()
