"""GPU parity of the token half of a refinement stage in training form (dir_amd/train/stage.py: sampler -> token MLPs -> P-GCN -> STE ->
RegressorOffset -> MANO, batch-statistics BatchNorm) -- forward outputs, the gradient into fusion_feat, every parameter gradient and the
BatchNorm running statistics -- against
  G17  torch autograd through the reference's Joint2BoneFeature in training mode (oracle/gen_golden.py::gen_stage_grad), and
  the oracle's float64 chain rule (oracle/stage_grad.py) at another batch size.
Tolerance 1e-5 of each gradient's maximum, per SURVEY.md 8f / VERDICT r1 item 4."""
import json
import os

import numpy as np
import pytest
import torch

from dir_amd import engine, synth
from dir_amd.train import ops as O
from dir_amd.train import stage as TS
from oracle.golden_inputs import stage_grad_inputs

pytestmark = pytest.mark.gpu
SEED = 1234
HERE = os.path.dirname(os.path.abspath(__file__))
ZERO = ('gconv.bias', 'gconv.e_0', 'filters.0.bias', 'pos_emb_left.0.bias', 'pos_emb_right.0.bias', 'global_pos_emb.0.bias')
MANO_KEYS = ('th_selected_comps', 'th_hands_mean', 'th_shapedirs', 'th_posedirs', 'th_v_template', 'th_J_regressor', 'th_weights')
OUT_KEYS = ('pd_offset', 'pd_mano_para_left', 'pd_mano_para_right', 'pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left',
            'pd_joint_xyz_right', 'pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_uv_left', 'pd_mesh_uv_right')


def setup(B):
    with open(os.path.join(HERE, 'golden', 'manifest_stage16.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth.synth_state_dict(shapes, SEED)
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items()}
    keep = []
    tabs = [engine.pack_mano(P, 'regressor.mano_layer_' + s, s, 0, keep) for s in ('left', 'right')]
    ins, cot = stage_grad_inputs(16, B)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    feat = dv(ins[0]).permute(0, 2, 3, 1).contiguous()
    prev = {'pd_joint_xyz_left': dv(ins[1]), 'pd_joint_xyz_right': dv(ins[2]), 'pd_joint_uv_left': dv(ins[3]), 'pd_joint_uv_right': dv(ins[4]),
            'pd_mano_para_left': dv(ins[5]), 'pd_mano_para_right': dv(ins[6]), 'pd_offset': dv(ins[7])}
    return sd, P, tabs, keep, ins, cot, feat, prev, dv


def run(P, tabs, feat, prev, cot, dv):
    out, ctx = TS.stage_tokens_forward(P, tabs, feat, prev)
    c = {k: dv(v) for k, v in cot.items() if k != 'joint_feat'}
    g_feat, G = TS.stage_tokens_backward(P, tabs, ctx, c, g_joint_feat=dv(cot['joint_feat']))
    return out, g_feat, G


def test_stage_tokens_backward_vs_reference_autograd(golden):
    from conftest import check_compact_grads
    g = golden('g17_stage_grad')
    sd, P, tabs, keep, ins, cot, feat, prev, dv = setup(4)
    out, g_feat, G = run(P, tabs, feat, prev, cot, dv)
    for k in OUT_KEYS + ('joint_feat',):
        ref = g['out.' + k]
        e = float(np.abs(out[k].cpu().numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max())
        assert e < 2e-5, (k, e)
    gf = g_feat.permute(0, 3, 1, 2).cpu().numpy().astype(np.float64)
    e_feat = max(np.abs(gf[:, ::4] - g['gfeat.ch4']).max() / np.abs(g['gfeat.ch4']).max(),
                 np.abs(gf.sum(1) - g['gfeat.chsum']).max() / g['gfeat.abssum'].max())
    assert e_feat < 1e-5, e_feat
    Gn = {k: v.cpu().numpy() for k, v in G.items()}
    Gn = {k: (v.reshape(2 * 21 * 128, 128) if k.endswith('gconv.W') else v) for k, v in Gn.items()}
    n_ref = {k[5:].split('.cols')[0].replace('.rowsum', '').replace('.colsum', '') for k in g if k.startswith('grad.')}
    assert n_ref <= set(Gn), sorted(n_ref - set(Gn))
    assert not any(k.startswith('interaction.STEblocks.0.') for k in Gn)
    worst = check_compact_grads(Gn, g, 1e-5, zero_suffixes=ZERO)
    for k in g:
        if k.startswith('after.'):
            e = float(np.abs(P[k[6:]].cpu().numpy() - g[k]).max())
            assert e < 1e-5 * max(1.0, float(np.abs(g[k]).max())), (k, e)
    print('stage token path (training mode) vs torch autograd through the reference: g fusion_feat %.2e, parameters worst %.2e (%d tensors)'
          % (e_feat, worst, len(Gn)))


def test_stage_tokens_backward_vs_oracle_and_determinism():
    """B = 3 against the float64 chain rule; two runs agree bit for bit (no atomics anywhere on the path)"""
    from oracle.stage_grad import stage_token_grads
    sd, P, tabs, keep, ins, cot, feat, prev, dv = setup(3)
    P2 = {k: v.clone() for k, v in P.items()}
    out, g_feat, G = run(P, tabs, feat, prev, cot, dv)
    out2, g_feat2, G2 = run(P2, tabs, feat, prev, cot, dv)
    assert torch.equal(g_feat, g_feat2) and all(torch.equal(G[k], G2[k]) for k in G)
    mano = [{k: sd['regressor.mano_layer_%s.%s' % (s, k)] for k in MANO_KEYS} for s in ('left', 'right')]
    tok, gf_ref, G_ref, running = stage_token_grads(sd, mano[0], mano[1], *ins, cot)
    gf = g_feat.permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(gf - gf_ref).max() < 1e-5 * np.abs(gf_ref).max()
    gmax = max(np.abs(v).max() for v in G_ref.values())
    worst = 0.0
    for k, ref in G_ref.items():
        got = G[k].cpu().numpy().reshape(ref.shape)
        if any(k.endswith(z) for z in ZERO):
            assert np.abs(got).max() < 1e-4 * gmax
            continue
        e = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    for k, v in running.items():
        assert np.abs(P[k].cpu().numpy() - v).max() < 1e-5 * max(1.0, np.abs(v).max()), k
    print('stage token path vs float64 oracle (B = 3): worst %.2e' % worst)


def test_grid_rows_roundtrip():
    """<rows, g> == <feat, grid_rows_bwd(g)> (adjoint identity), and out-of-range taps contribute nothing"""
    torch.manual_seed(0)
    B, S, C = 5, 16, 256
    feat = torch.randn(B, S, S, C, device='cuda')
    uv = (torch.rand(B, 21, 2, device='cuda') * 2.6 - 1.3).contiguous()
    rows = O.grid_rows_fwd(feat, uv)
    ref = torch.nn.functional.grid_sample(feat.permute(0, 3, 1, 2), uv.unsqueeze(1), align_corners=False).squeeze(2).permute(0, 2, 1).reshape(B * 21, C)
    assert (rows - ref).abs().max() < 1e-5
    g = torch.randn_like(rows)
    gf = O.grid_rows_bwd([g], [uv], B, S, C)
    lhs, rhs = (rows.double() * g.double()).sum(), (feat.double() * gf.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-6 * float(rows.double().abs().mul(g.double().abs()).sum())


def test_token_stage_train_step_feeds_flat_adamw():
    """loss gradients -> stage backward -> FlatAdamW.flat_grad (the data-parallel bucket) -> AdamW step, all through library calls:
    the bucket holds exactly the stage gradients, parameters without gradient (STE block 0, buffers) stay put, the rest move by
    lr * sign(g) (Adam's first step) + decoupled decay"""
    from dir_amd.optim import FlatAdamW
    from dir_amd.train import step as TSTEP
    B = 4
    sd, P, tabs, keep, ins, cot, feat, prev, dv = setup(B)
    params = {'decoder.projecter_4.' + k: torch.nn.Parameter(v.clone()) for k, v in P.items() if 'running_' not in k and 'num_batches' not in k
              and 'mano_layer' not in k and k != 'img_gird'}
    named = dict(params)
    for k, v in P.items():
        if 'decoder.projecter_4.' + k not in named:
            named['decoder.projecter_4.' + k] = v.clone()                      # buffers (running statistics, MANO tables)
    opt = FlatAdamW(list(params.values()), lr=1e-4)
    # token_stage_train_step covers the token half only: the image half's parameters are kept out of this step, like parameters torch
    # leaves with grad None (the whole network: dir_amd.train.step.train_step, tests/test_gpu_full_bwd.py)
    opt.set_inactive(TSTEP.inactive_parameters(params) + [p for k, p in params.items() if '.proj_feat_emb.' in k or '.fusion.' in k])
    before = {k: p.detach().clone() for k, p in params.items()}
    rng = np.random.RandomState(3)
    target, meta = {}, {}
    for s in ('left', 'right'):
        target['joint_2d_' + s] = dv(rng.uniform(0, 256, (B, 21, 2)).astype(np.float32))
        target['mesh_2d_' + s] = dv(rng.uniform(0, 256, (B, 778, 2)).astype(np.float32))
        target['joint_3d_' + s] = dv(rng.normal(0, 0.05, (B, 21, 3)).astype(np.float32))
        target['mesh_3d_' + s] = dv(rng.normal(0, 0.05, (B, 778, 3)).astype(np.float32))
        meta['center_' + s] = dv(rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32))
    faces = [dv(synth.mano_buffers(s, SEED)['th_faces'].astype(np.int64)) for s in ('left', 'right')]
    # reference gradients of the same step, computed without the optimiser
    Pd = {k[len('decoder.projecter_4.'):]: (v.data.clone() if isinstance(v, torch.nn.Parameter) else v.clone()) for k, v in named.items()}
    from dir_amd.models import loss as L
    out0, ctx0 = TS.stage_tokens_forward(Pd, tabs, feat, prev)
    G0 = TS.stage_tokens_backward(Pd, tabs, ctx0, L.stage_loss_grads(out0, target, meta, faces))[1]
    out, g_feat = TSTEP.token_stage_train_step(named, 'decoder.projecter_4.', tabs, feat, prev, target, meta, faces, opt)
    assert torch.equal(out['pd_mesh_xyz_left'], out0['pd_mesh_xyz_left'])
    moved = 0
    for k, p in params.items():
        rel = k[len('decoder.projecter_4.'):]
        if rel in G0:
            assert torch.equal(p.grad.reshape(-1), G0[rel].reshape(-1)), k              # the bucket holds the stage gradient, bit for bit
            big = G0[rel].reshape(p.shape).abs() > 1e-6                        # >> Adam's eps 1e-8
            if big.any():
                d = (before[k] * (1 - 1e-4 * 1e-2) - p.detach())[big] / 1e-4           # = sign(g) * |g| / (|g| + eps) ~ sign(g)
                assert (d * torch.sign(G0[rel].reshape(p.shape)[big]) > 0.5).all(), k
                moved += 1
        else:
            assert torch.equal(p.detach(), before[k]), k                                # no gradient: untouched (no decay either)
            assert 'STEblocks.0.' in k or 'proj_feat_emb' in k or 'fusion' in k, k
    assert moved > 100
    print('token-stage train step: %d parameter tensors updated through flat_grad (%d floats in the bucket)' % (moved, opt.flat_grad.numel()))
