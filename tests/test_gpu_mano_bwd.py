"""GPU parity of dir_mano_backward_pair (the first link of the backward pass behind the loss gradients) against
  G13  torch autograd through the reference's own manopth ManoLayer + projection_batch_xy (oracle/gen_golden.py::gen_mano_grad), and
  the oracle: central differences in float64 on the numpy forward (oracle/grad.py) -- gradients by definition.
Tolerance 1e-5 of each gradient's maximum (fp32 kernel; 778-term reductions)."""
import numpy as np
import pytest
import torch

from dir_amd import engine, synth
from dir_amd import functional as F
from oracle import grad as OG
from oracle.golden_inputs import MANO_GRAD_CASES, mano_grad_inputs

pytestmark = pytest.mark.gpu
SEED = 1234


def tables(side, center):
    buf = synth.mano_buffers(side, SEED)
    sd = {('m.' + k): torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in buf.items()}
    keep = []
    return engine.pack_mano(sd, 'm', side, None if center < 0 else center, keep), keep, buf


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('side', ['left', 'right'])
def test_mano_backward_vs_reference_autograd(golden, side):
    g = golden('g13_mano_grad')
    worst = 0.0
    for case, center in MANO_GRAD_CASES:
        T, keep, _ = tables(side, center)
        para, cot = mano_grad_inputs(case, side)
        dp = dev(para)
        for sel in ('all', 'verts', 'joints', 'joint_uv', 'mesh_uv'):
            kw = {('g_' + k): [dev(cot[k])] for k in cot if sel in ('all', k)}
            got = F.mano_backward([T], [dp], **kw)[0].cpu().numpy()
            ref = g['%s_%s_c%d.%s' % (side, case, center, sel)]
            e = float(np.abs(got - ref).max() / np.abs(ref).max())
            worst = max(worst, e)
            assert e < 1e-5, (case, center, sel, e)
    print('mano backward vs torch autograd through the reference (%s): worst %.2e of the gradient maximum' % (side, worst))


def test_mano_backward_pair_vs_oracle_and_determinism():
    """both hands in one launch, B = 37 (not a multiple of anything), against the float64 finite-difference oracle; two launches agree
    bit for bit (fixed-order reductions)"""
    B = 37
    rng = np.random.RandomState(5)
    Ts, bufs, paras, cots = [], [], [], []
    for side in ('left', 'right'):
        T, keep, buf = tables(side, 0)
        Ts.append(T); bufs.append((buf, keep))
        p = rng.normal(0, 0.5, (B, 64)).astype(np.float32)
        p[:, 61] = 1.0 + 0.2 * p[:, 61]
        paras.append(p)
        cots.append({'verts': rng.normal(0, 1, (B, 778, 3)).astype(np.float32), 'joints': rng.normal(0, 1, (B, 21, 3)).astype(np.float32),
                     'joint_uv': rng.normal(0, 1, (B, 21, 2)).astype(np.float32), 'mesh_uv': rng.normal(0, 1, (B, 778, 2)).astype(np.float32)})
    kw = {('g_' + k): [dev(c[k]) for c in cots] for k in cots[0]}
    dp = [dev(p) for p in paras]
    got = F.mano_backward(Ts, dp, **kw)
    again = F.mano_backward(Ts, dp, **kw)
    for h, side in enumerate(('left', 'right')):
        assert torch.equal(got[h], again[h])
        ref = OG.mano_vjp(bufs[h][0], paras[h].astype(np.float64), side, 0, cots[h]['verts'], cots[h]['joints'], cots[h]['joint_uv'], cots[h]['mesh_uv'])
        e = float(np.abs(got[h].cpu().numpy() - ref).max() / np.abs(ref).max())
        print('mano backward vs float64 central differences (%s, B = %d): %.2e' % (side, B, e))
        assert e < 1e-5


def test_mano_backward_without_projection():
    """cam = NULL (ManoLayer alone): only verts / joints cotangents, the cam slot of the gradient stays zero"""
    import ctypes as C
    from dir_amd import _capi
    T, keep, buf = tables('right', 9)
    para, cot = mano_grad_inputs('normal', 'right')
    dp = dev(para)
    gv, gj = dev(cot['verts']), dev(cot['joints'])
    out = torch.zeros(3, 64, device='cuda')
    P1 = C.c_void_p * 1
    _capi.check(_capi.lib().dir_mano_backward_pair((_capi.ManoTables * 1)(T), P1(dp.data_ptr()), 64, P1(dp.data_ptr() + 51 * 4), 64, None, 0,
                                                   P1(gv.data_ptr()), P1(gj.data_ptr()), None, None, P1(out.data_ptr()), 64,
                                                   P1(out.data_ptr() + 51 * 4), 64, None, 0, 1, 3, _capi.stream_ptr()), 'mano_bwd')
    ref = OG.numeric_vjp(lambda p: OG.OM.mano_forward({k: np.asarray(v, np.float64) for k, v in buf.items() if np.asarray(v).dtype.kind == 'f'},
                                                      p[:, :51], p[:, 51:61], 'right', 9), para.astype(np.float64), (cot['verts'], cot['joints']))
    got = out.cpu().numpy()
    assert np.abs(got[:, :61] - ref[:, :61]).max() / np.abs(ref).max() < 1e-5 and float(np.abs(got[:, 61:]).max()) == 0.0


def test_regressor_backward_chain_vs_reference_autograd(golden):
    """G14 = torch autograd through the reference's RegressorOffset (three Linears -> two MANO layers -> four projections).  HIP chain:
    dir_regress_forward -> dir_mano_backward_pair (cotangents of the MANO outputs -> g mano_para) -> dir_regress_backward (parameter
    gradients in nn.Linear layout + the gradient w.r.t. the joint tokens)."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN, maxabs
    from dir_amd import _capi
    from oracle import nnops as N
    from oracle import tokens as OT
    from oracle.golden_inputs import regress_grad_inputs
    g = golden('g14_regress_grad')
    with open(os.path.join(GOLDEN, 'manifest_regressor.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sdn = synth.synth_state_dict(shapes, SEED)
    sd = {('r.' + k): dev(v) for k, v in sdn.items()}
    keep = []
    Ts = [engine.pack_mano(sd, 'r.mano_layer_left', 'left', 0, keep), engine.pack_mano(sd, 'r.mano_layer_right', 'right', 0, keep)]
    ins, cot = regress_grad_inputs()
    B = ins['feat_l'].shape[0]
    tok = dev(np.concatenate([ins['feat_l'], ins['feat_r']], 1))
    # forward of the three Linears on the GPU: the regressor kernel of the engine (k-major packed weights)
    R = _capi.RegressParams()
    t = dict(wt=torch.cat([sd['r.mano_left.weight'].t(), sd['r.mano_right.weight'].t()], 1).contiguous(), bl=sd['r.mano_left.bias'], br=sd['r.mano_right.bias'],
             wo=sd['r.offset.weight'], bo=sd['r.offset.bias'])
    R.mano_wt, R.mano_b[0], R.mano_b[1], R.off_w, R.off_b = t['wt'].data_ptr(), t['bl'].data_ptr(), t['br'].data_ptr(), t['wo'].data_ptr(), t['bo'].data_ptr()
    zero = {k: torch.zeros(64 if k != 'w2t' else 64 * 64, device='cuda') for k in ('w1t', 's1', 'b1', 'w2t', 'b2')}
    zero['w1t'] = torch.zeros(64 * 64, device='cuda')
    R.emb = _capi.TokenMlp(*(zero[k].data_ptr() for k in ('w1t', 's1', 'b1', 'w2t', 'b2')))
    pl, pr, off, emb = (torch.empty(B, 64, device='cuda'), torch.empty(B, 64, device='cuda'), torch.empty(B, 3, device='cuda'), torch.empty(B, 42, 64, device='cuda'))
    dpl, dpr, doff = dev(ins['para_l']), dev(ins['para_r']), dev(ins['offset'].reshape(B, 3))
    _capi.check(_capi.lib().dir_regress_forward(C.byref(R), _capi.ptr(tok), _capi.ptr(dpl), _capi.ptr(dpr), _capi.ptr(doff), _capi.ptr(pl), _capi.ptr(pr),
                                                _capi.ptr(off), _capi.ptr(emb), B, _capi.stream_ptr()), 'regress')
    assert maxabs(pl.cpu().numpy(), g['out.pd_mano_para_left']) < 2e-5 and maxabs(off.cpu().numpy(), g['out.pd_offset']) < 2e-5
    gp = F.mano_backward(Ts, [pl, pr], g_verts=[dev(cot['pd_mesh_xyz_left']), dev(cot['pd_mesh_xyz_right'])],
                         g_joints=[dev(cot['pd_joint_xyz_left']), dev(cot['pd_joint_xyz_right'])],
                         g_joint_uv=[dev(cot['pd_joint_uv_left']), dev(cot['pd_joint_uv_right'])],
                         g_mesh_uv=[dev(cot['pd_mesh_uv_left']), dev(cot['pd_mesh_uv_right'])])
    got = F.regress_backward(sd['r.mano_left.weight'], sd['r.mano_right.weight'], sd['r.offset.weight'], tok, dpl, dpr, doff, gp[0], gp[1],
                             dev(cot['pd_offset']))
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())  # noqa: E731
    gt = got['tok'].cpu().numpy()
    worst = max(rel(gt[:, :21], g['grad.feat_l']), rel(gt[:, 21:], g['grad.feat_r']))
    for k in ('mano_left.bias', 'mano_right.bias', 'offset.weight', 'offset.bias'):
        worst = max(worst, rel(got[k].cpu().numpy(), g['grad.' + k]))
    for k in ('mano_left.weight', 'mano_right.weight'):
        w = got[k].cpu().numpy()
        worst = max(worst, rel(w[:, ::4], g['grad.' + k + '.cols4']), rel(w.astype(np.float64).sum(1), g['grad.' + k + '.rowsum']),
                    rel(w.astype(np.float64).sum(0), g['grad.' + k + '.colsum']))
    print('regressor backward chain vs torch autograd through the reference: worst %.2e of each gradient maximum' % worst)
    assert worst < 1e-5
    # and against the float64 oracle (analytic Linears + central differences through MANO)
    from oracle import grad as OG
    P = N.Params(sdn)
    ref = OG.regressor_vjp(P, OT.mano_bufs(P, 'left'), OT.mano_bufs(P, 'right'), ins['feat_l'], ins['feat_r'], ins['para_l'], ins['para_r'], ins['offset'], cot)
    assert rel(gp[0].cpu().numpy(), ref['g_para_left']) < 1e-5 and rel(got['mano_left.weight'].cpu().numpy(), ref['mano_left.weight']) < 1e-5
    assert rel(gt[:, 21:], ref['feat_r']) < 1e-5
