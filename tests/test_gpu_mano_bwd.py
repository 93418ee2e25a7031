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
