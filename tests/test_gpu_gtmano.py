"""GPU parity of the ground-truth MANO layer (models/manolayer.py:251-323): dir_gt_mano_forward through
the drop-in dir_amd.models.manolayer.ManoLayer against the reference class's own outputs (tests/golden/g10_gtmano.npz) and
the oracle.  Tolerance: vertex / joint positions within 1e-7 m (1e-4 mm, the north-star budget); the `trans` cases sit at
0.7 m where one fp32 ulp is 6e-8 m."""
import numpy as np
import pytest
import torch

from oracle import gt_mano as G
from oracle.golden_inputs import GTMANO_CASES, gtmano_inputs

pytestmark = pytest.mark.gpu
TOL = 1.2e-7


def _layer(side, case):
    from dir_amd.models.manolayer import ManoLayer
    return ManoLayer.synthetic(side, center_idx=case[2], use_pca=case[1] > 0, new_skel=case[5]).cuda()


def _run(layer, ins):
    T = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    R, pose, shape, trans, scale = ins
    v, j = layer(T(R), T(pose), T(shape), trans=T(trans), scale=T(scale))
    torch.cuda.synchronize()
    return v.cpu().numpy(), j.cpu().numpy()


@pytest.mark.parametrize('side', ['left', 'right'])
@pytest.mark.parametrize('case', GTMANO_CASES, ids=[c[0] for c in GTMANO_CASES])
def test_gt_mano_vs_reference_golden(golden, side, case):
    g = golden('g10_gtmano')
    v, j = _run(_layer(side, case), gtmano_inputs(case))
    assert np.abs(v - g['%s.%s.verts' % (side, case[0])]).max() <= TOL
    assert np.abs(j - g['%s.%s.joints' % (side, case[0])]).max() <= TOL


def test_gt_mano_fp64_arbitration_and_batch():
    """B = 257 (ragged vs any chunking), fp64 oracle as the arbiter: the kernel is within 1e-7 m of the exact value"""
    case = GTMANO_CASES[0]
    ins = gtmano_inputs(case, B=257)
    v, j = _run(_layer('right', case), ins)
    T = {k: a.astype(np.float64) for k, a in G.tables('right').items()}
    R, pose, shape, trans, scale = [None if a is None else a.astype(np.float64) for a in ins]
    v64, j64 = G.gt_mano_forward(T, R, pose, shape, trans, scale, center_idx=case[2], use_pca=True, new_skel=False)
    assert np.abs(v - v64).max() <= TOL and np.abs(j - j64).max() <= TOL
    # per-sample independence: sample 100 alone
    sub = tuple(None if a is None else a[100:101] for a in ins)
    v1, j1 = _run(_layer('right', case), sub)
    assert np.array_equal(v1[0], v[100]) and np.array_equal(j1[0], j[100])


def test_gt_mano_feeds_eval_regressor():
    """the eval loop builds Jr from this layer's J_regressor (apps/eval.py:116-117): Jr(verts) reproduces the 16 chain joints'
    regression and the 5 tip vertices of the layer's own output ordering"""
    from dir_amd.apps.eval import Jr
    case = ('x', 45, None, True, False, False)
    layer = _layer('left', case)
    ins = gtmano_inputs(('normal',) + case[1:], B=4)
    T = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    v, j = layer(T(ins[0]), T(ins[1]), T(ins[2]), trans=T(ins[3]))
    jr = Jr(layer.J_regressor)
    jj = jr(v).cpu().numpy()
    tips = [4, 8, 12, 16, 20]                                   # new_order positions of the five tip vertices
    assert np.abs(jj[:, tips] - j.cpu().numpy()[:, tips]).max() <= 1e-7


def test_gt_mano_rejects_bad_arguments(tmp_path):
    from dir_amd import _capi
    from dir_amd.models.manolayer import ManoLayer
    with pytest.raises(FileNotFoundError):
        ManoLayer(str(tmp_path / 'MANO_RIGHT.pkl'))
    layer = ManoLayer.synthetic('right', center_idx=None).cuda()
    R = torch.eye(3).repeat(2, 1, 1)
    with pytest.raises(_capi.DirHipError):                      # CPU tensors: no fallback
        layer(R, torch.zeros(2, 45), torch.zeros(2, 10))
    with pytest.raises(_capi.DirHipError):
        layer(R.cuda(), torch.zeros(2, 46).cuda(), torch.zeros(2, 10).cuda())   # more PCA coefficients than components
