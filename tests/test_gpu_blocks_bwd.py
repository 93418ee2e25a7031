"""GPU parity of the image half's residual blocks in training form (dir_amd/train/blocks.py: Bottleneck, hourglass Residual; batch-statistics
BatchNorm2d, Conv2d data / weight gradients) against
  G18  torch autograd through the reference's own Bottleneck / Residual classes in training mode (oracle/gen_golden.py::gen_block_grad), and
  the oracle's float64 chain rule (oracle/block_grad.py) at a batch size where BatchNorm takes the chunked-reduction path (R > 2048).
Tolerance 1e-5 of each gradient's maximum."""
import json
import os

import numpy as np
import pytest
import torch

from dir_amd import synth
from dir_amd.train import blocks as TB
from oracle import block_grad as OB
from oracle.golden_inputs import BLOCK_GRAD_CASES, block_grad_inputs

pytestmark = pytest.mark.gpu
SEED = 1234
HERE = os.path.dirname(os.path.abspath(__file__))
ZERO = ('conv1.conv.bias', 'conv2.conv.bias')


def params(name):
    with open(os.path.join(HERE, 'golden', 'manifest_blocks.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f)[name].items()}
    sd = synth.synth_state_dict(shapes, SEED)
    return sd, {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items() if 'num_batches' not in k}


def run(name, P, x, gy):
    kind, stride = BLOCK_GRAD_CASES[name][:2]
    xg = torch.from_numpy(x).cuda().permute(0, 2, 3, 1).contiguous()
    gg = torch.from_numpy(gy).cuda().permute(0, 2, 3, 1).contiguous()
    if kind == 'bottleneck':
        y, ctx = TB.bottleneck_forward(P, xg, stride)
        gx, G = TB.bottleneck_backward(P, ctx, gg)
    else:
        y, ctx = TB.residual_forward(P, xg)
        gx, G = TB.residual_backward(P, ctx, gg)
    return y.permute(0, 3, 1, 2), gx.permute(0, 3, 1, 2), G


@pytest.mark.parametrize('name', list(BLOCK_GRAD_CASES))
def test_block_backward_vs_reference_autograd(golden, name):
    from conftest import check_compact_grads
    g = golden('g18_block_grad_' + name)
    sd, P = params(name)
    x, gy = block_grad_inputs(name)
    y, gx, G = run(name, P, x, gy)
    yn, gn = y.cpu().numpy().astype(np.float64), gx.cpu().numpy().astype(np.float64)
    assert np.abs(yn[:, ::8] - g['y.ch8']).max() < 2e-5 * np.abs(g['y.ch8']).max()
    e_x = max(np.abs(gn[:, ::8] - g['gx.ch8']).max() / np.abs(g['gx.ch8']).max(), np.abs(gn.sum(1) - g['gx.chsum']).max() / g['gx.abssum'].max())
    assert e_x < 1e-5, e_x
    Gn = {k: v.cpu().numpy() for k, v in G.items()}
    ref_names = {k[5:].split('.cols')[0].replace('.rowsum', '').replace('.colsum', '') for k in g if k.startswith('grad.')}
    assert ref_names == set(Gn), (sorted(ref_names - set(Gn)), sorted(set(Gn) - ref_names))
    worst = check_compact_grads(Gn, g, 1e-5, zero_suffixes=ZERO)
    for k in g:
        if k.startswith('after.'):
            assert np.abs(P[k[6:]].cpu().numpy() - g[k]).max() < 1e-5 * max(1.0, float(np.abs(g[k]).max())), k
    print('%s (training mode) vs torch autograd through the reference: g x %.2e, parameters worst %.2e' % (name, e_x, worst))


@pytest.mark.parametrize('name', ['bneck_down', 'res_skip'])
def test_block_backward_vs_oracle_large_batch(name):
    """B = 12 at 16x16: 3072 rows per BatchNorm -> the chunked (deterministic) reductions; two runs agree bit for bit"""
    kind, stride = BLOCK_GRAD_CASES[name][:2]
    sd, P = params(name)
    x, gy = block_grad_inputs(name, batch=12)
    P2 = {k: v.clone() for k, v in P.items()}
    y, gx, G = run(name, P, x, gy)
    y2, gx2, G2 = run(name, P2, x, gy)
    assert torch.equal(gx, gx2) and all(torch.equal(G[k], G2[k]) for k in G)
    yr, gxr, Gr, Rr = OB.bottleneck(sd, x, gy, stride) if kind == 'bottleneck' else OB.residual(sd, x, gy)
    assert np.abs(gx.cpu().numpy() - gxr).max() < 1e-5 * np.abs(gxr).max()
    gmax = max(np.abs(v).max() for v in Gr.values())
    for k, ref in Gr.items():
        got = G[k].cpu().numpy()
        if k in ZERO:
            assert np.abs(got).max() < 1e-4 * gmax
            continue
        assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max(), k
    for k, v in Rr.items():
        assert np.abs(P[k].cpu().numpy() - v).max() < 1e-5 * max(1.0, np.abs(v).max()), k
