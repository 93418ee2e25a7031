"""GPU parity (a1, layer2 / layer3 bottlenecks): dir_bottleneck_tail_forward -- conv3 1x1 + bn3 + identity + ReLU of block i and
conv1 1x1 + bn1 + ReLU of block i+1 in one launch (models/backbone/resnet.py:122-124,132-140) -- vs the numpy oracle with the same
bf16 rounding points, and vs the unfused dir_conv2d_forward pair it replaces (same rounding points; the GEMMs are computed
transposed, D[channel][pixel], so the fp32 sums may differ in the last bit and an output that sits on a bf16 rounding boundary by
one bf16 ulp)."""
import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import functional as F
from dir_amd import synth
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234
BF = torch.bfloat16
GEOM = [(128, 128), (128, 256), (256, 256)]


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(BF).float().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(a.transpose(0, 2, 3, 1)).to(BF)


def make(tag, B, H, W, P, N2):
    g = lambda n, shp, **k: synth.synth_input('%s.%s' % (tag, n), shp, SEED, **k)  # noqa: E731
    C4 = 4 * P
    p = dict(y2=np.maximum(bf16_round(g('y2', (B, P, H, W))), 0), res=np.maximum(bf16_round(g('res', (B, C4, H, W))), 0),
             w3=bf16_round(g('w3', (C4, P, 1, 1)) * np.float32(np.sqrt(2.0 / P))),
             w1=bf16_round(g('w1', (N2, C4, 1, 1)) * np.float32(np.sqrt(2.0 / C4))))
    for k, c in (('3', C4), ('1', N2)):
        p['s' + k] = g('s' + k, (c,), kind='uniform', lo=0.5, hi=1.5)
        p['h' + k] = g('h' + k, (c,)) * np.float32(0.3)
    return p


def oracle_tail(p):
    aff = lambda t, s, h: t * s.reshape(1, -1, 1, 1) + h.reshape(1, -1, 1, 1)  # noqa: E731
    o = aff(N.conv2d(p['y2'].astype(np.float64), p['w3'].astype(np.float64)), p['s3'], p['h3']) + p['res']
    out = bf16_round(np.maximum(o, 0).astype(np.float32))
    y1n = np.maximum(aff(N.conv2d(out.astype(np.float64), p['w1'].astype(np.float64)), p['s1'], p['h1']), 0)
    return out, y1n


def run_fused(p, waves=8):
    C4, P = p['w3'].shape[:2]
    N2 = p['w1'].shape[0]
    return F.bottleneck_tail(nhwc(p['y2']), dev(p['w3'].reshape(C4, P)), dev(p['s3']), dev(p['h3']), nhwc(p['res']),
                             dev(p['w1'].reshape(N2, C4)), dev(p['s1']), dev(p['h1']), waves=waves)


@pytest.mark.parametrize('waves', [8, 4])
@pytest.mark.parametrize('geom', GEOM)
@pytest.mark.parametrize('shape', [(1, 8, 8), (2, 16, 16), (3, 32, 32)])
def test_tail_matches_oracle(geom, shape, waves):
    """one tile, fewer tiles than CUs, and several tiles per workgroup; both workgroup shapes (8 waves x 64 pixels, 4 waves x 32)"""
    (P, N2), (B, H, W) = geom, shape
    p = make('tail.%d_%d.%d_%d_%d' % (geom + shape), B, H, W, P, N2)
    ref_out, ref_y1n = oracle_tail(p)
    out, y1n = run_fused(p, waves)
    got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert relerr(got, ref_out) < 1e-2                     # bf16 output: one ulp of the output scale
    d = np.abs(got - ref_out)
    assert d.max() <= np.abs(ref_out).max() * 2.0 ** -6
    assert np.mean(d > np.maximum(np.abs(ref_out), 0.05) * 2.0 ** -6) < 1e-3
    got1 = y1n.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert relerr(got1, ref_y1n) < 1.5e-2                  # its input (the block output) may differ from the oracle's by a bf16 ulp


@pytest.mark.parametrize('waves', [8, 4])
@pytest.mark.parametrize('geom', GEOM)
def test_tail_vs_the_unfused_pair_full_size(geom, waves):
    """full-size: B = 64 at the layer's resolution (65 536 / 16 384 pixels: 4 / 1 tiles per workgroup on 256 CUs)"""
    P, N2 = geom
    B, H, W = (64, 32, 32) if P == 128 else (64, 16, 16)
    p = make('tail.full.%d_%d' % geom, B, H, W, P, N2)
    out, y1n = run_fused(p, waves)
    y2, res = nhwc(p['y2']), nhwc(p['res'])
    o = F.conv2d_nhwc(y2, F.pack_conv_weight(dev(p['w3']), BF), 1, 0, dev(p['s3']), dev(p['h3']), relu=True, residual=res)
    n1 = F.conv2d_nhwc(o, F.pack_conv_weight(dev(p['w1']), BF), 1, 0, dev(p['s1']), dev(p['h1']), relu=True)
    torch.cuda.synchronize()
    d = (out.float() - o.float()).abs()
    print('tail vs unfused: %.4f %% of the block outputs differ, max %.3e' % (100 * float((d > 0).float().mean()), float(d.max())))
    assert float(d.max()) <= float(o.float().abs().max()) * 2.0 ** -6     # at most a bf16 ulp of the output scale
    assert float((d > 0).float().mean()) < 0.02
    assert relerr(y1n.float().cpu().numpy(), n1.float().cpu().numpy()) < 1e-2
    # and a second launch on the same inputs reproduces itself (persistent workgroups, DMA double buffer, register ring)
    out2, y1n2 = run_fused(p, waves)
    assert torch.equal(out, out2) and torch.equal(y1n, y1n2)


def test_tail_rejects_bad_arguments():
    from dir_amd._capi import DirHipError
    p = make('tail.bad', 1, 8, 8, 128, 128)
    with pytest.raises(DirHipError):                       # M = 36 pixels: not a multiple of the 64-pixel tile
        F.bottleneck_tail(torch.zeros(1, 6, 6, 128, device='cuda', dtype=BF), dev(p['w3'].reshape(512, 128)), dev(p['s3']), dev(p['h3']),
                          torch.zeros(1, 6, 6, 512, device='cuda', dtype=BF), dev(p['w1'].reshape(128, 512)), dev(p['s1']), dev(p['h1']))
    with pytest.raises(AssertionError):                    # geometry the kernel is not built for
        F.bottleneck_tail(torch.zeros(1, 8, 8, 64, device='cuda', dtype=BF), torch.zeros(256, 64, device='cuda'), dev(p['s3'][:256]), dev(p['h3'][:256]),
                          torch.zeros(1, 8, 8, 256, device='cuda', dtype=BF), torch.zeros(128, 256, device='cuda'), dev(p['s1']), dev(p['h1']))
