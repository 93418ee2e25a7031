"""GPU parity (a1, layer1 bottlenecks): dir_bottleneck_chain_forward -- conv2 3x3 + bn2 + ReLU + conv3 1x1 + bn3 + identity + ReLU of
block i and conv1 1x1 + bn1 + ReLU of block i+1 in one launch (models/backbone/resnet.py:122-140) -- vs the numpy oracle with the
same bf16 rounding points, and vs the unfused dir_conv2d_forward sequence it replaces."""
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


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(BF).float().numpy()


def make(tag, B, H, W):
    g = lambda n, shp, **k: synth.synth_input('%s.%s' % (tag, n), shp, SEED, **k)  # noqa: E731
    p = dict(y1=np.maximum(bf16_round(g('y1', (B, 64, H, W))), 0), res=bf16_round(g('res', (B, 256, H, W))),
             w2=bf16_round(g('w2', (64, 64, 3, 3)) * np.float32(np.sqrt(2.0 / 576))),
             w3=bf16_round(g('w3', (256, 64, 1, 1)) * np.float32(np.sqrt(2.0 / 64))),
             w1=bf16_round(g('w1', (64, 256, 1, 1)) * np.float32(np.sqrt(2.0 / 256))))
    for k, c in (('2', 64), ('3', 256), ('1', 64)):
        p['s' + k] = g('s' + k, (c,), kind='uniform', lo=0.5, hi=1.5)
        p['h' + k] = g('h' + k, (c,)) * np.float32(0.3)
    return p


def oracle_chain(p, res, nxt):
    aff = lambda t, s, h: t * s.reshape(1, -1, 1, 1) + h.reshape(1, -1, 1, 1)  # noqa: E731
    y2 = bf16_round(np.maximum(aff(N.conv2d(p['y1'].astype(np.float64), p['w2'].astype(np.float64), None, 1, 1), p['s2'], p['h2']), 0)
                    .astype(np.float32))
    o = aff(N.conv2d(y2.astype(np.float64), p['w3'].astype(np.float64), None, 1, 0), p['s3'], p['h3'])
    if res:
        o = o + p['res']
    out = bf16_round(np.maximum(o, 0).astype(np.float32))
    y1n = None
    if nxt:
        y1n = np.maximum(aff(N.conv2d(out.astype(np.float64), p['w1'].astype(np.float64), None, 1, 0), p['s1'], p['h1']), 0)
    return out, y1n


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(a.transpose(0, 2, 3, 1)).to(BF)


def run_fused(p, res, nxt):
    w2 = F.pack_conv_weight(dev(p['w2']), BF)
    w3 = dev(p['w3'].reshape(256, 64)).to(BF)
    w1 = dev(p['w1'].reshape(64, 256)).to(BF)
    return F.bottleneck_chain(nhwc(p['y1']), w2, dev(p['s2']), dev(p['h2']), w3, dev(p['s3']), dev(p['h3']),
                              residual=nhwc(p['res']) if res else None, nxt=(w1, dev(p['s1']), dev(p['h1'])) if nxt else None)


@pytest.mark.parametrize('shape', [(2, 16, 32), (1, 64, 64), (3, 8, 16)])
@pytest.mark.parametrize('res', [True, False])
@pytest.mark.parametrize('nxt', [True, False])
def test_chain_matches_oracle(shape, res, nxt):
    B, H, W = shape
    p = make('bneck.%d_%d_%d' % shape, B, H, W)
    ref_out, ref_y1n = oracle_chain(p, res, nxt)
    out, y1n = run_fused(p, res, nxt)
    got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert relerr(got, ref_out) < 1e-2                     # bf16 output: one ulp of the output scale
    d = np.abs(got - ref_out)
    assert d.max() <= np.abs(ref_out).max() * 2.0 ** -6            # never off by more than ~a bf16 ulp of the output scale
    assert np.mean(d > np.maximum(np.abs(ref_out), 0.05) * 2.0 ** -6) < 1e-3   # and almost always within an ulp of the value itself
    if nxt:
        got1 = y1n.float().cpu().numpy().transpose(0, 3, 1, 2)
        assert relerr(got1, ref_y1n) < 1.5e-2              # input (the block output) may differ from the oracle's by a bf16 ulp
    else:
        assert y1n is None


def test_chain_vs_unfused_sequence():
    B, H, W = 4, 64, 64
    p = make('bneck.seq', B, H, W)
    out, y1n = run_fused(p, True, True)
    y1 = nhwc(p['y1'])
    y2 = F.conv2d_nhwc(y1, F.pack_conv_weight(dev(p['w2']), BF), 1, 1, dev(p['s2']), dev(p['h2']), relu=True)
    o = F.conv2d_nhwc(y2, F.pack_conv_weight(dev(p['w3']), BF), 1, 0, dev(p['s3']), dev(p['h3']), relu=True, residual=nhwc(p['res']))
    n1 = F.conv2d_nhwc(o, F.pack_conv_weight(dev(p['w1']), BF), 1, 0, dev(p['s1']), dev(p['h1']), relu=True)
    d = (out.float() - o.float()).abs()
    assert float(d.max()) <= float(o.float().abs().max()) * 2.0 ** -6     # same rounding points; K grouping inside the MFMA differs
    assert float((d > 0).float().mean()) < 0.05
    assert relerr(y1n.float().cpu().numpy(), n1.float().cpu().numpy()) < 1e-2


@pytest.mark.parametrize('nxt', [True, False])
def test_chain_projection_shortcut(nxt):
    """first block of layer1: conv3 + downsample conv as one GEMM (BN scales folded into the bf16 weight rows, as DualConvOp does)"""
    B, H, W = 2, 16, 32
    p = make('bneck.dual', B, H, W)
    g = lambda n, shp, **k: synth.synth_input('bneck.dual.%s' % n, shp, SEED, **k)  # noqa: E731
    x0 = bf16_round(g('x0', (B, 64, H, W)))
    wd = g('wd', (256, 64, 1, 1)) * np.float32(np.sqrt(2.0 / 64))
    sd, hd = g('sd', (256,), kind='uniform', lo=0.5, hi=1.5), g('hd', (256,)) * np.float32(0.3)
    w3f = bf16_round(p['w3'].reshape(256, 64) * p['s3'][:, None])            # folded rows, then bf16 (what the kernel multiplies)
    wdf = bf16_round(wd.reshape(256, 64) * sd[:, None])
    shift = (p['h3'] + hd).astype(np.float32)
    aff = lambda t, s, h: t * s.reshape(1, -1, 1, 1) + h.reshape(1, -1, 1, 1)  # noqa: E731
    y2 = bf16_round(np.maximum(aff(N.conv2d(p['y1'].astype(np.float64), p['w2'].astype(np.float64), None, 1, 1), p['s2'], p['h2']), 0)
                    .astype(np.float32))
    o = (N.conv2d(y2.astype(np.float64), w3f.reshape(256, 64, 1, 1).astype(np.float64), None, 1, 0)
         + N.conv2d(x0.astype(np.float64), wdf.reshape(256, 64, 1, 1).astype(np.float64), None, 1, 0) + shift.reshape(1, -1, 1, 1))
    ref_out = bf16_round(np.maximum(o, 0).astype(np.float32))
    out, y1n = F.bottleneck_chain(nhwc(p['y1']), F.pack_conv_weight(dev(p['w2']), BF), dev(p['s2']), dev(p['h2']), dev(w3f).to(BF),
                                  torch.ones(256, device='cuda'), dev(shift),
                                  nxt=(dev(p['w1'].reshape(64, 256)).to(BF), dev(p['s1']), dev(p['h1'])) if nxt else None,
                                  dual=(nhwc(x0), dev(wdf).to(BF)))
    got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert relerr(got, ref_out) < 1e-2
    assert np.abs(got - ref_out).max() <= np.abs(ref_out).max() * 2.0 ** -6
    if nxt:
        ref1 = np.maximum(aff(N.conv2d(ref_out.astype(np.float64), p['w1'].astype(np.float64), None, 1, 0), p['s1'], p['h1']), 0)
        assert relerr(y1n.float().cpu().numpy().transpose(0, 3, 1, 2), ref1) < 1.5e-2


def test_chain_next_conv1_128_channels():
    """last layer1 block -> layer2's first conv1 (256 -> 128)"""
    B, H, W = 2, 16, 32
    p = make('bneck.n128', B, H, W)
    g = lambda n, shp, **k: synth.synth_input('bneck.n128.%s' % n, shp, SEED, **k)  # noqa: E731
    w1 = bf16_round(g('w1b', (128, 256, 1, 1)) * np.float32(np.sqrt(2.0 / 256)))
    s1, h1 = g('s1b', (128,), kind='uniform', lo=0.5, hi=1.5), g('h1b', (128,)) * np.float32(0.3)
    ref_out, _ = oracle_chain(p, True, False)
    ref1 = np.maximum(N.conv2d(ref_out.astype(np.float64), w1.astype(np.float64), None, 1, 0) * s1.reshape(1, -1, 1, 1)
                      + h1.reshape(1, -1, 1, 1), 0)
    out, y1n = F.bottleneck_chain(nhwc(p['y1']), F.pack_conv_weight(dev(p['w2']), BF), dev(p['s2']), dev(p['h2']),
                                  dev(p['w3'].reshape(256, 64)).to(BF), dev(p['s3']), dev(p['h3']), residual=nhwc(p['res']),
                                  nxt=(dev(w1.reshape(128, 256)).to(BF), dev(s1), dev(h1)))
    assert tuple(y1n.shape) == (B, H, W, 128)
    assert relerr(out.float().cpu().numpy().transpose(0, 3, 1, 2), ref_out) < 1e-2
    assert relerr(y1n.float().cpu().numpy().transpose(0, 3, 1, 2), ref1) < 1.5e-2


def test_chain_rejects_bad_shape():
    from dir_amd._capi import DirHipError
    p = make('bneck.bad', 1, 8, 16)
    y1 = torch.zeros(1, 12, 16, 64, device='cuda', dtype=BF)
    w2 = F.pack_conv_weight(dev(p['w2']), BF)
    w3 = dev(p['w3'].reshape(256, 64)).to(BF)
    with pytest.raises(DirHipError):
        F.bottleneck_chain(y1, w2, dev(p['s2']), dev(p['h2']), w3, dev(p['s3']), dev(p['h3']))


@pytest.mark.parametrize('shape', [(2, 16, 32), (1, 64, 64)])
def test_chain_decimated_output_is_the_even_pixels_of_the_full_one(shape):
    """dir_bneck_chain_params.out_decimate (round 4): the last layer1 block's output is read by layer2's stride-2 projection shortcut only, so
    only its even (y, x) pixels are written -- bit for bit the even pixels of the full output; the fused next conv1 is unchanged"""
    B, H, W = shape
    p = make('bneck.dec.%d_%d_%d' % shape, B, H, W)
    w2 = F.pack_conv_weight(dev(p['w2']), BF)
    w3 = dev(p['w3'].reshape(256, 64)).to(BF)
    w1 = dev(p['w1'].reshape(64, 256)).to(BF)
    args = (nhwc(p['y1']), w2, dev(p['s2']), dev(p['h2']), w3, dev(p['s3']), dev(p['h3']))
    kw = dict(residual=nhwc(p['res']), nxt=(w1, dev(p['s1']), dev(p['h1'])))
    full, y1n = F.bottleneck_chain(*args, **kw)
    dec, y1n_d = F.bottleneck_chain(*args, decimate=True, **kw)
    assert dec.shape == (B, H // 2, W // 2, 256)
    assert torch.equal(dec, full[:, ::2, ::2].contiguous())
    assert torch.equal(y1n_d, y1n)
