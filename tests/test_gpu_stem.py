"""GPU parity (a1, stem): dir_stem_pool_forward -- conv1 7x7/s2/p3 + bn1 + ReLU + MaxPool2d(3,2,1) in one launch (bf16 mode) --
vs the numpy oracle (conv2d -> scale/shift -> ReLU -> bf16 rounding -> maxpool3x3s2p1) on bf16-rounded operands, vs the staged
path it replaces (dir_stem_prep_s2d + dir_conv2d_forward + dir_maxpool3x3s2), and uint8 input (apps/eval.py:59-61 fused)
against the float path bit for bit.  models/backbone/resnet.py:244-247."""
import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import functional as F
from dir_amd import synth
from oracle import image_prep as IP
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


def params(tag):
    w = synth.synth_input(tag + '.w', (64, 3, 7, 7), SEED) * np.float32(np.sqrt(2.0 / 147))
    scale = synth.synth_input(tag + '.s', (64,), SEED, kind='uniform', lo=0.5, hi=1.5)
    shift = synth.synth_input(tag + '.b', (64,), SEED) * np.float32(0.3)
    return w, scale, shift


def oracle_stem(x, w, scale, shift):
    """x, w already bf16-rounded; fp64 conv, fp32-style epilogue, bf16 storage of the conv output (as the kernel keeps it)"""
    c = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, 2, 3)
    c = np.maximum(c * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1), 0)
    return N.maxpool3x3s2p1(bf16_round(c.astype(np.float32)))


@pytest.mark.parametrize('shape', [(2, 256, 256), (3, 36, 52), (1, 8, 12), (2, 64, 32)])
def test_stem_pool_matches_oracle(shape):
    B, H, W = shape
    tag = 'stem.%d_%d_%d' % shape
    w, scale, shift = params(tag)
    x = bf16_round(synth.synth_input(tag + '.x', (B, 3, H, W), SEED))
    ref = oracle_stem(x, bf16_round(w), scale, shift)
    y = F.stem_pool(torch.from_numpy(x).cuda(), F.pack_stem_weight(torch.from_numpy(w).cuda()), torch.from_numpy(scale).cuda(),
                    torch.from_numpy(shift).cuda())
    got = y.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert relerr(got, ref) < 1e-2            # one bf16 ulp of the output scale (bf16 conv output, summation order)
    # elementwise: never more than one bf16 ulp away from the oracle value
    assert np.all(np.abs(got - ref) <= np.maximum(np.abs(ref), 1e-3) * 2.0 ** -7)


def test_stem_pool_vs_staged_path():
    B, H = 2, 256
    w, scale, shift = params('stem.staged')
    x = synth.synth_input('stem.staged.x', (B, 3, H, H), SEED)
    dx = torch.from_numpy(x).cuda()
    ds, dh = torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda()
    y = F.stem_pool(dx, F.pack_stem_weight(torch.from_numpy(w).cuda()), ds, dh).float()
    # the staged path through the engine's own stem operator
    from dir_amd import _capi, engine as E
    L = _capi.lib()
    xp = torch.empty(B, 131, 132, 16, device='cuda', dtype=torch.bfloat16)
    _capi.check(L.dir_stem_prep_s2d(_capi.ptr(dx), _capi.ptr(xp), B, H, H, 131, 132, 1, _capi.stream_ptr()), 'prep')
    stem = E.stem_conv_op(torch.from_numpy(w).cuda(), ds, dh, torch.bfloat16)
    s1 = stem(xp)
    z = torch.empty(B, 64, 64, 64, device='cuda', dtype=torch.bfloat16)
    _capi.check(L.dir_maxpool3x3s2(_capi.ptr(s1), _capi.ptr(z), B, 128, 128, 64, 1, _capi.stream_ptr()), 'pool')
    z = z.float()
    # same operand rounding, same epilogue; only the K summation order differs -> at most one bf16 ulp, and rare
    d = (y - z).abs()
    assert float(d.max()) <= float(z.abs().max()) * 2.0 ** -7
    assert float((d > 0).float().mean()) < 0.02


def test_stem_pool_uint8_equals_float_path():
    B, H, W = 2, 256, 256
    w, scale, shift = params('stem.u8')
    rng = np.random.RandomState(7)
    img = rng.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    x = IP.normalize_u8_bgr(img)                                  # fp32 NCHW, the reference's arithmetic
    pw = F.pack_stem_weight(torch.from_numpy(w).cuda())
    ds, dh = torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda()
    y8 = F.stem_pool(torch.from_numpy(img).cuda(), pw, ds, dh)
    yf = F.stem_pool(torch.from_numpy(x).cuda(), pw, ds, dh)
    assert torch.equal(y8.view(torch.int16), yf.view(torch.int16))


def test_stem_pool_rejects_bad_sizes():
    from dir_amd._capi import DirHipError
    w, scale, shift = params('stem.bad')
    pw = F.pack_stem_weight(torch.from_numpy(w).cuda())
    with pytest.raises(DirHipError):
        F.stem_pool(torch.zeros(1, 3, 30, 32, device='cuda'), pw, torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda())
