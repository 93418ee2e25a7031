"""GPU parity of dir_conv1x1_stream_forward (the HBM-bound 1x1 convolutions as a streaming kernel: hourglass.Residual conv1 with its
pre-activation and conv3 + skip_layer, models/backbone/hourglass.py:55-70; Bottleneck conv1 / conv3 + projection shortcut,
models/backbone/resnet.py:117-140) vs the fp64 oracle on bf16-rounded operands and vs the tiled kernels it can replace."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(a.transpose(0, 2, 3, 1)).to(BF)


CASES = [  # B, H, W, Cin, Cout, pre, relu, (Cin2, stride2)
    (2, 16, 16, 512, 128, True, True, None),          # Residual.conv1 @16
    (1, 32, 32, 512, 128, True, True, None),          # Residual.conv1 @32
    (2, 16, 16, 2304, 128, True, True, None),         # fusion_layer4.conv1 (longest pre-activation vector)
    (3, 8, 8, 1024, 512, False, True, None),          # layer4.0.conv1, Cout = 2 N-chunks of 256
    (2, 16, 16, 128, 256, False, False, (512, 1)),    # Residual.conv3 + skip_layer
    (2, 16, 16, 128, 512, False, True, (256, 2)),     # layer2.0: conv3 + strided projection shortcut
    (1, 5, 7, 64, 128, False, True, None),            # ragged M (35 pixels: one partial workgroup)
    (3, 9, 11, 128, 384, True, False, (64, 1)),       # ragged M, Cout = 3 N-chunks of 128, both features
]


@pytest.mark.parametrize('case', CASES)
def test_stream_conv_vs_oracle(case):
    B, H, W, Cin, Cout, pre, relu, src2 = case
    g = lambda n, shp, **k: synth.synth_input('stream.%s.%s' % ('_'.join(map(str, case[:5])), n), shp, SEED, **k)  # noqa: E731
    x = bf16_round(g('x', (B, Cin, H, W)))
    w1 = bf16_round(g('w1', (Cout, Cin)) * np.float32(np.sqrt(2.0 / Cin)))
    scale, shift = g('s', (Cout,), kind='uniform', lo=0.5, hi=1.5), g('h', (Cout,)) * np.float32(0.3)
    ps, pb = g('ps', (Cin,), kind='uniform', lo=0.5, hi=1.5), g('pb', (Cin,)) * np.float32(0.3)
    a = x.astype(np.float64)
    if pre:     # the kernel rounds the pre-activated operand to bf16 (it is the MFMA operand), like the tiled kernels
        a = bf16_round(np.maximum(np.float32(x) * ps.reshape(1, -1, 1, 1) + pb.reshape(1, -1, 1, 1), 0)).astype(np.float64)
    ref = N.conv2d(a, w1.reshape(Cout, Cin, 1, 1).astype(np.float64))
    wk = w1
    x2d = None
    if src2 is not None:
        Cin2, s2 = src2
        x2 = bf16_round(g('x2', (B, Cin2, H * s2, W * s2)))
        w2 = bf16_round(g('w2', (Cout, Cin2)) * np.float32(np.sqrt(2.0 / Cin2)))
        ref = ref + N.conv2d(x2.astype(np.float64), w2.reshape(Cout, Cin2, 1, 1).astype(np.float64), None, s2, 0)
        wk = np.concatenate([w1, w2], 1)
        x2d = nhwc(x2)
    ref = ref * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    if relu:
        ref = np.maximum(ref, 0)
    # input inside a wider buffer at a channel offset, output into a slice of a wider buffer (the concat buffers of models/dir.py:444,461)
    xb = torch.zeros(B, H, W, Cin + 24, device='cuda', dtype=BF)
    xb[..., 16:16 + Cin] = nhwc(x)
    out = torch.full((B, H, W, Cout + 24), 3.0, device='cuda', dtype=BF)
    F.conv1x1_stream(xb, dev(wk), dev(scale), dev(shift), relu=relu, pre_scale=dev(ps) if pre else None, pre_shift=dev(pb) if pre else None,
                     pre_relu=pre, x2=x2d, stride2=src2[1] if src2 else 1, out=out, out_coff=8, in_coff=16, cin=Cin)
    got = out[..., 8:8 + Cout].float().permute(0, 3, 1, 2).cpu().numpy()
    assert relerr(got, ref) < 1e-2                                  # bf16 output
    assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -7
    assert float((out[..., :8].float() - 3).abs().max()) == 0 and float((out[..., 8 + Cout:].float() - 3).abs().max()) == 0


def test_stream_conv_vs_tiled_kernels_full_size():
    """B = 64 @32x32, K = 512 -> 128 with the pre-activation: against dir_conv2d_forward on the same operands"""
    B, H, W, Cin, Cout = 64, 32, 32, 512, 128
    gen = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=gen).to(BF)
    w = (torch.randn(Cout, Cin, device='cuda', generator=gen) * (2.0 / Cin) ** 0.5)
    s, h = torch.rand(Cout, device='cuda', generator=gen) + 0.5, torch.randn(Cout, device='cuda', generator=gen) * 0.3
    ps, pb = torch.rand(Cin, device='cuda', generator=gen) + 0.5, torch.randn(Cin, device='cuda', generator=gen) * 0.3
    a = F.conv1x1_stream(x, w, s, h, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=True)
    b = F.conv2d_nhwc(x, F.pack_conv_weight(w.reshape(Cout, Cin, 1, 1), BF), 1, 0, s, h, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=True)
    d = (a.float() - b.float()).abs()
    print('stream vs tiled: %.4f %% differ, max %.3e' % (100 * float((d > 0).float().mean()), float(d.max())))
    assert float(d.max()) <= float(b.float().abs().max()) * 2.0 ** -6 and float((d > 0).float().mean()) < 0.02
    a2 = F.conv1x1_stream(x, w, s, h, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=True)
    assert torch.equal(a, a2)
    a3 = F.conv1x1_stream(x, w, s, h, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=True, variant=22)        # 64-pixel workgroups (round 4)
    assert torch.equal(a, a3)


@pytest.mark.parametrize('case', CASES)
def test_stream_conv_64_pixel_workgroups_are_bit_identical(case):
    """DIR_CONV_VARIANT 22: the streaming kernel on 64-pixel workgroups (twice as many, half as long) -- the same MFMA k-slots in the same order,
    so every output bit equals the 128-pixel form's, ragged pixel counts, channel slices, second sources and pre-activation included"""
    B, H, W, Cin, Cout, pre, relu, src2 = case
    gen = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, H, W, Cin + 24, device='cuda', generator=gen).to(BF)
    K = Cin + (src2[0] if src2 else 0)
    w = torch.randn(Cout, K, device='cuda', generator=gen) * (2.0 / K) ** 0.5
    s, h = torch.rand(Cout, device='cuda', generator=gen) + 0.5, torch.randn(Cout, device='cuda', generator=gen) * 0.3
    ps, pb = torch.rand(Cin, device='cuda', generator=gen) + 0.5, torch.randn(Cin, device='cuda', generator=gen) * 0.3
    x2 = torch.randn(B, H * src2[1], W * src2[1], src2[0], device='cuda', generator=gen).to(BF) if src2 else None
    kw = dict(relu=relu, pre_scale=ps if pre else None, pre_shift=pb if pre else None, pre_relu=pre, x2=x2, stride2=src2[1] if src2 else 1, in_coff=16, cin=Cin)
    a = F.conv1x1_stream(x, w, s, h, **kw)
    for v in (22, 23, 24):          # 64- and 32-pixel workgroups; 24 (round 5): the pipelined form with a DMA producer wave (falls back to the plain kernel under a pre-activation)
        assert torch.equal(a, F.conv1x1_stream(x, w, s, h, variant=v, **kw)), v


def test_stream_conv_rejects_bad_arguments():
    from dir_amd._capi import DirHipError
    x = torch.zeros(1, 8, 8, 96, device='cuda', dtype=BF)
    with pytest.raises((DirHipError, AssertionError)):           # Cin not a multiple of 64
        F.conv1x1_stream(x, torch.zeros(128, 96, device='cuda'))
    with pytest.raises((DirHipError, AssertionError)):           # Cout not a multiple of 128
        F.conv1x1_stream(torch.zeros(1, 8, 8, 64, device='cuda', dtype=BF), torch.zeros(64, 64, device='cuda'))


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(64, 32, 32, 128, 256, 512, 1), (64, 16, 16, 256, 1024, 0, 1), (64, 32, 32, 128, 512, 256, 2), (7, 16, 16, 2304, 128, 0, 1), (3, 9, 11, 64, 384, 0, 1)])
def test_pipelined_stream_kernel_full_size_is_bit_identical(shape, dt):
    """DIR_CONV_VARIANT 24 (round 5: a fifth wave feeds a 4-deep ring of activation chunks by LDS-DMA, the MFMA waves only stream weights): the
    layers it serves at full size (B = 64), odd / even chunk counts, second sources with stride, ragged pixel counts, both 16-bit storage kinds --
    every output bit equals the register-staged kernel's, twice in a row (ring re-use across launches)"""
    B, H, W, Cin, Cout, Cin2, s2 = shape
    gen = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=gen).to(dt)
    K = Cin + Cin2
    w = torch.randn(Cout, K, device='cuda', generator=gen) * (2.0 / K) ** 0.5
    s, h = torch.rand(Cout, device='cuda', generator=gen) + 0.5, torch.randn(Cout, device='cuda', generator=gen) * 0.3
    x2 = torch.randn(B, H * s2, W * s2, Cin2, device='cuda', generator=gen).to(dt) if Cin2 else None
    kw = dict(relu=True, x2=x2, stride2=s2)
    a = F.conv1x1_stream(x, w, s, h, **kw)
    b1 = F.conv1x1_stream(x, w, s, h, variant=24, **kw)
    b2 = F.conv1x1_stream(x, w, s, h, variant=24, **kw)
    assert torch.equal(a, b1) and torch.equal(a, b2)
