"""GPU parity of the split-precision convolution arithmetic (DIR_DT_F16X3, include/dir_hip.h): fp32 tensors, every product evaluated on
the f16 matrix cores as hi*hi + lo*hi + hi*lo with fp32 accumulation.  Compared through the C ABI against the float64 oracle conv2d on the
SAME fp32 operands (nothing is pre-rounded: the mode has to carry fp32 operands, that is its point) and against the exact-fp32 MFMA kernel:
tolerance 5e-6 of the output scale (K = 18432 reaches 2.4e-6) -- four times tighter than the 2e-5 the exact-fp32 path is held to (its own accumulation-order slack) --
and never more than 4x the exact-fp32 kernel's own distance from float64."""
import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import _capi
from dir_amd import functional as F
from dir_amd import synth
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234
TOL = 5e-6

CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 1, 1, 0),
    (3, 15, 13, 64, 96, 3, 2, 1),       # odd sizes, stride 2, Cout tail inside a tile, M tail
    (2, 32, 32, 256, 128, 1, 2, 0),     # strided 1x1 (ResNet downsample)
    (1, 8, 8, 2048, 200, 3, 1, 1),      # long K (InitRegressor attention shape, Cout cut down)
    (2, 9, 9, 64, 130, 3, 1, 1),        # two N tiles with a ragged second one
    (2, 32, 32, 256, 6, 1, 1, 0),       # the merged seg / dense head: Cout = 6
]


def to_nhwc(a):
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1))


@pytest.mark.parametrize('case', CASES)
def test_f16x3_conv_matches_float64_oracle(case):
    B, H, W, Ci, Co, k, s, p = case
    tag = 'conv.%s' % '_'.join(map(str, case))
    x = synth.synth_input(tag + '.x', (B, Ci, H, W), SEED)
    w = synth.synth_input(tag + '.w', (Co, Ci, k, k), SEED) * np.float32(np.sqrt(2.0 / (k * k * Ci)))
    scale = synth.synth_input(tag + '.s', (Co,), SEED, kind='uniform', lo=0.5, hi=1.5)
    shift = synth.synth_input(tag + '.b', (Co,), SEED) * np.float32(0.3)
    ref = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, s, p)
    ref = np.maximum(ref * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1), 0)
    dx = torch.from_numpy(to_nhwc(x)).cuda()
    dw = F.pack_conv_weight(torch.from_numpy(w).cuda(), torch.float32)
    sc, sh = torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda()
    for variant in (0, 1, 2, 3, 4):                                   # heuristic tile and the four explicit tiles
        y = F.conv2d_nhwc(dx, dw, s, p, sc, sh, relu=True, arith='f16x3', variant=variant)
        e3 = relerr(y.cpu().numpy().transpose(0, 3, 1, 2), ref)
        assert e3 < TOL, (variant, e3)
    for variant in (0, 1, 2, 3, 4, 17, 18, 20, 8, 9, 10):              # activations pre-split (DIR_DT_F16X3P): DMA path, incl. the 3-buffer ring tiles and
                                                                       # the 8-wave pipelined kernel (conv_pipe.hip, XM = 3)
        yp = F.conv2d_nhwc(dx, dw, s, p, sc, sh, relu=True, arith='f16x3', variant=variant, presplit=True)
        ep = relerr(yp.cpu().numpy().transpose(0, 3, 1, 2), ref)
        assert ep < TOL, ('presplit', variant, ep)
        if variant == 0:
            assert torch.equal(yp, F.conv2d_nhwc(dx, dw, s, p, sc, sh, relu=True, arith='f16x3', variant=0))      # same operands, same order: same bits
    e32 = relerr(F.conv2d_nhwc(dx, dw, s, p, sc, sh, relu=True).cpu().numpy().transpose(0, 3, 1, 2), ref)
    print('%s: f16x3 %.2e  exact fp32 %.2e  (of the output scale)' % (tag, e3, e32))
    assert e3 < 4 * e32 + 2e-7


def test_f16x3_weight_packing_is_an_exact_split():
    """hi + lo reproduces the pre-scaled weight to 2^-22 relative; the prescale is a power of two and comes back out through `scale`"""
    torch.manual_seed(0)
    w = torch.randn(40, 96, device='cuda') * torch.logspace(-6, 2, 40, device='cuda')[:, None]
    w[3] = 0                                                          # an all-zero row must not produce inf / nan
    packed, sc = F.pack_f16x3_weights(w, torch.full((40,), 3.0, device='cuda'))
    assert packed.dtype == torch.float16 and packed.shape == (40, 3, 2, 32) and torch.isfinite(packed.float()).all()
    hi, lo = packed[:, :, 0].reshape(40, 96).double(), packed[:, :, 1].reshape(40, 96).double()
    p = 3.0 / sc.double()
    assert torch.equal(torch.frexp(p.cpu())[0], torch.full((40,), 0.5, dtype=torch.float64))    # powers of two (frexp on the host)
    amax = (w.abs().amax(1).double() * p)
    assert ((amax >= 4096) & (amax < 8192) | (amax == 0)).all()
    err = ((hi + lo) - w.double() * p[:, None]).abs().amax(1)
    assert (err <= amax * 2.0 ** -22 + 1e-30).all(), (err / amax.clamp_min(1e-30)).max()


def test_f16x3_prologue_residual_and_concat_slices():
    """pre-activation BN + ReLU on the input (hourglass.Residual conv1), residual add, channel-slice I/O: same case as the fp32 test"""
    B, H, W, Ci, Co = 2, 12, 12, 128, 64
    xbuf = synth.synth_input('convp.x', (B, Ci + 64, H, W), SEED)
    w = synth.synth_input('convp.w', (Co, Ci, 3, 3), SEED) * np.float32(0.05)
    ps = synth.synth_input('convp.ps', (Ci,), SEED, kind='uniform', lo=0.5, hi=1.5)
    pb = synth.synth_input('convp.pb', (Ci,), SEED) * np.float32(0.5)
    res = synth.synth_input('convp.res', (B, Co, H, W), SEED)
    bias = synth.synth_input('convp.bias', (Co,), SEED)
    act = np.maximum(xbuf[:, 64:].astype(np.float64) * ps.reshape(1, -1, 1, 1) + pb.reshape(1, -1, 1, 1), 0)
    ref = N.conv2d(act, w.astype(np.float64), bias.astype(np.float64), 1, 1) + res
    dx = torch.from_numpy(to_nhwc(xbuf)).cuda()
    dw = F.pack_conv_weight(torch.from_numpy(w).cuda(), torch.float32)
    out = torch.full((B, H, W, Co + 32), 7.0, device='cuda')
    dres = torch.from_numpy(to_nhwc(res)).cuda()
    F.conv2d_nhwc(dx, dw, 1, 1, None, torch.from_numpy(bias).cuda(), residual=dres, pre_scale=torch.from_numpy(ps).cuda(),
                  pre_shift=torch.from_numpy(pb).cuda(), pre_relu=True, out=out, out_coff=32, in_coff=64, cin=Ci, arith='f16x3')
    got = out.cpu().numpy()
    assert np.all(got[..., :32] == 7.0)                               # untouched slice of the concat buffer
    e = relerr(got[..., 32:].transpose(0, 3, 1, 2), ref)
    print('f16x3 prologue + residual: %.2e' % e)
    assert e < TOL
    out2 = torch.full((B, H, W, Co + 32), 7.0, device='cuda')          # the same through the pre-split pass (pre-activation applied there)
    F.conv2d_nhwc(dx, dw, 1, 1, None, torch.from_numpy(bias).cuda(), residual=dres, pre_scale=torch.from_numpy(ps).cuda(),
                  pre_shift=torch.from_numpy(pb).cuda(), pre_relu=True, out=out2, out_coff=32, in_coff=64, cin=Ci, arith='f16x3', presplit=True)
    assert torch.equal(out2, out)


@pytest.mark.parametrize('shape', [(3, 16, 64, 64, 256, 1), (2, 32, 128, 256, 512, 2), (5, 8, 256, 512, 96, 2)])
def test_f16x3_dual_source_conv(shape):
    """dir_conv2d_dual_scaled_forward: relu((conv1x1(y) + conv1x1_stride(x)) * scale + shift) -- the projection shortcut folded into conv3
    (models/backbone/resnet.py:117-119,137-140) with the split rows' power-of-two prescale coming back out through `scale`"""
    B, S, c1, c2, cout, stride = shape
    rng = np.random.default_rng(sum(shape))
    y = rng.standard_normal((B, c1, S, S)).astype(np.float32)
    x = rng.standard_normal((B, c2, S * stride, S * stride)).astype(np.float32)
    w3 = (rng.standard_normal((cout, c1, 1, 1)) * 0.05).astype(np.float32)
    wd = (rng.standard_normal((cout, c2, 1, 1)) * 0.05).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    ref = N.conv2d(y.astype(np.float64), w3.astype(np.float64), None, 1, 0) + \
        N.conv2d(x.astype(np.float64), wd.astype(np.float64), None, stride, 0) + shift[None, :, None, None]
    ref = np.maximum(ref, 0)
    yd = torch.from_numpy(y).permute(0, 2, 3, 1).contiguous().cuda()
    xd = torch.from_numpy(x).permute(0, 2, 3, 1).contiguous().cuda()
    rows = torch.cat([torch.from_numpy(w3).flatten(1), torch.from_numpy(wd).flatten(1)], 1).contiguous().cuda()
    w, sc = F.pack_f16x3_weights(rows)
    sh = torch.from_numpy(shift).cuda()
    out = torch.empty(B, S, S, cout, device='cuda')
    for variant in (0, 1, 4):
        out.zero_()
        d = _capi.ConvDesc(B, S, S, c1, c1, 0, cout, cout, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F16X3, _capi.DT_F32, 1 | (variant << 8), 0, 0)
        d2 = _capi.ConvSrc2(S * stride, S * stride, c2, c2, 0, stride)
        _capi.check(_capi.lib().dir_conv2d_dual_scaled_forward(d, _capi.ptr(yd), d2, _capi.ptr(xd), _capi.ptr(w), _capi.ptr(sc), _capi.ptr(sh),
                                                               _capi.ptr(out), _capi.stream_ptr()), 'dual')
        e = relerr(out.cpu().numpy().transpose(0, 3, 1, 2), ref)
        assert e < TOL, (variant, e)


def test_f16x3_small_and_large_magnitudes():
    """activations far outside the f16 range in either direction: the per-layer power-of-two input scale (dir_conv_desc.in_scale, chosen
    from the data here exactly as DirEngine.calibrate does) centres them, so the result keeps its accuracy at every magnitude; and with
    NO input scale, values beyond 65504 saturate instead of turning into inf / nan"""
    torch.manual_seed(3)
    w = torch.randn(64, 1, 1, 128, device='cuda') * 0.1
    for mag in (1e-6, 1e-3, 1.0, 3e3, 1e6):
        x = torch.randn(2, 8, 8, 128, device='cuda') * mag
        ref = torch.einsum('bhwc,oc->bhwo', x.double(), w.double().reshape(64, 128))
        y = F.conv2d_nhwc(x, w, arith='f16x3')
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        print('magnitude %g: %.2e' % (mag, e))
        assert e < TOL, (mag, e)
    x = torch.randn(2, 8, 8, 128, device='cuda') * 1e6
    wp, sc = F.pack_f16x3_weights(w.reshape(64, 128))
    y = torch.empty(2, 8, 8, 64, device='cuda')
    d = _capi.ConvDesc(2, 8, 8, 128, 128, 0, 64, 64, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F16X3, _capi.DT_F32, 0)       # in_scale = 0 -> 1
    _capi.check(_capi.lib().dir_conv2d_forward(d, _capi.ptr(x), _capi.ptr(wp), _capi.ptr(sc), None, None, None, None, _capi.ptr(y),
                                               _capi.stream_ptr()), 'conv')
    ref = torch.einsum('bhwc,oc->bhwo', x.clamp(-65504, 65504).double(), w.double().reshape(64, 128))
    assert torch.isfinite(y).all() and float((y.double() - ref).abs().max() / ref.abs().max()) < TOL


@pytest.mark.parametrize('shape', [(64, 32, 256, 256, 3), (64, 8, 2048, 512, 1)])
def test_f16x3_full_size_chunk_consistency(shape):
    """BASELINE config-2 sizes: the full batch equals, bit for bit, the 8-image chunks the oracle comparison covers (repeated)"""
    B, H, Ci, Co, k = shape
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, H, H, Ci, device='cuda', generator=g)
    w = torch.randn(Co, k, k, Ci, device='cuda', generator=g) * 0.05
    ref = torch.cat([F.conv2d_nhwc(x[i:i + 8].contiguous(), w, 1, k // 2, arith='f16x3') for i in range(0, B, 8)], 0)
    assert torch.isfinite(ref).all()
    for _ in range(3):
        assert torch.equal(F.conv2d_nhwc(x, w, 1, k // 2, arith='f16x3'), ref)


@pytest.mark.parametrize('case', CASES[:4])
def test_f16_hi_only_conv(case):
    """DIR_DT_F16X1, the "fp16 MFMA path": one f16 MFMA per product on the same fp32 tensors and packing.  Against the float64 convolution
    of the operands ROUNDED to f16 the way the kernel rounds them (activations and pre-scaled weights: exact powers of two cancel), only
    fp32 accumulation noise remains (5e-6); against the unrounded operands the error is the f16 rounding itself, a few 1e-4 of the output
    scale -- several times below bf16 operands'."""
    B, H, W, Ci, Co, k, s, p = case
    tag = 'conv.%s' % '_'.join(map(str, case))
    x = synth.synth_input(tag + '.x', (B, Ci, H, W), SEED)
    w = synth.synth_input(tag + '.w', (Co, Ci, k, k), SEED) * np.float32(np.sqrt(2.0 / (k * k * Ci)))
    dx = torch.from_numpy(to_nhwc(x)).cuda()
    dw = F.pack_conv_weight(torch.from_numpy(w).cuda(), torch.float32)
    y = F.conv2d_nhwc(dx, dw, s, p, arith='f16').cpu().numpy().transpose(0, 3, 1, 2)
    ref = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, s, p)
    # the kernel's roundings: x * in_scale -> f16 ; w * p_n -> f16 (both scales are powers of two: rounding commutes with them up to range)
    import math
    sx = 2.0 ** (10 - math.frexp(float(np.abs(x).max()))[1])
    xr = (torch.from_numpy(x * np.float32(sx)).half().double().numpy()) / sx
    amax = np.abs(w).reshape(Co, -1).max(1)
    pw = 2.0 ** (13 - np.frexp(amax)[1])
    wr = (torch.from_numpy(w * pw[:, None, None, None].astype(np.float32)).half().double().numpy()) / pw[:, None, None, None]
    ref_r = N.conv2d(xr, wr, None, s, p)
    e_r, e_x = relerr(y, ref_r), relerr(y, ref)
    xb, wb = torch.from_numpy(x).to(torch.bfloat16).double().numpy(), torch.from_numpy(w).to(torch.bfloat16).double().numpy()
    e_bf = relerr(N.conv2d(xb, wb, None, s, p), ref)
    print('%s: f16 arithmetic vs f16-rounded operands %.2e, vs exact operands %.2e (bf16 operands: %.2e)' % (tag, e_r, e_x, e_bf))
    assert e_r < 5e-6 and e_x < 1.5e-3 and e_x < 0.5 * e_bf


def test_device_weight_packing_equals_host_packing():
    """dir_pack_f16x3_weights (one launch, for weights that change every optimiser step) == functional.pack_f16x3_weights, bit for bit"""
    torch.manual_seed(1)
    w = torch.randn(70, 9 * 64, device='cuda') * torch.logspace(-5, 3, 70, device='cuda')[:, None]
    w[5] = 0
    sc_in = torch.rand(70, device='cuda') + 0.5
    ph, sh = F.pack_f16x3_weights(w, sc_in)
    pd, sd = F.pack_f16x3_weights_device(w, sc_in)
    assert torch.equal(ph.view(torch.int16), pd.view(torch.int16)) and torch.equal(sh, sd)
