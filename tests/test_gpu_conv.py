"""GPU parity (a1/a2/a3/a11): MFMA implicit-GEMM convolution through the C ABI vs the numpy oracle conv2d.
fp32 path = v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate): tolerance 2e-5 relative to the output
scale (summation order differs from BLAS).  bf16 path: inputs/weights rounded to bf16, fp32 accumulate; compared
against the oracle run on the bf16-rounded operands, tolerance 1e-2 of the output scale when the output is bf16."""
import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import functional as F
from dir_amd import synth
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234

CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 1, 1, 0),
    (3, 15, 13, 64, 96, 3, 2, 1),       # odd sizes, stride 2, Cout tail inside a tile, M tail
    (2, 32, 32, 256, 128, 1, 2, 0),     # strided 1x1 (ResNet downsample)
    (1, 8, 8, 2048, 200, 3, 1, 1),      # long K (InitRegressor attention shape, Cout cut down)
    (2, 9, 9, 64, 130, 3, 1, 1),        # two N tiles with a ragged second one
]


def to_nhwc(a):
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1))


def bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv_matches_oracle(case, dtype):
    B, H, W, Ci, Co, k, s, p = case
    tag = 'conv.%s' % '_'.join(map(str, case))
    x = synth.synth_input(tag + '.x', (B, Ci, H, W), SEED)
    w = synth.synth_input(tag + '.w', (Co, Ci, k, k), SEED) * np.float32(np.sqrt(2.0 / (k * k * Ci)))
    scale = synth.synth_input(tag + '.s', (Co,), SEED, kind='uniform', lo=0.5, hi=1.5)
    shift = synth.synth_input(tag + '.b', (Co,), SEED) * np.float32(0.3)
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    if dtype == 'bf16':
        x, w = bf16_round(x), bf16_round(w)
    ref = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, s, p)
    ref = ref * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    ref_relu = np.maximum(ref, 0)
    dx = torch.from_numpy(to_nhwc(x)).cuda().to(tdt)
    dw = F.pack_conv_weight(torch.from_numpy(w).cuda(), tdt)
    y = F.conv2d_nhwc(dx, dw, s, p, torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda(), relu=True)
    got = y.float().cpu().numpy().transpose(0, 3, 1, 2)
    tol = 2e-5 if dtype == 'f32' else 1e-2
    assert relerr(got, ref_relu) < tol
    if dtype == 'bf16':     # fp32 output from bf16 operands: only accumulation-order error remains
        y32 = F.conv2d_nhwc(dx, dw, s, p, torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda(),
                            out_dtype=torch.float32)
        assert relerr(y32.cpu().numpy().transpose(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv_prologue_residual_and_concat_slices(dtype):
    """pre-activation BN+ReLU on the input (hourglass.Residual), residual add, and channel-slice I/O."""
    B, H, W, Ci, Co = 2, 12, 12, 128, 64
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    xbuf = synth.synth_input('convp.x', (B, Ci + 64, H, W), SEED)
    w = synth.synth_input('convp.w', (Co, Ci, 3, 3), SEED) * np.float32(0.05)
    ps = synth.synth_input('convp.ps', (Ci,), SEED, kind='uniform', lo=0.5, hi=1.5)
    pb = synth.synth_input('convp.pb', (Ci,), SEED) * np.float32(0.5)
    res = synth.synth_input('convp.res', (B, Co, H, W), SEED)
    bias = synth.synth_input('convp.bias', (Co,), SEED)
    if dtype == 'bf16':
        xbuf, w, res = bf16_round(xbuf), bf16_round(w), bf16_round(res)
    xin = xbuf[:, 64:]                                       # the conv reads channels [64, 64+Ci)
    act = np.maximum(xin.astype(np.float64) * ps.reshape(1, -1, 1, 1) + pb.reshape(1, -1, 1, 1), 0)
    if dtype == 'bf16':
        act = bf16_round(act.astype(np.float32)).astype(np.float64)   # the kernel re-rounds the activated input
    ref = N.conv2d(act, w.astype(np.float64), bias.astype(np.float64), 1, 1) + res
    dx = torch.from_numpy(to_nhwc(xbuf)).cuda().to(tdt)
    dw = F.pack_conv_weight(torch.from_numpy(w).cuda(), tdt)
    out = torch.full((B, H, W, Co + 32), 7.0, device='cuda', dtype=tdt)
    dres = torch.from_numpy(to_nhwc(res)).cuda().to(tdt)
    F.conv2d_nhwc(dx, dw, 1, 1, None, torch.from_numpy(bias).cuda(), residual=dres, pre_scale=torch.from_numpy(ps).cuda(),
                  pre_shift=torch.from_numpy(pb).cuda(), pre_relu=True, out=out, out_coff=32, in_coff=64, cin=Ci)
    got = out.float().cpu().numpy()
    assert np.all(got[..., :32] == 7.0)                     # untouched slice of the concat buffer
    assert relerr(got[..., 32:].transpose(0, 3, 1, 2), ref) < (2e-5 if dtype == 'f32' else 1e-2)


def test_conv_linearity_full_size():
    """size-independent property at a BASELINE config-2 shape (B=64, fusion conv 2560->256 @16x16, bf16):
    conv(x1 + x2) == conv(x1) + conv(x2) up to bf16 output rounding, and zero input -> shift only."""
    B, S, Ci, Co = 64, 16, 2560, 256
    g = torch.Generator(device='cuda').manual_seed(1)
    x1 = (torch.randn(B, S, S, Ci, device='cuda', generator=g) * 0.5).bfloat16()
    x2 = torch.zeros_like(x1)
    x2[:, ::3, ::2, ::5] = 1.0
    w = (torch.randn(Co, 3, 3, Ci, device='cuda', generator=g) * 0.01).bfloat16()
    y1 = F.conv2d_nhwc(x1, w, 1, 1, out_dtype=torch.float32)
    y2 = F.conv2d_nhwc(x2, w, 1, 1, out_dtype=torch.float32)
    y12 = F.conv2d_nhwc((x1.float() + x2.float()).bfloat16(), w, 1, 1, out_dtype=torch.float32)
    # x1+x2 is re-rounded to bf16: compare against conv of the rounded sum's decomposition instead
    xs = (x1.float() + x2.float()).bfloat16()
    y_rest = F.conv2d_nhwc((xs.float() - x2.float()).bfloat16(), w, 1, 1, out_dtype=torch.float32)
    scale = float(y12.abs().max())
    assert float((y12 - (y_rest + y2)).abs().max()) / scale < 2e-2
    assert float((y1 - y_rest).abs().max()) / scale < 2e-2
    z = F.conv2d_nhwc(torch.zeros_like(x1), w, 1, 1, shift=torch.arange(Co, device='cuda', dtype=torch.float32),
                      out_dtype=torch.float32)
    assert torch.equal(z[3, 5, 7], torch.arange(Co, device='cuda', dtype=torch.float32))


@pytest.mark.parametrize('shape', [(64, 64, 64, 256, 1), (64, 64, 128, 256, 1), (64, 64, 192, 256, 1), (64, 64, 64, 256, 3),
                                   (64, 32, 512, 128, 1), (64, 8, 2048, 512, 1)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_conv_full_size_chunk_consistency(shape, dtype):
    """BASELINE config-2 sizes (B=64, thousands of workgroups in flight).  Size-independent property: every image is
    convolved independently, so the full-batch result must equal, bit for bit, the result of running 8-image chunks
    (a size the oracle comparison above covers) -- repeated, because a software-pipelining bug shows up only
    intermittently and only under a full grid."""
    B, H, Ci, Co, k = shape
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, H, H, Ci, device='cuda', generator=g).to(dtype)
    w = (torch.randn(Co, k, k, Ci, device='cuda', generator=g) * 0.05).to(dtype)
    ref = torch.cat([F.conv2d_nhwc(x[i:i + 8].contiguous(), w, 1, k // 2) for i in range(0, B, 8)], 0)
    assert torch.isfinite(ref.float()).all()
    for _ in range(4):
        y = F.conv2d_nhwc(x, w, 1, k // 2)
        assert torch.equal(y, ref)


@pytest.mark.parametrize('shape', [(16, 32, 256, 256, 3, 1), (8, 32, 512, 384, 1, 1), (9, 30, 128, 200, 3, 2), (64, 32, 256, 256, 3, 1)])
@pytest.mark.parametrize('odt', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('pre', [False, True])
def test_big_tile_variant_is_bit_identical(shape, odt, pre):
    """DIR_CONV_VARIANT 11 (conv_big.hip, 256 x 256 block tile) and 15 (conv_pipe.hip, 128 x 64 tile on eight waves, deep ring): the same K order and fp32 accumulation order as the 4-wave kernel, so the
    same bits -- whole and ragged tiles in both directions, stride 2, scale / shift / residual / ReLU, bf16 and fp32 outputs, a channel-slice output."""
    B, H, Ci, Co, k, stride = shape
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    x = torch.randn(B, H, H, Ci, device='cuda', generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, device='cuda', generator=g) * 0.03).bfloat16()
    sc = torch.rand(Co, device='cuda', generator=g) + 0.5
    sh = torch.randn(Co, device='cuda', generator=g)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Co, device='cuda', generator=g).to(odt)
    ps, pb = torch.rand(Ci, device='cuda', generator=g) + 0.5, torch.randn(Ci, device='cuda', generator=g) * 0.5     # (pre-activation: 11 falls back)
    outs = []
    for variant in (1, 11, 15):
        out = torch.full((B, Ho, Ho, Co + 8), 3.0, device='cuda', dtype=odt)
        F.conv2d_nhwc(x, w, stride, k // 2, sc, sh, relu=True, residual=res, out=out, out_coff=8, variant=variant,
                      pre_scale=ps if pre else None, pre_shift=pb if pre else None, pre_relu=pre)
        outs.append(out)
    assert torch.isfinite(outs[0].float()).all() and float(outs[0][..., 8:].float().abs().max()) > 0.5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert bool((outs[1][..., :8] == 3.0).all()) and bool((outs[2][..., :8] == 3.0).all())


@pytest.mark.parametrize('dt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize('shape', [(3, 16, 64, 64, 256, 1), (2, 32, 128, 256, 512, 2), (5, 8, 256, 512, 96, 2)])
def test_dual_source_conv_vs_oracle(dt, tol, shape):
    """dir_conv2d_dual_forward: relu(conv1x1(y) + conv1x1_stride(x) + shift), the projection shortcut folded into conv3
    (models/backbone/resnet.py:117-119,137-140), against the float64 composition of the two convolutions."""
    from dir_amd import _capi
    from oracle import nnops as N
    B, S, c1, c2, cout, stride = shape
    rng = np.random.default_rng(sum(shape))
    y = rng.standard_normal((B, c1, S, S)).astype(np.float32)
    x = rng.standard_normal((B, c2, S * stride, S * stride)).astype(np.float32)
    w3 = (rng.standard_normal((cout, c1, 1, 1)) * 0.05).astype(np.float32)
    wd = (rng.standard_normal((cout, c2, 1, 1)) * 0.05).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32)
    q = (lambda a: torch.from_numpy(a).to(dt).float().numpy())
    ref = N.conv2d(q(y).astype(np.float64), q(w3).astype(np.float64), None, 1, 0) + \
        N.conv2d(q(x).astype(np.float64), q(wd).astype(np.float64), None, stride, 0) + shift[None, :, None, None]
    ref = np.maximum(ref, 0)
    yd = torch.from_numpy(y).permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    xd = torch.from_numpy(x).permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    w = torch.cat([torch.from_numpy(w3).flatten(1), torch.from_numpy(wd).flatten(1)], 1).contiguous().to(dt).cuda()
    sh = torch.from_numpy(shift).cuda()
    out = torch.empty(B, S, S, cout, device='cuda', dtype=dt)
    code = 0 if dt == torch.float32 else 1
    for variant in (0, 1, 4, 19):
        out.zero_()
        d = _capi.ConvDesc(B, S, S, c1, c1, 0, cout, cout, 0, 0, 0, 1, 1, 1, 0, code, code, 1 | (variant << 8), 0, 0)
        d2 = _capi.ConvSrc2(S * stride, S * stride, c2, c2, 0, stride)
        _capi.check(_capi.lib().dir_conv2d_dual_forward(d, _capi.ptr(yd), d2, _capi.ptr(xd), _capi.ptr(w), _capi.ptr(sh), _capi.ptr(out),
                                                        _capi.stream_ptr()), 'dual')
        got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (variant, np.abs(got - ref).max())
    d2 = _capi.ConvSrc2(S * stride + 1, S * stride, c2, c2, 0, stride + 1)            # wrong geometry is rejected
    assert _capi.lib().dir_conv2d_dual_forward(d, _capi.ptr(yd), d2, _capi.ptr(xd), _capi.ptr(w), _capi.ptr(sh), _capi.ptr(out),
                                               _capi.stream_ptr()) != 0
