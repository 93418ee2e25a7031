"""GPU parity of the f16 STORAGE mode (include/dir_hip.h: DIR_DT_F16; round 5): every bf16 feature-map kernel has a twin
instantiated on IEEE f16 -- same bytes, layouts, LDS-DMA path and accumulation order, v_mfma_*_f16 instead of _bf16, stores saturating at
+-65504 through MODE.FP16_OVFL.  Each kernel is held to the float64 oracle on f16-rounded operands (the tolerances of the bf16 tests scaled
by the 8x finer rounding), the kernel variants to each other bit for bit, the fused chain / tail / stem / stream kernels to the unfused
f16 sequences, and the saturation to exact values.  The end-to-end gate (every stage of both hands inside 0.01 mm of the reference's own
forward on trained-like weights) is tests/test_gpu_dir.py::test_engine_vs_reference_golden_trained_like_weights[f16s].

Replaces the same reference calls as the bf16 kernels: models/backbone/resnet.py:117-140,243-255, models/backbone/hourglass.py:55-70,
models/dir.py:57-62,227-241,404-420."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import functional as F
from dir_amd import synth
from dir_amd.engine import DirEngine, ForwardPipeline
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234
H16 = torch.float16
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def h_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(H16).float().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(a.transpose(0, 2, 3, 1)).to(H16)


CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 1, 1, 0),
    (3, 15, 13, 64, 96, 3, 2, 1),       # odd sizes, stride 2, Cout tail inside a tile, M tail
    (2, 32, 32, 256, 128, 1, 2, 0),
    (1, 8, 8, 2048, 200, 3, 1, 1),      # long K
    (2, 9, 9, 64, 130, 3, 1, 1),
]


@pytest.mark.parametrize('case', CASES)
def test_f16_conv_matches_oracle(case):
    B, H, W, Ci, Co, k, s, p = case
    tag = 'conv.%s' % '_'.join(map(str, case))
    x = h_round(synth.synth_input(tag + '.x', (B, Ci, H, W), SEED))
    w = h_round(synth.synth_input(tag + '.w', (Co, Ci, k, k), SEED) * np.float32(np.sqrt(2.0 / (k * k * Ci))))
    scale = synth.synth_input(tag + '.s', (Co,), SEED, kind='uniform', lo=0.5, hi=1.5)
    shift = synth.synth_input(tag + '.b', (Co,), SEED) * np.float32(0.3)
    ref = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, s, p) * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    dx, dw = nhwc(x), F.pack_conv_weight(dev(w), H16)
    y = F.conv2d_nhwc(dx, dw, s, p, dev(scale), dev(shift), relu=True)
    assert y.dtype == H16
    assert relerr(y.float().cpu().numpy().transpose(0, 3, 1, 2), np.maximum(ref, 0)) < 1.5e-3       # f16 output: 2^-11 per value (bf16 test: 1e-2)
    y32 = F.conv2d_nhwc(dx, dw, s, p, dev(scale), dev(shift), out_dtype=torch.float32)            # fp32 output: accumulation order only
    assert relerr(y32.cpu().numpy().transpose(0, 3, 1, 2), ref) < 2e-5


def test_f16_conv_prologue_residual_and_concat_slices():
    """pre-activation BN + ReLU on the input (hourglass.Residual), residual add, channel-slice input and output -- the f16 instantiation"""
    B, H, W, Ci, Co = 2, 12, 12, 128, 64
    xbuf = h_round(synth.synth_input('convp.x', (B, Ci + 64, H, W), SEED))
    w = h_round(synth.synth_input('convp.w', (Co, Ci, 3, 3), SEED) * np.float32(0.05))
    ps = synth.synth_input('convp.ps', (Ci,), SEED, kind='uniform', lo=0.5, hi=1.5)
    pb = synth.synth_input('convp.pb', (Ci,), SEED) * np.float32(0.5)
    res = h_round(synth.synth_input('convp.res', (B, Co, H, W), SEED))
    bias = synth.synth_input('convp.bias', (Co,), SEED)
    act = np.maximum(xbuf[:, 64:].astype(np.float64) * ps.reshape(1, -1, 1, 1) + pb.reshape(1, -1, 1, 1), 0)
    act = h_round(act.astype(np.float32)).astype(np.float64)            # the kernel re-rounds the activated input
    ref = N.conv2d(act, w.astype(np.float64), bias.astype(np.float64), 1, 1) + res
    out = torch.zeros(B, H, W, Co + 32, device='cuda', dtype=H16)
    F.conv2d_nhwc(nhwc(xbuf), F.pack_conv_weight(dev(w), H16), 1, 1, None, dev(bias), pre_scale=dev(ps), pre_shift=dev(pb), pre_relu=True,
                  residual=nhwc(res), out=out, out_coff=32, in_coff=64, cin=Ci)
    assert relerr(out[..., 32:].float().cpu().numpy().transpose(0, 3, 1, 2), ref) < 1.5e-3
    assert float(out[..., :32].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(16, 32, 256, 256, 3, 1), (8, 16, 512, 384, 1, 1), (16, 16, 256, 256, 3, 1), (32, 8, 512, 512, 3, 1)])
@pytest.mark.parametrize('odt', [H16, torch.float32])
def test_f16_conv_variants_are_bit_identical(shape, odt):
    """every kernel variant (4-wave tiles, 8-wave pipelined, halo-reuse incl. the 16x16 / 8x8 maps whose patch swizzle was re-keyed in round 5,
    256x256 tile) accumulates in the same order: the f16 instantiations give the same bits, as the bf16 ones do"""
    B, S, Ci, Co, k, s = shape
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, S, S, Ci, device='cuda', generator=g).to(H16)
    w = (torch.randn(Co, k, k, Ci, device='cuda', generator=g) * (2.0 / (k * k * Ci)) ** 0.5).to(H16)
    sc = torch.rand(Co, device='cuda', generator=g) + 0.5
    sh = torch.randn(Co, device='cuda', generator=g) * 0.3
    ref = F.conv2d_nhwc(x, w, s, k // 2, sc, sh, relu=True, out_dtype=odt, variant=1)
    for v in (2, 3, 4, 17, 18, 20, 8, 9, 10, 11, 12, 13, 14, 15):
        y = F.conv2d_nhwc(x, w, s, k // 2, sc, sh, relu=True, out_dtype=odt, variant=v)
        assert torch.equal(y, ref), 'variant %d' % v


def test_f16_stores_saturate_instead_of_overflowing():
    """MODE.FP16_OVFL (csrc/conv_common.h: half_kernel_init): a result beyond the f16 range is stored as +-65504, never as inf, at no VALU
    cost -- through the coalesced epilogue of the 4-wave kernel, the 8-wave kernels' shared epilogue and the streaming kernel"""
    B, S, Ci, Co = 4, 16, 128, 128
    x = torch.full((B, S, S, Ci), 64.0, device='cuda', dtype=H16)
    w = torch.full((Co, 1, 1, Ci), 16.0, device='cuda', dtype=H16)           # 128 * 64 * 16 = 131 072 > 65504
    w[1::2] = -16.0
    for variant in (1, 9, 15):
        y = F.conv2d_nhwc(x, w, 1, 0, variant=variant)
        assert bool(torch.isfinite(y).all()), variant
        assert float(y[..., 0::2].min()) == 65504.0 and float(y[..., 1::2].max()) == -65504.0, variant
        yr = F.conv2d_nhwc(x, w, 1, 0, relu=True, variant=variant)
        assert float(yr[..., 0::2].min()) == 65504.0 and float(yr[..., 1::2].abs().max()) == 0.0, variant
    ys = F.conv1x1_stream(x, w.reshape(Co, Ci).float())
    assert float(ys[..., 0::2].min()) == 65504.0 and float(ys[..., 1::2].max()) == -65504.0
    # in range, the conversion is the ordinary round-to-nearest-even
    w2 = (torch.randn(Co, 1, 1, Ci, device='cuda') * 0.01).to(H16)
    y32 = F.conv2d_nhwc(x, w2, 1, 0, out_dtype=torch.float32)
    assert torch.equal(F.conv2d_nhwc(x, w2, 1, 0), y32.to(H16))


STREAM_CASES = [(2, 16, 16, 256, 128, False), (3, 16, 16, 512, 256, True), (1, 32, 32, 128, 512, False)]


@pytest.mark.parametrize('case', STREAM_CASES)
def test_f16_stream_conv_equals_the_tiled_kernels(case):
    """dir_conv1x1_stream_forward on f16: the same k-slots in the same order as the tiled kernels -> bit-identical, with and without the
    pre-activation, for the 128- / 64- / 32-pixel workgroup variants"""
    B, H, W, Ci, Co, pre = case
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g).to(H16)
    w = (torch.randn(Co, Ci, device='cuda', generator=g) * (2.0 / Ci) ** 0.5)
    sc, sh = torch.rand(Co, device='cuda', generator=g) + 0.5, torch.randn(Co, device='cuda', generator=g) * 0.3
    ps, pb = (torch.rand(Ci, device='cuda', generator=g) + 0.5, torch.randn(Ci, device='cuda', generator=g) * 0.3) if pre else (None, None)
    ref = F.conv2d_nhwc(x, w.to(H16).reshape(Co, 1, 1, Ci).contiguous(), 1, 0, sc, sh, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=pre, variant=1)
    for v in (0, 22, 23):
        y = F.conv1x1_stream(x, w, sc, sh, relu=True, pre_scale=ps, pre_shift=pb, pre_relu=pre, variant=v)
        assert y.dtype == H16 and torch.equal(y, ref), v


@pytest.mark.parametrize('geom', [(128, 128), (128, 256), (256, 256)])
@pytest.mark.parametrize('waves', [8, 4])
def test_f16_tail_vs_the_unfused_pair(geom, waves):
    """dir_bottleneck_tail_forward (dtype = DIR_DT_F16) against the two f16 dir_conv2d_forward launches it replaces: same rounding points;
    the GEMMs are computed transposed, so a sum may differ in its last bit and an output on a rounding boundary by one f16 ulp"""
    P, N2 = geom
    B, H, W = (8, 32, 32) if P == 128 else (16, 16, 16)
    C4 = 4 * P
    gi = lambda n, shp, **k: synth.synth_input('tailh.%d_%d.%s' % (P, N2, n), shp, SEED, **k)  # noqa: E731
    y2, res = np.maximum(h_round(gi('y2', (B, P, H, W))), 0), np.maximum(h_round(gi('res', (B, C4, H, W))), 0)
    w3 = h_round(gi('w3', (C4, P, 1, 1)) * np.float32(np.sqrt(2.0 / P)))
    w1 = h_round(gi('w1', (N2, C4, 1, 1)) * np.float32(np.sqrt(2.0 / C4)))
    s3, h3 = gi('s3', (C4,), kind='uniform', lo=0.5, hi=1.5), gi('h3', (C4,)) * np.float32(0.3)
    s1, h1 = gi('s1', (N2,), kind='uniform', lo=0.5, hi=1.5), gi('h1', (N2,)) * np.float32(0.3)
    out, y1n = F.bottleneck_tail(nhwc(y2), dev(w3.reshape(C4, P)), dev(s3), dev(h3), nhwc(res), dev(w1.reshape(N2, C4)), dev(s1), dev(h1), waves=waves)
    assert out.dtype == H16 and y1n.dtype == H16
    o = F.conv2d_nhwc(nhwc(y2), F.pack_conv_weight(dev(w3), H16), 1, 0, dev(s3), dev(h3), relu=True, residual=nhwc(res))
    n1 = F.conv2d_nhwc(o, F.pack_conv_weight(dev(w1), H16), 1, 0, dev(s1), dev(h1), relu=True)
    d = (out.float() - o.float()).abs()
    assert float(d.max()) <= float(o.float().abs().max()) * 2.0 ** -9      # at most an f16 ulp of the output scale
    assert float((d > 0).float().mean()) < 0.02
    assert relerr(y1n.float().cpu().numpy(), n1.float().cpu().numpy()) < 1.5e-3
    # and against the float64 oracle with the same rounding points
    aff = lambda t, s, h: t * s.reshape(1, -1, 1, 1) + h.reshape(1, -1, 1, 1)  # noqa: E731
    ro = h_round(np.maximum(aff(N.conv2d(y2.astype(np.float64), w3.astype(np.float64)), s3, h3) + res, 0).astype(np.float32))
    assert relerr(out.float().cpu().numpy().transpose(0, 3, 1, 2), ro) < 1.5e-3


@pytest.mark.parametrize('res,nxt,dual', [(True, True, False), (False, False, False), (False, True, True)])
def test_f16_bneck_chain_vs_unfused_sequence(res, nxt, dual):
    """dir_bottleneck_chain_forward (dtype = DIR_DT_F16) against the unfused f16 conv2 -> conv3 (+ residual | + projection shortcut) -> next conv1"""
    B, H, W = 3, 16, 32
    gi = lambda n, shp, **k: synth.synth_input('bneckh.%s' % n, shp, SEED, **k)  # noqa: E731
    y1 = np.maximum(h_round(gi('y1', (B, 64, H, W))), 0)
    rs = h_round(gi('res', (B, 256, H, W)))
    x2 = np.maximum(h_round(gi('x2', (B, 64, H, W))), 0)
    w2 = h_round(gi('w2', (64, 64, 3, 3)) * np.float32(np.sqrt(2.0 / 576)))
    w3 = h_round(gi('w3', (256, 64, 1, 1)) * np.float32(np.sqrt(2.0 / 64)))
    wd = h_round(gi('wd', (256, 64, 1, 1)) * np.float32(np.sqrt(2.0 / 64)))
    w1 = h_round(gi('w1', (64, 256, 1, 1)) * np.float32(np.sqrt(2.0 / 256)))
    s2, h2 = gi('s2', (64,), kind='uniform', lo=0.5, hi=1.5), gi('h2', (64,)) * np.float32(0.3)
    s3, h3 = (np.ones(256, np.float32) if dual else gi('s3', (256,), kind='uniform', lo=0.5, hi=1.5)), gi('h3', (256,)) * np.float32(0.3)
    s1, h1 = gi('s1', (64,), kind='uniform', lo=0.5, hi=1.5), gi('h1', (64,)) * np.float32(0.3)
    out, y1n = F.bottleneck_chain(nhwc(y1), F.pack_conv_weight(dev(w2), H16), dev(s2), dev(h2), dev(w3.reshape(256, 64)).to(H16), dev(s3), dev(h3),
                                  residual=nhwc(rs) if res else None, nxt=(dev(w1.reshape(64, 256)).to(H16), dev(s1), dev(h1)) if nxt else None,
                                  dual=(nhwc(x2), dev(wd.reshape(256, 64)).to(H16)) if dual else None)
    y2 = F.conv2d_nhwc(nhwc(y1), F.pack_conv_weight(dev(w2), H16), 1, 1, dev(s2), dev(h2), relu=True)
    aff = lambda t, s, h: t * s.reshape(1, -1, 1, 1) + h.reshape(1, -1, 1, 1)  # noqa: E731
    y2o = h_round(np.maximum(aff(N.conv2d(y1.astype(np.float64), w2.astype(np.float64), None, 1, 1), s2, h2), 0).astype(np.float32))
    o = aff(N.conv2d(y2o.astype(np.float64), w3.astype(np.float64)), s3, h3)
    if res:
        o = o + rs
    if dual:
        o = o + N.conv2d(x2.astype(np.float64), wd.astype(np.float64))
    oo = h_round(np.maximum(o, 0).astype(np.float32))
    assert relerr(y2.float().cpu().numpy().transpose(0, 3, 1, 2), y2o) < 1.5e-3
    assert out.dtype == H16 and relerr(out.float().cpu().numpy().transpose(0, 3, 1, 2), oo) < 1.5e-3
    if nxt:
        n1 = np.maximum(aff(N.conv2d(oo.astype(np.float64), w1.astype(np.float64)), s1, h1), 0)
        assert y1n.dtype == H16 and relerr(y1n.float().cpu().numpy().transpose(0, 3, 1, 2), n1) < 1.5e-3


@pytest.mark.parametrize('u8', [False, True])
def test_f16_stem_pool_vs_oracle(u8):
    """dir_stem_pool_forward_dt(out_dtype = DIR_DT_F16): conv1 7x7/2 + bn1 + ReLU + MaxPool(3, 2, 1) on f16 operands (models/backbone/resnet.py:244-247;
    uint8 frames: apps/eval.py:59-61 inside the kernel) against the float64 oracle on the f16-rounded image and weights"""
    B, H, W = 2, 64, 96
    w = synth.synth_input('stemh.w', (64, 3, 7, 7), SEED) * np.float32(np.sqrt(2.0 / 147))
    sc, sh = synth.synth_input('stemh.s', (64,), SEED, kind='uniform', lo=0.5, hi=1.5), synth.synth_input('stemh.b', (64,), SEED) * np.float32(0.3)
    if u8:
        rng = np.random.default_rng(5)
        frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
        img = ((frames[..., ::-1].astype(np.float32) / np.float32(255.0) - mean) / std).transpose(0, 3, 1, 2)
        dimg = torch.from_numpy(frames).cuda()
    else:
        img = synth.synth_input('stemh.img', (B, 3, H, W), SEED)
        dimg = dev(img)
    y = F.stem_pool(dimg, F.pack_stem_weight(dev(w), H16), dev(sc), dev(sh))
    assert y.dtype == H16 and y.shape == (B, H // 4, W // 4, 64)
    c = N.conv2d(h_round(img).astype(np.float64), h_round(w).astype(np.float64), None, 2, 3) * sc.reshape(1, -1, 1, 1) + sh.reshape(1, -1, 1, 1)
    c = h_round(np.maximum(c, 0).astype(np.float32))
    ref = N.maxpool3x3s2p1(c)
    assert relerr(y.float().cpu().numpy().transpose(0, 3, 1, 2), ref) < 1.5e-3


# ------------------------------------------------------------------------------------------------------------ the whole network
@pytest.fixture(scope='module')
def dir_state():
    with open(os.path.join(GOLDEN, 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, SEED).items()}
    img = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), SEED)).cuda()
    return sd, img


def _snap(o):
    return [o[i][k].clone() for i in range(3) for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_uv_left', 'pd_offset')] + [o[3]['seg'].clone(), o[3]['proj_feat'].clone()]


def test_f16_engine_autotuned_and_unfused_paths_agree(dir_state):
    """the f16 engine: autotuned kernel variants give the bits of the library heuristic; the golden-pinned images keep their outputs as rows
    0 / 63 of a 64-image batch (batch-size independence), four forwards in flight reproduce one at a time; the refined stages stay inside the
    bf16 mode's envelope of the reference golden (random weights: G7)"""
    sd, img = dir_state
    g = np.load(os.path.join(GOLDEN, 'g7_dir.npz'))
    eng = DirEngine(sd, dtype=torch.float16)
    small = _snap(eng.forward(img))
    big = torch.randn(64, 3, 256, 256, device='cuda', generator=torch.Generator(device='cuda').manual_seed(64))
    big[0], big[63] = img[0], img[1]
    o = _snap(eng.forward(big))
    for a, b in zip(small, o):
        assert torch.equal(a, b[[0, 63]])
    eng.autotune(big)
    o2 = _snap(eng.forward(big))
    assert all(torch.equal(a, b) for a, b in zip(o, o2))
    pipe = ForwardPipeline(eng, [big.clone() for _ in range(4)])
    for _ in range(3):
        for s in range(4):
            pipe.launch(s)
    torch.cuda.synchronize()
    for s in range(4):
        assert all(torch.equal(a, b) for a, b in zip(o, _snap(pipe.wait(s)))), s
    for side in ('left', 'right'):
        d = eng.forward(img)[2]['pd_joint_xyz_' + side].cpu().numpy() - g['s2.pd_joint_xyz_' + side]
        assert float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3 < 0.01
    # uint8 frames through the f16 stem == the float path on the same pixels (the LUT holds f16 values of the reference's own fp32 arithmetic)
    rng = np.random.default_rng(3)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)).cuda()
    from dir_amd.apps.eval import normalize_images
    a, b = _snap(eng.forward(frames)), _snap(eng.forward(normalize_images(frames)))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
