"""nn.Conv2d's gradients from libdir_hip.so (dir_amd/train/conv.py: dir_conv2d_wgrad_f32, the data gradient as a forward convolution
with flipped / transposed weights) against torch autograd on the same fp32 tensors, for every convolution geometry on the DIR path
(models/backbone/resnet.py, models/backbone/hourglass.py, models/dir.py:57-62,227-241,404-419).  Tolerance 1e-5 of each gradient's
maximum."""
import pytest
import torch

from dir_amd.train import conv as TC

pytestmark = pytest.mark.gpu

CASES = [  # B, H, Cin, Cout, k, stride, pad
    (4, 16, 64, 128, 3, 1, 1),        # bottleneck conv2
    (3, 16, 256, 64, 1, 1, 0),        # bottleneck conv1
    (2, 32, 128, 128, 3, 2, 1),       # stage-entry conv2 (stride 2)
    (2, 32, 256, 512, 1, 2, 0),       # projection shortcut (1x1 stride 2)
    (2, 64, 3, 64, 7, 2, 3),          # stem (no data gradient: the image is a leaf)
    (2, 32, 128, 3, 1, 1, 0),         # seg / dense head (3 outputs)
    (5, 9, 96, 40, 3, 1, 1),          # ragged channel counts and an odd map
    (64, 8, 512, 512, 3, 1, 1),       # layer4 at the benchmark batch (one pixel chunk per 2048 pixels)
]


@pytest.mark.parametrize('case', CASES)
def test_conv_backward_matches_autograd(case):
    B, H, Cin, Cout, k, stride, pad = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, H, H, Cin, device='cuda')
    w = torch.randn(Cout, k, k, Cin, device='cuda') * (k * k * Cin) ** -0.5
    b = torch.randn(Cout, device='cuda')
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.permute(0, 3, 1, 2).clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, br, stride=stride, padding=pad)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    if k * k <= 32:                                        # (the 7x7 stem runs forward through the space-to-depth path of the engine)
        y = TC.conv_fwd(x, w, b, stride, pad)
        assert (y - yr.detach().permute(0, 2, 3, 1)).abs().max() < 2e-5 * yr.abs().max()
    need_gx = Cin != 3
    gx, gw, gb = TC.conv_bwd(x, w, gy.permute(0, 2, 3, 1).contiguous(), stride, pad, need_gx=need_gx)
    e_w = float((gw - wr.grad.permute(0, 2, 3, 1)).abs().max() / wr.grad.abs().max())
    e_b = float((gb - br.grad).abs().max() / br.grad.abs().max())
    assert e_w < 1e-5 and e_b < 1e-5, (e_w, e_b)
    if need_gx:
        e_x = float((gx - xr.grad.permute(0, 2, 3, 1)).abs().max() / xr.grad.abs().max())
        assert e_x < 1e-5, e_x
    gw2 = TC.conv_wgrad(x, gy.permute(0, 2, 3, 1).contiguous(), w.shape, stride, pad)
    assert torch.equal(gw, gw2)                            # fixed chunk order: reproducible bit for bit
    acc = gw.clone()
    TC.conv_wgrad(x, gy.permute(0, 2, 3, 1).contiguous(), w.shape, stride, pad, out=acc, accumulate=True)
    assert (acc - 2 * gw).abs().max() <= 1e-6 * gw.abs().max()


X3_CASES = [  # B, H, Cin, Cout, k, stride, pad, |gy| scale
    (8, 32, 128, 128, 3, 1, 1, 1.0),      # 128 x 128 tiles, several pixel chunks
    (4, 16, 192, 64, 1, 1, 0, 3e-7),      # 64 x 128 tiles with a ragged second tile; tiny gradients (the scale does the work)
    (3, 17, 36, 260, 3, 2, 1, 40.0),      # odd map, stride 2, ragged tiles on both sides
    (16, 32, 2560, 256, 3, 1, 1, 1e-3),   # the bone-fusion convolution's shape (models/dir.py:57-62)
    (4, 64, 64, 128, 3, 2, 1, 1.0),       # stride 2 onto a 32-wide map: the uniform-row address form (Wo % 32 == 0) with column stride 2
    (2, 64, 256, 96, 1, 2, 0, 0.1),       # the 1x1 / stride 2 projection shortcut, 32-wide output, ragged output-channel tile
]


@pytest.mark.parametrize('case', X3_CASES)
def test_split_precision_weight_gradient_vs_float64(case):
    """dir_conv2d_wgrad_f16x3 against the float64 gradient: as close as the exact fp32 kernel (both are bound by fp32 accumulation)"""
    B, H, Cin, Cout, k, stride, pad, gs = case
    torch.manual_seed(sum(int(v) for v in case[:7]))
    x = torch.randn(B, H, H, Cin, device='cuda') * 3.0
    Ho = (H + 2 * pad - k) // stride + 1
    gy = torch.randn(B, Ho, Ho, Cout, device='cuda') * gs
    wr = torch.zeros(Cout, Cin, k, k, device='cuda', dtype=torch.float64, requires_grad=True)
    yr = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wr, None, stride=stride, padding=pad)
    yr.backward(gy.permute(0, 3, 1, 2).double())
    ref = wr.grad.permute(0, 2, 3, 1)
    saved = TC.WGRAD_ARITH
    try:
        TC.WGRAD_ARITH = 'f16x3'
        g3 = TC.conv_wgrad(x, gy, (Cout, k, k, Cin), stride, pad)
        g3b = TC.conv_wgrad(x, gy, (Cout, k, k, Cin), stride, pad)
        TC.WGRAD_ARITH = 'f32'
        g1 = TC.conv_wgrad(x, gy, (Cout, k, k, Cin), stride, pad)
    finally:
        TC.WGRAD_ARITH = saved
    assert torch.equal(g3, g3b)                            # deterministic
    e3 = float((g3.double() - ref).abs().max() / ref.abs().max())
    e1 = float((g1.double() - ref).abs().max() / ref.abs().max())
    assert e3 < max(2.0 * e1, 2e-6), (e3, e1)
    acc = g3.clone()
    TC.WGRAD_ARITH = 'f16x3'
    try:
        TC.conv_wgrad(x, gy, (Cout, k, k, Cin), stride, pad, out=acc, accumulate=True)
    finally:
        TC.WGRAD_ARITH = saved
    assert (acc - 2 * g3).abs().max() <= 1e-6 * g3.abs().max()


def test_split_precision_weight_gradient_rejects_bad_arguments():
    """dir_conv2d_wgrad_f16x3: scales that are not powers of two, channel counts that are not multiples of 4 and a short workspace are
    argument errors (DIR_E_ARG with a message), never a launch"""
    from dir_amd import _capi
    L = _capi.lib()
    x = torch.randn(2, 8, 8, 64, device='cuda')
    gy = torch.randn(2, 8, 8, 64, device='cuda')
    gw = torch.empty(64, 1, 1, 64, device='cuda')
    d = _capi.ConvDesc(2, 8, 8, 64, 64, 0, 64, 64, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F32, _capi.DT_F32, 0)
    ws = torch.empty(max(L.dir_conv2d_wgrad_f16x3_workspace_bytes(d), 4) // 4, device='cuda')

    def call(desc, sx, sg, nbytes, acc=0):
        return L.dir_conv2d_wgrad_f16x3(desc, _capi.ptr(x), _capi.ptr(gy), _capi.ptr(gw), acc, _capi.ptr(ws), nbytes, sx, sg, _capi.stream_ptr())
    assert call(d, 1.0, 1.0, ws.numel() * 4) == 0
    assert call(d, 3.0, 1.0, ws.numel() * 4) != 0 and b'powers of two' in L.dir_last_error()
    assert call(d, 1.0, 0.0, ws.numel() * 4) != 0
    assert call(d, 1.0, 1.0, 0, acc=1) != 0 and b'workspace' in L.dir_last_error()
    d2 = _capi.ConvDesc(2, 8, 8, 62, 64, 0, 64, 64, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F32, _capi.DT_F32, 0)
    assert call(d2, 1.0, 1.0, ws.numel() * 4) != 0 and b'multiples of 4' in L.dir_last_error()
    torch.cuda.synchronize()


def test_weight_pack_is_the_per_call_packing_bit_for_bit():
    """dir_train_pack_conv_weights (one launch for every convolution weight of a step, both operand forms, straight from OIHW) against what the
    per-call path builds: OHWI copy -> dir_pack_f16x3_weights for the forward; flip, (Cin <-> Cout) transpose, Cout padded to 32 -> the same
    packing for the data gradient.  Then with in_scales folded in: the scales are the per-call ones divided by the power of two."""
    from dir_amd import functional as F
    from dir_amd.train import conv as TC
    gen = torch.Generator(device='cuda').manual_seed(5)
    shapes = [(64, 64, 3, 3), (256, 64, 1, 1), (6, 256, 1, 1), (1, 96, 3, 3), (128, 160, 3, 3), (40, 12, 1, 1), (512, 1024, 1, 1)]
    ws = [torch.randn(*s, device='cuda', generator=gen) * (10.0 ** (i - 3)) for i, s in enumerate(shapes)]
    ws[1][5] = 0                                              # an all-zero output channel (p = 1)
    pk = TC.WeightPack(ws)
    pk.refresh()
    for w, e in zip(ws, pk.entries):
        Cout, Cin, kh, kw = w.shape
        if Cin % 32 == 0:
            ref, sc = F.pack_f16x3_weights_device(w.permute(0, 2, 3, 1).contiguous().reshape(Cout, -1))
            assert torch.equal(e.fwd.view(-1), ref.view(-1)) and torch.equal(e.fwd_scale, sc), w.shape
        else:
            assert e.fwd is None
        wt = TC._pad_last(w.permute(0, 2, 3, 1).flip(1, 2).permute(3, 1, 2, 0).contiguous(), 32)
        ref, sc = F.pack_f16x3_weights_device(wt.reshape(Cin, -1))
        assert torch.equal(e.dgrad.view(-1), ref.view(-1)) and torch.equal(e.dgrad_scale, sc), w.shape
    e = pk.entries[0]
    before = (e.fwd_scale.clone(), e.dgrad_scale.clone())
    e.want = [4.0, 0.125]
    ws[0].mul_(3.0)                                           # and the refresh sees the new weight values
    pk.refresh()
    ref, sc = F.pack_f16x3_weights_device(ws[0].permute(0, 2, 3, 1).contiguous().reshape(64, -1))
    assert torch.equal(e.fwd.view(-1), ref.view(-1)) and torch.equal(e.fwd_scale, sc / 4.0) and e.applied == [4.0, 0.125]
    assert not torch.equal(e.fwd_scale, before[0]) and torch.equal(pk.entries[1].fwd_scale, F.pack_f16x3_weights_device(ws[1].reshape(256, -1))[1])


def test_training_convolution_variants_are_bit_identical():
    """The forward kernels' variants (DIR_CONV_VARIANT: tile shapes / pipelines) all accumulate in the same order: the split-precision forward and
    data-gradient convolutions of the training step give the SAME BITS whichever runs -- on a 3x3, a strided 3x3, a 1x1 with a residual operand, a
    wide-K 3x3, a 1-channel head and a wide 1x1.  (Round 4 timed a per-shape choice among them for the training step, as DirEngine.autotune does per
    layer: 0.0400 -> 0.0397 s, within noise -- the heuristic already picks well for fp32 maps -- so the step keeps the heuristic.)"""
    from dir_amd import functional as F
    candidates = (0, 1, 2, 3, 4, 17, 18, 20, 8, 9, 10, 11, 12, 13, 14, 15)
    gen = torch.Generator(device='cuda').manual_seed(11)
    cases = [(4, 32, 128, 128, 3, 1, 1, False), (4, 32, 128, 256, 3, 2, 1, False), (4, 32, 512, 128, 1, 1, 0, True), (2, 16, 2560, 256, 3, 1, 1, False),
             (4, 32, 128, 1, 1, 1, 0, False), (2, 8, 512, 2048, 1, 1, 0, False)]
    for B, H, Cin, Cout, k, stride, pad, res in cases:
        x = torch.randn(B, H, H, Cin, device='cuda', generator=gen)
        w = torch.randn(Cout, k, k, Cin, device='cuda', generator=gen) * 0.05
        Ho = (H + 2 * pad - k) // stride + 1
        r = torch.randn(B, Ho, Ho, Cout, device='cuda', generator=gen) if res else None
        presplit = k >= 3 or (Cout >= 512 and Cin >= 128)
        ref = F.conv2d_nhwc(x, w, stride=stride, pad=pad, arith='f16x3', in_scale=4.0, device_pack=True, presplit=presplit, residual=r, variant=0)
        for v in candidates[1:]:
            y = F.conv2d_nhwc(x, w, stride=stride, pad=pad, arith='f16x3', in_scale=4.0, device_pack=True, presplit=presplit, residual=r, variant=v)
            assert torch.equal(y, ref), ((B, H, Cin, Cout, k, stride), v)
