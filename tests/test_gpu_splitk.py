"""dir_conv2d_splitk_forward (the reduction of one output tile shared by several workgroups) against the unsplit kernel and a float64
reference of the same bf16 operands: same result to bf16 rounding, independent of the arrival order (bit-identical across repeated
launches on one workspace), counters left ready for the next launch, tails in M and N, pre-activation, residual + ReLU epilogue."""
import pytest
import torch

from dir_amd import _capi
from dir_amd import functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, Cin, Cout, k, stride, splits, residual, relu, pre
    (64, 8, 512, 512, 3, 1, 4, False, True, False),       # ResNet layer4 conv2
    (64, 8, 2048, 512, 1, 1, 4, False, True, False),      # layer4 conv1
    (64, 16, 512, 512, 3, 2, 3, False, True, False),      # layer4 block 0 conv2 (stride 2), uneven slab shares
    (64, 16, 128, 128, 3, 1, 2, False, True, False),      # decoder Residual conv2 @16x16
    (64, 16, 1024, 128, 1, 1, 4, False, True, True),      # decoder Residual conv1: pre-activation BN + ReLU
    (5, 9, 256, 200, 3, 1, 5, True, True, False),         # M = 405 and N = 200: tails in both directions, residual add
    (3, 16, 64, 128, 1, 1, 1, True, False, False),        # splits = 1 degenerates to the plain kernel
]


def ref64(x, w, stride, pad, scale, shift, relu, residual, pre):
    xf = x.double()
    if pre is not None:
        xf = torch.relu(xf * pre[0].double() + pre[1].double()).to(torch.bfloat16).double()     # the kernel rounds the activated operand to bf16
    y = torch.nn.functional.conv2d(xf.permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    y = y * scale.double() + shift.double()
    if residual is not None:
        y = y + residual.double()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize('case', CASES)
def test_splitk_matches_unsplit_and_reference(case):
    B, H, Cin, Cout, k, stride, S, use_res, relu, use_pre = case
    torch.manual_seed(sum(case[:6]))
    pad = k // 2
    x = torch.randn(B, H, H, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') * (k * k * Cin) ** -0.5).to(torch.bfloat16)
    scale = torch.rand(Cout, device='cuda') + 0.5
    shift = torch.randn(Cout, device='cuda')
    pre = (torch.rand(Cin, device='cuda') + 0.5, torch.randn(Cin, device='cuda') * 0.3) if use_pre else None
    Ho = (H + 2 * pad - k) // stride + 1
    res = torch.randn(B, Ho, Ho, Cout, device='cuda').to(torch.bfloat16) if use_res else None
    kw = dict(stride=stride, pad=pad, scale=scale, shift=shift, relu=relu, residual=res,
              pre_scale=None if pre is None else pre[0], pre_shift=None if pre is None else pre[1], pre_relu=use_pre)
    plain = F.conv2d_nhwc(x, w, **kw)
    ws = torch.zeros(_capi.lib().dir_conv2d_splitk_workspace_bytes(
        _capi.ConvDesc(B, H, H, Cin, Cin, 0, Cout, Cout, 0, 0, 0, k, k, stride, pad, _capi.DT_BF16, _capi.DT_BF16, 0), S), dtype=torch.uint8, device='cuda')
    outs = [F.conv2d_nhwc(x, w, splits=S, workspace=ws, **kw).clone() for _ in range(4)]      # same workspace, back to back
    torch.cuda.synchronize()
    assert all(torch.equal(o, outs[0]) for o in outs[1:]), 'split-K result depends on the arrival order / counters not reset'
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    ref = ref64(x, w, stride, pad, scale, shift, relu, res, pre)
    ulp = ref.abs().clamp_min(2.0 ** -6) * 2.0 ** -8                                 # one bf16 ulp (8 significant bits)
    e_split = float(((outs[0].double() - ref).abs() / ulp).max())
    e_plain = float(((plain.double() - ref).abs() / ulp).max())
    assert e_split < 1.01 and e_plain < 1.01, (e_split, e_plain)                      # both: correctly rounded up to fp32 summation noise
    differ = float((outs[0] != plain).float().mean())
    assert differ < 0.02, differ                                                      # a different summation order: rare last-bit flips only
    if S == 1:
        assert torch.equal(outs[0], plain)


def test_splitk_rejects_bad_workspace():
    x = torch.randn(2, 8, 8, 64, device='cuda').to(torch.bfloat16)
    w = torch.randn(128, 1, 1, 64, device='cuda').to(torch.bfloat16)
    with pytest.raises(_capi.DirHipError):
        F.conv2d_nhwc(x, w, splits=2, workspace=torch.zeros(1024, dtype=torch.uint8, device='cuda'))
    with pytest.raises(_capi.DirHipError):
        F.conv2d_nhwc(x, w, splits=2)                                                # K = 64: one slab cannot be shared by two workgroups


def test_splitk_stress_many_launches_two_streams():
    """the partial-tile hand-over rests on drained sc1 stores + a relaxed ticket, not on a formal release / acquire pair.  Stress
    it: 300 launches per stream on two concurrent streams (own workspace each, as the API requires), every result bit-identical to the
    first."""
    torch.manual_seed(7)
    B, H, Cin, Cout, k, S = 64, 8, 512, 512, 3, 4
    x = torch.randn(B, H, H, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device='cuda') * (k * k * Cin) ** -0.5).to(torch.bfloat16)
    d = _capi.ConvDesc(B, H, H, Cin, Cin, 0, Cout, Cout, 0, 0, 0, k, k, 1, 1, _capi.DT_BF16, _capi.DT_BF16, 0)
    nbytes = _capi.lib().dir_conv2d_splitk_workspace_bytes(d, S)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    wss = [torch.zeros(nbytes, dtype=torch.uint8, device='cuda') for _ in streams]
    ref = F.conv2d_nhwc(x, w, pad=1, relu=True, splits=S, workspace=torch.zeros(nbytes, dtype=torch.uint8, device='cuda')).clone()
    torch.cuda.synchronize()
    bad = torch.zeros(2, dtype=torch.int64, device='cuda')
    for it in range(300):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                o = F.conv2d_nhwc(x, w, pad=1, relu=True, splits=S, workspace=wss[i])
                bad[i] += (o != ref).sum()
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0], bad.tolist()
