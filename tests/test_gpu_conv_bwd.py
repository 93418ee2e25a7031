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
