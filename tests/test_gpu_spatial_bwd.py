"""GPU parity of the spatial backward kernels of the image half (dir_amd/train/spatial.py):
  max pool 3x3/2/1, bilinear 2x upsample -- the reference uses stock nn.MaxPool2d / nn.Upsample (models/backbone/resnet.py:247,
      models/dir.py:392,398): compared with torch autograd through the same ATen operators, incl. ties (post-ReLU zeros);
  InitRegressor's attention pooling (models/dir.py:263-270) -- torch autograd through the reference's expressions;
  bone_proj (models/dir.py:146-174) -- G19, torch autograd through the reference's own method, and the float64 oracle.
Tolerance 1e-5 of each gradient's maximum (g uv of bone_proj: 1e-4, the reference's own fp32 noise near a joint)."""
import numpy as np
import pytest
import torch

from dir_amd.train import spatial as TSP
from oracle.golden_inputs import bone_grad_inputs

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_maxpool_backward_matches_autograd_with_ties():
    torch.manual_seed(0)
    for B, H, C in ((2, 128, 64), (3, 17, 8)):
        x = torch.relu(torch.randn(B, C, H, H, device='cuda'))               # ~half zeros: plenty of tied windows
        x.requires_grad_(True)
        y = torch.nn.functional.max_pool2d(x, 3, 2, 1)
        gy = torch.randn_like(y)
        y.backward(gy)
        xn = x.detach().permute(0, 2, 3, 1).contiguous()
        assert torch.equal(TSP.maxpool_fwd(xn), y.detach().permute(0, 2, 3, 1))
        gx = TSP.maxpool_bwd(xn, gy.permute(0, 2, 3, 1).contiguous())
        assert rel(gx, x.grad.permute(0, 2, 3, 1)) < 1e-6


def test_upsample_backward_matches_autograd():
    torch.manual_seed(1)
    for B, H, C in ((2, 8, 2048), (3, 16, 256), (2, 5, 8)):
        x = torch.randn(B, C, H, H, device='cuda', requires_grad=True)
        y = torch.nn.functional.interpolate(x, scale_factor=2, mode='bilinear')
        gy = torch.randn_like(y)
        y.backward(gy)
        # the upsampled map is the first half of a concatenation (models/dir.py:444): its gradient is a channel slice
        buf = torch.randn(B, 2 * H, 2 * H, C + 64, device='cuda')
        buf[..., 32:32 + C] = gy.permute(0, 2, 3, 1)
        gx = TSP.upsample_bwd(buf, C, coff=32)
        assert rel(gx, x.grad.permute(0, 2, 3, 1)) < 1e-5
        assert rel(TSP.upsample_fwd(x.detach().permute(0, 2, 3, 1).contiguous()), y.detach().permute(0, 2, 3, 1)) < 1e-6


def test_attention_pooling_backward_matches_autograd():
    torch.manual_seed(2)
    B, H, C = 5, 8, 2048
    feat = torch.randn(B, C, H, H, device='cuda', requires_grad=True)
    logit = torch.randn(B, 1, H, H, device='cuda', requires_grad=True)
    attn = torch.sigmoid(logit)
    pooled = (feat * attn).sum(-1).sum(-1) / (attn.sum(-1).sum(-1) + 1e-8)    # models/dir.py:264-265
    mean = feat.mean(-1).mean(-1)                                            # models/dir.py:269
    gp, gm = torch.randn_like(pooled), torch.randn_like(mean)
    ((pooled * gp).sum() + (mean * gm).sum()).backward()
    fn = feat.detach().permute(0, 2, 3, 1).contiguous()
    p, a, m = TSP.attn_pool_fwd(fn, logit.detach().reshape(B, H * H).contiguous(), want_mean=True)
    assert rel(p, pooled.detach()) < 1e-5 and rel(m, mean.detach()) < 1e-5
    g_feat, g_logit = TSP.attn_pool_bwd(fn, a, p, gp, gm)
    assert rel(g_feat, feat.grad.permute(0, 2, 3, 1)) < 1e-5
    assert rel(g_logit, logit.grad.reshape(B, H * H)) < 1e-5
    base = torch.randn_like(fn)
    acc, _ = TSP.attn_pool_bwd(fn, a, p, gp, gm, g_feat=base.clone(), need_logit=False)
    assert rel(acc - base, g_feat) < 1e-5


@pytest.mark.parametrize('S,dist', [(16, 1), (32, 2)])
def test_bone_proj_backward_vs_reference_autograd_and_oracle(golden, S, dist):
    from oracle.spatial_grad import bone_proj_backward
    g = golden('g19_bone_grad')
    uv, feat, gi = bone_grad_inputs(S)
    B = uv.shape[0]
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    # the kernel handles both hands: the golden's single hand goes in as left AND (with another cotangent) right
    uv2, feat2, gi2 = bone_grad_inputs(S, B)[0][::-1].copy(), feat[::-1].copy(), gi[::-1].copy()
    emb = dv(np.concatenate([feat, feat2], 1))                                # [B,42,64]
    g_img = np.concatenate([gi, gi2], 1)                                      # [B,2560,S,S]: channel (hand*20 + bone)*64 + c
    g_nhwc = dv(g_img).permute(0, 2, 3, 1).contiguous()
    img = TSP.bone_proj_fwd(dv(uv), dv(uv2), emb, S, dist)
    assert np.abs(img[..., :1280].double().sum((1, 2)).cpu().numpy() - g['S%d.img.sum' % S]).max() < 1e-4 * np.abs(g['S%d.img.sum' % S]).max()
    g_emb, gul, gur = TSP.bone_proj_bwd(dv(uv), dv(uv2), emb, g_nhwc, S, dist)
    rf, ru = g['S%d.g_feat' % S], g['S%d.g_uv' % S]
    e_f = float(np.abs(g_emb[:, :21].cpu().numpy() - rf).max() / np.abs(rf).max())
    e_u = float(np.abs(gul.cpu().numpy() - ru).max() / np.abs(ru).max())
    assert e_f < 1e-5 and e_u < 1e-4, (e_f, e_u)
    ou, of = bone_proj_backward(uv2, feat2, gi2, S, dist)                     # the right hand against the float64 oracle
    assert np.abs(g_emb[:, 21:].cpu().numpy() - of).max() < 1e-5 * np.abs(of).max()
    assert np.abs(gur.cpu().numpy() - ou).max() < 1e-4 * np.abs(ou).max()
    g2 = TSP.bone_proj_bwd(dv(uv), dv(uv2), emb, g_nhwc, S, dist)
    assert torch.equal(g_emb, g2[0]) and torch.equal(gul, g2[1])
    print('bone_proj backward S=%d vs torch autograd through the reference: g feat %.2e, g uv %.2e' % (S, e_f, e_u))


@pytest.mark.parametrize('S,dist,B', [(16, 1, 3), (32, 2, 2), (32, 2, 5)])
def test_factorised_bone_fusion_forward_and_backward_vs_float64_composition(S, dist, B):
    """dir_bone_fusion_prepare / _forward (exact fp32) + dir_bone_fusion_backward -- the training step's bone_proj + fusion.0 -- against the
    composition they replace: the float64 oracle's bone_proj, a float64 3x3 convolution differentiated by torch autograd (CPU) and the float64
    oracle's bone_proj backward (models/dir.py:57,132-174).  Tolerance 1e-5 of each tensor's maximum (g uv 1e-4, as bone_proj's own test)."""
    from oracle.tokens import bone_proj
    from oracle.spatial_grad import bone_proj_backward
    rng = np.random.default_rng(S * 100 + B)
    uv_l = bone_grad_inputs(S, B)[0]
    uv_r = uv_l[::-1].copy()
    feat = rng.standard_normal((B, 42, 64)).astype(np.float32)
    W = (rng.standard_normal((256, 2560, 3, 3)) * 0.02).astype(np.float32)
    bias = rng.standard_normal(256).astype(np.float32)
    gy = rng.standard_normal((B, S, S, 256)).astype(np.float32)
    # float64 reference
    img = np.concatenate([np.asarray(bone_proj(uv, feat[:, 21 * h:21 * h + 21], S, dist), np.float64) for h, uv in enumerate((uv_l, uv_r))], 1)   # [B,2560,S,S]
    ti = torch.from_numpy(img).requires_grad_(True)
    tw = torch.from_numpy(W.astype(np.float64)).requires_grad_(True)
    y_ref = torch.nn.functional.conv2d(ti, tw, torch.from_numpy(bias.astype(np.float64)), padding=1)
    y_ref.backward(torch.from_numpy(gy.astype(np.float64)).permute(0, 3, 1, 2))
    g_img, g_w_ref = ti.grad.numpy(), tw.grad.numpy()
    gu, gf = zip(*[bone_proj_backward(uv, feat[:, 21 * h:21 * h + 21], g_img[:, 1280 * h:1280 * h + 1280], S, dist) for h, uv in enumerate((uv_l, uv_r))])
    # library
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    w_d = dv(W)
    y, ctx = TSP.bone_fusion_fwd(dv(uv_l), dv(uv_r), dv(feat), TSP.fusion_w_g(w_d), dv(bias), S, dist)
    yr = y_ref.detach().permute(0, 2, 3, 1).numpy()
    e_y = float(np.abs(y.cpu().numpy() - yr).max() / np.abs(yr).max())
    g_w_g, g_emb, gul, gur = TSP.bone_fusion_bwd(ctx, dv(gy))
    g_w = TSP.fusion_w_g_grad_to_oihw(g_w_g).cpu().numpy()
    e_w = float(np.abs(g_w - g_w_ref).max() / np.abs(g_w_ref).max())
    gf_ref = np.concatenate(gf, 1)
    e_f = float(np.abs(g_emb.cpu().numpy() - gf_ref).max() / np.abs(gf_ref).max())
    e_u = max(float(np.abs(a.cpu().numpy() - r).max() / np.abs(r).max()) for a, r in ((gul, gu[0]), (gur, gu[1])))
    print('factorised fusion S=%d B=%d vs float64: y %.2e, g weight %.2e, g emb %.2e, g uv %.2e' % (S, B, e_y, e_w, e_f, e_u))
    assert e_y < 1e-5 and e_w < 1e-5 and e_f < 1e-5 and e_u < 1e-4, (e_y, e_w, e_f, e_u)
    again = TSP.bone_fusion_bwd(ctx, dv(gy))
    assert all(torch.equal(a, b) for a, b in zip((g_w_g, g_emb, gul, gur), again))          # deterministic
