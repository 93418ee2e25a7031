"""Tensor front ends of the spatial backward kernels (include/dir_hip.h: dir_maxpool3x3s2_backward, dir_upsample2x_bilinear_backward,
dir_attn_pool_*, dir_bone_proj_backward) and fp32 forwards of the same operators through the inference entry points."""
import ctypes as C

import torch

from . import ops as O
from .. import _capi


def maxpool_fwd(x):
    B, H, W, Cc = x.shape
    y = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, device=x.device)
    _capi.check(_capi.lib().dir_maxpool3x3s2(_capi.ptr(x), _capi.ptr(y), B, H, W, Cc, _capi.DT_F32, _capi.stream_ptr()), 'dir_maxpool3x3s2')
    return y


def maxpool_bwd(x, gy):
    O._chk(x, gy)
    B, H, W, Cc = x.shape
    gx = torch.empty_like(x)
    _capi.check(_capi.lib().dir_maxpool3x3s2_backward(_capi.ptr(x), _capi.ptr(gy), _capi.ptr(gx), B, H, W, Cc, _capi.stream_ptr()), 'dir_maxpool3x3s2_backward')
    return gx


def upsample_fwd(x, out=None, out_coff=0):
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty(B, 2 * H, 2 * W, Cc, device=x.device)
    _capi.check(_capi.lib().dir_upsample2x_bilinear(_capi.ptr(x), _capi.ptr(out), B, H, W, Cc, out.shape[3], out_coff, _capi.DT_F32, _capi.stream_ptr()),
                'dir_upsample2x_bilinear')
    return out


def upsample_bwd(gy, C_, coff=0):
    """gy [B,2H,2W,Cbuf] (the gradient of a concat buffer; channels coff .. coff + C_ belong to the upsampled half) -> g x [B,H,W,C_]"""
    O._chk(gy)
    B, H2, W2, cs = gy.shape
    gx = torch.empty(B, H2 // 2, W2 // 2, C_, device=gy.device)
    _capi.check(_capi.lib().dir_upsample2x_bilinear_backward(_capi.ptr(gy), _capi.ptr(gx), B, H2 // 2, W2 // 2, C_, cs, coff, _capi.stream_ptr()),
                'dir_upsample2x_bilinear_backward')
    return gx


def attn_pool_fwd(feat, logit, want_mean=False):
    """feat [B,H,W,C], logit [B,H,W(,1)] -> (pooled [B,C], attn [B,HW], mean [B,C] or None)"""
    O._chk(feat, logit)
    B, H, W, Cc = feat.shape
    attn, pooled = torch.empty(B, H * W, device=feat.device), torch.empty(B, Cc, device=feat.device)
    mean = torch.empty(B, Cc, device=feat.device) if want_mean else None
    _capi.check(_capi.lib().dir_attn_pool_forward(_capi.ptr(feat), _capi.ptr(logit), _capi.ptr(attn), _capi.ptr(pooled), _capi.ptr(mean), B, H * W, Cc,
                                                  _capi.stream_ptr()), 'dir_attn_pool_forward')
    return pooled, attn, mean


def attn_pool_bwd(feat, attn, pooled, g_pooled, g_mean=None, g_feat=None, need_logit=True):
    """-> (g feat [B,H,W,C] (added onto g_feat if given), g logit [B,HW] or None)"""
    O._chk(feat, attn, pooled, g_pooled, g_mean, g_feat)
    B, H, W, Cc = feat.shape
    acc = g_feat is not None
    if g_feat is None:
        g_feat = torch.empty_like(feat)
    g_logit = torch.empty(B, H * W, device=feat.device) if need_logit else None
    _capi.check(_capi.lib().dir_attn_pool_backward(_capi.ptr(feat), _capi.ptr(attn), _capi.ptr(pooled), _capi.ptr(g_pooled), _capi.ptr(g_mean), _capi.ptr(g_feat),
                                                   _capi.ptr(g_logit), B, H * W, Cc, int(acc), _capi.stream_ptr()), 'dir_attn_pool_backward')
    return g_feat, g_logit


def bone_proj_fwd(uv_l, uv_r, emb, S, distance, want_vis=False):
    """-> NHWC fp32 [B,S,S,2560] (dir_bone_proj_forward); want_vis: also `vis_img_feat` = left + right maps, NCHW fp32 [B,1280,S,S]
    (models/dir.py:128, returned as outs_list[3]['proj_feat'], models/dir.py:481)"""
    B = emb.shape[0]
    out = torch.empty(B, S, S, 2560, device=emb.device)
    vis = torch.empty(B, 1280, S, S, device=emb.device) if want_vis else None
    _capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(uv_l), _capi.ptr(uv_r), _capi.ptr(emb), _capi.ptr(out), _capi.ptr(vis), None, B, S, float(distance),
                                                  _capi.DT_F32, _capi.stream_ptr()), 'dir_bone_proj_forward')
    return (out, vis) if want_vis else out


def bone_proj_vis(uv_l, uv_r, emb, S, distance):
    """only `vis_img_feat` (proj_feat, models/dir.py:128,481), NCHW fp32 [B,1280,S,S]: the factorised fusion needs no bone map"""
    B = emb.shape[0]
    vis = torch.empty(B, 1280, S, S, device=emb.device)
    _capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(uv_l), _capi.ptr(uv_r), _capi.ptr(emb), None, _capi.ptr(vis), None, B, S, float(distance),
                                                  _capi.DT_F32, _capi.stream_ptr()), 'dir_bone_proj_forward')
    return vis


def bone_proj_bwd(uv_l, uv_r, emb, g_img, S, distance, coff=0):
    """g_img [B,S,S,Cbuf] -> (g emb [B,42,64], g uv_left [B,21,2], g uv_right [B,21,2])"""
    O._chk(uv_l, uv_r, emb, g_img)
    B = emb.shape[0]
    g_emb = torch.empty(B, 42, 64, device=emb.device)
    g_uv = [torch.empty(B, 21, 2, device=emb.device) for _ in range(2)]
    scratch = torch.empty(_capi.lib().dir_bone_proj_backward_scratch_bytes(B, 2) // 4, device=emb.device)
    P2 = C.c_void_p * 2
    _capi.check(_capi.lib().dir_bone_proj_backward(P2(uv_l.data_ptr(), uv_r.data_ptr()), _capi.ptr(emb), _capi.ptr(g_img), g_img.shape[3], coff, float(distance),
                                                   _capi.ptr(g_emb), P2(g_uv[0].data_ptr(), g_uv[1].data_ptr()), _capi.ptr(scratch), B, S, 2, _capi.stream_ptr()),
                'dir_bone_proj_backward')
    return g_emb, g_uv[0], g_uv[1]


# ---------------------------------------------------------------------------------------------- factorised bone fusion, training form
def fusion_w_g(w_oihw):
    """fusion.0.weight [256, 2560, 3, 3] -> w_g [9][40][64][256] (include/dir_hip.h: dir_bone_fusion_params.w_g), unrounded"""
    return w_oihw.reshape(256, 40, 64, 9).permute(3, 1, 2, 0).contiguous()


def fusion_w_g_grad_to_oihw(g_w_g):
    return g_w_g.permute(3, 1, 2, 0).reshape(256, 2560, 3, 3).contiguous()


def bone_fusion_fwd(uv_l, uv_r, emb, w_g, bias, S, distance):
    """bone_proj + fusion.0 (models/dir.py:57,132-174) without the [B,S,S,2560] map, exact fp32: -> (y [B,S,S,256] = the raw convolution + bias,
    ctx for bone_fusion_bwd).  dir_bone_fusion_prepare + dir_bone_fusion_forward (exact_f32)."""
    O._chk(uv_l, uv_r, emb, w_g, bias)
    B, L = emb.shape[0], _capi.lib()
    par = _capi.BoneFusionParams(w_g.data_ptr(), None, None if bias is None else bias.data_ptr(), 1, 0.0)
    g = torch.empty(L.dir_bone_fusion_scratch_bytes(B) // 4, device=emb.device)
    y = torch.empty(B, S, S, 256, device=emb.device)
    if _capi.PROFILE is not None:
        _capi.annotate(family='bone_fusion', flops=2.0 * B * 9 * 80 * 64 * 256, bytes=4.0 * (9 * 40 * 64 * 256 + B * 9 * 80 * 256), shape='G B=%d' % B)
    _capi.check(L.dir_bone_fusion_prepare(par, _capi.ptr(emb), _capi.ptr(g), B, _capi.stream_ptr()), 'dir_bone_fusion_prepare')
    if _capi.PROFILE is not None:
        _capi.annotate(family='bone_fusion', flops=2.0 * B * S * S * 256 * 720, bytes=4.0 * (B * 9 * 80 * 256 + B * S * S * 256), shape='fuse B=%d S=%d f32' % (B, S))
    _capi.check(L.dir_bone_fusion_forward(par, _capi.ptr(uv_l), _capi.ptr(uv_r), _capi.ptr(g), _capi.ptr(y), B, S, float(distance), 256, 0, 0, _capi.stream_ptr()),
                'dir_bone_fusion_forward')
    return y, dict(uv=(uv_l, uv_r), emb=emb, w_g=w_g, g=g, S=S, distance=distance)


def bone_fusion_bwd(ctx, gy, need_uv=True):
    """gy [B,S,S,256] -> (g w_g [9,40,64,256], g emb [B,42,64], g uv_left, g uv_right [B,21,2])   (dir_bone_fusion_backward)"""
    O._chk(gy)
    emb, S, L = ctx['emb'], ctx['S'], _capi.lib()
    B = emb.shape[0]
    g_w_g = torch.empty_like(ctx['w_g'])
    g_emb = torch.empty(B, 42, 64, device=emb.device)
    g_uv = [torch.empty(B, 21, 2, device=emb.device) if need_uv else None for _ in range(2)]
    n = L.dir_bone_fusion_backward_workspace_bytes(B, S)
    ws = torch.empty(n // 4, device=emb.device)
    if _capi.PROFILE is not None:
        _capi.annotate(family='bone_fusion', flops=2.0 * B * (2 * (S + 2) ** 2 * 720 * 256 + 2 * 720 * 64 * 256), bytes=4.0 * B * (2 * 720 * 256 + (S + 2) ** 2 * 336),
                       shape='bwd B=%d S=%d' % (B, S))
    _capi.check(L.dir_bone_fusion_backward(_capi.ptr(ctx['w_g']), _capi.ptr(emb), _capi.ptr(ctx['uv'][0]), _capi.ptr(ctx['uv'][1]), _capi.ptr(ctx['g']), _capi.ptr(gy),
                                           float(ctx['distance']), _capi.ptr(g_w_g), _capi.ptr(g_emb), _capi.ptr(g_uv[0]), _capi.ptr(g_uv[1]), _capi.ptr(ws), n, B, S,
                                           _capi.stream_ptr()), 'dir_bone_fusion_backward')
    return g_w_g, g_emb, g_uv[0], g_uv[1]
