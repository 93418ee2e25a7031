"""Thin torch-tensor front ends of the C ABI (plumbing only: pointer/shape marshalling, no arithmetic).
Feature maps are NHWC torch tensors (float32 or bfloat16) on the GPU."""
import torch

from . import _capi
from ._capi import CONV_PRE_RELU, CONV_RELU, DT_BF16, DT_F16, DT_F16X1, DT_F16X1P, DT_F16X3, DT_F16X3P, DT_F32, ConvDesc


def _dt(t):
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:         # f16 STORAGE (include/dir_hip.h: DIR_DT_F16)
        return DT_F16
    raise _capi.DirHipError('unsupported dtype %s' % t.dtype)


def pack_conv_weight(w_oihw, dtype=torch.float32):
    """nn.Conv2d weight [Cout,Cin,kh,kw] -> [Cout,kh,kw,Cin] contiguous in the compute dtype."""
    return w_oihw.detach().permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_f16x3_weights(w_rows, scale=None):
    """The weight operand of DIR_DT_F16X3 (include/dir_hip.h) from fp32 rows [Cout, K] (K % 32 == 0; K = kh*kw*Cin of an OHWI tensor, or
    [kh*kw*Cin | Cin2] of a dual convolution): every row is scaled by a power of two p_n so that max |w_n| p_n lies in [2^12, 2^13)
    (exact), split into hi = f16(w p) and lo = f16(w p - hi) and stored per 32-column slab as [32 hi | 32 lo].
    -> (float16 tensor [Cout, K/32, 2, 32] -- the bytes of the fp32 tensor it replaces --, scale / p as fp32 [Cout])"""
    w = w_rows.detach().float()
    Cout, K = w.shape
    assert K % 32 == 0
    amax = w.abs().amax(1).cpu()                             # exponents on the host: the device's frexp / ldexp are not exact
    _, e = torch.frexp(amax)                                 # amax = m * 2^e, m in [0.5, 1)
    e = torch.where(amax > 0, e, torch.full_like(e, 13)).clamp(-100, 100)
    p = torch.pow(torch.tensor(2.0, dtype=torch.float64), (13 - e).double()).float().to(w.device)     # amax * p in [2^12, 2^13)
    ws = w * p[:, None]                                      # exact: a power of two
    hi = ws.half()
    lo = (ws - hi.float()).half()
    packed = torch.stack([hi.view(Cout, K // 32, 32), lo.view(Cout, K // 32, 32)], 2).contiguous()
    sc = (torch.ones_like(p) if scale is None else scale.detach().float().to(p.device)) / p
    return packed, sc.contiguous()


def pack_f16x3_weights_device(w_rows, scale=None):
    """pack_f16x3_weights without leaving the GPU (dir_pack_f16x3_weights: one launch, no host synchronisation) -- for weights that change
    every optimiser step.  Same result, bit for bit."""
    w = _capi.f32c(w_rows.detach())
    _capi.require_cuda(w, scale)
    N, K = w.shape
    packed = torch.empty(N, K // 32, 2, 32, device=w.device, dtype=torch.float16)
    sc = torch.empty(N, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        _capi.check(_capi.lib().dir_pack_f16x3_weights(_capi.ptr(w), _capi.ptr(packed), _capi.ptr(sc), _capi.ptr(scale), N, K, _capi.stream_ptr()),
                    'dir_pack_f16x3_weights')
    return packed, sc


def split_f16(x, C, in_coff=0, pre_scale=None, pre_shift=None, pre_relu=False, in_scale=1.0, hi_only=False, out=None):
    """dir_split_f16_forward: channels [in_coff, in_coff + C) of the fp32 NHWC tensor x -> the pre-split operand of DIR_DT_F16X3P / F16X1P
    (a float32-TYPED tensor [B,H,W,C] whose bytes are f16 hi | lo slabs)"""
    _capi.require_cuda(x, pre_scale, pre_shift, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and C % 32 == 0
    if out is None:
        out = torch.empty(x.shape[:-1] + (C,), device=x.device, dtype=torch.float32)
    pixels = x.numel() // x.shape[-1]
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dir_split_f16_forward(_capi.ptr(x), _capi.ptr(out), pixels, C, x.shape[-1], in_coff, _capi.ptr(pre_scale), _capi.ptr(pre_shift),
                                                      int(bool(pre_relu)), float(in_scale), int(bool(hi_only)), _capi.stream_ptr()), 'dir_split_f16_forward')
    return out


def pow2_in_scale(x, pre_scale=None, pre_shift=None):
    """the power of two that puts the largest |activation| of x (through an optional pre-activation) at [2^9, 2^10) -- a host synchronisation"""
    import math
    amax = float(x.abs().max())
    if pre_scale is not None:
        amax = amax * float(pre_scale.abs().max()) + float(pre_shift.abs().max())
    return 2.0 ** (10 - math.frexp(amax)[1]) if amax > 0 and math.isfinite(amax) else 1.0


def conv2d_nhwc(x, w_ohwi, stride=1, pad=0, scale=None, shift=None, relu=False, residual=None, pre_scale=None,
                pre_shift=None, pre_relu=False, out=None, out_coff=0, in_coff=0, cin=None, out_dtype=None,
                res_coff=0, splits=1, workspace=None, arith=None, variant=0, presplit=False, in_scale=None, device_pack=False, prepacked=None,
                stats_out=None, mask=None, bn_bwd=None):
    """x [B,H,W,Cbuf] NHWC; reads channels [in_coff, in_coff+cin).  Returns/updates `out` [B,Ho,Wo,Cobuf].
    splits > 1: dir_conv2d_splitk_forward (workspace: uint8 tensor of dir_conv2d_splitk_workspace_bytes, first 16 KiB zero; made here
    if None).  arith='f16x3' (fp32 tensors only): split-precision arithmetic, DIR_DT_F16X3 -- the fp32 weights are packed here."""
    if prepacked is not None:
        # (packed f16x3 weights [Cout][K/32][2][32], their epilogue scale WITH 1 / in_scale folded in, (Cout, kh, kw, Cin)): the training
        # step's WeightPack (dir_amd/train/conv.py) -- nothing is packed or divided here
        assert arith == 'f16x3' and w_ohwi is None and scale is None and in_scale is not None
        w_ohwi, pscale, wshape = prepacked
    else:
        wshape = w_ohwi.shape
    _capi.require_cuda(x, w_ohwi, scale, shift, residual, pre_scale, pre_shift, out)
    assert x.is_contiguous() and w_ohwi.is_contiguous() and x.dim() == 4
    B, H, W, cbuf = x.shape
    Cout, kh, kw, Cin = wshape
    if cin is None:
        cin = Cin
    assert cin == Cin and (prepacked is not None or w_ohwi.dtype == x.dtype)
    assert arith in (None, 'f16x3', 'f16')               # 'f16': the hi parts only (DIR_DT_F16X1), same packing
    if arith is None:
        in_scale = 0.0
    if prepacked is not None:
        assert x.dtype == torch.float32 and splits == 1
        scale = pscale
    elif arith is not None:
        assert x.dtype == torch.float32 and splits == 1
        w_ohwi, scale = (pack_f16x3_weights_device if device_pack else pack_f16x3_weights)(w_ohwi.reshape(Cout, -1), scale)
        if in_scale is None:
            # input scale as DirEngine.calibrate picks it: the largest |activation| the split sees lands in [2^9, 2^10)  (host synchronisation)
            in_scale = pow2_in_scale(x[..., in_coff:in_coff + Cin], pre_scale, pre_shift)
        scale = scale / in_scale
    if arith is not None:
        if presplit:     # activations split ONCE by dir_split_f16_forward (pre-activation and in_scale applied there), both operands by DMA
            xs = split_f16(x, Cin, in_coff, pre_scale, pre_shift, pre_relu, in_scale, hi_only=(arith == 'f16'))
            x, cbuf, in_coff, pre_scale, pre_shift, pre_relu = xs, Cin, 0, None, None, False
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    if out is None:
        odt = out_dtype or x.dtype
        out = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=odt)
    assert out.is_contiguous() and out.shape[:3] == (B, Ho, Wo)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == out.dtype and residual.shape[:3] == (B, Ho, Wo)
    for v in (scale, shift):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cout and v.is_contiguous())
    for v in (pre_scale, pre_shift):
        assert v is None or (v.dtype == torch.float32 and v.numel() == Cin and v.is_contiguous())
    d = ConvDesc(B, H, W, Cin, cbuf, in_coff, Cout, out.shape[3], out_coff,
                 residual.shape[3] if residual is not None else 0, res_coff, kh, kw, stride, pad, (DT_F16X3P if presplit else DT_F16X3) if arith == 'f16x3' else (DT_F16X1P if presplit else DT_F16X1) if arith == 'f16' else _dt(x), _dt(out),
                 (CONV_RELU if relu else 0) | (CONV_PRE_RELU if pre_relu else 0) | ((variant & 0xff) << 8), 0, 0, in_scale)      # variant: DIR_CONV_VARIANT code
    with torch.cuda.device(x.device):
        if splits > 1:
            if workspace is None:
                workspace = torch.zeros(_capi.lib().dir_conv2d_splitk_workspace_bytes(d, splits), dtype=torch.uint8, device=x.device)
            _capi.check(_capi.lib().dir_conv2d_splitk_forward(d, _capi.ptr(x), _capi.ptr(w_ohwi), _capi.ptr(scale), _capi.ptr(shift),
                                                              _capi.ptr(pre_scale), _capi.ptr(pre_shift), _capi.ptr(residual), _capi.ptr(out),
                                                              splits, _capi.ptr(workspace), workspace.numel(), _capi.stream_ptr()),
                        'dir_conv2d_splitk_forward')
            return out
        if _capi.PROFILE is not None:
            _capi.annotate(family='conv', flops=2.0 * B * Ho * Wo * Cout * kh * kw * Cin, bytes=float(x.numel() * x.element_size() + out.numel() * out.element_size()),
                           shape='conv M=%d N=%d K=%d k%d s%d %s' % (B * Ho * Wo, Cout, kh * kw * Cin, kh, stride, arith or str(x.dtype)[6:]))
        if stats_out is not None and residual is None and not relu and out.dtype == torch.float32 and out.shape[3] == Cout and out_coff == 0:
            # round 5: the chunk partials of the BatchNorm (training mode) that follows, formed in the epilogue (dir_conv2d_forward_stats);
            # stats_out (a list) receives (p1, p2, rows per chunk) -- rows 0: the kernel that ran does not form them
            import ctypes as C
            nch = (B * Ho * Wo + 63) // 64
            nch += (nch + 31) // 32            # + room for the two-level pooling of dir_bn_train_stats_from_partials
            part = torch.empty(2, nch, Cout, device=x.device, dtype=torch.float32)
            rows = C.c_int(0)
            rc = _capi.lib().dir_conv2d_forward_stats(d, _capi.ptr(x), _capi.ptr(w_ohwi), _capi.ptr(scale), _capi.ptr(shift), _capi.ptr(pre_scale),
                                                      _capi.ptr(pre_shift), _capi.ptr(out), _capi.ptr(part[0]), _capi.ptr(part[1]), C.byref(rows),
                                                      _capi.stream_ptr())
            _capi.check(rc, 'dir_conv2d_forward_stats')
            stats_out.append((part[0], part[1], rows.value))
            return out
        if bn_bwd is not None and out.dtype == torch.float32 and out.shape[3] == Cout and out_coff == 0 and Cout % 4 == 0 and not relu:
            # round 5 (dir_conv2d_forward_ex): this is a data-gradient convolution whose output is the gradient of a training-mode BatchNorm's (+ ReLU's)
            # output; bn_bwd = {'z', 'mean', 'rstd', 'w', 'b', 'relu', 'out': []}: the epilogue also forms that BatchNorm backward's chunk partials,
            # bn_bwd['out'] receives (p1, p2, chunks) -- nothing when the kernel that ran does not form them
            import ctypes as C
            from ._capi import ConvBnBwd
            z = bn_bwd['z']
            assert z.dtype == torch.float32 and z.is_contiguous() and z.numel() == out.numel()
            M_ = B * Ho * Wo
            part = torch.empty(2, (M_ + 63) // 64, Cout, device=x.device, dtype=torch.float32)
            bb = ConvBnBwd(z.data_ptr(), bn_bwd['mean'].data_ptr(), bn_bwd['rstd'].data_ptr(), bn_bwd['w'].data_ptr() if bn_bwd.get('w') is not None else None,
                           bn_bwd['b'].data_ptr() if bn_bwd.get('b') is not None else None, int(bool(bn_bwd.get('relu'))), part[0].data_ptr(), part[1].data_ptr())
            rows = C.c_int(0)
            if mask is not None:
                assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.shape == out.shape
            rc = _capi.lib().dir_conv2d_forward_ex(d, _capi.ptr(x), _capi.ptr(w_ohwi), _capi.ptr(scale), _capi.ptr(shift), _capi.ptr(pre_scale), _capi.ptr(pre_shift),
                                                   _capi.ptr(residual), _capi.ptr(mask), _capi.ptr(out), C.byref(bb), C.byref(rows), _capi.stream_ptr())
            _capi.check(rc, 'dir_conv2d_forward_ex')
            if rows.value > 0:
                bn_bwd['out'].append((part[0], part[1], (M_ + rows.value - 1) // rows.value))
            return out
        if mask is not None:               # y = mask > 0 ? conv (+ residual) : 0 (dir_conv2d_forward_masked: a ReLU backward applied where the gradient is written)
            assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.shape == out.shape and out_coff == 0
            rc = _capi.lib().dir_conv2d_forward_masked(d, _capi.ptr(x), _capi.ptr(w_ohwi), _capi.ptr(scale), _capi.ptr(shift), _capi.ptr(pre_scale),
                                                       _capi.ptr(pre_shift), _capi.ptr(residual), _capi.ptr(mask), _capi.ptr(out), _capi.stream_ptr())
            _capi.check(rc, 'dir_conv2d_forward_masked')
            return out
        rc = _capi.lib().dir_conv2d_forward(d, _capi.ptr(x), _capi.ptr(w_ohwi), _capi.ptr(scale), _capi.ptr(shift),
                                            _capi.ptr(pre_scale), _capi.ptr(pre_shift), _capi.ptr(residual),
                                            _capi.ptr(out), _capi.stream_ptr())
    _capi.check(rc, 'dir_conv2d_forward')
    return out


def pack_stem_weight(w_oihw, dtype=torch.bfloat16):
    """conv1.weight [64,3,7,7] -> bf16 [64][ky 7][kx 8][c 4] for dir_stem_pool_forward (zero for kx = 7, c = 3)"""
    wk = torch.zeros(64, 7, 8, 4, device=w_oihw.device, dtype=torch.float32)
    wk[:, :, :7, :3] = w_oihw.float().permute(0, 2, 3, 1)
    return wk.to(dtype).contiguous()


def stem_pool(img, w_packed, scale, shift, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """models/backbone/resnet.py:244-247 in one launch (bf16 operands): conv1 7x7/s2/p3 + folded bn1 + ReLU + MaxPool2d(3,2,1).
    img: float32 NCHW [B,3,H,W] (normalised) or uint8 BGR HWC [B,H,W,3] (apps/eval.py:59-61 fused) -> bf16 NHWC [B,H/4,W/4,64]"""
    import ctypes as C
    _capi.require_cuda(img)
    u8 = img.dtype == torch.uint8
    if u8:
        B, H, W = img.shape[0], img.shape[1], img.shape[2]
    else:
        img = _capi.f32c(img)
        B, H, W = img.shape[0], img.shape[2], img.shape[3]
    img = img.contiguous()
    y = torch.empty(B, H // 4, W // 4, 64, device=img.device, dtype=w_packed.dtype)          # bf16 | f16 storage: the packed weights' kind
    _capi.check(_capi.lib().dir_stem_pool_forward_dt(_capi.ptr(img), 2 if u8 else 0, _dt(w_packed), (C.c_float * 3)(*mean), (C.c_float * 3)(*std),
                                                  _capi.ptr(w_packed), _capi.ptr(_capi.f32c(scale)), _capi.ptr(_capi.f32c(shift)),
                                                  _capi.ptr(y), B, H, W, _capi.stream_ptr()), 'dir_stem_pool_forward')
    return y


def bottleneck_chain(y1, w2, s2, h2, w3, s3, h3, residual=None, nxt=None, dual=None, decimate=False):
    """models/backbone/resnet.py:126-140 (+ :122-124 of the next block) in one launch, layer1 geometry, bf16 NHWC.
    y1 [B,H,W,64]; w2 packed [64,3,3,64] bf16; w3 [256,64] bf16; nxt = (w1n [64,256] bf16, scale, shift) or None;
    dual = (x2 [B,H,W,64], wd [256,64] bf16) = projection shortcut as 64 more K of conv3 (BN scales folded into w3 / wd, s3 = 1).
    decimate: only the even (y, x) pixels of the block output are written, as out [B,H/2,W/2,256] (dir_bneck_chain_params.out_decimate).
    returns (out [B,H,W,256], y1_next [B,H,W,64] or None)"""
    _capi.require_cuda(y1)
    B, H, W, _ = y1.shape
    out = torch.empty((B, H // 2, W // 2, 256) if decimate else (B, H, W, 256), device=y1.device, dtype=y1.dtype)
    y1n = torch.empty(B, H, W, nxt[0].shape[0], device=y1.device, dtype=y1.dtype) if nxt is not None else None
    keep = [_capi.f32c(t) for t in (s2, h2, s3, h3)] + ([_capi.f32c(nxt[1]), _capi.f32c(nxt[2])] if nxt is not None else [])
    p = _capi.BneckChainParams(_capi.ptr(w2), _capi.ptr(keep[0]), _capi.ptr(keep[1]), _capi.ptr(w3), _capi.ptr(keep[2]), _capi.ptr(keep[3]),
                               _capi.ptr(nxt[0]) if nxt is not None else None, _capi.ptr(keep[4]) if nxt is not None else None,
                               _capi.ptr(keep[5]) if nxt is not None else None, _capi.ptr(dual[1]) if dual is not None else None,
                               nxt[0].shape[0] if nxt is not None else 0, 1 if decimate else 0, _dt(y1))
    import ctypes as C
    _capi.check(_capi.lib().dir_bottleneck_chain_forward(C.byref(p), _capi.ptr(y1), _capi.ptr(residual) if residual is not None else None,
                                                         _capi.ptr(dual[0]) if dual is not None else None, _capi.ptr(out), _capi.ptr(y1n) if y1n is not None else None, B, H, W,
                                                         _capi.stream_ptr()), 'dir_bottleneck_chain_forward')
    return out, y1n


def bottleneck_tail(y2, w3, s3, h3, residual, w1n, s1n, h1n, waves=8):
    """models/backbone/resnet.py:132-140 (+ :122-124 of the next block) in one launch, layer2 / layer3 geometry, bf16 NHWC.
    y2 [B,H,W,P]; w3 [4P,P]; residual [B,H,W,4P] (the block input); w1n [N2,4P]  ->  (out [B,H,W,4P], y1_next [B,H,W,N2])"""
    import ctypes as C
    from .engine import pack_tail_stream
    _capi.require_cuda(y2, residual)
    B, H, W, P = y2.shape
    C4, N2 = w3.shape[0], w1n.shape[0]
    stream = pack_tail_stream(w3, w1n, waves, dtype=y2.dtype)
    keep = [_capi.f32c(t) for t in (s3, h3, s1n, h1n)]
    out = torch.empty(B, H, W, C4, device=y2.device, dtype=y2.dtype)
    y1n = torch.empty(B, H, W, N2, device=y2.device, dtype=y2.dtype)
    p = _capi.BneckTailParams(_capi.ptr(stream), _capi.ptr(keep[0]), _capi.ptr(keep[1]), _capi.ptr(keep[2]), _capi.ptr(keep[3]), P, N2, waves, _dt(y2))
    _capi.check(_capi.lib().dir_bottleneck_tail_forward(C.byref(p), _capi.ptr(y2.contiguous()), _capi.ptr(residual.contiguous()), _capi.ptr(out),
                                                        _capi.ptr(y1n), B * H * W, _capi.stream_ptr()), 'dir_bottleneck_tail_forward')
    return out, y1n


def conv1x1_stream(x, w_nk, scale=None, shift=None, relu=False, pre_scale=None, pre_shift=None, pre_relu=False, x2=None, stride2=1,
                   out=None, out_coff=0, in_coff=0, cin=None, variant=0):
    """dir_conv1x1_stream_forward: y = act(scale * (x[..., in_coff:in_coff+cin] . W1^T + x2[::stride2, ::stride2] . W2^T) + shift), bf16 NHWC.
    w_nk: fp32/bf16 [Cout, cin (+ Cin2)] (second source's columns appended); variant 22: 64-pixel workgroups (same bits)"""
    from .engine import pack_stream_weights
    _capi.require_cuda(x, x2, out)
    B, H, W, cbuf = x.shape
    Cout, K = w_nk.shape
    cin2 = x2.shape[3] if x2 is not None else 0
    cin = cin if cin is not None else K - cin2
    assert cin + cin2 == K
    ws = pack_stream_weights(w_nk, dtype=x.dtype)
    if out is None:
        out = torch.empty(B, H, W, Cout, device=x.device, dtype=x.dtype)
    d = ConvDesc(B, H, W, cin, cbuf, in_coff, Cout, out.shape[3], out_coff, 0, 0, 1, 1, 1, 0, _dt(x), _dt(out),
                 (CONV_RELU if relu else 0) | (CONV_PRE_RELU if pre_relu else 0) | ((variant & 0xff) << 8))
    d2 = _capi.ConvSrc2(x2.shape[1], x2.shape[2], cin2, x2.shape[3], 0, stride2) if x2 is not None else None
    keep = [None if t is None else _capi.f32c(t) for t in (scale, shift, pre_scale, pre_shift)]
    _capi.check(_capi.lib().dir_conv1x1_stream_forward(d, _capi.ptr(x), d2, _capi.ptr(x2), _capi.ptr(ws), _capi.ptr(keep[0]), _capi.ptr(keep[1]),
                                                       _capi.ptr(keep[2]), _capi.ptr(keep[3]), _capi.ptr(out), _capi.stream_ptr()),
                'dir_conv1x1_stream_forward')
    return out


def mano_backward(tables_lr, para_lr, g_verts=None, g_joints=None, g_joint_uv=None, g_mesh_uv=None):
    """dir_mano_backward_pair on 64-vectors (pose 51 | betas 10 | cam 3, models/dir.py:352-363).  tables_lr: list of _capi.ManoTables
    (1 or 2 hands); para_lr: list of fp32 [B,64] tensors; g_*: lists of cotangent tensors (or None).  Returns the list of g_para [B,64]."""
    import ctypes as C
    hands = len(para_lr)
    B = para_lr[0].shape[0]
    para = [_capi.f32c(p) for p in para_lr]
    _capi.require_cuda(*para)
    P = C.c_void_p * hands

    def arr(ts, off=0):
        if ts is None:
            return None
        keep.extend(t for t in ts if t is not None)
        return P(*[(None if t is None else t.data_ptr() + off) for t in ts])
    keep = []
    gs = [[None if t is None else _capi.f32c(t) for t in (g if g is not None else [None] * hands)] for g in (g_verts, g_joints, g_joint_uv, g_mesh_uv)]   # a hand without a cotangent: NULL pointer
    out = [torch.zeros(B, 64, device=para[0].device) for _ in range(hands)]
    tabs = (_capi.ManoTables * hands)(*tables_lr)
    _capi.check(_capi.lib().dir_mano_backward_pair(
        tabs, arr(para), 64, arr(para, 51 * 4), 64, arr(para, 61 * 4), 64,
        arr(gs[0]) if g_verts is not None else None, arr(gs[1]) if g_joints is not None else None,
        arr(gs[2]) if g_joint_uv is not None else None, arr(gs[3]) if g_mesh_uv is not None else None,
        arr(out), 64, arr(out, 51 * 4), 64, arr(out, 61 * 4), 64, hands, B, _capi.stream_ptr()), 'dir_mano_backward_pair')
    return out


def regress_backward(w_left, w_right, w_offset, tok, prev_para_left, prev_para_right, prev_offset, g_para_left, g_para_right, g_offset):
    """dir_regress_backward -> dict of the six parameter gradients (nn.Linear layout) and g_tok [B,42,64]"""
    ts = [_capi.f32c(t) for t in (w_left, w_right, w_offset, tok, prev_para_left, prev_para_right, prev_offset.reshape(-1, 3), g_para_left, g_para_right, g_offset)]
    _capi.require_cuda(*ts)
    B, dev = ts[3].shape[0], ts[3].device
    out = {'mano_left.weight': torch.empty(64, 1408, device=dev), 'mano_left.bias': torch.empty(64, device=dev),
           'mano_right.weight': torch.empty(64, 1408, device=dev), 'mano_right.bias': torch.empty(64, device=dev),
           'offset.weight': torch.empty(3, 2691, device=dev), 'offset.bias': torch.empty(3, device=dev), 'tok': torch.empty(B, 42, 64, device=dev)}
    _capi.check(_capi.lib().dir_regress_backward(*[_capi.ptr(t) for t in ts], _capi.ptr(out['mano_left.weight']), _capi.ptr(out['mano_left.bias']),
                                                 _capi.ptr(out['mano_right.weight']), _capi.ptr(out['mano_right.bias']), _capi.ptr(out['offset.weight']),
                                                 _capi.ptr(out['offset.bias']), _capi.ptr(out['tok']), B, _capi.stream_ptr()), 'dir_regress_backward')
    return out
