"""nn.Conv2d in training form on libdir_hip.so: fp32 NHWC forward (dir_conv2d_forward; split precision f16x3 by default, exact fp32 with
DIR_TRAIN_ARITH=f32) and its two gradients.

    y = conv_fwd(x, w, bias, stride, pad)                      x [B,H,W,Cin], w [Cout,kh,kw,Cin] (pack_conv_weight layout), y [B,Ho,Wo,Cout]
    gx, gw, gb = conv_bwd(x, w, gy, stride, pad, need_gx)      what autograd returns for nn.Conv2d (models/backbone/resnet.py:23-40,
                                                               models/backbone/hourglass.py:14, models/dir.py:58-61,229-232,404-419)
* weight gradient: dir_conv2d_wgrad_f16x3 (wgrad_x3.hip; layers with >= 32 channels on both sides) or dir_conv2d_wgrad_f32 (train_ops.hip: the
  3-channel stem, the 1- / 3- / 6-channel heads, DIR_TRAIN_WGRAD_ARITH=f32), bias gradient: dir_colsum_f32;
* data gradient: the transposed convolution is the SAME forward kernel run on gy with the flipped, (Cin <-> Cout)-transposed weights
  and padding k - 1 - pad; for stride 2, gy is first spread onto the even positions of a zero [2 Ho, 2 Wo] map.  The flips, the zero
  insertion and the channel padding to the kernel's 32-channel granularity are copies -- no arithmetic happens outside the library.
"""
import os
import warnings

import torch

from . import ops as O
from .. import _capi
from .. import functional as F

# Arithmetic of the forward and data-gradient convolutions: 'f16x3' = the split-precision f16 matrix-core path (DIR_DT_F16X3: ~2^-22 per
# product, below fp32 accumulation noise; 2.5x the exact kernel's speed), 'f32' = exact fp32 MFMA (rounds 1-2).  DIR_TRAIN_ARITH overrides.
ARITH = os.environ.get('DIR_TRAIN_ARITH', 'f16x3')
WGRAD_ARITH = os.environ.get('DIR_TRAIN_WGRAD_ARITH', ARITH)          # the weight gradient's arithmetic (dir_conv2d_wgrad_f16x3 | _f32)
# The split needs one power-of-two input scale per call site (include/dir_hip.h: in_scale).  Measuring it costs a host synchronisation, so it
# is measured on the FIRST step (and again every RECALIBRATE steps: gradient magnitudes drift as the loss falls) and re-used in between: the
# convolution calls of a training step happen in a fixed order, so the call counter identifies the site.  64x headroom + saturation at the
# f16 maximum make a stale scale a (bounded) precision loss, never an inf / nan.
RECALIBRATE = int(os.environ.get('DIR_TRAIN_RECALIBRATE', '50'))
HEADROOM = 64.0             # pow2_in_scale puts the largest |operand| in [2^9, 2^10): 64x below the f16 maximum
# (A step's backward must follow its own forward before another model's forward starts: the cache bound by begin_step stays active until
# the next begin_step.)
# One cache per trained model, held BY THE CALLER'S OWNER OBJECT (the optimizer in dir_amd.train.step.train_step, the nn.Module in
# dir_amd.models.dir.DIR): it lives exactly as long as that object.  Rounds 2-3 keyed a module-level table by the address of the model's first
# parameter -- an address the caching allocator hands to the NEXT model once the first one is freed, which then inherited scales measured on
# other weights (found by the round-4 frozen-BatchNorm gradient gate: 1.7e-2 instead of 6e-4 when other tests had run before it).  Without
# an owner (owner=None: direct calls of dir_amd.train.net.forward) nothing is cached: every convolution measures its scale on what it is given.
_scales, _state = [], {'call': 0, 'step': 0, 'cached': False}


def begin_step(owner=None):
    """called by dir_amd.train.net.forward at the start of every training step; `owner`: the object that owns this model's operand-scale
    cache (any object that accepts attributes), or None = no cache"""
    global _scales, _state
    if owner is None:
        _scales, _state = [], {'call': 0, 'step': 1, 'cached': False}
        return
    cache = getattr(owner, '_dir_conv_scale_cache', None)
    if cache is None:
        cache = ([], {'call': 0, 'step': 0, 'cached': True})
        setattr(owner, '_dir_conv_scale_cache', cache)
    _scales, _state = cache
    _state['call'] = 0
    _state['step'] += 1
    if RECALIBRATE > 0 and _state['step'] % RECALIBRATE == 1 and _state['step'] > 1:
        _state['previous'] = list(_scales)       # kept for this one step: _site_scale compares what it measures now with what was in use
        del _scales[:]
    else:
        _state.pop('previous', None)


def end_step():
    """called by dir_amd.train.net.backward when a step's gradients are complete: convolution calls outside a step (block-level callers, tests)
    measure their scales again instead of walking on in the finished step's cache"""
    global _scales, _state
    _scales, _state = [], {'call': 0, 'step': 1, 'cached': False}


def reset_scales(owner=None):
    """forget the cached operand scales (of `owner`, and the ones bound right now): the next step calibrates again on the batch it sees"""
    if owner is not None and getattr(owner, '_dir_conv_scale_cache', None) is not None:
        del owner._dir_conv_scale_cache[0][:]
        owner._dir_conv_scale_cache[1].update(call=0, step=0)
    end_step()


def _site_scale(x):
    if not _state.get('cached', False):          # no owner: measure (one host synchronisation per call site)
        return F.pow2_in_scale(x)
    i = _state['call']
    _state['call'] += 1
    if i < len(_scales) and _scales[i][0] == tuple(x.shape):
        return _scales[i][1]
    s = F.pow2_in_scale(x)
    prev = _state.get('previous')
    if prev is not None and i < len(prev) and prev[i][0] == tuple(x.shape) and prev[i][1] >= HEADROOM * s:
        # the operand grew past the headroom the stale scale left: values were clamped at the f16 maximum somewhere in the last RECALIBRATE steps
        warnings.warn('dir_amd.train.conv: call site %d (%s) outgrew its cached f16x3 operand scale (%.3g in use, %.3g needed now): operands '
                      'saturated during the last %d steps; lower DIR_TRAIN_RECALIBRATE or call reset_scales() after a learning-rate change'
                      % (i, 'x'.join(map(str, x.shape)), prev[i][1], s, RECALIBRATE), RuntimeWarning, stacklevel=3)
    del _scales[i:]
    _scales.append((tuple(x.shape), s))
    return s


def _conv(x, w, stride, pad, shift=None):
    if ARITH != 'f16x3':
        return F.conv2d_nhwc(x, w, stride=stride, pad=pad, shift=shift)
    kh = w.shape[1]
    return F.conv2d_nhwc(x, w, stride=stride, pad=pad, shift=shift, arith='f16x3', in_scale=_site_scale(x), device_pack=True,
                         presplit=(kh >= 3 or (w.shape[0] >= 512 and w.shape[3] >= 128)))


def conv_fwd(x, w, bias=None, stride=1, pad=0):
    cin = w.shape[3]
    if cin % 32:                                           # the 3-channel image: channels padded to the kernel's K granularity
        x, w = _pad_last(x, 32), _pad_last(w, 32)
    return _conv(x, w, stride, pad, bias)


def _pad_last(t, mult):
    c = t.shape[-1]
    if c % mult == 0:
        return t
    out = torch.zeros(*t.shape[:-1], (c + mult - 1) // mult * mult, device=t.device, dtype=t.dtype)
    out[..., :c] = t
    return out


def conv_dgrad(w, gy, stride, pad, H, W):
    """d loss / d x [B,H,W,Cin] of y = conv(x, w, stride, pad) from gy [B,Ho,Wo,Cout]"""
    Cout, kh, kw, Cin = w.shape
    assert kh == kw and stride in (1, 2)
    wt = _pad_last(w.flip(1, 2).permute(3, 1, 2, 0).contiguous(), 32)          # [Cin][kh][kw][Cout (padded)]
    g = gy
    if stride == 2:
        B, Ho, Wo, _ = gy.shape
        g = torch.zeros(B, 2 * Ho, 2 * Wo, Cout, device=gy.device)
        g[:, ::2, ::2] = gy
    g = _pad_last(g.contiguous(), 32)
    gx = _conv(g, wt, 1, kh - 1 - pad)
    if gx.shape[1] != H or gx.shape[2] != W:               # odd H / W under stride 2
        assert gx.shape[1] >= H and gx.shape[2] >= W
        gx = gx[:, :H, :W].contiguous()
    return gx


def conv_wgrad(x, gy, w_shape, stride, pad, out=None, accumulate=False):
    Cout, kh, kw, Cin = w_shape
    B, H, W, cs = x.shape
    O._chk(x, gy, out)
    d = _capi.ConvDesc(B, H, W, Cin, cs, 0, Cout, gy.shape[3], 0, 0, 0, kh, kw, stride, pad, _capi.DT_F32, _capi.DT_F32, 0)
    if out is None:
        assert not accumulate
        out = torch.empty(Cout, kh, kw, Cin, device=x.device)
    # split-precision kernel for every layer wide enough to fill its 64-channel tiles (the 3-channel stem and the 1- / 3- / 6-channel heads keep
    # the exact fp32 kernel); one power-of-two scale per operand and call site, cached like the forward's (_site_scale)
    if WGRAD_ARITH == 'f16x3' and Cin % 4 == 0 and Cout % 4 == 0 and cs % 4 == 0 and gy.shape[3] % 4 == 0 and Cin >= 32 and Cout >= 32:
        sx, sg = _site_scale(x), _site_scale(gy)
        n = _capi.lib().dir_conv2d_wgrad_f16x3_workspace_bytes(d)
        if accumulate:
            n = max(n, out.numel() * 4)
        ws = torch.empty(max(n, 4) // 4, device=x.device)
        if _capi.PROFILE is not None:
            _capi.annotate(family='wgrad', flops=2.0 * B * gy.shape[1] * gy.shape[2] * Cout * kh * kw * Cin, bytes=4.0 * (x.numel() + gy.numel() + Cout * kh * kw * Cin),
                           shape='wgrad M=%d Cout=%d Cin=%d k%d s%d' % (B * gy.shape[1] * gy.shape[2], Cout, Cin, kh, stride))
        _capi.check(_capi.lib().dir_conv2d_wgrad_f16x3(d, _capi.ptr(x), _capi.ptr(gy), _capi.ptr(out), int(accumulate), _capi.ptr(ws), n, sx, sg,
                                                       _capi.stream_ptr()), 'dir_conv2d_wgrad_f16x3')
        return out
    n = _capi.lib().dir_conv2d_wgrad_workspace_bytes(d)
    if accumulate:
        n = max(n, out.numel() * 4)
    ws = torch.empty(max(n, 4) // 4, device=x.device)
    if _capi.PROFILE is not None:
        _capi.annotate(family='wgrad', flops=2.0 * B * gy.shape[1] * gy.shape[2] * Cout * kh * kw * Cin, bytes=4.0 * (x.numel() + gy.numel() + Cout * kh * kw * Cin),
                       shape='wgrad M=%d Cout=%d Cin=%d k%d s%d' % (B * gy.shape[1] * gy.shape[2], Cout, Cin, kh, stride))
    _capi.check(_capi.lib().dir_conv2d_wgrad_f32(d, _capi.ptr(x), _capi.ptr(gy), _capi.ptr(out), int(accumulate), _capi.ptr(ws), n,
                                                 _capi.stream_ptr()), 'dir_conv2d_wgrad_f32')
    return out


def conv_bwd(x, w, gy, stride=1, pad=0, need_gx=True, has_bias=True):
    gy = gy.contiguous()
    gw = conv_wgrad(x, gy, w.shape, stride, pad)
    gb = O.colsum(gy.view(-1, gy.shape[3])) if has_bias else None
    gx = conv_dgrad(w, gy, stride, pad, x.shape[1], x.shape[2]) if need_gx else None
    return gx, gw, gb
