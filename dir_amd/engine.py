"""Execution plan of DIR.forward (eval) on MI355X: parameter packing + the kernel sequence.

Host-side plumbing only: every arithmetic step is a call into libdir_hip.so (include/dir_hip.h).  The plan keeps
feature maps NHWC in the compute dtype (bf16 = BASELINE.json config 2, or fp32 = exact-parity mode), the joint
tokens / MANO path in fp32, folds eval-mode BatchNorm + conv bias into per-channel scale/shift epilogues
(models/backbone/resnet.py:120-140) or prologues (pre-activation hourglass.Residual, models/backbone/hourglass.py:55-70),
lets convs write straight into channel slices of the concat buffers (models/dir.py:444,455,461,470), and launches
on torch's current stream so the whole forward can be captured in a HIP graph (torch.cuda.graph).

State-dict keys are the reference's (963 keys, tests/golden/manifest_dir.json).
"""
import ctypes as C
import math
import json
import os
import threading

import torch

from . import _capi
from ._capi import CONV_PRE_RELU, CONV_RELU, DT_BF16, DT_F16, DT_F16X1, DT_F16X1P, DT_F16X3, DT_F16X3P, DT_F32, ConvDesc

F32 = torch.float32
IMAGENET_MEAN = (C.c_float * 3)(0.485, 0.456, 0.406)      # apps/eval.py:49-50
IMAGENET_STD = (C.c_float * 3)(0.229, 0.224, 0.225)
# Live per-launch measurement: set _capi.PROFILE to a list and every library call is bracketed by HIP events and recorded with the
# kernel names it launched plus the algorithmic work announced here with _capi.annotate() (bench.py roofline, DirEngine.autotune).
_TLS = threading.local()      # .variant: DIR_CONV_VARIANT every conv of THIS thread is forced to (autotune); per thread, not global


def _forced_variant():
    return getattr(_TLS, 'variant', None)


def _packing_arith():
    """'f16x3' while a DirEngine(dtype=float32, arith='f16x3') packs its parameters (per thread): every fp32 convolution built meanwhile
    takes the split-precision arithmetic (include/dir_hip.h: DIR_DT_F16X3), else None"""
    return getattr(_TLS, 'arith', None)


def _ann(family, flops, nbytes, shape):
    """algorithmic work of the next library call (minimum HBM bytes = inputs + parameters + outputs once)"""
    if _capi.PROFILE is not None:
        _capi.annotate(family=family, flops=float(flops), bytes=float(nbytes), shape=shape)


HALF = (torch.bfloat16, torch.float16)      # the two 16-bit STORAGE kinds of the throughput modes: DIR_DT_BF16 | DIR_DT_F16 (round 5: f16 storage)


def _dt(dtype):
    return DT_F32 if dtype == torch.float32 else DT_F16 if dtype == torch.float16 else DT_BF16


def _tok_wdt(dtype):
    """weight dtype of the token path (P-GCN / STE Linears) for a feature-map dtype: the f16-storage mode keeps the bf16 weights (autocast
    semantics) -- its kernels take F32 | BF16, and the refined stages they feed are already inside 0.004 mm in the bf16 mode"""
    return torch.bfloat16 if dtype in HALF else torch.float32


def bn_fold(sd, prefix, conv_bias=None, eps=1e-5):
    """eval BatchNorm (after an optional conv bias) as y = x*scale + shift; folded in fp64."""
    g, b = sd[prefix + '.weight'].double(), sd[prefix + '.bias'].double()
    m, v = sd[prefix + '.running_mean'].double(), sd[prefix + '.running_var'].double()
    scale = g / torch.sqrt(v + eps)
    shift = b - m * scale
    if conv_bias is not None:
        shift = shift + conv_bias.double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


STREAM_VARIANT = 21      # DIR_CONV_VARIANT code: dir_conv1x1_stream_forward (1x1, bf16, Cout % 128 == 0), chosen per layer by autotune
STREAM64_VARIANT = 22    # the same kernel on 64-pixel workgroups (twice as many, half as long)
STREAM32_VARIANT = 23    # ... on 32-pixel workgroups (the 16x16 / 8x8 stages: 128-pixel tiles do not even cover the CUs there)
STREAMP_VARIANT = 24     # round 5: the pipelined form -- a fifth (producer) wave feeds a ring of activation chunks by LDS-DMA (no pre-activation form: those layers run 21)
STREAM_VARIANTS = (STREAM_VARIANT, STREAM64_VARIANT, STREAM32_VARIANT, STREAMP_VARIANT)
# round 6: the activation-stationary kernel for the small maps (conv_as.hip, dir_conv2d_as_forward): DIR_CONV_VARIANT -> (A = 32-channel blocks per
# wave, PB = 32-pixel blocks per workgroup).  Bit-identical to the tiled kernels (same K order and k-slots), so autotune may pick it per layer.
AS_VARIANTS = {25: (2, 2), 26: (2, 4), 27: (4, 2), 28: (1, 2)}


def pack_as_weights(w_ohwi, A):
    """dir_conv2d_as_forward's weight stream (include/dir_hip.h) from W [Cout][kh][kw][Cin] (16-bit): [Cout / (128 A)][4 waves][kh kw Cin / 64 steps]
    [4 k-steps][A][64 lanes][8], step = (64-channel slab) * (kh kw) + tap -- every wave's MFMA A fragments in the order it consumes them, with the
    k-slot assignment of conv.hip's MFMAs (lanes 0-31: channels 8 ks .. + 8 of the slab, lanes 32-63: 32 + 8 ks .. + 8)."""
    N, kh, kw, Cin = w_ohwi.shape
    assert N % (128 * A) == 0 and Cin % 64 == 0
    nt = kh * kw
    dev = w_ohwi.device
    w = w_ohwi.reshape(N, nt * Cin)
    lane = torch.arange(64, device=dev)
    l32, h = lane & 31, lane >> 5
    e = torch.arange(8, device=dev)
    g, wv, st, ks, cb = torch.meshgrid(torch.arange(N // (128 * A), device=dev), torch.arange(4, device=dev), torch.arange(nt * Cin // 64, device=dev),
                                       torch.arange(4, device=dev), torch.arange(A, device=dev), indexing='ij')
    row = (g * (128 * A) + (wv * A + cb) * 32)[..., None] + l32                       # [G,4,steps,4,A,64]
    slab, tap = st // nt, st % nt
    k0 = (tap * Cin + 64 * slab + 8 * ks)[..., None] + 32 * h
    return w[row[..., None], k0[..., None] + e].contiguous()


def pack_stream_weights(w_nk, dtype=torch.bfloat16):
    """dir_conv1x1_stream_forward's weight stream (include/dir_hip.h) from W [Cout, K] (K = Cin, or Cin + Cin2 for two sources):
    bf16 [Cout/NWG][4 waves][K/64][4 k-steps][NCB][64 lanes][8]."""
    out_dev = w_nk.device
    w = w_nk.detach().float().cpu()
    N, K = w.shape
    assert N % 128 == 0 and K % 64 == 0
    ncb = 2 if N % 256 == 0 else 1
    dev = w.device
    lane = torch.arange(64, device=dev)
    l32, h = lane & 31, lane >> 5
    e = torch.arange(8, device=dev)
    g, wv, c, ks, cb = torch.meshgrid(torch.arange(N // (128 * ncb), device=dev), torch.arange(4, device=dev), torch.arange(K // 64, device=dev),
                                      torch.arange(4, device=dev), torch.arange(ncb, device=dev), indexing='ij')
    row = (g * (128 * ncb) + (wv * ncb + cb) * 32)[..., None] + l32                    # [G,4,nk,4,ncb,64]
    k0 = (64 * c + 8 * ks)[..., None] + 32 * h            # the k-slot assignment of conv.hip's MFMAs (bit-identical sums)
    return w[row[..., None], k0[..., None] + e].to(dtype).contiguous().to(out_dev)


class ConvOp(object):
    """one dir_conv2d_forward call with packed parameters"""
    def __init__(self, w_oihw, dtype, stride=1, pad=0, scale=None, shift=None, relu=False, pre=None, pre_relu=False,
                 out_dtype=None, arith=None):
        self.w = w_oihw.detach().permute(0, 2, 3, 1).contiguous().to(dtype)
        self.cout, self.kh, self.kw, self.cin = self.w.shape
        self.stride, self.pad, self.dtype = stride, pad, dtype
        self.out_dtype = out_dtype or dtype
        self.arith = (arith or _packing_arith()) if dtype == torch.float32 else None
        self.in_code = DT_F16X3 if self.arith == 'f16x3' else DT_F16X1 if self.arith == 'f16' else _dt(dtype)
        self.in_scale = 1.0                # f16x3: power of two applied to the activations before the split (set_in_scale / DirEngine.calibrate)
        if self.arith in ('f16x3', 'f16'):  # fp32 tensors, f16 hi / lo split arithmetic ('f16': hi only): weights split + pre-scaled here, 1 / p_n into the scale
            from .functional import pack_f16x3_weights
            self.w, scale = pack_f16x3_weights(self.w.reshape(self.cout, -1), scale)
            self.scale0 = scale.float().contiguous().clone()  # the epilogue scale at in_scale = 1 (a copy: set_in_scale overwrites self.scale in place)
        self.scale = None if scale is None else scale.float().contiguous()
        self.shift = None if shift is None else shift.float().contiguous()
        self.pre_scale, self.pre_shift = (None, None) if pre is None else (pre[0].contiguous(), pre[1].contiguous())
        self.flags = (CONV_RELU if relu else 0) | (CONV_PRE_RELU if pre_relu else 0)
        self.ho = self.wo = 0
        self.in_cs_override = None
        # f16 arithmetic modes: layers whose activations are worth splitting ONCE (dir_split_f16_forward) so that both operands go global ->
        # LDS by DMA: every 3x3 (each input pixel is read by 9 taps and by every output-channel tile) and the 1x1 layers with >= 4
        # output-channel tiles; the rest (HBM-bound 1x1 layers with one or two tiles) keep converting while staging.  DIR_PRESPLIT=0|1|2: never / rule / always
        self.presplit = self.arith is not None and (ConvOp.PRESPLIT == 2 or (ConvOp.PRESPLIT == 1 and (
            self.kh * self.kw >= 9 or (self.cout >= 512 and self.cin >= 128))))
        self.split_consumer = None                         # f16 arithmetic modes: the ONE convolution reading this op's whole output (link_split)
        self.alg_k = self.kh * self.kw * self.cin          # reduction length the reference computes (stem: 147)
        self.variant = {}                                  # batch size -> DIR_CONV_VARIANT code chosen by DirEngine.autotune
        self.split, self._ws = {}, {}                      # batch size -> split-K factor; (B, S, stream) -> workspace
        self._w_as = {}                                    # A -> the activation-stationary kernel's weight stream (pack_as_weights), packed on first use
        # streaming alternative for the HBM-bound 1x1 layers (dir_conv1x1_stream_forward), taken when autotune prefers it
        self.w_stream = None
        if (dtype in HALF and self.out_dtype == dtype and self.kh == 1 and self.kw == 1 and stride == 1 and pad == 0
                and self.cin % 64 == 0 and self.cout % 128 == 0 and self.cin <= 2304):
            self.w_stream = pack_stream_weights(self.w.reshape(self.cout, self.cin), dtype)

    PRESPLIT = int(os.environ.get('DIR_PRESPLIT', '1'))
    OUT_SPLIT = os.environ.get('DIR_OUT_SPLIT', '1') != '0'      # producers write the next convolution's pre-split operand (link_split)

    def link_split(self, consumer):
        """f16 arithmetic modes: declare that `consumer` (a ConvOp without pre-activation) is the ONLY reader of this convolution's whole fp32
        output: the epilogue then writes that tensor directly as the consumer's pre-split operand (f16 hi | lo slabs times the consumer's
        in_scale -- the same bytes as fp32), and the consumer takes the two-operand DMA path without a dir_split_f16_forward pass."""
        if (self.arith is not None and isinstance(consumer, ConvOp) and consumer.arith == self.arith and consumer.pre_scale is None
                and consumer.in_cs_override is None and self.cout == consumer.cin and self.cout % 32 == 0 and self.out_dtype == F32):
            self.split_consumer = consumer

    def set_in_scale(self, s):
        """f16x3: multiply the activations by the power of two `s` before the hi / lo split; 1 / s goes into the epilogue scale (exact)"""
        assert self.arith in ('f16x3', 'f16') and s > 0 and math.frexp(s)[0] == 0.5
        self.in_scale = float(s)
        self.scale.copy_(self.scale0 / s)

    def _calibrate(self, xs):
        """DirEngine.calibrate: pick in_scale from this batch -- the largest |activation| the layer reads lands in [2^9, 2^10)"""
        amax = max(float(t.abs().max()) for t in xs)
        if self.pre_scale is not None:     # the split sees relu(x * ps + pb): bound it by |x| max * |ps| max + |pb| max
            amax = amax * float(self.pre_scale.abs().max()) + float(self.pre_shift.abs().max())
        self.set_in_scale(2.0 ** (10 - math.frexp(amax)[1]) if amax > 0 and math.isfinite(amax) else 1.0)

    def __call__(self, x, out=None, out_coff=0, in_coff=0, residual=None, res_coff=0, bbox=None, _replay_split=None):
        B, H, W, cbuf = x.shape
        x_arg, in_coff_arg = x, in_coff
        if self.arith is not None and getattr(_TLS, 'calibrating', False):
            self._calibrate([x[..., in_coff:in_coff + self.cin]] if self.in_cs_override is None else [x])
        ho = self.ho or (H + 2 * self.pad - self.kh) // self.stride + 1
        wo = self.wo or (W + 2 * self.pad - self.kw) // self.stride + 1
        out_given, residual_ok = out, True
        if out is None:
            out = torch.empty(B, ho, wo, self.cout, device=x.device, dtype=self.out_dtype)
        pre_scale, pre_shift, flags, in_code, in_cs = self.pre_scale, self.pre_shift, self.flags, self.in_code, self.in_cs_override or cbuf
        if getattr(x, '_dir_split', False):
            # the producer's epilogue already wrote this tensor as f16 hi | lo slabs scaled by OUR in_scale (link_split): straight to the DMA path
            assert in_coff == 0 and cbuf == self.cin and pre_scale is None and bbox is None
            in_code = DT_F16X1P if self.arith == 'f16' else DT_F16X3P
        elif self.presplit and self.in_cs_override is None and bbox is None:
            # the activations' hi | lo split (with in_scale and the pre-activation) as its own HBM-bound pass; the convolution then reads
            # both operands by DMA (DIR_DT_F16X3P / F16X1P)
            xs = torch.empty(B, H, W, self.cin, device=x.device, dtype=F32)
            _ann('split_f16', 0, 2 * xs.numel() * 4, 'M=%d C=%d fp32 -> f16 hi | lo' % (B * H * W, self.cin))
            _capi.check(_capi.lib().dir_split_f16_forward(_capi.ptr(x), _capi.ptr(xs), B * H * W, self.cin, cbuf, in_coff, _capi.ptr(pre_scale), _capi.ptr(pre_shift),
                                                          1 if (flags & CONV_PRE_RELU) else 0, self.in_scale, 1 if self.arith == 'f16' else 0, _capi.stream_ptr()),
                        'dir_split_f16_forward')
            x, in_cs, in_coff, pre_scale, pre_shift = xs, self.cin, 0, None, None
            flags &= ~CONV_PRE_RELU
            in_code = DT_F16X1P if self.arith == 'f16' else DT_F16X3P
        d = ConvDesc(B, H, W, self.cin, in_cs, in_coff, self.cout, out.shape[3], out_coff,
                     residual.shape[3] if residual is not None else 0, res_coff, self.kh, self.kw, self.stride, self.pad,
                     in_code, _dt(out.dtype), flags, self.ho, self.wo, self.in_scale)
        cons = self.split_consumer
        write_split = (cons is not None and out_given is None and residual_ok and not getattr(_TLS, 'calibrating', False)
                       and not getattr(_TLS, 'no_out_split', False) and ConvOp.OUT_SPLIT)
        if _replay_split is not None:
            write_split = _replay_split
        if getattr(_TLS, 'capture', None) is not None:       # autotune_energy: this call, replayable (same output buffer, same hand-over format)
            _TLS.capture.append((self, (x_arg,), dict(out=out, out_coff=out_coff, in_coff=in_coff_arg, residual=residual, res_coff=res_coff, bbox=bbox,
                                                  _replay_split=write_split)))
        if write_split:      # the only reader is the next convolution: write its pre-split operand instead of fp32 (same bytes, no split pass)
            d.out_split_scale = -cons.in_scale if cons.arith == 'f16' else cons.in_scale
        v = _forced_variant() if _forced_variant() is not None else self.variant.get(B, self.variant.get(getattr(_TLS, 'parent_batch', None), 0))
        d.flags |= (v & 0xff) << 8
        if _capi.PROFILE is not None:
            nbytes = (B * H * W * self.cin * x.element_size() + self.w.numel() * self.w.element_size()
                      + B * ho * wo * self.cout * out.element_size() * (2 if residual is not None else 1))
            _capi.annotate(family='conv', flops=2.0 * B * ho * wo * self.cout * self.alg_k, bytes=nbytes, op=self, flops_real=2.0 * B * ho * wo * self.cout * self.alg_k * getattr(self, 'alg_scale', 1.0),
                           dtype=self.arith or ('f32' if self.dtype == F32 else 'bf16'),        # (roofline class: f16 storage runs at the bf16 MFMA rate)
                           shape='M=%d N=%d K=%d k%dx%d s%d' % (B * ho * wo, self.cout, self.kh * self.kw * self.cin, self.kh,
                                                              self.kw, self.stride))
        if v in AS_VARIANTS and bbox is None and pre_scale is None and self.dtype in HALF and out.dtype == self.dtype and self.arith is None:
            A_, PB_ = AS_VARIANTS[v]
            if _capi.lib().dir_conv2d_as_supported(d, A_, PB_):
                was = self._w_as.get(A_)
                if was is None:
                    was = self._w_as[A_] = pack_as_weights(self.w, A_)
                d.flags &= 0xff
                _capi.check(_capi.lib().dir_conv2d_as_forward(d, _capi.ptr(x), _capi.ptr(was), _capi.ptr(self.scale), _capi.ptr(self.shift), _capi.ptr(residual),
                                                              _capi.ptr(out), A_, PB_, _capi.stream_ptr()), 'dir_conv2d_as_forward')
                return out
            d.flags &= 0xff                                    # does not apply to this layer: the library's heuristic, like every other variant
        if v in STREAM_VARIANTS and self.w_stream is not None and residual is None and bbox is None and out.dtype == self.dtype:
            d.flags = (d.flags & 0xff) | ((v & 0xff) << 8 if v != STREAM_VARIANT else 0)
            _capi.check(_capi.lib().dir_conv1x1_stream_forward(d, _capi.ptr(x), None, None, _capi.ptr(self.w_stream), _capi.ptr(self.scale),
                                                               _capi.ptr(self.shift), _capi.ptr(self.pre_scale), _capi.ptr(self.pre_shift),
                                                               _capi.ptr(out), _capi.stream_ptr()), 'dir_conv1x1_stream_forward')
            return out
        S = self.splits(B, ho, wo) if (bbox is None and out.dtype in HALF and self.dtype in HALF) else 1
        if S > 1 and (_forced_variant() in (None, 0)):
            d.flags &= 0xff
            ws = self._splitk_ws(d, S, B, x.device)
            _capi.check(_capi.lib().dir_conv2d_splitk_forward(d, _capi.ptr(x), _capi.ptr(self.w), _capi.ptr(self.scale), _capi.ptr(self.shift),
                                                              _capi.ptr(self.pre_scale), _capi.ptr(self.pre_shift), _capi.ptr(residual),
                                                              _capi.ptr(out), S, _capi.ptr(ws), ws.numel(), _capi.stream_ptr()),
                        'dir_conv2d_splitk_forward')
            return out
        if bbox is not None:
            rc = _capi.lib().dir_conv2d_sparse_forward(d, _capi.ptr(x), _capi.ptr(self.w), _capi.ptr(self.scale),
                                                       _capi.ptr(self.shift), _capi.ptr(residual), _capi.ptr(out),
                                                       _capi.ptr(bbox), _capi.stream_ptr())
        else:
            rc = _capi.lib().dir_conv2d_forward(d, _capi.ptr(x), _capi.ptr(self.w), _capi.ptr(self.scale),
                                                _capi.ptr(self.shift), _capi.ptr(pre_scale),
                                                _capi.ptr(pre_shift), _capi.ptr(residual), _capi.ptr(out),
                                                _capi.stream_ptr())
        _capi.check(rc, 'dir_conv2d_forward')
        if write_split:
            out._dir_split = True
        return out


    # ---- split-K (dir_conv2d_splitk_forward) for layers whose M x Cout grid of 128x128 tiles covers at most half of the 256 CUs at
    # this batch size (ResNet layer4 at 8x8, the decoder's 16x16 Residual blocks).  OFF by default: measured at B = 64
    # (tools/bench_splitk.py) it wins only on the 2304 -> 128 pre-activation 1x1 (46 -> 39 us) and ties on layer4's 3x3 (36 -> 34 us); every
    # other candidate is slower than its best tiled variant -- each extra split costs 3 - 5 us of partial-tile traffic through the
    # device-coherent level -- and its sums are not bit-identical to the tiled kernels'.  DIR_SPLITK=1 enables the shape heuristic
    # (shapes only, never timing); op.split[B] = S forces a factor.
    SPLITK = os.environ.get('DIR_SPLITK', '0') == '1'

    def splits(self, B, ho, wo):
        S = self.split.get(B)
        if S is None:
            S = 1
            tiles = -(-(B * ho * wo) // 128) * -(-self.cout // 128)
            nk = self.kh * self.kw * self.cin // 64
            if ConvOp.SPLITK and self.cout % 8 == 0 and tiles <= 128 and nk >= 8:
                S = max(1, min(16, nk // 4, -(-512 // tiles)))
            self.split[B] = S
        return S

    def _splitk_ws(self, d, S, B, dev):
        key = (B, S, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            n = _capi.lib().dir_conv2d_splitk_workspace_bytes(d, S)
            assert n > 0
            ws = torch.empty(n, dtype=torch.uint8, device=dev)
            ws[:16384].zero_()                             # the arrival counters; the kernel leaves them zero
            self._ws[key] = ws
        return ws


class DualConvOp(object):
    """dir_conv2d_dual_forward: a bottleneck's conv3 + BN with the projection shortcut (downsample conv + BN, stride s) folded
    in as a second K range -- relu(bn3(conv3(y)) + bn_ds(conv_ds(x))) in one launch, the identity tensor never exists
    (models/backbone/resnet.py:117-119,137-140).  Both BatchNorm scales are multiplied into the weight rows."""
    def __init__(self, w3, s3, h3, wds, sds, hds, stride2, dtype, relu=True):
        self.relu = relu
        self.cout, self.cin = w3.shape[0], w3.shape[1]
        self.cin2, self.stride2, self.dtype = wds.shape[1], stride2, dtype
        w = torch.cat([w3.float().flatten(1) * s3.float()[:, None], wds.float().flatten(1) * sds.float()[:, None]], 1)
        self.w = w.contiguous().to(dtype)                                    # [Cout][Cin + Cin2]
        self.arith = _packing_arith() if dtype == torch.float32 else None
        self.scale = None
        if self.arith is not None:         # split-precision rows; their power-of-two prescale comes back out through a scale vector
            from .functional import pack_f16x3_weights
            self.w, self.scale = pack_f16x3_weights(self.w)
            self.scale0 = self.scale.clone()
        self.in_scale = 1.0
        self.shift = (h3.float() + hds.float()).contiguous()
        self.kh = self.kw = self.stride = 1
        self.pre_scale = None
        self.variant = {}
        self.w_stream = None
        if dtype in HALF and self.cin % 64 == 0 and self.cin2 % 64 == 0 and self.cout % 128 == 0:
            self.w_stream = pack_stream_weights(self.w, dtype)

    set_in_scale = ConvOp.set_in_scale
    pre_scale = None

    def __call__(self, y, x, out=None, out_coff=0, x_decimated=False):
        """x_decimated: x already holds only the pixels the stride-`stride2` shortcut reads ([B,H,W,Cin2] at y's resolution)"""
        B, H, W, cbuf = y.shape
        if self.arith is not None and getattr(_TLS, 'calibrating', False):
            ConvOp._calibrate(self, [y[..., :self.cin], x[..., :self.cin2]])
        if out is None:
            out = torch.empty(B, H, W, self.cout, device=y.device, dtype=self.dtype)
        if getattr(_TLS, 'capture', None) is not None:
            _TLS.capture.append((self, (y, x), dict(out=out, out_coff=out_coff, x_decimated=x_decimated)))
        d = ConvDesc(B, H, W, self.cin, cbuf, 0, self.cout, out.shape[3], out_coff, 0, 0, 1, 1, 1, 0,
                     DT_F16X3 if self.arith == 'f16x3' else DT_F16X1 if self.arith == 'f16' else _dt(self.dtype), _dt(self.dtype),
                     CONV_RELU if self.relu else 0, 0, 0, self.in_scale)
        v = _forced_variant() if _forced_variant() is not None else self.variant.get(B, self.variant.get(getattr(_TLS, 'parent_batch', None), 0))
        d.flags |= (v & 0xff) << 8
        d2 = _capi.ConvSrc2(x.shape[1], x.shape[2], self.cin2, x.shape[3], 0, 1 if x_decimated else self.stride2)
        if _capi.PROFILE is not None:
            es, m = y.element_size(), B * H * W
            _capi.annotate(family='conv', flops=2.0 * m * self.cout * (self.cin + self.cin2), op=self,
                           bytes=(m * self.cin + m * self.cin2 + self.w.numel() + m * self.cout) * es,
                           dtype=self.arith or ('f32' if self.dtype == F32 else 'bf16'),
                           shape='M=%d N=%d K=%d+%d dual s%d' % (m, self.cout, self.cin, self.cin2, self.stride2))
        if v in STREAM_VARIANTS and self.w_stream is not None:
            d.flags = (d.flags & 0xff) | ((v & 0xff) << 8 if v != STREAM_VARIANT else 0)
            _capi.check(_capi.lib().dir_conv1x1_stream_forward(d, _capi.ptr(y), d2, _capi.ptr(x), _capi.ptr(self.w_stream), None, _capi.ptr(self.shift),
                                                               None, None, _capi.ptr(out), _capi.stream_ptr()), 'dir_conv1x1_stream_forward')
            return out
        if self.scale is not None:
            _capi.check(_capi.lib().dir_conv2d_dual_scaled_forward(d, _capi.ptr(y), d2, _capi.ptr(x), _capi.ptr(self.w), _capi.ptr(self.scale),
                                                                   _capi.ptr(self.shift), _capi.ptr(out), _capi.stream_ptr()),
                        'dir_conv2d_dual_scaled_forward')
            return out
        _capi.check(_capi.lib().dir_conv2d_dual_forward(d, _capi.ptr(y), d2, _capi.ptr(x), _capi.ptr(self.w), _capi.ptr(self.shift),
                                                        _capi.ptr(out), _capi.stream_ptr()), 'dir_conv2d_dual_forward')
        return out


def pack_token_mlp(sd, prefix, keep):
    """nn.Sequential(Conv1d(k=1), BatchNorm1d, ReLU, Conv1d(k=1)) -> dir_token_mlp (k-major weights, folded BN)."""
    w1 = sd[prefix + '.0.weight'][:, :, 0]
    s1, b1 = bn_fold(sd, prefix + '.1', sd[prefix + '.0.bias'])
    t = dict(w1t=w1.t().contiguous().float(), s1=s1, b1=b1,
             w2t=sd[prefix + '.3.weight'][:, :, 0].t().contiguous().float(), b2=sd[prefix + '.3.bias'].float().contiguous())
    keep.append(t)
    return _capi.TokenMlp(*(t[k].data_ptr() for k in ('w1t', 's1', 'b1', 'w2t', 'b2')))


def pack_pgcn(sd, prefix, keep, num_layers=4, weight_dtype=torch.float32):
    """weight_dtype float32: exact fp32 matmuls (reference layout); bfloat16: bf16 matmuls = autocast semantics (W transposed)"""
    arr = (_capi.PgcnLayer * num_layers)()
    for i in range(num_layers):
        p = '%s.gconv_layers.%d' % (prefix, i)
        s, b = bn_fold(sd, p + '.bn')
        W = sd[p + '.gconv.W']
        W = W.float().contiguous() if weight_dtype == torch.float32 else W.detach().transpose(2, 3).contiguous().to(torch.bfloat16)
        t = dict(W=W, e1=sd[p + '.gconv.e_1'].float().reshape(-1).contiguous(),
                 bias=sd[p + '.gconv.bias'].float().contiguous(), s=s, b=b)
        keep.append(t)
        arr[i] = _capi.PgcnLayer(t['W'].data_ptr(), t['e1'].data_ptr(), t['bias'].data_ptr(), s.data_ptr(), b.data_ptr(), 1,
                                 _dt(weight_dtype))
    return arr


def pack_ste(sd, prefix, keep, depth=4, weight_dtype=torch.float32):
    """weight_dtype float32: exact fp32 Linears (k-major weights); bfloat16: bf16 Linears = autocast semantics ([out][in])."""
    def f(k):
        t = sd[prefix + '.' + k].float().contiguous()
        keep.append(t)
        return t.data_ptr()

    def ft(k):
        w = sd[prefix + '.' + k]
        t = w.float().t().contiguous() if weight_dtype == torch.float32 else w.detach().to(torch.bfloat16).contiguous()
        keep.append(t)
        return t.data_ptr()
    P = _capi.SteParams()
    P.weight_dtype = _dt(weight_dtype)
    pe = sd[prefix + '.spatial_pos_embed'].float().reshape(42, 128).contiguous()
    keep.append(pe)
    P.pos_embed = pe.data_ptr()
    for i in range(1, depth):                       # block 0 is never executed (transformer/mixSTE.py:197)
        b = 'STEblocks.%d.' % i
        P.blocks[i - 1] = _capi.SteBlock(f(b + 'norm1.weight'), f(b + 'norm1.bias'), ft(b + 'attn.qkv.weight'),
                                         f(b + 'attn.qkv.bias'), ft(b + 'attn.proj.weight'), f(b + 'attn.proj.bias'),
                                         f(b + 'norm2.weight'), f(b + 'norm2.bias'), ft(b + 'mlp.fc1.weight'),
                                         f(b + 'mlp.fc1.bias'), ft(b + 'mlp.fc2.weight'), f(b + 'mlp.fc2.bias'))
    P.num_blocks = depth - 1
    P.snorm_w, P.snorm_b = f('spatial_norm.weight'), f('spatial_norm.bias')
    P.head_ln_w, P.head_ln_b = f('head.0.weight'), f('head.0.bias')
    P.head_wt, P.head_b = ft('head.1.weight'), f('head.1.bias')
    return P


def _pad_rows(t, width=2336):
    """[K, 2334] -> contiguous [K, 2336] (zero padded): 16-byte aligned rows for the MANO kernel's float4 reads"""
    out = torch.zeros(t.shape[0], width, device=t.device, dtype=torch.float32)
    out[:, :t.shape[1]] = t
    return out


def pack_mano(sd, prefix, side, center_idx, keep):
    f = lambda k: sd[prefix + '.' + k].float()  # noqa: E731
    t = dict(shapedirs_t=_pad_rows(f('th_shapedirs').reshape(2334, 10).t()),
             posedirs_t=_pad_rows(f('th_posedirs').reshape(2334, 135).t()),
             v_template=f('th_v_template').reshape(2334).contiguous(),
             j_template=(f('th_J_regressor').double() @ f('th_v_template').double().reshape(778, 3)).float().contiguous(),
             j_shapedirs=torch.einsum('jv,vck->jck', f('th_J_regressor').double(),
                                      f('th_shapedirs').double()).float().contiguous(),
             weights=f('th_weights').contiguous(), hands_mean=f('th_hands_mean').reshape(45).contiguous(),
             comps=f('th_selected_comps').contiguous())
    keep.append(t)
    return _capi.ManoTables(t['shapedirs_t'].data_ptr(), t['posedirs_t'].data_ptr(), t['v_template'].data_ptr(),
                            t['j_template'].data_ptr(), t['j_shapedirs'].data_ptr(), t['weights'].data_ptr(), t['hands_mean'].data_ptr(),
                            t['comps'].data_ptr(), 0 if side == 'right' else 1,
                            -1 if center_idx is None else int(center_idx), 0)


class BneckChainOp(object):
    """dir_bottleneck_chain_forward: conv2 + bn2 + ReLU + conv3 + bn3 + identity + ReLU of a layer1 bottleneck and, optionally,
    the next block's conv1 + bn1 + ReLU in one launch (models/backbone/resnet.py:122-140).  Built from the blocks' ConvOps;
    carries the attributes autotune / export_tuning read from a conv op (it has a single kernel: every variant code is a no-op)."""

    def __init__(self, c2, c3, c1n=None, dual=None):
        """c3: the block's conv3 ConvOp, or None with dual = DualConvOp (conv3 + projection shortcut, BN scales folded into rows)"""
        self.c2, self.c3, self.c1n, self.dual = c2, c3, c1n, dual
        self.cout, self.cin, self.kh, self.kw, self.stride = 256, c2.cin, 3, 3, 1
        self.variant = {}
        dev = c2.w.device
        if dual is None:
            self.w3 = c3.w.reshape(c3.cout, c3.cin).contiguous()
            self.wd, self.s3, self.h3 = None, c3.scale, c3.shift
        else:
            self.w3 = dual.w[:, :64].contiguous()
            self.wd = dual.w[:, 64:].contiguous()
            self.s3, self.h3 = torch.ones(256, device=dev, dtype=F32), dual.shift
        self.w1n = c1n.w.reshape(c1n.cout, c1n.cin).contiguous() if c1n is not None else None
        n = (lambda t: None) if c1n is None else _capi.ptr        # noqa: E731
        self.params = _capi.BneckChainParams(_capi.ptr(c2.w), _capi.ptr(c2.scale), _capi.ptr(c2.shift), _capi.ptr(self.w3),
                                             _capi.ptr(self.s3), _capi.ptr(self.h3), n(self.w1n),
                                             n(c1n.scale if c1n is not None else None), n(c1n.shift if c1n is not None else None),
                                             _capi.ptr(self.wd) if self.wd is not None else None, c1n.cout if c1n is not None else 0, 0, _dt(c2.dtype))

    @staticmethod
    def applies(c2, c3, c1n, dtype, dual=None):
        ok = dtype in HALF and c2.cin == 64 and c2.cout == 64 and c2.kh == 3 and c2.stride == 1 and c2.scale is not None
        if dual is None:
            ok = ok and c3.cin == 64 and c3.cout == 256 and c3.scale is not None
        else:
            ok = ok and dual.cin == 64 and dual.cin2 == 64 and dual.cout == 256 and dual.stride2 == 1 and dual.relu
        return ok and (c1n is None or (c1n.cin == 256 and c1n.cout in (64, 128) and c1n.kh == 1 and c1n.stride == 1 and c1n.scale is not None))

    def __call__(self, y1, x, decimate=False):
        """x: the block input -- the identity residual, or (dual) the projection shortcut's source.  decimate: write only the even (y, x)
        pixels of the block output, as [B,H/2,W/2,256] (the last layer1 block when nobody reads c1: dir_bneck_chain_params.out_decimate)"""
        residual, x2 = (None, x) if self.dual is not None else (x, None)
        B, H, W, _ = y1.shape
        if getattr(_TLS, 'capture_fused', None) is not None:       # tools/energy_profile.py: replayable (allocates its own outputs)
            _TLS.capture_fused.append((self, (y1, x), dict(decimate=decimate)))
        self.params.out_decimate = 1 if decimate else 0
        out = torch.empty((B, H // 2, W // 2, 256) if decimate else (B, H, W, 256), device=y1.device, dtype=y1.dtype)
        y1n = torch.empty(B, H, W, self.c1n.cout, device=y1.device, dtype=y1.dtype) if self.c1n is not None else None
        if _capi.PROFILE is not None:
            m, nx = B * H * W, self.c1n is not None
            n2 = self.c1n.cout if nx else 0
            _capi.annotate(family='conv', flops=2.0 * m * (64 * 576 + 256 * 64 * (2 if x2 is not None else 1) + n2 * 256), op=self, dtype='bf16',
                           shape='M=%d chain 3x3(64)+1x1(256)%s' % (m, '+1x1(%d)' % n2 if nx else ''),
                           bytes=(m * (64 + 256 * (1 if residual is not None else 0) + (64 if decimate else 256) + n2 + (64 if x2 is not None else 0))
                                  + self.c2.w.numel() + self.w3.numel() * (2 if x2 is not None else 1) + (self.w1n.numel() if nx else 0)) * 2)
        _capi.check(_capi.lib().dir_bottleneck_chain_forward(C.byref(self.params), _capi.ptr(y1), _capi.ptr(residual), _capi.ptr(x2), _capi.ptr(out),
                                                             _capi.ptr(y1n), B, H, W, _capi.stream_ptr()), 'dir_bottleneck_chain_forward')
        return out, y1n


def pack_tail_stream(w3, w1n, waves=8, dtype=torch.bfloat16):
    """dir_bottleneck_tail_forward's weight stream (include/dir_hip.h): conv3.weight [4P, P] and the next conv1.weight [N2, 4P]
    as bf16 MFMA A-operand fragments in the order the kernel's waves consume them, [4P/512][waves][NBF + NCF][64 lanes][8]
    (waves = 8: 64-pixel tiles, one workgroup per CU; waves = 4: the thin variant, 32-pixel tiles, two workgroups per CU)."""
    out_dev = w3.device
    w3 = w3.detach().float().reshape(w3.shape[0], -1).cpu()        # packed on the host: hundreds of small gathers, once per engine
    w1n = w1n.detach().float().reshape(w1n.shape[0], -1).cpu()
    C4, P = w3.shape
    N2 = w1n.shape[0]
    assert w1n.shape[1] == C4 and C4 == 4 * P and C4 % 512 == 0 and N2 in (128, 256)
    dev = w3.device
    lane = torch.arange(64, device=dev)
    l32, h, l16, g16 = lane & 31, lane >> 5, lane & 15, lane >> 4
    e = torch.arange(8, device=dev)
    KBS = P // 16
    out = []
    if waves == 4:
        ncc = N2 // 128
        for hf in range(C4 // 512):
            for w in range(4):
                for cb in range(4):
                    for ks in range(KBS):
                        out.append(w3[(hf * 512 + 128 * w + 32 * cb + l32)[:, None], (16 * ks + 8 * h)[:, None] + e])
                for ks in range(32):
                    for cc in range(ncc):
                        out.append(w1n[((N2 // 4) * w + 32 * cc + l32)[:, None], (hf * 512 + 16 * ks + 8 * h)[:, None] + e])
        return torch.stack(out).to(dtype).contiguous().to(out_dev)
    for hf in range(C4 // 512):
        for w in range(8):
            for cb in range(2):
                for ks in range(KBS):
                    ch = hf * 512 + 64 * w + 32 * cb + l32
                    k0 = 16 * ks + 8 * h
                    out.append(w3[ch[:, None], k0[:, None] + e])
            if N2 == 128:
                for fc in range(16):
                    out.append(w1n[(16 * w + l16)[:, None], (hf * 512 + 32 * fc + 8 * g16)[:, None] + e])
            else:
                for fc in range(32):
                    out.append(w1n[(32 * w + l32)[:, None], (hf * 512 + 16 * fc + 8 * h)[:, None] + e])
    return torch.stack(out).to(dtype).contiguous().to(out_dev)


class BneckTailOp(object):
    """dir_bottleneck_tail_forward: conv3 + bn3 + identity + ReLU of a layer2 / layer3 bottleneck and the next block's conv1 + bn1 +
    ReLU in one launch (models/backbone/resnet.py:132-140,122-124): the block output is written once and not read back.  Built from
    the blocks' ConvOps; carries the attributes autotune / export_tuning read from a conv op (single kernel: variant codes are no-ops)."""
    GEOMETRIES = ((128, 128), (128, 256), (256, 256))
    # the thin variant is built and parity-tested but measured slower at every size (layer3, B = 64: 41.8 vs 34.7 us; it streams the
    # weights twice as often): off unless DIR_TAIL_THIN_MAX_TILES says otherwise
    THIN_MAX_TILES = int(os.environ.get('DIR_TAIL_THIN_MAX_TILES', '0'))
    force_waves = int(os.environ.get('DIR_TAIL_WAVES', '0'))

    def __init__(self, c3, c1n):
        self.c3, self.c1n = c3, c1n
        self.cout, self.cin, self.kh, self.kw, self.stride = c3.cout, c3.cin, 1, 1, 1
        self.variant = {}
        self.stream = {n: pack_tail_stream(c3.w.reshape(c3.cout, c3.cin), c1n.w.reshape(c1n.cout, c1n.cin), n, dtype=c3.dtype) for n in (8, 4)}
        self.params = {n: _capi.BneckTailParams(_capi.ptr(self.stream[n]), _capi.ptr(c3.scale), _capi.ptr(c3.shift), _capi.ptr(c1n.scale),
                                                _capi.ptr(c1n.shift), c3.cin, c1n.cout, n, _dt(c3.dtype)) for n in (8, 4)}

    @staticmethod
    def applies(c3, c1n, dtype):
        return (dtype in HALF and c3.kh == 1 and c3.stride == 1 and c3.scale is not None and c3.cout == 4 * c3.cin
                and c1n.kh == 1 and c1n.stride == 1 and c1n.cin == c3.cout and c1n.scale is not None and c1n.pre_scale is None
                and (c3.cin, c1n.cout) in BneckTailOp.GEOMETRIES)

    def __call__(self, y2, x, out=None, y1n=None):
        """y2: conv2's output [B,H,W,P]; x: the block input [B,H,W,4P] (identity residual) -> (block output, next y1); out / y1n: optional
        contiguous destinations (the sub-batched high-resolution half writes slices of the whole batch's tensors)"""
        B, H, W, P = y2.shape
        M = B * H * W
        if getattr(_TLS, 'capture_fused', None) is not None:
            _TLS.capture_fused.append((self, (y2, x), {}))
        if out is None:
            out = torch.empty(B, H, W, self.c3.cout, device=y2.device, dtype=y2.dtype)
        if y1n is None:
            y1n = torch.empty(B, H, W, self.c1n.cout, device=y2.device, dtype=y2.dtype)
        if _capi.PROFILE is not None:
            c4, n2 = self.c3.cout, self.c1n.cout
            _capi.annotate(family='conv', flops=2.0 * M * (P * c4 + c4 * n2), op=self, dtype='bf16',
                           shape='M=%d tail 1x1(%d->%d)+res+1x1(->%d)' % (M, P, c4, n2),
                           bytes=(M * (P + 2 * c4 + n2) + c4 * P + n2 * c4) * 2)
        # thin variant (32-pixel tiles, two workgroups per CU) while the fat one would leave workgroups with a single tile
        waves = self.force_waves or (4 if M // 64 <= self.THIN_MAX_TILES else 8)
        _capi.check(_capi.lib().dir_bottleneck_tail_forward(C.byref(self.params[waves]), _capi.ptr(y2), _capi.ptr(x), _capi.ptr(out), _capi.ptr(y1n),
                                                            M, _capi.stream_ptr()), 'dir_bottleneck_tail_forward')
        return out, y1n


def stem_conv_op(w, scale, shift, dtype):
    """7x7/2 stem over 2x2 space-to-depth blocks (dir_stem_prep_s2d): a 4x4 stride-1 convolution whose K-slab is a block-row
    window of 4 blocks x 16 channels; w'[n][j*16 + (dy*2+dx)*4 + c][r] = w[n, c, 2r+dy-1, 2j+dx-1]   (w = conv1.weight [64,3,7,7])"""
    wp = torch.zeros(64, 64, 4, 1, device=w.device, dtype=F32)        # [Cout, Cin', kh, kw=1]
    for r in range(4):
        for dy in range(2):
            ky = 2 * r + dy - 1
            if not 0 <= ky <= 6:
                continue
            for j in range(4):
                for dx in range(2):
                    kx = 2 * j + dx - 1
                    if 0 <= kx <= 6:
                        c0 = j * 16 + (dy * 2 + dx) * 4
                        wp[:, c0:c0 + 3, r, 0] = w[:, :, ky, kx]
    op = ConvOp(wp, dtype, stride=1, pad=0, scale=scale, shift=shift, relu=True)
    op.ho = op.wo = 128
    op.in_cs_override = 16
    op.alg_k = 147
    return op


class BackboneOp(object):
    """ResNet-50 pyramid (models/backbone/resnet.py:243-255): stem as a 4x4 implicit GEMM over 2x2 space-to-depth blocks,
    maxpool, 16 bottlenecks with BN folded into the conv epilogues, the residual add + ReLU fused, and the four projection
    shortcuts folded into their block's conv3 as a second K range (dir_conv2d_dual_forward)."""
    fold_downsample = os.environ.get('DIR_FOLD_DOWNSAMPLE', '1') != '0'
    fused_stem = os.environ.get('DIR_FUSED_STEM', '1') != '0'       # bf16 mode: conv1 + bn1 + ReLU + maxpool in one launch
    bneck_chain = os.environ.get('DIR_BNECK_CHAIN', '1') != '0'     # bf16 mode, layer1: conv2 + conv3 (+ next conv1) in one launch
    bneck_tail = os.environ.get('DIR_BNECK_TAIL', '1') != '0'       # bf16 mode, layer2 / layer3: conv3 + residual + next conv1 in one launch

    def __init__(self, sd, p, dtype, device):
        dt = self.dtype = dtype
        self.device = device
        # dir_stem_pool_forward: conv1.weight [64,3,7,7] -> bf16 [64][ky 7][kx 8][c 4] (zero for kx = 7, c = 3)
        w = sd[p + '.conv1.weight']
        wk = torch.zeros(64, 7, 8, 4, device=w.device, dtype=F32)
        wk[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        self.stem_w = wk.to(dtype if dtype in HALF else torch.bfloat16).contiguous().to(device)
        s_, h_ = bn_fold(sd, p + '.bn1')
        self.stem_scale, self.stem_shift = s_.float().contiguous().to(device), h_.float().contiguous().to(device)
        self.stem = stem_conv_op(sd[p + '.conv1.weight'], s_, h_, dt)     # staged path (fp32 mode, DIR_FUSED_STEM=0)
        self.layers = []
        for li, n in enumerate((3, 4, 6, 3), start=1):
            blocks = []
            for b in range(n):
                q = '%s.layer%d.%d' % (p, li, b)
                stride = 2 if (b == 0 and li > 1) else 1
                s1, h1 = bn_fold(sd, q + '.bn1')
                s2, h2 = bn_fold(sd, q + '.bn2')
                s3, h3 = bn_fold(sd, q + '.bn3')
                blk = dict(c1=ConvOp(sd[q + '.conv1.weight'], dt, scale=s1, shift=h1, relu=True),
                           c2=ConvOp(sd[q + '.conv2.weight'], dt, stride=stride, pad=1, scale=s2, shift=h2, relu=True),
                           c3=ConvOp(sd[q + '.conv3.weight'], dt, scale=s3, shift=h3, relu=True), ds=None)
                if (q + '.downsample.0.weight') in sd:
                    sd_, hd_ = bn_fold(sd, q + '.downsample.1')
                    if self.fold_downsample:
                        blk['dual'] = DualConvOp(sd[q + '.conv3.weight'], s3, h3, sd[q + '.downsample.0.weight'], sd_, hd_, stride, dt)
                    else:
                        blk['ds'] = ConvOp(sd[q + '.downsample.0.weight'], dt, stride=stride, scale=sd_, shift=hd_)
                blk['c1'].link_split(blk['c2'])
                if 'dual' not in blk and blk['ds'] is None:
                    blk['c2'].link_split(blk['c3'])
                blocks.append(blk)
            self.layers.append(blocks)
        # layer1 (HBM-bound): blocks without a projection shortcut run conv2 -> conv3 -> (next block's conv1) as one kernel
        if self.bneck_chain:
            l1 = self.layers[0]
            for i, blk in enumerate(l1):
                nxt = l1[i + 1]['c1'] if i + 1 < len(l1) else self.layers[1][0]['c1']      # last block: layer2's first conv1
                if 'dual' in blk:
                    if BneckChainOp.applies(blk['c2'], None, nxt, dt, dual=blk['dual']):
                        blk['chain'] = BneckChainOp(blk['c2'], None, nxt, dual=blk['dual'])
                elif blk['ds'] is None and BneckChainOp.applies(blk['c2'], blk['c3'], nxt, dt):
                    blk['chain'] = BneckChainOp(blk['c2'], blk['c3'], nxt)

        # layer2 / layer3 (HBM- / latency-bound 1x1 convs): identity blocks hand their output to the next block's conv1 on chip
        if self.bneck_tail:
            for li in (1, 2):
                blocks = self.layers[li]
                for i, blk in enumerate(blocks):
                    if 'dual' in blk or blk['ds'] is not None or 'chain' in blk:
                        continue
                    nxt = blocks[i + 1]['c1'] if i + 1 < len(blocks) else self.layers[li + 1][0]['c1']
                    if BneckTailOp.applies(blk['c3'], nxt, dt):
                        blk['tail'] = BneckTailOp(blk['c3'], nxt)

    def __call__(self, img):
        L, dt, dev = _capi.lib(), self.dtype, self.device
        B = img.shape[0]
        if self.fused_stem and dt in HALF:
            x = torch.empty(B, 64, 64, 64, device=dev, dtype=dt)
            u8 = img.dtype == torch.uint8
            _ann('stem', 2.0 * B * 128 * 128 * 64 * 147, img.numel() * img.element_size() + x.numel() * 2 + self.stem_w.numel() * 2,
                 'B=%d 7x7/2 conv + bn + relu + maxpool' % B)
            nb = self.subbatch
            last2 = self.layers[1][-1]
            if nb and nb < B and B % nb == 0 and 'tail' in last2 and not getattr(_TLS, 'capture_fused', None) and not getattr(_TLS, 'no_out_split', False):
                # Depth-first over sub-batches of `nb` images through the high-resolution half (stem -> layer1 -> layer2): the 64 x 64 and 32 x 32 maps
                # of one sub-batch (<= 33 MB each at nb = 16) are written and read back while still in the 256 MB Infinity Cache instead of making
                # HBM round trips of 134 MB per map; layer3 onwards runs on the whole batch.  Same kernels on the same rows: bit-identical.
                c2 = torch.empty(B, 32, 32, 512, device=dev, dtype=dt)
                y1n = torch.empty(B, 32, 32, last2['tail'].c1n.cout, device=dev, dtype=dt)
                _TLS.parent_batch = B
                try:
                    for b0 in range(0, B, nb):
                        isub = img[b0:b0 + nb]
                        xs = x[b0:b0 + nb]
                        _ann('stem', 2.0 * nb * 128 * 128 * 64 * 147, isub.numel() * isub.element_size() + xs.numel() * 2 + self.stem_w.numel() * 2,
                             'B=%d 7x7/2 conv + bn + relu + maxpool' % nb)
                        _capi.check(L.dir_stem_pool_forward_dt(_capi.ptr(isub), 2 if u8 else 0, _dt(dt), IMAGENET_MEAN, IMAGENET_STD, _capi.ptr(self.stem_w),
                                                            _capi.ptr(self.stem_scale), _capi.ptr(self.stem_shift), _capi.ptr(xs), nb, 256, 256,
                                                            _capi.stream_ptr()), 'dir_stem_pool_forward')
                        self._layers(xs, upto=2, out_last=(c2[b0:b0 + nb], y1n[b0:b0 + nb]))
                finally:
                    _TLS.parent_batch = None
                return [None, c2] + self._layers(c2, start=2, y1=y1n)
            _capi.check(L.dir_stem_pool_forward_dt(_capi.ptr(img), 2 if u8 else 0, _dt(dt), IMAGENET_MEAN, IMAGENET_STD, _capi.ptr(self.stem_w),
                                                _capi.ptr(self.stem_scale), _capi.ptr(self.stem_shift), _capi.ptr(x), B, 256, 256,
                                                _capi.stream_ptr()), 'dir_stem_pool_forward')
            return self._layers(x)
        Hs, Ws = 131, 132                                                        # blocks Y, X = 0 .. 130 (+1 column: even rows)
        xp = torch.empty(B, Hs, Ws, 16, device=dev, dtype=dt)
        _ann('stem', 0, img.numel() * img.element_size() + xp.numel() * xp.element_size(), 'B=%d space-to-depth staging' % B)
        if img.dtype == torch.uint8:     # [B,256,256,3] BGR as decoded: normalisation fused into the staging (apps/eval.py:59-61)
            _capi.check(L.dir_stem_prep_s2d_u8(_capi.ptr(img), _capi.ptr(xp), IMAGENET_MEAN, IMAGENET_STD, B, 256, 256, Hs, Ws,
                                               _dt(dt), _capi.stream_ptr()), 'dir_stem_prep_s2d_u8')
        else:
            _capi.check(L.dir_stem_prep_s2d(_capi.ptr(img), _capi.ptr(xp), B, 256, 256, Hs, Ws, _dt(dt), _capi.stream_ptr()),
                        'dir_stem_prep_s2d')
        s1 = self.stem(xp)                                                       # [B,128,128,64]
        x = torch.empty(B, 64, 64, 64, device=dev, dtype=dt)
        _ann('stem', 0, (s1.numel() + x.numel()) * x.element_size(), 'B=%d maxpool 3x3/2' % B)
        _capi.check(L.dir_maxpool3x3s2(_capi.ptr(s1), _capi.ptr(x), B, 128, 128, 64, _dt(dt), _capi.stream_ptr()),
                    'dir_maxpool3x3s2')
        return self._layers(x)

    decimate_c1 = os.environ.get('DIR_DECIMATE_C1', '1') != '0'     # bf16 mode: c1 is not produced unless asked for (taps / backbone_standalone)
    subbatch = int(os.environ.get('DIR_SUBBATCH', '0'))             # bf16 mode: images per depth-first pass of stem + layer1 + layer2 (0 = whole batch)

    def _layers(self, x, start=0, upto=4, y1=None, out_last=None):
        """layers [start, upto) of the pyramid; y1: the first block's conv1 output when the previous layer's last launch already made it;
        out_last: (block output, next conv1 output) destinations of the LAST block's tail launch (sub-batched half)"""
        feats = []
        x_dec = False                                                # x holds only the even pixels of the previous layer's output (decimate_c1)
        for li in range(start, upto):
            blocks = self.layers[li]
            for bi, blk in enumerate(blocks):
                if 'chain' in blk:
                    # the last layer1 block's output is read by layer2's stride-2 projection shortcut only (its conv1 is fused into this launch):
                    # when nobody asks for c1, a quarter of its pixels are written (dir_bneck_chain_params.out_decimate)
                    dec = (self.decimate_c1 and li == 0 and bi == len(blocks) - 1 and blk['chain'].c1n is not None and 'dual' in self.layers[1][0]
                           and self.layers[1][0]['dual'].stride2 == 2 and 'chain' not in self.layers[1][0] and not getattr(_TLS, 'no_out_split', False))
                    x, y1 = blk['chain'](y1 if y1 is not None else blk['c1'](x), x, decimate=dec)
                    x_dec = dec
                    continue
                if 'tail' in blk and (x.shape[0] * x.shape[1] * x.shape[2]) % 64 == 0:
                    if out_last is not None and li == upto - 1 and bi == len(blocks) - 1:
                        x, y1 = blk['tail'](blk['c2'](y1 if y1 is not None else blk['c1'](x)), x, out=out_last[0], y1n=out_last[1])
                        continue
                    x, y1 = blk['tail'](blk['c2'](y1 if y1 is not None else blk['c1'](x)), x)
                elif 'dual' in blk:                                  # conv3 + projection shortcut in one launch
                    x, y1 = blk['dual'](blk['c2'](y1 if y1 is not None else blk['c1'](x)), x, x_decimated=x_dec), None
                    x_dec = False
                else:
                    idn = blk['ds'](x) if blk['ds'] is not None else x
                    x, y1 = blk['c3'](blk['c2'](y1 if y1 is not None else blk['c1'](x)), residual=idn), None
            feats.append(None if x_dec else x)                     # (c1 is not produced when its only reader is the strided shortcut)
        return feats


def backbone_standalone(module, x, compute_dtype=torch.float32):
    """ResNet.forward of the mirror module: NCHW float32 in, [c1..c4] NCHW float32 out."""
    _capi.require_cuda(x)
    if module.training:
        raise NotImplementedError('dir_amd implements the inference path (eval-mode BatchNorm); call .eval()')
    sd = {'b.' + k: v.detach() for k, v in module.state_dict().items()}
    op = BackboneOp(sd, 'b', compute_dtype, x.device)
    op.decimate_c1 = False                                           # ResNet.forward returns c1
    with torch.cuda.device(x.device):
        feats = op(_capi.f32c(x.detach()))
    return [f.permute(0, 3, 1, 2).float() for f in feats]


# ------------------------------------------------------------------------------------------------------------- HRNet-W48 (f4)
def _padc(c):
    """physical channel count of a logical width: the convolution kernels reduce in 64-channel (bf16) / 32-channel (fp32) slabs and the
    HRNet widths 48 and 96 are neither: 48 -> 64, 96 -> 128 (zero weights / zero BatchNorm scale and shift on the padding: exact zeros)"""
    return (c + 63) // 64 * 64


def _pad_conv_bn(sd, conv_key, bn_prefix, dtype, stride=1, relu=True):
    """Conv2d(bias=False) + BatchNorm2d (+ ReLU) with both channel dimensions padded to _padc"""
    w = sd[conv_key]
    co, ci, kh, kw = w.shape
    wp = torch.zeros(_padc(co), _padc(ci), kh, kw, device=w.device, dtype=torch.float32)
    wp[:co, :ci] = w.float()
    sc, sh = bn_fold(sd, bn_prefix)
    scp, shp = torch.zeros(_padc(co), device=w.device), torch.zeros(_padc(co), device=w.device)
    scp[:co], shp[:co] = sc.to(w.device), sh.to(w.device)
    op = ConvOp(wp, dtype, stride=stride, pad=kh // 2, scale=scp, shift=shp, relu=relu)
    op.alg_scale = (co * ci) / float(_padc(co) * _padc(ci))      # share of the launch's MFMA work that multiplies real channels (bench.py: pad_waste)
    return op


class HRNetOp(object):
    """HRNet-W48 pyramid (dir_amd/models/backbone/hrnet.py; SURVEY.md 8f rank 4, no reference counterpart) on the same convolution entry
    points as the ResNet: every Conv + BN (+ ReLU, + residual) is one dir_conv2d_forward (layer1's projection shortcut folded as in
    BackboneOp), a fuse layer's sum is accumulated term by term -- the strided-conv terms through the convolution's residual input, the
    identity / upsampled terms by dir_add_upsampled (nearest x 2^(j-i), the ReLU on the last term).  Returns [c1, c2, c3, c4] NHWC."""
    WIDTHS = (48, 96, 192, 384)
    MODULES = ((2, 1), (3, 4), (4, 3))

    def __init__(self, sd, p, dtype, device):
        dt = self.dtype = dtype
        self.device = device
        # stem conv1 (3x3 / 2 on 3 channels): over the zero-bordered NHWC4 image of dir_stem_prep as a kh = 3, kw = 1 reduction over windows
        # of 16 pixels x 4 channels (weights zero beyond the 3 x 3 real ones): K = 192 instead of 9 x 64
        w = sd[p + '.conv1.weight'].float()
        wp = torch.zeros(64, 64, 3, 1, device=w.device)
        for j in range(3):
            wp[:, 4 * j:4 * j + 3, :, 0] = w[:, :, :, j]
        s1, h1 = bn_fold(sd, p + '.bn1')
        self.conv1 = ConvOp(wp, dt, stride=2, pad=0, scale=s1, shift=h1, relu=True)
        self.conv1.ho = self.conv1.wo = 0          # set per call (H / 2)
        self.conv1.in_cs_override = 4
        self.conv1.alg_k = 27
        s2, h2 = bn_fold(sd, p + '.bn2')
        self.conv2 = ConvOp(sd[p + '.conv2.weight'], dt, stride=2, pad=1, scale=s2, shift=h2, relu=True)
        self.layer1 = []
        for b in range(4):
            q = '%s.layer1.%d' % (p, b)
            sa, ha = bn_fold(sd, q + '.bn1'); sb, hb = bn_fold(sd, q + '.bn2'); sc, hc = bn_fold(sd, q + '.bn3')
            blk = dict(c1=ConvOp(sd[q + '.conv1.weight'], dt, scale=sa, shift=ha, relu=True),
                       c2=ConvOp(sd[q + '.conv2.weight'], dt, pad=1, scale=sb, shift=hb, relu=True))
            if (q + '.downsample.0.weight') in sd:
                sd_, hd_ = bn_fold(sd, q + '.downsample.1')
                blk['dual'] = DualConvOp(sd[q + '.conv3.weight'], sc, hc, sd[q + '.downsample.0.weight'], sd_, hd_, 1, dt)
            else:
                blk['c3'] = ConvOp(sd[q + '.conv3.weight'], dt, scale=sc, shift=hc, relu=True)
            blk['c1'].link_split(blk['c2'])
            if 'c3' in blk:
                blk['c2'].link_split(blk['c3'])
            self.layer1.append(blk)
        self.trans = {(1, 0): _pad_conv_bn(sd, p + '.transition1.0.0.weight', p + '.transition1.0.1', dt),
                      (1, 1): _pad_conv_bn(sd, p + '.transition1.1.0.weight', p + '.transition1.1.1', dt, stride=2),
                      (2, 2): _pad_conv_bn(sd, p + '.transition2.0.weight', p + '.transition2.1', dt, stride=2),
                      (3, 3): _pad_conv_bn(sd, p + '.transition3.0.weight', p + '.transition3.1', dt, stride=2)}
        self.stages = []
        for st, n in self.MODULES:
            mods = []
            for m in range(n):
                q = '%s.stage%d.%d' % (p, st, m)
                branches = [[(_pad_conv_bn(sd, '%s.branches.%d.%d.conv1.weight' % (q, b, k), '%s.branches.%d.%d.bn1' % (q, b, k), dt),
                              _pad_conv_bn(sd, '%s.branches.%d.%d.conv2.weight' % (q, b, k), '%s.branches.%d.%d.bn2' % (q, b, k), dt))
                             for k in range(4)] for b in range(st)]
                for br in branches:
                    for c1, c2 in br:
                        c1.link_split(c2)
                fuse = {}
                for i in range(st):
                    for j in range(st):
                        f = '%s.fuse_layers.%d.%d' % (q, i, j)
                        if j > i:
                            fuse[(i, j)] = [_pad_conv_bn(sd, f + '.0.weight', f + '.1', dt, relu=False)]
                        elif j < i:
                            fuse[(i, j)] = [_pad_conv_bn(sd, '%s.%d.0.weight' % (f, t), '%s.%d.1' % (f, t), dt, stride=2, relu=(t < i - j - 1))
                                            for t in range(i - j)]
                mods.append((branches, fuse))
            self.stages.append(mods)
        self.incre = [_pad_conv_bn(sd, '%s.incre.%d.0.weight' % (p, b), '%s.incre.%d.1' % (p, b), dt) for b in range(4)]

    def _fuse_sum(self, base, srcs, factors, relu, out=None):
        """out = act(base + sum_t nearest_upsample(srcs[t], factors[t])) in one pass (dir_fuse_sum); out None: a new map"""
        import ctypes as C
        B, H, W, Cc = base.shape
        out = torch.empty_like(base) if out is None else out
        n = len(srcs)
        _ann('hr_fuse', 0, (2 * base.numel() + sum(t.numel() for t in srcs)) * base.element_size(), 'sum %dx%dx%d (%d terms)' % (H, W, Cc, n + 1))
        ps = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in srcs])
        fs = (C.c_int * max(n, 1))(*factors)
        _capi.check(_capi.lib().dir_fuse_sum(_capi.ptr(out), _capi.ptr(base), ps, fs, n, B, H, W, Cc, 1 if relu else 0, _dt(self.dtype), _capi.stream_ptr()), 'dir_fuse_sum')
        return out

    def _module(self, xs, mod):
        branches, fuse = mod
        nb = len(xs)
        ys = []
        for b in range(nb):
            y = xs[b]
            for c1, c2 in branches[b]:
                y = c2(c1(y), residual=y)                     # relu(bn2(conv2(.)) + x): the epilogue adds the residual before the ReLU
            ys.append(y)
        outs = []
        for i in range(nb):
            # row i: the strided-conv terms (j < i) accumulate through the last convolution's residual input; the identity term and the
            # upsampled 1x1-conv terms (j > i) are summed with them in ONE pass (round 5: dir_fuse_sum; rounds 3-4 added them one launch
            # and two passes over the row's map at a time, after a copy of the branch output)
            acc = None
            for j in range(i):
                t = ys[j]
                chain = fuse[(i, j)]
                for op in chain[:-1]:
                    t = op(t)
                acc = chain[-1](t) if acc is None else chain[-1](t, residual=acc)
            ups = [fuse[(i, j)][0](ys[j]) for j in range(i + 1, nb)]
            facs = [2 ** (j - i) for j in range(i + 1, nb)]
            if acc is None:
                outs.append(self._fuse_sum(ys[i], ups, facs, True))
            else:
                outs.append(self._fuse_sum(acc, [ys[i]] + ups, [1] + facs, True, out=acc))
        return outs

    def __call__(self, img):
        L, dt, dev = _capi.lib(), self.dtype, self.device
        assert img.dtype == F32, 'the HRNet path takes the normalised float image (NCHW)'
        B, _, H, W = img.shape
        Hp, Wp = H + 2, W + 2
        xp = torch.empty(B, Hp, Wp, 4, device=dev, dtype=dt)
        _ann('stem', 0, img.numel() * 4 + xp.numel() * xp.element_size(), 'B=%d NHWC4 staging' % B)
        _capi.check(L.dir_stem_prep(_capi.ptr(img), _capi.ptr(xp), B, H, W, Hp, Wp, 1, _dt(dt), _capi.stream_ptr()), 'dir_stem_prep')
        self.conv1.ho, self.conv1.wo = H // 2, W // 2
        x = self.conv2(self.conv1(xp))
        for blk in self.layer1:
            y = blk['c2'](blk['c1'](x))
            x = blk['dual'](y, x) if 'dual' in blk else blk['c3'](y, residual=x)
        xs = [self.trans[(1, 0)](x), self.trans[(1, 1)](x)]
        for si, (st, n) in enumerate(self.MODULES):
            if st > 2:
                xs = xs + [self.trans[(st - 1, st - 1)](xs[-1])]
            for mod in self.stages[si]:
                xs = self._module(xs, mod)
        return [self.incre[b](xs[b]) for b in range(4)]


def hrnet_standalone(module, x, compute_dtype=torch.float32):
    """HRNetW48.forward of the mirror module: NCHW float32 in, [c1..c4] NCHW float32 out."""
    _capi.require_cuda(x)
    sd = {'b.' + k: v.detach() for k, v in module.state_dict().items()}      # (.train() with autograd on goes through HRNetW48._train_forward)
    op = HRNetOp(sd, 'b', compute_dtype, x.device)
    with torch.cuda.device(x.device):
        feats = op(_capi.f32c(x.detach()))
    return [f.permute(0, 3, 1, 2).float() for f in feats]


class ResidualOp(object):
    """hourglass.Residual (models/backbone/hourglass.py:33-70), all convs carry a bias"""

    def __init__(self, sd, p, dtype):
        w = lambda k: sd['%s.%s.conv.weight' % (p, k)]  # noqa: E731
        b = lambda k: sd['%s.%s.conv.bias' % (p, k)]  # noqa: E731
        self.need_skip = w('skip_layer').shape[0] != w('skip_layer').shape[1]
        self.skip = ConvOp(w('skip_layer'), dtype, shift=b('skip_layer'))
        s2, h2 = bn_fold(sd, p + '.bn2', b('conv1'))
        self.c1 = ConvOp(w('conv1'), dtype, scale=s2, shift=h2, relu=True, pre=bn_fold(sd, p + '.bn1'), pre_relu=True)
        s3, h3 = bn_fold(sd, p + '.bn3', b('conv2'))
        self.c2 = ConvOp(w('conv2'), dtype, pad=1, scale=s3, shift=h3, relu=True)
        self.c3 = ConvOp(w('conv3'), dtype, shift=b('conv3'))
        # out = conv3(h) + skip_layer(x): the 1x1 skip projection is a second K range of conv3 (dir_conv2d_dual_forward)
        self.dual = None
        if self.need_skip and self.fold_skip:
            one = torch.ones(w('conv3').shape[0], device=w('conv3').device)
            self.dual = DualConvOp(w('conv3'), one, b('conv3'), w('skip_layer'), one, b('skip_layer'), 1, dtype, relu=False)
        self.c1.link_split(self.c2)
        if self.dual is None:
            self.c2.link_split(self.c3)

    fold_skip = os.environ.get('DIR_FOLD_SKIP', '1') != '0'

    def __call__(self, x, out=None, out_coff=0):
        if self.dual is not None:
            return self.dual(self.c2(self.c1(x)), x, out=out, out_coff=out_coff)
        res = self.skip(x) if self.need_skip else x
        return self.c3(self.c2(self.c1(x)), out=out, out_coff=out_coff, residual=res)


class StageOp(object):
    """Joint2BoneFeature (models/dir.py:19-174) packed for one pyramid level"""

    def __init__(self, sd, p, S, distance, dtype, root_joint, keep):
        self.S, self.distance, self.dtype = S, float(distance), dtype
        self.img2joint = (_capi.TokenMlp * 2)(pack_token_mlp(sd, p + '.img2joint_left.filters', keep),
                                              pack_token_mlp(sd, p + '.img2joint_right.filters', keep))
        self.pos_emb = (_capi.TokenMlp * 2)(pack_token_mlp(sd, p + '.pos_emb_left', keep),
                                            pack_token_mlp(sd, p + '.pos_emb_right', keep))
        self.gpos = pack_token_mlp(sd, p + '.global_pos_emb', keep)
        self.gcn = (pack_pgcn(sd, p + '.gcn_left', keep, weight_dtype=_tok_wdt(dtype)), pack_pgcn(sd, p + '.gcn_right', keep, weight_dtype=_tok_wdt(dtype)))
        self.ste = pack_ste(sd, p + '.interaction', keep, weight_dtype=_tok_wdt(dtype))
        R = _capi.RegressParams()
        t = dict(wt=torch.cat([sd[p + '.regressor.mano_left.weight'].float().t(),
                               sd[p + '.regressor.mano_right.weight'].float().t()], 1).contiguous(),     # [1408][128]
                 bl=sd[p + '.regressor.mano_left.bias'].float().contiguous(),
                 br=sd[p + '.regressor.mano_right.bias'].float().contiguous(),
                 wo=sd[p + '.regressor.offset.weight'].float().contiguous(),
                 bo=sd[p + '.regressor.offset.bias'].float().contiguous())
        keep.append(t)
        R.mano_wt, R.mano_b[0], R.mano_b[1] = (t[k].data_ptr() for k in ('wt', 'bl', 'br'))
        R.off_w, R.off_b = t['wo'].data_ptr(), t['bo'].data_ptr()
        R.emb = pack_token_mlp(sd, p + '.proj_feat_emb', keep)
        self.reg = R
        self.mano = (pack_mano(sd, p + '.regressor.mano_layer_left', 'left', root_joint, keep),
                     pack_mano(sd, p + '.regressor.mano_layer_right', 'right', root_joint, keep))
        s, h = bn_fold(sd, p + '.fusion.1', sd[p + '.fusion.0.bias'])
        self.fusion0 = ConvOp(sd[p + '.fusion.0.weight'], dtype, pad=1, scale=s, shift=h, relu=True)
        self.bone_fusion = None
        if dtype in HALF or _packing_arith() is not None:
            # factorised bone fusion (dir_bone_fusion_forward): w_g[tap][hb][c][n] = weight[n, hb*64+c, ky, kx]; bf16 mode: rounded to
            # bf16, bf16 matrix cores; f16 arithmetic modes: unrounded fp32 operands (exact_f32 = 1), on the exact fp32 matrix cores until
            # DirEngine.calibrate has measured G (g_scale = 0), in split precision on the f16 matrix cores afterwards
            exact = dtype == torch.float32
            w = sd[p + '.fusion.0.weight'].detach()
            w = w.float() if exact else w.to(dtype).float()                                   # [256, 2560, 3, 3]  (rounded to the 16-bit storage kind)
            s_f, h_f = bn_fold(sd, p + '.fusion.1', sd[p + '.fusion.0.bias'])                   # (fusion0.scale carries the f16x3 prescale)
            t = dict(w_g=w.reshape(256, 40, 64, 9).permute(3, 1, 2, 0).contiguous(), scale=s_f.to(w.device), shift=h_f.to(w.device))
            keep.append(t)
            self.bone_fusion = _capi.BoneFusionParams(t['w_g'].data_ptr(), t['scale'].data_ptr(), t['shift'].data_ptr(), 1 if exact else 2 if dtype == torch.float16 else 0)
        self.fusion3 = ConvOp(sd[p + '.fusion.3.weight'], dtype, shift=sd[p + '.fusion.3.bias'])


def run_mano_pair(tables_lr, para_l, para_r, B, flags=None, mesh_uv=False):
    """MANO + projection for both hands in one launch, straight out of the 64-wide parameter vectors
    (pose = para[:, :51], betas = para[:, 51:61], cam = para[:, 61:64]; models/dir.py:272-280).  flags: optional int32 [2, B],
    set to 1 where the 6D root rotation has det < 0 (the reference asserts there, rot6d.py:50).  mesh_uv: also return pd_mesh_uv
    [B,778,2] per hand (fourth entry), which only the training loss reads."""
    dev = para_l.device
    out = [[torch.empty(B, 778, 3, device=dev, dtype=F32), torch.empty(B, 21, 3, device=dev, dtype=F32),
            torch.empty(B, 21, 2, device=dev, dtype=F32)] for _ in range(2)]
    if mesh_uv:
        for o in out:
            o.append(torch.empty(B, 778, 2, device=dev, dtype=F32))
    P2 = C.c_void_p * 2
    base = (para_l.data_ptr(), para_r.data_ptr())
    tabs = (_capi.ManoTables * 2)(tables_lr[0], tables_lr[1])
    _ann('mano', 2.0 * B * 1.2e6, 2 * B * (64 + 778 * 3 + 21 * 3 + 21 * 2) * 4 + 2 * (145 * 2336 + 2334 + 778 * 16 + 16 * 33 + 45 * 46) * 4,
         'B=%d x 2 hands (rot6d + PCA + blend shapes + LBS + projection)' % B)
    rc = _capi.lib().dir_mano_forward_pair(
        tabs, P2(base[0], base[1]), 64, P2(base[0] + 51 * 4, base[1] + 51 * 4), 64, P2(base[0] + 61 * 4, base[1] + 61 * 4), 64,
        P2(out[0][0].data_ptr(), out[1][0].data_ptr()), P2(out[0][1].data_ptr(), out[1][1].data_ptr()),
        P2(out[0][2].data_ptr(), out[1][2].data_ptr()), P2(out[0][3].data_ptr(), out[1][3].data_ptr()) if mesh_uv else None,
        None if flags is None else P2(flags[0].data_ptr(), flags[1].data_ptr()), B, _capi.stream_ptr())
    _capi.check(rc, 'dir_mano_forward_pair')
    return out


class DirEngine(object):
    def __init__(self, state_dict, dtype=torch.bfloat16, root_joint=0, device='cuda', sparse_fusion=True, arith=None):
        """dtype bfloat16: the throughput mode (BASELINE config 2); float32: fp32 feature maps, exact fp32 matrix-core arithmetic
        everywhere (the parity mode of rounds 1-2); float32 with arith='f16x3': the same fp32 feature maps and token path, the
        convolutions on the f16 matrix cores in split precision (3 products per multiply, DIR_DT_F16X3) -- meets the same 1e-4 mm
        budget several times faster."""
        assert dtype in (torch.bfloat16, torch.float16, torch.float32)       # float16: f16 STORAGE (round 5), the bf16 data path with 11-bit significands
        assert arith in (None, 'f16x3', 'f16') and (arith is None or dtype == torch.float32)     # 'f16': one f16 MFMA per product (DIR_DT_F16X1)
        self.arith = arith
        self.tuned_batches = set()
        self.sparse_fusion = sparse_fusion     # skip all-zero (tap, bone) K-slabs in the fusion conv (bit-identical)
        _capi.lib()
        self.dtype, self.device = dtype, torch.device(device)
        sd = {k: v.detach().to(self.device) for k, v in state_dict.items()}
        self.keep = []
        self._pgcn_ws = {}
        _TLS.arith = arith
        try:
            self._pack(sd, root_joint)
        finally:
            _TLS.arith = None

    # ------------------------------------------------------------------------------------------ packing
    def _pack(self, sd, root_joint):
        dt, keep = self.dtype, self.keep
        hr = 'backbone.stage4.0.fuse_layers.0.1.0.weight' in sd              # f4: HRNet-W48 instead of ResNet-50 (no reference counterpart)
        self.bb = HRNetOp(sd, 'backbone', dt, self.device) if hr else BackboneOp(sd, 'backbone', dt, self.device)
        # InitRegressor
        p = 'init_regressor'
        H = _capi.InitHeadParams()
        t = {}
        # both attention branches read c4: ONE 3x3 conv with N = 2 x 1024 (left | right), A operand streamed once
        aw, asc, ash = [], [], []
        for i, side in enumerate(('left', 'right')):
            a = '%s.attention_%s' % (p, side)
            s, h = bn_fold(sd, a + '.1', sd[a + '.0.bias'])
            aw.append(sd[a + '.0.weight']); asc.append(s); ash.append(h)
            t['aw%d' % i] = sd[a + '.3.weight'].float().reshape(-1).contiguous()
            H.attn_w[i] = t['aw%d' % i].data_ptr()
            H.attn_b[i] = float(sd[a + '.3.bias'].float().item())
            t['mb%d' % i] = sd['%s.mano_%s.bias' % (p, side)].float().contiguous()
            H.mano_b[i] = t['mb%d' % i].data_ptr()
        self.attn = ConvOp(torch.cat(aw, 0), dt, pad=1, scale=torch.cat(asc), shift=torch.cat(ash), relu=True)
        t['mwt'] = torch.cat([sd[p + '.mano_left.weight'].float().t(), sd[p + '.mano_right.weight'].float().t()],
                             1).contiguous()                               # [2048][128] k-major
        H.mano_wt = t['mwt'].data_ptr()
        t['ow'], t['ob'] = sd[p + '.offset.weight'].float().contiguous(), sd[p + '.offset.bias'].float().contiguous()
        H.off_w, H.off_b = t['ow'].data_ptr(), t['ob'].data_ptr()
        keep.append(t)
        self.init_head = H
        self.init_mano = (pack_mano(sd, p + '.mano_layer_left', 'left', root_joint, keep),
                          pack_mano(sd, p + '.mano_layer_right', 'right', root_joint, keep))
        # decoder
        d = 'decoder'
        self.res = {k: ResidualOp(sd, '%s.%s' % (d, k), dt) for k in (
            'skip_layer4', 'fusion_layer4', 'enhance_layer4', 'skip_layer3', 'fusion_layer3', 'enhance_layer3')}
        self.stage4 = StageOp(sd, d + '.projecter_4', 16, 1, dt, root_joint, keep)
        self.stage3 = StageOp(sd, d + '.projecter_3', 32, 2, dt, root_joint, keep)
        # f4: further refinement iterations at 32x32 (no reference counterpart; dir_amd.models.dir.FusionJointInterIterDecoder extra_stages)
        self.stages_x, self.res_x = [], []
        while ('%s.projecter_x.%d.fusion.0.weight' % (d, len(self.stages_x))) in sd:
            i = len(self.stages_x)
            self.stages_x.append(StageOp(sd, '%s.projecter_x.%d' % (d, i), 32, 2, dt, root_joint, keep))
            self.res_x.append(ResidualOp(sd, '%s.enhance_layer_x.%d' % (d, i), dt))
        s, h = bn_fold(sd, d + '.conv_final.1')
        self.final0 = ConvOp(sd[d + '.conv_final.0.weight'], dt, pad=1, scale=s, shift=h, relu=True)
        self.final3 = ConvOp(sd[d + '.conv_final.3.weight'], dt, shift=sd[d + '.conv_final.3.bias'])
        # seg / dense heads (models/dir.py:425-433): both read `feat`, so their 3x3 convs run as ONE N = 128 + 128 GEMM and their
        # 1x1 -> 3 convs as one block-diagonal N = 6 GEMM over the 256 merged channels (identical arithmetic per output)
        w0, s0, h0, w3, b3 = [], [], [], torch.zeros(6, 256, 1, 1, device=self.device), []
        for i, k in enumerate(('seg', 'dense')):
            s, h = bn_fold(sd, '%s.%s.1' % (d, k), sd['%s.%s.0.bias' % (d, k)])
            w0.append(sd['%s.%s.0.weight' % (d, k)]); s0.append(s); h0.append(h)
            w3[3 * i:3 * i + 3, 128 * i:128 * i + 128] = sd['%s.%s.3.weight' % (d, k)]
            b3.append(sd['%s.%s.3.bias' % (d, k)])
        self.heads0 = ConvOp(torch.cat(w0, 0), dt, pad=1, scale=torch.cat(s0), shift=torch.cat(h0), relu=True)
        self.heads3 = ConvOp(w3, dt, shift=torch.cat(b3), out_dtype=F32)
        self.final0.link_split(self.final3)
        self.final3.link_split(self.heads0)
        self.heads0.link_split(self.heads3)

    # ------------------------------------------------------------------------------------------ pieces
    def init_regressor(self, c4):
        L, dev = _capi.lib(), self.device
        B = c4.shape[0]
        hh = self.attn(c4)                                   # [B,8,8,2048] = (left 1024 | right 1024)
        ch = hh.shape[3] // 2
        esz = hh.element_size()
        para_l = torch.empty(B, 64, device=dev, dtype=F32)
        para_r = torch.empty(B, 64, device=dev, dtype=F32)
        off = torch.empty(B, 3, device=dev, dtype=F32)
        npx = c4.shape[1] * c4.shape[2]
        _ann('init_head', 2.0 * B * (2 * npx * ch + 2 * npx * c4.shape[3] + c4.shape[3] * 128),
             (c4.numel() + hh.numel()) * esz + (c4.shape[3] * 128 + 2 * ch + 131 * 3) * 4 + B * 131 * 4, 'B=%d attention pooling + 3 Linears' % B)
        _capi.check(L.dir_init_head_forward(self.init_head, _capi.ptr(c4), C.c_void_p(hh.data_ptr()),
                                            C.c_void_p(hh.data_ptr() + ch * esz), 2 * ch, _capi.ptr(para_l),
                                            _capi.ptr(para_r), _capi.ptr(off), B, c4.shape[1] * c4.shape[2], c4.shape[3],
                                            ch, _dt(self.dtype), _capi.stream_ptr()), 'dir_init_head_forward')
        return self.mano_outputs(self.init_mano, para_l, para_r, off, self._flags[0] if self._flags is not None else None)

    def mano_outputs(self, tables, para_l, para_r, off, flags=None):
        B = para_l.shape[0]
        (vl, jl, uvl), (vr, jr, uvr) = run_mano_pair(tables, para_l, para_r, B, flags)
        return {'pd_offset': off, 'pd_mano_para_left': para_l, 'pd_mano_para_right': para_r,
                'pd_proj_left': para_l[:, 61:64], 'pd_proj_right': para_r[:, 61:64],
                'pd_mesh_xyz_left': vl, 'pd_mesh_xyz_right': vr, 'pd_joint_xyz_left': jl, 'pd_joint_xyz_right': jr,
                'pd_joint_uv_left': uvl, 'pd_joint_uv_right': uvr, 'pd_rel_joint': None}

    def stage(self, st, feat_buf, feat_cs, prev, img_out, img_coff, want_vis):
        """Joint2BoneFeature.forward.  feat_buf: NHWC buffer whose channels [0,256) are fusion_feat; the stage's
        img_feat is written into img_out[..., img_coff:img_coff+256]."""
        L, dev = _capi.lib(), self.device
        B, S = feat_buf.shape[0], st.S
        sp = _capi.stream_ptr()
        x0 = torch.empty(2, B, 21, 128, device=dev, dtype=F32)
        gp = torch.empty(2, B, 21, 128, device=dev, dtype=F32)
        es = feat_buf.element_size()
        mlp = lambda cin: cin * 128 + 128 * 128          # noqa: E731  (Conv1d cin->128, Conv1d 128->128)
        _ann('grid_tokens', 2.0 * 2 * B * 21 * (mlp(256) + 2 * mlp(3)), 2 * B * 21 * 4 * 256 * es + (2 * mlp(256) + 3 * mlp(3)) * 4 + 4 * B * 21 * 128 * 4,
             'B=%d S=%d bilinear gather + 3 token MLPs, 2 hands' % (B, S))
        _capi.check(L.dir_grid_tokens_forward(
            _capi.ptr(feat_buf), _dt(self.dtype), S, 256, feat_cs, 0, _capi.ptr(prev['pd_joint_uv_left']),
            _capi.ptr(prev['pd_joint_uv_right']), _capi.ptr(prev['pd_joint_xyz_left']),
            _capi.ptr(prev['pd_joint_xyz_right']), _capi.ptr(prev['pd_offset']), st.img2joint, st.pos_emb,
            C.byref(st.gpos), _capi.ptr(x0), _capi.ptr(gp), B, sp), 'dir_grid_tokens_forward')
        tok = torch.empty(B, 42, 128, device=dev, dtype=F32)
        scratch = torch.empty(4, B, 21, 256, device=dev, dtype=F32)
        wes = 2 if self.dtype in HALF else 4
        _ann('pgcn', 2.0 * 4 * 2 * B * 21 * 2 * 128 * 128, 4 * 2 * (2 * 21 * 128 * 128 * wes + 2 * B * 21 * 128 * 4),
             'B=%d 4 layers x 2 hands (per-node W0/W1 %s + neighbour mix + BN + ReLU)' % (B, 'bf16' if wes == 2 else 'f32'))
        if self.pgcn_fused:      # the whole stack of both hands in one launch, layers separated by per-node flags (tokens.hip: pgcn_fused_kernel)
            _capi.check(L.dir_pgcn_stack_forward_fused(st.gcn[0], st.gcn[1], 4, _capi.ptr(x0), _capi.ptr(gp), _capi.ptr(tok), _capi.ptr(scratch),
                                                       _capi.ptr(self._pgcn_sync()), 0, B, sp), 'dir_pgcn_stack_forward_fused')
        else:
            _capi.check(L.dir_pgcn_stack_forward_pair(st.gcn[0], st.gcn[1], 4, _capi.ptr(x0), _capi.ptr(gp), _capi.ptr(tok),
                                                      _capi.ptr(scratch), B, sp), 'dir_pgcn_stack_forward_pair')
        y = torch.empty(B, 42, 64, device=dev, dtype=F32)
        blk = 42 * 128 * 384 + 4 * 2 * 42 * 42 * 32 + 42 * 128 * 128 + 2 * 42 * 128 * 256
        _ann('ste', 2.0 * B * (3 * blk + 42 * 128 * 64), B * 42 * (128 + 64) * 4 + (3 * (128 * 384 + 128 * 128 + 2 * 128 * 256) + 128 * 64) * wes,
             'B=%d 42 tokens x 128, 3 blocks + head' % B)
        _capi.check(L.dir_ste_forward(C.byref(st.ste), _capi.ptr(tok), None, _capi.ptr(y), B, sp), 'dir_ste_forward')
        para_l = torch.empty(B, 64, device=dev, dtype=F32)
        para_r = torch.empty(B, 64, device=dev, dtype=F32)
        off = torch.empty(B, 3, device=dev, dtype=F32)
        emb = torch.empty(B, 42, 64, device=dev, dtype=F32)
        _ann('regress', 2.0 * B * (2 * 1408 * 64 + 2691 * 3 + 42 * 2 * 64 * 64), (2 * 1408 * 64 + 2691 * 3 + 2 * 64 * 64) * 4 + B * (42 * 64 * 2 + 2 * 64 * 2 + 6) * 4,
             'B=%d 2 x Linear 1408->64 + Linear 2691->3 + proj_feat_emb' % B)
        _capi.check(L.dir_regress_forward(C.byref(st.reg), _capi.ptr(y), _capi.ptr(prev['pd_mano_para_left']),
                                          _capi.ptr(prev['pd_mano_para_right']), _capi.ptr(prev['pd_offset']),
                                          _capi.ptr(para_l), _capi.ptr(para_r), _capi.ptr(off), _capi.ptr(emb), B, sp),
                    'dir_regress_forward')
        factorised = st.bone_fusion is not None and self.factorised_fusion
        if factorised:
            # bf16 throughput mode: bone_proj + fusion conv as a K = 720 reduction, no [B,S,S,2560] bone map.  The per-sample
            # G tensors need the token features only: they are computed on the side stream, beside the MANO layer.
            scratch = torch.empty(L.dir_bone_fusion_scratch_bytes(B), device=dev, dtype=torch.uint8)
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _ann('bone_fusion', 2.0 * B * 9 * 80 * 64 * 256, 9 * 40 * 64 * 256 * 4 + B * 42 * 64 * 4 + B * 9 * 80 * 256 * (4 if self.dtype == F32 else 2),
                     'B=%d G = f_end . W (9 taps x 80 bone ends x 256)' % B)
                _capi.check(L.dir_bone_fusion_prepare(st.bone_fusion, _capi.ptr(emb), _capi.ptr(scratch), B, _capi.stream_ptr()),
                            'dir_bone_fusion_prepare')
        res = self.mano_outputs(st.mano, para_l, para_r, off, self._flags[self._stage_index(st)] if self._flags is not None else None)
        vis = torch.empty(B, 1280, S, S, device=dev, dtype=F32) if want_vis else None
        if factorised:
            main.wait_stream(side)
            if self.arith is not None and getattr(_TLS, 'calibrating', False):
                # split-precision fusion (dir_bone_fusion_params.g_scale): the largest |G| of this batch lands in [2^9, 2^10) of the f16 range
                amax = float(scratch.view(F32)[:B * 9 * 40 * 256 * 2].abs().max())
                st.bone_fusion.g_scale = 2.0 ** (10 - math.frexp(amax)[1]) if amax > 0 and math.isfinite(amax) else 1.0
            fused = torch.empty(B, S, S, 256, device=dev, dtype=self.dtype)
            _ann('bone_fusion', 2.0 * B * S * S * 256 * 720, (B * 9 * 80 * 256 + B * S * S * 256) * (4 if self.dtype == F32 else 2) + B * 42 * 2 * 4,
                 'B=%d S=%d factorised bone_proj + 3x3 fusion conv (K=720)' % (B, S))
            _capi.check(L.dir_bone_fusion_forward(st.bone_fusion, _capi.ptr(res['pd_joint_uv_left']),
                                                  _capi.ptr(res['pd_joint_uv_right']), _capi.ptr(scratch),
                                                  _capi.ptr(fused), B, S, st.distance, 256, 0, 1, sp), 'dir_bone_fusion_forward')
            if want_vis:                                        # proj_feat output only (models/dir.py:128,481)
                _ann('proj_feat', 30.0 * B * S * S * 40, B * 1280 * S * S * 4 + B * 42 * 64 * 4, 'B=%d S=%d proj_feat output (fp32 NCHW)' % (B, S))
                _capi.check(L.dir_bone_proj_forward(_capi.ptr(res['pd_joint_uv_left']), _capi.ptr(res['pd_joint_uv_right']),
                                                    _capi.ptr(emb), None, _capi.ptr(vis), None, B, S, st.distance,
                                                    _dt(self.dtype), sp), 'dir_bone_proj_forward')
            st.fusion3(fused, out=img_out, out_coff=img_coff)
        else:
            bone = torch.empty(B, S, S, 2560, device=dev, dtype=self.dtype)
            bbox = torch.empty(B, 40, 4, device=dev, dtype=torch.int32)
            _ann('bone_proj', 30.0 * B * S * S * 40, bone.numel() * bone.element_size() + (B * 1280 * S * S * 4 if want_vis else 0) + B * 42 * 64 * 4,
                 'B=%d S=%d materialised bone map' % (B, S))
            _capi.check(L.dir_bone_proj_forward(_capi.ptr(res['pd_joint_uv_left']), _capi.ptr(res['pd_joint_uv_right']),
                                                _capi.ptr(emb), _capi.ptr(bone), _capi.ptr(vis), _capi.ptr(bbox), B, S,
                                                st.distance, _dt(self.dtype), sp), 'dir_bone_proj_forward')
            st.fusion3(st.fusion0(bone, bbox=bbox if self.sparse_fusion else None), out=img_out, out_coff=img_coff)
        res['joint_feat'] = emb
        res['vis_img_feat'] = vis
        return res

    # The P-GCN stack of both hands as ONE launch (tokens.hip: pgcn_fused_kernel, layers separated by per-node flags): built, bit-identical, and
    # measured no faster than the five launches it replaces (B = 64: 29.9 us with 4 splits against 31.4 us; 38 us with the 2 splits that keep six
    # concurrent launches co-resident) -- a device-coherent hand-off costs what a kernel boundary costs.  Off unless DIR_PGCN_FUSED=1.
    pgcn_fused = os.environ.get('DIR_PGCN_FUSED', '0') == '1'

    def _pgcn_sync(self):
        """the fused P-GCN launch's flag words: zeroed once, one buffer per stream that runs forwards of this engine (two streams' launches must
        not share flags); never reset afterwards -- every launch raises its flags by one (tokens.hip).  Allocated outside any capture when the
        stream has run an eager forward first (ForwardPipeline does); allocated inside a capture it is re-zeroed by every replay, which is
        equally correct."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._pgcn_ws.get(key)
        if ws is None:
            ws = torch.zeros(int(_capi.lib().dir_pgcn_fused_sync_bytes()) // 4, dtype=torch.int32, device=self.device)
            self._pgcn_ws[key] = ws
        return ws

    def pgcn_sync_error(self):
        """True if a fused P-GCN workgroup of any stream ever gave up waiting for a neighbour (host read; tests and bench.py look)"""
        return any(int(ws[-4].item()) != 0 for ws in self._pgcn_ws.values())

    def _stage_index(self, st):
        """position of the stage's dict in outs_list (0 = the init regression)"""
        return 1 if st is self.stage4 else 2 if st is self.stage3 else 3 + self.stages_x.index(st)

    def upsample_into(self, x, out, coff):
        B, H, W, Cc = x.shape
        _ann('upsample', 0, 5 * x.numel() * x.element_size(), 'B=%d %dx%dx%d -> 2x' % (B, H, W, Cc))
        _capi.check(_capi.lib().dir_upsample2x_bilinear(_capi.ptr(x), _capi.ptr(out), B, H, W, Cc, out.shape[3], coff,
                                                        _dt(self.dtype), _capi.stream_ptr()), 'dir_upsample2x_bilinear')

    factorised_fusion = os.environ.get('DIR_FACTORISED_FUSION', '1') != '0'    # bf16 mode: dir_bone_fusion_forward
    # Side-stream fork / join inside one forward (skip branches + the fusion's G tensors beside the token path): OFF by default.
    # It is how the packed-FP32 hazard was found ("Packed FP32 beside another kernel"; fixed at build level), and with
    # two whole forwards in flight (ForwardPipeline) it no longer pays: 2.44 ms per forward without it against 2.51 ms with it.
    overlap = os.environ.get('DIR_OVERLAP', '0') == '1'

    def _side_stream(self):
        if not self.overlap:
            return torch.cuda.current_stream()
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # kernel variants a layer can be forced to (include/dir_hip.h: DIR_CONV_VARIANT); 0 = the library's heuristic.  (Code 19, the
    # 64x128 tile on the 3-buffer ring, is not offered: see conv.hip, DIR_RING_64x128.)
    # All of them -- including the streaming 1x1 kernel (STREAM_VARIANT, stream.hip), which feeds the MFMA the same k-slots in the
    # same order as the tiled kernels -- accumulate identically: outputs are bit-identical whichever is chosen
    # (tools/check_stream_layers.py, tests/test_gpu_dir.py::test_autotuned_engine_is_bit_identical).
    # (STREAMP_VARIANT = 24, the pipelined streaming kernel of round 5, is built, bit-identical and tested but NOT offered: measured slower than 21 / 22 on
    #  every layer it could serve -- one producer wave cannot issue 16 KB of LDS-DMA per 0.25 us of MFMAs; tools/bench_stream.py)
    CONV_VARIANTS = (0, 1, 2, 3, 4, 17, 18, 20, 8, 9, 10, 11, 12, 13, 14, 15, STREAM_VARIANT, STREAM64_VARIANT, STREAM32_VARIANT, 25, 26, 27, 28)

    def _profiled_forwards(self, img, n):
        """n eager forwards with every library call timed; returns the records of the conv family (those that carry an `op`)"""
        _capi.PROFILE = []
        try:
            for _ in range(n):
                self.forward(img)
            torch.cuda.synchronize()
            return [r for r in _capi.PROFILE if r.get('op') is not None]
        finally:
            _capi.PROFILE = None

    # margin by which an 8-wave one-workgroup-per-CU variant (pipe / patch) must beat the best 4-wave variant in isolation (A/B aid)
    PIPE_MARGIN = float(os.environ.get('DIR_TUNE_PIPE_MARGIN', '0.03'))
    TUNE_EXCLUDE = tuple(int(v) for v in os.environ.get('DIR_TUNE_EXCLUDE', '').split(',') if v)   # variants autotune skips (A/B aid)
    STREAM_MARGIN = float(os.environ.get('DIR_TUNE_STREAM_MARGIN', '0.03'))     # same for the streaming 1x1 kernel (negative: preferred even when slower alone)

    def autotune(self, img, reps=2):
        """Pick the convolution kernel variant per layer for this batch size by timing every candidate inside real
        forwards (HIP events around each conv launch, side stream off, realistic cache state).  All variants accumulate in
        the same order, so the outputs are bit-identical whatever is chosen; a variant that does not apply to a layer falls
        back to the heuristic inside the library.  ~14 x (reps+1) eager forwards, once per (engine, batch size)."""
        B = img.shape[0]
        saved_overlap, saved_profile = self.overlap, _capi.PROFILE
        self.overlap = False
        best = {}
        try:
            for v in self.CONV_VARIANTS:
                if v in self.TUNE_EXCLUDE:
                    continue
                _TLS.variant = v
                self._profiled_forwards(img, 1)                    # warm-up (allocator, instruction cache)
                acc = {}
                for rec in self._profiled_forwards(img, reps):
                    acc.setdefault(rec['op'], []).append(rec['e0'].elapsed_time(rec['e1']))
                for op, ts in acc.items():
                    t = min(ts)
                    margin = self.PIPE_MARGIN if v in (8, 9, 10, 11, 12, 13, 14, 15) else self.STREAM_MARGIN if v in STREAM_VARIANTS else 0.03
                    if op not in best or t < best[op][0] * (1.0 - margin):  # a challenger must win by 3 % (timing noise)
                        best[op] = (t, v)
        finally:
            _TLS.variant = None
            _capi.PROFILE = saved_profile
            self.overlap = saved_overlap
        for op, (t, v) in best.items():
            op.variant[B] = v
        self.tuned_batches.add(B)
        self._tuned_order = getattr(self, '_tuned_order', {})
        self._tuned_order[B] = [op for op in best]          # first-call order of one forward: stable for a given engine
        return {op: v for op, (t, v) in best.items()}

    def autotune_energy(self, img, seconds=None, slack=2.6, idle_w=None, log=None, max_calls=None, min_saving=0.0, near=1.12):
        """The per-layer kernel choice for THROUGHPUT with several forwards in flight.  Four bs-64 forwards in flight run the socket at its
        power cap (1.3-1.4 kW of 1.4 kW, 2.9 J per forward), so what raises images/s is the variant that costs the fewest joules
        above idle, not the one that finishes first alone: typically a larger tile on fewer CUs (less L2 -> LDS and LDS -> register traffic
        per MFMA), the idle CUs being filled by the other forwards.  Every convolution call of one forward is captured and replayed back to
        back on its real tensors, per variant, for `seconds`, and its energy is read from the socket's energy accumulator (amdsmi: exact joules
        over the window; default 0.2 s) or, where that is not available, from rocm-smi's averaged power sampled over the last 40 % of a longer
        window (default 0.8 s: the reading lags by about a second); the choice minimises joules above idle per launch among the variants within
        `slack` x the fastest.  ~50 calls x ~10 variants x seconds: 2 minutes with the counter, 5-8 without -- run it
        once per (GPU model, batch size) and keep export_tuning()'s table (dir_amd/tuning/, load_tuning_table).  Results stay bit-identical
        (same argument as autotune).  One forward alone gets ~15 % slower with this table: latency-bound callers keep autotune().
        min_saving (round 6): a variant slower than `near` x the fastest is taken only if it saves at least this fraction of the joules of the best
        variant that IS within `near` x the fastest.  With 0 (rounds 3-5) the table held a dozen launches that were 1.4 - 2.5x slower one at a time for
        1 - 5 % fewer joules -- inside the 0.2 s windows' noise -- which cost 0.17 ms of one-forward latency and a tenth of the per-launch roofline
        fraction for nothing measurable at four in flight (profiles/r06_energy_rule_ab.txt)."""
        import time as _time
        from . import power
        B = img.shape[0]
        if B not in getattr(self, '_tuned_order', {}):
            self.autotune(img)                                         # the time-tuned choice first: op order, and the fallback for every op not replayed here
        counter = power.energy_joules() is not None
        if not counter and power.smi_sample() is None:
            raise RuntimeError('autotune_energy: neither the amdsmi energy counter nor rocm-smi gives a reading on this machine')
        if seconds is None:
            seconds = 0.2 if counter else 0.8
        saved_overlap, self.overlap = self.overlap, False
        torch.cuda.synchronize(self.device)
        if idle_w is None:
            idle_w = power.IDLE_W       # (the reading is a moving average with a time constant near a second: an "idle" sample taken here would still carry load)
        rows = []
        try:
            _TLS.capture = []
            self.forward(img)
            torch.cuda.synchronize(self.device)
            calls, _TLS.capture = _TLS.capture, None
            seen = set()
            for op, args, kw in (calls if max_calls is None else calls[:max_calls]):      # (max_calls: tests rate the first few calls only)
                if id(op) in seen:                                     # an op called twice per forward keeps one choice: the first call's
                    continue
                seen.add(id(op))
                times = {}
                for v in self.CONV_VARIANTS:
                    if v in self.TUNE_EXCLUDE:
                        continue
                    _TLS.variant = v
                    for _ in range(3):
                        op(*args, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        op(*args, **kw)
                    e1.record()
                    torch.cuda.synchronize(self.device)
                    times[v] = e0.elapsed_time(e1) / 10 * 1e3
                tb = min(times.values())
                row = {}
                for v, us in times.items():
                    if us > slack * tb:
                        continue
                    _TLS.variant = v
                    burst = max(10, min(100, int(0.02 / (us * 1e-6))))                  # ~20 ms of launches between host synchronisations
                    if counter:
                        for _ in range(burst):                                          # the loop is already running when the counter is read
                            op(*args, **kw)
                        torch.cuda.synchronize(self.device)
                        e0 = power.energy_joules()
                    else:
                        smp = power.Sampler(skip=0.6 * seconds, period=0.03).start()    # the first 60 % still carries the previous variant's level
                    t0, n = _time.perf_counter(), 0
                    while _time.perf_counter() - t0 < seconds:
                        for _ in range(burst):
                            op(*args, **kw)
                        torch.cuda.synchronize(self.device)
                        n += burst
                    dt = _time.perf_counter() - t0
                    if counter:
                        e1 = power.energy_joules()
                        w = (e1[0] - e0[0]) / dt if e0 and e1 else float('nan')
                    else:
                        w = power.median(smp.stop(), 'w')
                    if w == w:                                         # (NaN: no reading in the window -- variant not rated)
                        row[v] = (dt / n * 1e6, w)
                _TLS.variant = None
                if not row:
                    continue
                joules = lambda v: row[v][0] * max(row[v][1] - idle_w, 1.0)  # noqa: E731
                fastest = min(row, key=lambda v: row[v][0])
                base = min((v for v in row if row[v][0] <= near * row[fastest][0]), key=joules)
                best = min(row, key=joules)
                if joules(best) > (1.0 - min_saving) * joules(base):
                    best = base
                op.variant[B] = best
                rows.append(dict(cout=op.cout, cin=getattr(op, 'cin', 0), kh=getattr(op, 'kh', 1), stride=getattr(op, 'stride', 1), chosen=best,
                                 us=round(row[best][0], 1), w=round(row[best][1]), fastest=fastest, fastest_us=round(row[fastest][0], 1),
                                 fastest_w=round(row[fastest][1])))
                if log is not None:
                    log(rows[-1])
        finally:
            _TLS.capture, _TLS.variant = None, None
            self.overlap = saved_overlap
        return {'idle_w': idle_w, 'layers': rows, 'instrument': 'amdsmi energy accumulator, %.2f s windows' % seconds if counter else 'rocm-smi power, last 40 %% of %.2f s windows' % seconds}

    TUNING_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuning')

    def load_tuning_table(self, img, name):
        """Apply dir_amd/tuning/<name>.json (written by tools/energy_tune.py from export_tuning) if it was made for this batch size and an
        identically built engine; returns the table's meta dict, or None when there is no such table / it does not match (the caller then
        keeps whatever autotune chose)."""
        path = os.path.join(self.TUNING_DIR, name + '.json')
        if not os.path.exists(path):
            return None
        with open(path) as f:
            t = json.load(f)
        if t.get('batch') != img.shape[0] or any(int(r[5]) not in self.CONV_VARIANTS for r in t['table']):
            return None
        try:
            self.import_tuning(img, t['table'])
        except ValueError:
            return None
        return t.get('meta', {})

    def export_tuning(self, B):
        """the variants autotune chose for batch size B, in the order the conv layers run (JSON-serialisable)"""
        return [[op.cout, op.cin, op.kh, op.kw, op.stride, op.variant.get(B, 0)] for op in self._tuned_order[B]]

    def import_tuning(self, img, table):
        """apply a table produced by export_tuning on an identically built engine (same layer order, checked by shape)"""
        B = img.shape[0]
        saved_overlap, saved_profile, self.overlap = self.overlap, _capi.PROFILE, False
        try:
            ops = []
            for rec in self._profiled_forwards(img, 1):
                if rec['op'] not in ops:
                    ops.append(rec['op'])
        finally:
            _capi.PROFILE, self.overlap = saved_profile, saved_overlap
        if len(ops) != len(table) or any([op.cout, op.cin, op.kh, op.kw, op.stride] != row[:5] for op, row in zip(ops, table)):
            raise ValueError('tuning table does not match this engine')
        for op, row in zip(ops, table):
            op.variant[B] = int(row[5])
        self.tuned_batches.add(B)
        self._tuned_order = getattr(self, '_tuned_order', {})
        self._tuned_order[B] = ops

    pending_tuning = None          # {batch size: export_tuning table} handed over by DIR.engine() when the weights were re-packed

    def export_all_tuning(self):
        return {B: self.export_tuning(B) for B in getattr(self, '_tuned_order', {})}

    def tune_for(self, img):
        """Make sure batch size img.shape[0] has a kernel choice: a table handed over from the previous engine, else the variants of
        the nearest batch size already tuned (the ragged last batch of an evaluation loader must not cost 45 forwards), else a
        timed autotune."""
        B = img.shape[0]
        if B in self.tuned_batches:
            return
        if self.pending_tuning and B in self.pending_tuning:
            try:
                self.import_tuning(img, self.pending_tuning[B])
                return
            except ValueError:
                pass
        order = getattr(self, '_tuned_order', {})
        if order:
            self.copy_tuning(min(order, key=lambda b: abs(b - B)), B)
        else:
            self.autotune(img)

    def copy_tuning(self, B_from, B_to):
        """reuse the variants tuned for batch size B_from at batch size B_to (e.g. the ragged last batch of an evaluation loader:
        the choice is bit-identical by construction, so only speed is at stake)"""
        for op in self._tuned_order.get(B_from, []):
            op.variant[B_to] = op.variant.get(B_from, 0)
        self._tuned_order[B_to] = self._tuned_order[B_from]
        self.tuned_batches.add(B_to)

    # ------------------------------------------------------------------------------------------ f16x3 calibration
    calibrated = False

    def calibrate(self, img):
        """arith='f16x3' only: one eager forward on `img` during which every convolution records the largest |activation| it reads and
        sets its power-of-two input scale so that this maximum lands in [2^9, 2^10) of the f16 range (include/dir_hip.h: in_scale): 64x
        headroom before values saturate at 65504, full 22-bit operand accuracy down to max / 4096.  Each layer is re-scaled BEFORE it
        runs, so the activations further down are already those of the calibrated network.  Host synchronisations: not capturable; call
        it once per engine on a representative batch (DIR.forward and bench.py do, on the first batch they see).  Without it in_scale is
        1 everywhere: correct for activations inside [2^-3, 65504), saturating beyond."""
        if self.arith is None:
            return
        _TLS.calibrating = True
        try:
            self.forward(img)
            torch.cuda.synchronize(self.device)
        finally:
            _TLS.calibrating = False
        self.calibrated = True

    # ------------------------------------------------------------------------------------------ forward
    _flags = None

    def forward(self, img, want_proj_feat=True, taps=None, reflection_flags=None):
        """img: float32 NCHW [B,3,256,256] on the GPU.  Returns outs_list exactly like DIR.forward (models/dir.py:521-540);
        tensors are engine-owned buffers (valid until the next forward when run under a captured graph).
        reflection_flags: optional int32 tensor [3 (+ extra) stages, 2 hands, B]; an entry is set to 1 where the predicted 6D root rotation
        is a reflection (det < 0) -- where the reference's `assert` fires (rot6d.py:50).  The caller decides when to look (a host
        read); dir_amd.models.dir.DIR.forward does, and raises AssertionError like the reference."""
        _capi.require_cuda(img)
        self._flags = reflection_flags
        _TLS.no_out_split = taps is not None          # the taps are read as fp32 maps
        try:
            return self._forward(img, want_proj_feat, taps)
        finally:
            self._flags = None
            _TLS.no_out_split = False

    def _forward(self, img, want_proj_feat, taps):
        if img.dtype == torch.uint8:     # decoded BGR frames [B,256,256,3]: the reference's normalisation runs inside the stem staging
            assert img.is_contiguous() and img.shape[1:] == (256, 256, 3)
        else:
            assert img.dtype == F32 and img.is_contiguous() and img.shape[1:] == (3, 256, 256)
        dt, dev = self.dtype, self.device
        B = img.shape[0]
        feats = self.bb(img)
        c1, c2, c3, c4 = feats
        cat4 = torch.empty(B, 16, 16, 2304, device=dev, dtype=dt)
        cat3 = torch.empty(B, 32, 32, 512, device=dev, dtype=dt)
        # The two skip branches depend on the backbone only.  With `overlap` (off by default, see the class attribute) they run on
        # a side stream: skip_layer4 beside init_head / MANO, skip_layer3 beside stage 1's token path.  Otherwise `side` is the
        # current stream and the waits below are no-ops.
        main = torch.cuda.current_stream()
        side = self._side_stream()
        ev_tok = torch.cuda.Event()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.res['skip_layer4'](c3, out=cat4, out_coff=2048)
            ev_s4 = torch.cuda.Event()
            ev_s4.record()
        init = self.init_regressor(c4)
        # ---- stage 1 @16x16 (models/dir.py:442-456)
        self.upsample_into(c4, cat4, 0)
        main.wait_event(ev_s4)
        enh4_in = torch.empty(B, 16, 16, 512, device=dev, dtype=dt)             # cat(fusion_feat, img_feat)
        self.res['fusion_layer4'](cat4, out=enh4_in, out_coff=0)
        ev_tok.record()
        with torch.cuda.stream(side):
            side.wait_event(ev_tok)
            self.res['skip_layer3'](c2, out=cat3, out_coff=256)
        r4 = self.stage(self.stage4, enh4_in, 512, init, enh4_in, 256, False)
        e4 = self.res['enhance_layer4'](enh4_in)
        # ---- stage 2 @32x32 (models/dir.py:459-471)
        self.upsample_into(e4, cat3, 0)
        main.wait_stream(side)
        enh3_in = torch.empty(B, 32, 32, 512, device=dev, dtype=dt)
        self.res['fusion_layer3'](cat3, out=enh3_in, out_coff=0)
        nx = len(self.stages_x)
        r3 = self.stage(self.stage3, enh3_in, 512, r4, enh3_in, 256, want_proj_feat and nx == 0)
        # f4: every further iteration reads the running 32x32 map from channels [0, 256) of its own cat buffer (written there by the
        # previous enhance Residual) and appends its img_feat to channels [256, 512)
        extra, buf, prev = [], enh3_in, r3
        for i in range(nx):
            nxt = torch.empty(B, 32, 32, 512, device=dev, dtype=dt)
            (self.res['enhance_layer3'] if i == 0 else self.res_x[i - 1])(buf, out=nxt, out_coff=0)
            prev = self.stage(self.stages_x[i], nxt, 512, prev, nxt, 256, want_proj_feat and i == nx - 1)
            extra.append(prev)
            buf = nxt
        e3 = (self.res['enhance_layer3'] if nx == 0 else self.res_x[nx - 1])(buf)
        # ---- heads (models/dir.py:474-476)
        feat = self.final3(self.final0(e3))
        sd6 = self.heads3(self.heads0(feat))                                     # NHWC fp32 [B,32,32,6] = seg | dense
        seg, dense = sd6[..., :3], sd6[..., 3:]
        if taps is not None:
            taps.update(c1=c1, c2=c2, c3=c3, c4=c4, fusion4=enh4_in[..., :256], proj4=enh4_in[..., 256:], enh4=e4,
                        fusion3=enh3_in[..., :256], proj3=enh3_in[..., 256:], enh3=e3, final=feat,
                        skip4=cat4[..., 2048:])
        outs = []
        for o in [init, r4, r3] + extra:
            outs.append({k: o[k] for k in ('pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_xyz_left',
                                           'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right',
                                           'pd_proj_left', 'pd_proj_right', 'pd_offset', 'pd_rel_joint')})
        outs.append({'dense': dense.permute(0, 3, 1, 2), 'seg': seg.permute(0, 3, 1, 2),
                     'proj_feat': (extra[-1] if extra else r3)['vis_img_feat']})
        return outs


class ForwardPipeline(object):
    """Several forwards in flight: one captured HIP graph of `DirEngine.forward` per slot, each with its own stream, input
    tensor and activation / output buffers (the weights are the engine's, shared).  Images are independent
    (models/dir.py:513-540 has no cross-sample op in eval mode), so two batches can overlap freely: the low-occupancy token
    kernels (64-336 workgroups) and every kernel's ramp / drain of one forward run under the convolutions of the other.
    Measured at B = 64, bf16: 2.93 ms per forward with one in flight, 2.44 ms with two (three: slower -- cache and LDS
    contention), `tools/two_stream_test.py`.

        pipe = ForwardPipeline(eng, [img_a, img_b])      # static input tensor per slot
        pipe.refill(0, batch0); pipe.launch(0); pipe.refill(1, batch1); pipe.launch(1)
        outs = pipe.wait(0)                               # host wait; valid until slot 0 is launched again

    Results are bit-identical to `eng.forward(img)`: the same kernels on the same inputs, only scheduled side by side."""

    def __init__(self, eng, imgs, want_proj_feat=True, streams=None):
        """streams: optional list of torch streams to run the slots on (one per slot).  HIP multiplexes streams onto a handful of
        hardware queues (GPU_MAX_HW_QUEUES, default 4 including the null stream's: set it to 8 before the runtime starts for four
        slots, as bench.py does); two slots whose streams share a queue do not overlap at all, so a second pipeline on the same engine
        should re-use the first one's streams (bench.py's proj_feat-less variant does)."""
        assert len(imgs) >= 1 and (streams is None or len(streams) == len(imgs))
        self.eng, self.imgs = eng, list(imgs)
        self.streams, self.graphs, self.outs, self.done = [], [], [], []
        cur = torch.cuda.current_stream()
        for i, img in enumerate(self.imgs):
            s = streams[i] if streams is not None else torch.cuda.Stream(device=eng.device)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                eng.forward(img, want_proj_feat)          # eager once on this stream: allocator warm-up before the capture
                s.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    o = eng.forward(img, want_proj_feat)
            self.streams.append(s); self.graphs.append(g); self.outs.append(o)
            self.done.append(torch.cuda.Event())
        torch.cuda.synchronize(eng.device)

    def __len__(self):
        return len(self.graphs)

    def refill(self, slot, batch):
        """Copies `batch` into the slot's input ON THE SLOT'S STREAM (ordered before its next launch by stream order alone).
        `batch` must be complete from the host's point of view (pinned host memory, or device memory produced before a host
        synchronise): every hand-over of the pipeline is stream order or a host wait, no cross-stream event edges."""
        with torch.cuda.stream(self.streams[slot]):
            self.imgs[slot].copy_(batch, non_blocking=True)

    def launch(self, slot):
        """Replays slot's forward on its stream (after whatever `refill` queued there)."""
        s = self.streams[slot]
        with torch.cuda.stream(s):
            self.graphs[slot].replay()
            self.done[slot].record(s)

    def wait(self, slot):
        """Host wait for slot's forward; returns its outs_list (valid until the slot is launched again)."""
        self.done[slot].synchronize()
        return self.outs[slot]
