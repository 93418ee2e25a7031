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
PREPACK = os.environ.get('DIR_TRAIN_PREPACK', '1') == '1'          # WeightPack below (0: pack per convolution call, rounds 2-3)
# round 5: conv_fwd(stats=[]) asks the convolution's epilogue for the chunk partials of the BatchNorm that follows it (dir_conv2d_forward_stats):
# that BatchNorm's statistics pass over the stored map is not run.  DIR_TRAIN_STATS_IN_EPILOGUE=0: the separate statistics kernel (rounds 2-4).
STATS_IN_EPILOGUE = os.environ.get('DIR_TRAIN_STATS_IN_EPILOGUE', '1') == '1'
# ... and conv_dgrad(bn_bwd=...) for the sums of a BatchNorm's BACKWARD pass from the data-gradient convolution that writes the gradient of its output
# (dir_conv2d_forward_ex).  DIR_TRAIN_BN_BWD_IN_EPILOGUE=0: the separate pass over (gradient, input).
BN_BWD_IN_EPILOGUE = os.environ.get('DIR_TRAIN_BN_BWD_IN_EPILOGUE', '1') == '1'
HEADROOM = 64.0             # pow2_in_scale puts the largest |operand| in [2^9, 2^10): 64x below the f16 maximum
# (A step's backward must follow its own forward before another model's forward starts: the cache bound by begin_step stays active until
# the next begin_step.)
# One cache per trained model, held BY THE CALLER'S OWNER OBJECT (the optimizer in dir_amd.train.step.train_step, the nn.Module in
# dir_amd.models.dir.DIR): it lives exactly as long as that object.  Rounds 2-3 keyed a module-level table by the address of the model's first
# parameter -- an address the caching allocator hands to the NEXT model once the first one is freed, which then inherited scales measured on
# other weights (found by the round-4 frozen-BatchNorm gradient gate: 1.7e-2 instead of 6e-4 when other tests had run before it).  Without
# an owner (owner=None: direct calls of dir_amd.train.net.forward) nothing is cached: every convolution measures its scale on what it is given.
_scales, _state = [], {'call': 0, 'step': 0, 'cached': False}

# Weight gradients on a SIDE STREAM (round 4, measured, OFF by default: DIR_TRAIN_SIDE_WGRAD=1 switches it on).  Nothing in the backward pass
# waits for a convolution's weight gradient -- only the optimiser does -- so inside dir_amd.train.net.backward (side_begin .. side_end) conv_bwd
# can issue it (and its OIHW copy, and the gradient moves into the flat bucket) on a second stream behind an event of the compute stream,
# which waits once at the end.  Same kernels on the same operands: bit-identical (the whole-step tests pass with it on).  It buys NOTHING at
# 32 images: 0.032 s per step with and without at equal stream priority -- the two streams share the compute units evenly while the data
# gradients run and the side stream is empty again by the time the joint-token path (the under-occupied part) starts; at the device's
# lowest stream priority (hipStreamCreateWithPriority) the step is SLOWER and erratic, 0.033-0.039 s.  Off while a HIP graph is captured.
SIDE_WGRAD = os.environ.get('DIR_TRAIN_SIDE_WGRAD', '0') == '1'
SIDE_IN_CAPTURE = os.environ.get('DIR_TRAIN_SIDE_IN_CAPTURE', '0') == '1'      # experiment: the side stream as a parallel branch of the captured graph
_side = {'streams': {}, 'active': None}


def side_begin():
    """-> the side stream for this backward pass, or None (switched off / graph capture)"""
    _side['active'] = None
    if not SIDE_WGRAD or not torch.cuda.is_available() or (torch.cuda.is_current_stream_capturing() and not SIDE_IN_CAPTURE):
        return None
    dev = torch.cuda.current_device()
    st = _side['streams'].get(dev)
    if st is None:
        st = _side['streams'][dev] = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream())
    _side['active'] = st
    return st


def side_run(fn, *reads):
    """fn() on the side stream once everything the compute stream has issued so far is done; `reads`: tensors of the compute stream's
    allocator that fn reads (kept from re-use until the side stream is through with them)"""
    st = _side['active']
    if st is None:
        return fn()
    st.wait_event(torch.cuda.current_stream().record_event())
    with torch.cuda.stream(st):
        out = fn()
    for t in reads:
        if t is not None:
            t.record_stream(st)
    return out


def side_end():
    st, _side['active'] = _side['active'], None
    if st is not None:
        torch.cuda.current_stream().wait_stream(st)


class _Packed:
    """one convolution weight's two packed operand forms inside a WeightPack (views of its buffers)"""
    __slots__ = ('shape', 'fwd', 'fwd_scale', 'dgrad', 'dgrad_scale', 'dshape', 'applied', 'want', 'index')


class WeightPack:
    """All convolution weights of a model in both DIR_DT_F16X3 operand forms, refreshed by ONE launch per step (dir_train_pack_conv_weights)
    instead of ~600 (per convolution call: layout copy, pack, scale division; flip, transpose, pad, pack, division for the data gradient).
    weights: the OIHW fp32 parameter tensors (their storage must stay where it is: FlatAdamW's views of the flat buffer do).  Each operand
    form's epilogue scale carries 1 / in_scale of the call site that used it last step (`applied`); a site that asks for another in_scale
    gets the ratio applied by one torch multiply and the table follows at the next refresh."""

    def __init__(self, weights):
        import ctypes as C
        dev = weights[0].device
        self.ptrs = tuple(w.data_ptr() for w in weights)
        self.weights = list(weights)
        n16 = nsc = 0
        plan = []
        for w in weights:
            Cout, Cin, kh, kw = w.shape
            T, Co32 = kh * kw, (Cout + 31) // 32 * 32
            has_f = Cin % 32 == 0
            plan.append((has_f, n16, nsc))
            n16 += (Cout * T * Cin * 2 if has_f else 0) + Cin * T * Co32 * 2
            nsc += (Cout if has_f else 0) + Cin
        self.buf16 = torch.empty(n16, device=dev, dtype=torch.float16)
        self.bufsc = torch.empty(nsc, device=dev, dtype=torch.float32)
        self.entries, rows = [], [0]
        for i, (w, (has_f, o16, osc)) in enumerate(zip(weights, plan)):
            Cout, Cin, kh, kw = w.shape
            T, Co32 = kh * kw, (Cout + 31) // 32 * 32
            e = _Packed()
            e.index, e.shape, e.dshape = i, (Cout, kh, kw, Cin), (Cin, kh, kw, Co32)
            e.fwd = e.fwd_scale = None
            if has_f:
                e.fwd = self.buf16[o16:o16 + Cout * T * Cin * 2].view(Cout, T * Cin // 32, 2, 32)
                e.fwd_scale = self.bufsc[osc:osc + Cout]
                o16, osc = o16 + Cout * T * Cin * 2, osc + Cout
            e.dgrad = self.buf16[o16:o16 + Cin * T * Co32 * 2].view(Cin, T * Co32 // 32, 2, 32)
            e.dgrad_scale = self.bufsc[osc:osc + Cin]
            e.applied, e.want = [1.0, 1.0], [None, None]          # in_scale folded into (forward, data-gradient) scales / asked for this step
            self.entries.append(e)
            rows.append(rows[-1] + (Cout if has_f else 0) + Cin // 4)          # workgroups: one per forward row, one per four data-gradient rows
        self.by_ptr = {p: e for p, e in zip(self.ptrs, self.entries)}
        self.total_rows = rows[-1]
        self.row_start = torch.tensor(rows, dtype=torch.int32).to(dev)
        self._ctable = (_capi.TrainWeight * len(weights))()
        for t, w, e in zip(self._ctable, weights, self.entries):
            t.w, t.Cout, t.Cin, t.kh, t.kw = w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], w.shape[3]
            t.fwd, t.fwd_scale = (e.fwd.data_ptr(), e.fwd_scale.data_ptr()) if e.fwd is not None else (None, None)
            t.dgrad, t.dgrad_scale = e.dgrad.data_ptr(), e.dgrad_scale.data_ptr()
            t.fwd_inv_in = t.dgrad_inv_in = 1.0
        self._nbytes = C.sizeof(self._ctable)
        self.table = torch.empty(self._nbytes, device=dev, dtype=torch.uint8)
        self._upload()

    def _upload(self):
        import ctypes as C
        if torch.cuda.is_current_stream_capturing():
            # a pageable host -> device copy (and its stream synchronisation) is not capturable: GraphedTrainStep calls sync_table() before it captures
            raise RuntimeError('WeightPack: operand scales changed inside a HIP-graph capture; call sync_table() before capturing (ADVICE r4)')
        host = torch.frombuffer(bytearray(C.string_at(C.addressof(self._ctable), self._nbytes)), dtype=torch.uint8)
        self.table.copy_(host)

    def _apply_pending(self):
        """operand scales asked for during the last step (want) become the applied ones; True if the device table must be re-uploaded"""
        dirty = False
        for t, e in zip(self._ctable, self.entries):
            for f in (0, 1):
                if e.want[f] is not None and e.want[f] != e.applied[f]:
                    e.applied[f] = e.want[f]
                    dirty = True
                e.want[f] = None
            t.fwd_inv_in, t.dgrad_inv_in = 1.0 / e.applied[0], 1.0 / e.applied[1]
        return dirty

    def sync_table(self):
        """apply pending operand-scale changes to the device table NOW.  GraphedTrainStep calls this before every capture: the eager step before it
        (warm-up or re-calibration) re-measured the scales, and refresh() inside the capture would otherwise find the table dirty and upload it
        from pageable host memory in the middle of the capture"""
        if self._apply_pending():
            self._upload()

    def refresh(self):
        """pack the CURRENT values of every weight (call once per step, before the first convolution)"""
        if self._apply_pending():
            self._upload()
        with torch.cuda.device(self.table.device):
            _capi.check(_capi.lib().dir_train_pack_conv_weights(_capi.ptr(self.table), _capi.ptr(self.row_start), len(self.entries), self.total_rows,
                                                                _capi.stream_ptr()), 'dir_train_pack_conv_weights')


_pack = None          # the WeightPack bound for the running step (begin_step .. end_step)


def conv_weights(P):
    """the convolution weights of a parameter dict that run through this module: every 4-D `.weight` except the 3-channel stem's"""
    return [v for k, v in P.items() if k.endswith('.weight') and torch.is_tensor(v) and v.dim() == 4 and v.shape[1] % 32 == 0
            and v.shape[2] == v.shape[3] and v.shape[2] in (1, 3)]          # (Cin % 32: whole reduction slabs; HRNet's 48-wide convolutions pad per call)


def begin_step(owner=None, P=None):
    """called by dir_amd.train.net.forward at the start of every training step; `owner`: the object that owns this model's operand-scale
    cache and packed weights (any object that accepts attributes), or None = no cache; P: the step's parameter dict (with an owner and
    f16x3 arithmetic: all its convolution weights are packed now, in one launch)"""
    global _scales, _state, _pack
    _pack = None
    if owner is None:
        _scales, _state = [], {'call': 0, 'step': 1, 'cached': False}
        return
    if P is not None and ARITH == 'f16x3' and PREPACK:
        ws = conv_weights(P)
        pk = getattr(owner, '_dir_weight_pack', None)
        if ws and (pk is None or pk.ptrs != tuple(w.data_ptr() for w in ws)):
            pk = WeightPack(ws)
            setattr(owner, '_dir_weight_pack', pk)
        if ws:
            pk.refresh()
            _pack = pk
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
    global _scales, _state, _pack
    _scales, _state, _pack = [], {'call': 0, 'step': 1, 'cached': False}, None


def reset_scales(owner=None):
    """forget the cached operand scales (of `owner`, and the ones bound right now): the next step calibrates again on the batch it sees"""
    if owner is not None and getattr(owner, '_dir_conv_scale_cache', None) is not None:
        del owner._dir_conv_scale_cache[0][:]
        owner._dir_conv_scale_cache[1].update(call=0, step=0)
    end_step()


def _measure(x, pre):
    """the power-of-two operand scale of x as the convolution sees it: through the pre-activation max(x ps + pb, 0) when there is one (measured
    exactly, on a temporary: this runs on calibration steps only)"""
    if pre is None:
        return F.pow2_in_scale(x)
    return F.pow2_in_scale(torch.relu(torch.addcmul(pre[1], x, pre[0])))


def _site_scale(x, pre=None):
    if not _state.get('cached', False):          # no owner: measure (one host synchronisation per call site)
        return _measure(x, pre)
    i = _state['call']
    _state['call'] += 1
    if i < len(_scales) and _scales[i][0] == tuple(x.shape):
        return _scales[i][1]
    s = _measure(x, pre)
    prev = _state.get('previous')
    if prev is not None and i < len(prev) and prev[i][0] == tuple(x.shape) and prev[i][1] >= HEADROOM * s:
        # the operand grew past the headroom the stale scale left: values were clamped at the f16 maximum somewhere in the last RECALIBRATE steps
        warnings.warn('dir_amd.train.conv: call site %d (%s) outgrew its cached f16x3 operand scale (%.3g in use, %.3g needed now): operands '
                      'saturated during the last %d steps; lower DIR_TRAIN_RECALIBRATE or call reset_scales() after a learning-rate change'
                      % (i, 'x'.join(map(str, x.shape)), prev[i][1], s, RECALIBRATE), RuntimeWarning, stacklevel=3)
    del _scales[i:]
    _scales.append((tuple(x.shape), s))
    return s


def _conv(x, w, stride, pad, shift=None, packed=None, residual=None, pre=None, stats=None, mask=None, bn_bwd=None):
    """w: OHWI fp32 tensor, or None with packed = (_Packed entry, form 0 forward | 1 data gradient); residual: added in the epilogue;
    pre = (pre_scale, pre_shift) [Cin]: the convolution reads max(x pre_scale + pre_shift, 0) (a BatchNorm + ReLU that is not materialised)"""
    pk = {} if pre is None else dict(pre_scale=pre[0], pre_shift=pre[1], pre_relu=True)
    if stats is not None and STATS_IN_EPILOGUE:
        pk['stats_out'] = stats          # the following BatchNorm's chunk partials from this convolution's epilogue (dir_conv2d_forward_stats)
    if mask is not None:
        pk['mask'] = mask                # y = mask > 0 ? y : 0 in the epilogue (dir_conv2d_forward_masked)
    if bn_bwd is not None and BN_BWD_IN_EPILOGUE:
        pk['bn_bwd'] = bn_bwd            # the chunk partials of the backward pass of the BatchNorm whose output's gradient this convolution writes (dir_conv2d_forward_ex)
    if ARITH != 'f16x3':
        return F.conv2d_nhwc(x, w, stride=stride, pad=pad, shift=shift, residual=residual, **pk)
    shape = w.shape if packed is None else (packed[0].shape, packed[0].dshape)[packed[1]]
    presplit = shape[1] >= 3 or (shape[0] >= 512 and shape[3] >= 128)
    s = _site_scale(x, pre)
    if packed is None:
        return F.conv2d_nhwc(x, w, stride=stride, pad=pad, shift=shift, arith='f16x3', in_scale=s, device_pack=True, presplit=presplit, residual=residual, **pk)
    e, f = packed
    sc = (e.fwd_scale, e.dgrad_scale)[f]
    if e.applied[f] != s:                                  # (a calibration step, or a weight shared by sites of different magnitude)
        sc = sc * (e.applied[f] / s)
        if e.want[f] is None:
            e.want[f] = s
    elif e.want[f] is None:
        e.want[f] = s
    return F.conv2d_nhwc(x, None, stride=stride, pad=pad, shift=shift, arith='f16x3', in_scale=s, presplit=presplit,
                         prepacked=((e.fwd, e.dgrad)[f], sc, shape), residual=residual, **pk)


def _packed_of(w_oihw):
    return _pack.by_ptr.get(w_oihw.data_ptr()) if _pack is not None else None


def _ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def conv_fwd(x, w, bias=None, stride=1, pad=0, oihw=False, residual=None, pre=None, stats=None):
    """w: OHWI [Cout,kh,kw,Cin], or with oihw=True the reference-layout parameter [Cout,Cin,kh,kw] itself (packed once per step when the
    step has a WeightPack: begin_step).  residual [B,Ho,Wo,Cout]: y = conv(x) + bias + residual in the convolution's epilogue (a Residual
    block's skip path: one launch and two passes over the map less than a separate dir_axpy_f32)"""
    if residual is not None:
        residual = residual.contiguous()
    if oihw:
        e = _packed_of(w)
        if e is not None and e.fwd is not None:
            return _conv(x, None, stride, pad, bias, packed=(e, 0), residual=residual, pre=pre, stats=stats)
        w = _ohwi(w)
    cin = w.shape[3]
    if cin % 32:                                           # the 3-channel image: channels padded to the kernel's K granularity
        assert pre is None
        x, w = _pad_last(x, 32), _pad_last(w, 32)
    return _conv(x, w, stride, pad, bias, residual=residual, pre=pre, stats=stats)


def _pad_last(t, mult):
    c = t.shape[-1]
    if c % mult == 0:
        return t
    out = torch.zeros(*t.shape[:-1], (c + mult - 1) // mult * mult, device=t.device, dtype=t.dtype)
    out[..., :c] = t
    return out


def conv_dgrad(w, gy, stride, pad, H, W, oihw=False, add=None, mask=None, bn_bwd=None):
    """d loss / d x [B,H,W,Cin] of y = conv(x, w, stride, pad) from gy [B,Ho,Wo,Cout]; add [B,H,W,Cin]: another gradient of x, summed in
    the convolution's epilogue (the identity / projection path of a residual block); mask [B,H,W,Cin]: the result is zeroed where mask <= 0 (x is
    the output of a ReLU and mask that output: the ReLU's backward, applied in the same epilogue)"""
    e = _packed_of(w) if oihw else None
    if oihw and e is None:
        w = _ohwi(w)
    Cout, kh, kw, Cin = w.shape if e is None else e.shape
    assert kh == kw and stride in (1, 2)
    wt = _pad_last(w.flip(1, 2).permute(3, 1, 2, 0).contiguous(), 32) if e is None else None          # [Cin][kh][kw][Cout (padded)]
    g = gy
    if stride == 2:
        B, Ho, Wo, _ = gy.shape
        g = torch.zeros(B, 2 * Ho, 2 * Wo, Cout, device=gy.device)
        g[:, ::2, ::2] = gy
    g = _pad_last(g.contiguous(), 32)
    p2 = kh - 1 - pad
    fits = g.shape[1] + 2 * p2 - kh + 1 == H and g.shape[2] + 2 * p2 - kw + 1 == W
    fused_mask = mask.contiguous() if (mask is not None and fits and mask.shape[3] % 4 == 0) else None
    # bn_bwd (see _conv): only when the epilogue writes the FINAL gradient (everything that is added to it is added there, any mask applied there)
    bb = bn_bwd if (bn_bwd is not None and fits and (mask is None or fused_mask is not None) and g.shape[3] % 4 == 0) else None
    gx = _conv(g, wt, 1, p2, packed=None if e is None else (e, 1), residual=add.contiguous() if (add is not None and fits) else None, mask=fused_mask, bn_bwd=bb)
    if not fits:                                           # odd H / W under stride 2
        assert gx.shape[1] >= H and gx.shape[2] >= W
        gx = gx[:, :H, :W].contiguous()
        if add is not None:
            O.axpy(gx, add.contiguous())
    if mask is not None and fused_mask is None:
        gx = O.relu_bwd(gx, mask)
    return gx


def conv_wgrad(x, gy, w_shape, stride, pad, out=None, accumulate=False, pre=None):
    """pre = (pre_scale, pre_shift): x is the INPUT of a BatchNorm + ReLU that the forward convolution applied on the fly (conv_fwd(pre=...)):
    the weight gradient applies it the same way (dir_conv2d_wgrad_f16x3_pre)"""
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
        sx, sg = _site_scale(x, pre), _site_scale(gy)
        n = _capi.lib().dir_conv2d_wgrad_f16x3_workspace_bytes(d)
        if accumulate:
            n = max(n, out.numel() * 4)
        ws = torch.empty(max(n, 4) // 4, device=x.device)
        if _capi.PROFILE is not None:
            _capi.annotate(family='wgrad', flops=2.0 * B * gy.shape[1] * gy.shape[2] * Cout * kh * kw * Cin, bytes=4.0 * (x.numel() + gy.numel() + Cout * kh * kw * Cin),
                           shape='wgrad M=%d Cout=%d Cin=%d k%d s%d' % (B * gy.shape[1] * gy.shape[2], Cout, Cin, kh, stride))
        if pre is not None:
            _capi.check(_capi.lib().dir_conv2d_wgrad_f16x3_pre(d, _capi.ptr(x), _capi.ptr(gy), _capi.ptr(out), int(accumulate), _capi.ptr(ws), n, sx, sg,
                                                               _capi.ptr(pre[0]), _capi.ptr(pre[1]), 1, _capi.stream_ptr()), 'dir_conv2d_wgrad_f16x3_pre')
            return out
        _capi.check(_capi.lib().dir_conv2d_wgrad_f16x3(d, _capi.ptr(x), _capi.ptr(gy), _capi.ptr(out), int(accumulate), _capi.ptr(ws), n, sx, sg,
                                                       _capi.stream_ptr()), 'dir_conv2d_wgrad_f16x3')
        return out
    if pre is not None:                                    # (the exact-fp32 weight gradient has no pre-activation: materialise the operand)
        x = torch.relu(torch.addcmul(pre[1], x, pre[0]))
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


def conv_bwd(x, w, gy, stride=1, pad=0, need_gx=True, has_bias=True, oihw=False, add_gx=None, gw_oihw=False, pre=None, mask_gx=None, bn_bwd=None):
    """-> (gx (+ add_gx), gw [Cout,kh,kw,Cin] ([Cout,Cin,kh,kw] with gw_oihw), gb); w OHWI, or the OIHW parameter with oihw=True (conv_fwd).
    Inside a backward pass (side_begin) gw is produced on the side stream: valid on the compute stream after side_end."""
    gy = gy.contiguous()
    shape = (w.shape[0], w.shape[2], w.shape[3], w.shape[1]) if oihw else w.shape

    def wgrad():
        g = conv_wgrad(x, gy, shape, stride, pad, pre=pre)
        return g.permute(0, 3, 1, 2).contiguous() if gw_oihw else g
    gw = side_run(wgrad, x, gy)
    gb = O.colsum(gy.view(-1, gy.shape[3])) if has_bias else None
    gx = conv_dgrad(w, gy, stride, pad, x.shape[1], x.shape[2], oihw=oihw, add=add_gx, mask=mask_gx, bn_bwd=bn_bwd) if need_gx else None
    return gx, gw, gb
