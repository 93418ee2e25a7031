"""Thin tensor front ends of the training building blocks of libdir_hip.so (include/dir_hip.h: dir_gemm_f32, dir_layernorm_*, dir_gelu_*,
dir_attention_*, dir_bn_train_*).  Plumbing only: shapes, strides, output allocation -- every arithmetic step is a library call."""
import os

import torch

from .. import _capi
from .._capi import GemmDesc, GemmGroups

F32 = torch.float32


def _chk(*ts):
    _capi.require_cuda(*ts)
    for t in ts:
        assert t is None or (t.dtype == F32 and t.is_contiguous()), 'fp32 contiguous tensors'


def gemm(A, B, ta=False, tb=False, bias=None, out=None, accumulate=False):
    """out (+)= op(A) op(B) (+ bias).  A, B: 2-D, or 3-D with a shared leading batch dimension (a 2-D operand is broadcast)."""
    _chk(A, B, bias, out)
    batch = max(A.shape[0] if A.dim() == 3 else 1, B.shape[0] if B.dim() == 3 else 1)
    a2, b2 = A.shape[-2:], B.shape[-2:]
    M, K = (a2[1], a2[0]) if ta else (a2[0], a2[1])
    K2, N = (b2[1], b2[0]) if tb else (b2[0], b2[1])
    assert K == K2, (A.shape, B.shape, ta, tb)
    if out is None:
        assert not accumulate
        out = torch.empty((batch, M, N) if batch > 1 or A.dim() == 3 or B.dim() == 3 else (M, N), device=A.device, dtype=F32)
    d = GemmDesc(M, N, K, a2[1], b2[1], N, int(ta), int(tb), int(accumulate), batch,
                 a2[0] * a2[1] if A.dim() == 3 else 0, b2[0] * b2[1] if B.dim() == 3 else 0, M * N if out.dim() == 3 else 0)
    assert out.shape[-2:] == (M, N) and (batch == 1 or out.dim() == 3)
    if _capi.PROFILE is not None:
        _capi.annotate(family='gemm', flops=2.0 * batch * M * N * K, bytes=4.0 * batch * (M * K + K * N + M * N), shape='gemm M=%d N=%d K=%d batch=%d ta=%d tb=%d' % (M, N, K, batch, ta, tb))
    if batch == 1 and K >= 512 and ((M + 63) // 64) * ((N + 63) // 64) <= 32:
        # a tall reduction into a small output (Linear weight gradients over B * 21 .. 42 token rows): K chunks on separate workgroups
        n = _capi.lib().dir_gemm_f32_splitk_workspace_bytes(d)
        ws = torch.empty(n // 4, device=A.device)
        _capi.check(_capi.lib().dir_gemm_f32_splitk(d, _capi.ptr(A), _capi.ptr(B), _capi.ptr(bias), _capi.ptr(out), _capi.ptr(ws), n, _capi.stream_ptr()),
                    'dir_gemm_f32_splitk')
        return out
    _capi.check(_capi.lib().dir_gemm_f32(d, _capi.ptr(A), _capi.ptr(B), _capi.ptr(bias), _capi.ptr(out), _capi.stream_ptr()), 'dir_gemm_f32')
    return out


def colsum(x, out=None, accumulate=False):
    """column sums of a 2-D tensor (bias gradients)"""
    _chk(x, out)
    R, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=F32)
    n = _capi.lib().dir_colsum_workspace_bytes(R, N)
    ws = torch.empty(n // 4, device=x.device) if n > 0 else None
    _capi.check(_capi.lib().dir_colsum_f32(_capi.ptr(x), _capi.ptr(out), R, N, N, int(accumulate), _capi.ptr(ws), n, _capi.stream_ptr()), 'dir_colsum_f32')
    return out


def linear_fwd(x, W, b, out=None, accumulate=False):
    """nn.Linear on rows: x [R,K], W [N,K] -> [R,N] (accumulate: added onto `out`, e.g. a residual trunk)"""
    return gemm(x, W, tb=True, bias=b, out=out, accumulate=accumulate)


def linear_bwd(gy, x, W, gW=None, gb=None, accumulate=False, need_gx=True):
    """-> (g x [R,K] or None, g W [N,K], g b [N]); accumulate: parameter gradients are added to gW / gb (shared parameters)"""
    gx = gemm(gy, W) if need_gx else None
    gW = gemm(gy, x, ta=True, out=gW, accumulate=accumulate and gW is not None)
    gb = colsum(gy, out=gb, accumulate=accumulate and gb is not None)
    return gx, gW, gb


def layernorm_fwd(x, w, b, eps):
    _chk(x, w, b)
    R, C = x.shape
    y, mean, rstd = torch.empty_like(x), torch.empty(R, device=x.device), torch.empty(R, device=x.device)
    _capi.check(_capi.lib().dir_layernorm_forward(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(y), _capi.ptr(mean), _capi.ptr(rstd), R, C, float(eps),
                                                  _capi.stream_ptr()), 'dir_layernorm_forward')
    return y, (mean, rstd)


def layernorm_bwd(gy, x, w, stats, gx=None, gw=None, gb=None, accumulate_x=False, accumulate_wb=False):
    """g x is written into (or, accumulate_x, added onto) gx; g w / g b likewise"""
    _chk(gy, x, w, gx, gw, gb)
    R, C = x.shape
    if gx is None:
        gx = torch.empty_like(x)
        accumulate_x = False
    if gw is None:
        gw, gb, accumulate_wb = torch.empty(C, device=x.device), torch.empty(C, device=x.device), False
    _capi.check(_capi.lib().dir_layernorm_backward(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(gx),
                                                   _capi.ptr(gw), _capi.ptr(gb), R, C, int(accumulate_x), int(accumulate_wb), _capi.stream_ptr()),
                'dir_layernorm_backward')
    return gx, gw, gb


def gelu_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    _capi.check(_capi.lib().dir_gelu_forward(_capi.ptr(x), _capi.ptr(y), x.numel(), _capi.stream_ptr()), 'dir_gelu_forward')
    return y


def gelu_bwd(gy, x):
    _chk(gy, x)
    gx = torch.empty_like(x)
    _capi.check(_capi.lib().dir_gelu_backward(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(gx), x.numel(), _capi.stream_ptr()), 'dir_gelu_backward')
    return gx


def attention_fwd(qkv, B, T, H, scale, save_probs=True):
    """qkv [B*T, 3*H*32] -> (out [B*T, H*32], probs [B,H,T,T] or None)"""
    _chk(qkv)
    out = torch.empty(B * T, H * 32, device=qkv.device)
    probs = torch.empty(B, H, T, T, device=qkv.device) if save_probs else None
    _capi.check(_capi.lib().dir_attention_forward(_capi.ptr(qkv), _capi.ptr(probs), _capi.ptr(out), B, T, H, float(scale), _capi.stream_ptr()),
                'dir_attention_forward')
    return out, probs


def attention_bwd(qkv, probs, gout, B, T, H, scale):
    _chk(qkv, probs, gout)
    gqkv = torch.empty_like(qkv)
    _capi.check(_capi.lib().dir_attention_backward(_capi.ptr(qkv), _capi.ptr(probs), _capi.ptr(gout), _capi.ptr(gqkv), B, T, H, float(scale),
                                                   _capi.stream_ptr()), 'dir_attention_backward')
    return gqkv


# BatchNorm with frozen statistics inside a training pass (every nn.BatchNorm* of the model in .eval() under model.train()): normalise with the
# running statistics, leave them alone; gradients w.r.t. weight / bias / input only (dir_bn_frozen_*).  The reference trains with batch
# statistics (train.py:64); this switch exists because the reference's whole-step gradient is reproducible to 4e-5 only in this form
# (tests/golden G20e, tests/test_gpu_full_bwd.py) -- and it is what fine-tuning with frozen BatchNorm runs.
BN_FROZEN = False


class frozen_batchnorm(object):
    """with frozen_batchnorm(): every bn_train_fwd / bn_train_bwd of the block uses the running statistics"""
    def __enter__(self):
        global BN_FROZEN
        self.prev, BN_FROZEN = BN_FROZEN, True

    def __exit__(self, *a):
        global BN_FROZEN
        BN_FROZEN = self.prev


# SyncBN (SURVEY.md 8e "optionally offer"; torch.nn.SyncBatchNorm semantics): with SYNC_BN on and more than one rank, every training-mode
# BatchNorm of the step pools its statistics over the ranks -- the reference's batch of 64 on one GPU (config.py:13-15) split over N GPUs then
# normalises exactly like the single-GPU batch.  Two small collectives per layer and direction (2 C + 4 floats gathered forward, 2 C floats
# reduced backward): not capturable in a HIP graph, so GraphedTrainStep is not for this mode.  Unmeasured on multi-GPU hardware (no node).
SYNC_BN = False


class sync_batchnorm(object):
    """with sync_batchnorm(): every bn_train_fwd / bn_train_bwd of the block pools its statistics over the ranks (a no-op with one rank)"""
    def __enter__(self):
        global SYNC_BN
        self.prev, SYNC_BN = SYNC_BN, True

    def __exit__(self, *a):
        global SYNC_BN
        SYNC_BN = self.prev


def _sync_active(C, *ts):
    from .. import dist as D
    import torch.distributed as dist
    # rank-INVARIANT facts only: a rank that fell back to the local path because one of its tensors happened to be misaligned would
    # leave the others waiting in all_gather / all_reduce.  Alignment is asserted instead (torch's caching allocator hands out 512-byte aligned
    # blocks; a sliced view with an odd offset is a caller error in this mode).
    on = SYNC_BN and not BN_FROZEN and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and C % 4 == 0 and D is not None
    if on:
        assert all(t is None or t.data_ptr() % 16 == 0 for t in ts), 'SyncBN: tensors must be 16-byte aligned (a rank cannot fall back on its own)'
    return on


def _sync_ws(R, C, dev):
    n = _capi.lib().dir_bn_sync_workspace_bytes(R, C)
    return torch.empty(n // 4, device=dev), n


def sync_bn_fwd(x, w, b, running_mean, running_var, eps, momentum, relu, residual):
    """bn_train_fwd with the statistics pooled over the ranks: local (mean | M2) -> all-gather -> exact pooling (dir_bn_sync_combine, the same
    bytes on every rank) -> normalisation with the pooled statistics (dir_bn_frozen_forward).  -> (y, (save_mean, save_rstd, rows_pooled))"""
    from .. import dist as D
    R, C = x.shape
    L = _capi.lib()
    part = torch.zeros(2 * C + 4, device=x.device)
    ws, n = _sync_ws(R, C, x.device)
    _capi.check(L.dir_bn_sync_local_stats(_capi.ptr(x), _capi.ptr(part), R, C, C, _capi.ptr(ws), n, _capi.stream_ptr()), 'dir_bn_sync_local_stats')
    part[2 * C] = float(R)
    parts = D.all_gather_rows(part).contiguous()
    mean, var = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    _capi.check(L.dir_bn_sync_combine(_capi.ptr(parts), parts.shape[0], C, _capi.ptr(mean), _capi.ptr(var), _capi.ptr(running_mean), _capi.ptr(running_var),
                                      float(momentum), _capi.stream_ptr()), 'dir_bn_sync_combine')
    y, sm, sr = torch.empty_like(x), torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    _capi.check(L.dir_bn_frozen_forward(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(y), _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(mean), _capi.ptr(var),
                                        R, C, C, float(eps), int(relu), _capi.ptr(residual), _capi.stream_ptr()), 'dir_bn_frozen_forward')
    written = [t for t in (running_mean, running_var) if t is not None]
    if written:
        torch._C._increment_version(written)
    # pooled row count: one host read per distinct (rows, world size) of this process instead of one per BatchNorm layer and step -- the
    # shards of a run are fixed (dir_amd.dist.shard_range), so the other ranks' row counts do not change between steps
    import torch.distributed as dist
    key = (R, dist.get_world_size())
    if key not in _POOLED_ROWS:
        _POOLED_ROWS[key] = float(parts[:, 2 * C].sum())
    return y, (sm, sr, _POOLED_ROWS[key])


_POOLED_ROWS = {}


def sync_bn_bwd(gy, x, w, stats, need_gx, b, relu):
    """bn_train_bwd for sync_bn_fwd: g w, g b are this rank's sums (the gradient exchange averages them); g x uses the sums pooled over the ranks"""
    from .. import dist as D
    R, C = x.shape
    L = _capi.lib()
    sums = torch.empty(2 * C, device=x.device)
    ws, n = _sync_ws(R, C, x.device)
    _capi.check(L.dir_bn_sync_backward_sums(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(sums),
                                            R, C, C, int(relu), _capi.ptr(ws), n, _capi.stream_ptr()), 'dir_bn_sync_backward_sums')
    gb, gw = sums[:C].clone(), sums[C:].clone()
    gx = None
    if need_gx:
        pooled = D.all_reduce_sum(sums)
        gx = torch.empty_like(x)
        _capi.check(L.dir_bn_sync_backward_apply(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(pooled),
                                                 _capi.ptr(gx), R, float(stats[2]), C, C, int(relu), _capi.stream_ptr()), 'dir_bn_sync_backward_apply')
    return gx, gw, gb


def _bn_ws(R, C, dev):
    n = _capi.lib().dir_bn_train_workspace_bytes(R, C)
    return (torch.empty(n // 4, device=dev) if n > 0 else None), n


def bn_train_fwd(x, w, b, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, relu=False, residual=None):
    """BatchNorm in training mode over x [R, C] (channels last).  -> (y, (save_mean, save_rstd)); running statistics updated in place.
    relu=True: y = max(BatchNorm(x), 0) in the same launch (bn_train_bwd(..., b=b, relu=True) re-computes the mask from x)"""
    _chk(x, w, b, running_mean, running_var, residual)
    assert residual is None or residual.shape == x.shape
    R, C = x.shape
    y, sm, sr = torch.empty_like(x), torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    if _sync_active(C, x, w, b, residual):
        return sync_bn_fwd(x, w, b, running_mean, running_var, eps, momentum, relu, residual)
    if BN_FROZEN:
        assert running_mean is not None and running_var is not None, 'frozen BatchNorm needs the running statistics'
        _capi.check(_capi.lib().dir_bn_frozen_forward(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(y), _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(running_mean),
                                                      _capi.ptr(running_var), R, C, C, float(eps), int(relu), _capi.ptr(residual), _capi.stream_ptr()), 'dir_bn_frozen_forward')
        return y, (sm, sr)
    ws, n = _bn_ws(R, C, x.device)
    _capi.check(_capi.lib().dir_bn_train_forward(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(y), _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(running_mean),
                                                 _capi.ptr(running_var), R, C, C, float(eps), float(momentum), int(relu), _capi.ptr(residual), _capi.ptr(ws), n, _capi.stream_ptr()),
                'dir_bn_train_forward')
    # the kernel wrote the running statistics through raw pointers: bump their version counters as an in-place torch op would, so that
    # caches keyed on (data_ptr, _version) -- DIR.engine()'s packed eval weights -- see the update (FlatAdamW.step does the same)
    written = [t for t in (running_mean, running_var) if t is not None]
    if written:
        torch._C._increment_version(written)
    return y, (sm, sr)


# round 5: a training-mode BatchNorm (+ ReLU) whose ONLY consumer is a convolution is not applied as a pass of its own: dir_bn_train_stats forms the
# statistics and the per-channel affine (pre_scale, pre_shift), and the consuming convolution (forward: dir_conv2d_forward's pre-activation or
# dir_split_f16_forward's; weight gradient: dir_conv2d_wgrad_f16x3_pre) applies max(x pre_scale + pre_shift, 0) where it reads x.  One launch and
# two passes over the map less per layer, and the normalised map is not stored for the backward pass.  DIR_TRAIN_FUSE_BN=0 switches it off.
FUSE_BN = os.environ.get('DIR_TRAIN_FUSE_BN', '1') == '1'
FUSE_BN_MIN_ROWS = 513          # dir_bn_train_stats: maps above BN_SMALL_R rows (smaller ones are one cooperative launch already)


def bn_can_fuse(x2d, w, b):
    R, C = x2d.shape
    return (FUSE_BN and not BN_FROZEN and R >= FUSE_BN_MIN_ROWS and C % 32 == 0 and w is not None and b is not None and x2d.data_ptr() % 16 == 0
            and not _sync_active(C, x2d, w, b))


def bn_train_stats(x, w, b, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """statistics of BatchNorm (training mode) over x [R, C] without the normalised map: -> ((save_mean, save_rstd), (pre_scale, pre_shift));
    running statistics updated in place like bn_train_fwd"""
    _chk(x, w, b, running_mean, running_var)
    R, C = x.shape
    sm, sr, ps, pb = (torch.empty(C, device=x.device) for _ in range(4))
    ws, n = _bn_ws(R, C, x.device)
    _capi.check(_capi.lib().dir_bn_train_stats(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(ps), _capi.ptr(pb),
                                               _capi.ptr(running_mean), _capi.ptr(running_var), R, C, C, float(eps), float(momentum), _capi.ptr(ws), n,
                                               _capi.stream_ptr()), 'dir_bn_train_stats')
    written = [t for t in (running_mean, running_var) if t is not None]
    if written:
        torch._C._increment_version(written)
    return (sm, sr), (ps, pb)


def bn_partials_usable(x2d, w, b, partials, residual=None):
    """partials: what conv_fwd(stats=[...]) collected -- usable when the convolution's kernel formed them and this BatchNorm runs the plain path"""
    R, C = x2d.shape
    return (bool(partials) and partials[-1][2] > 0 and FUSE_BN and not BN_FROZEN and R >= FUSE_BN_MIN_ROWS and C % 4 == 0 and w is not None and b is not None
            and x2d.data_ptr() % 16 == 0 and (residual is None or residual.data_ptr() % 16 == 0) and not _sync_active(C, x2d, w, b))


def bn_train_stats_from_partials(partials, R, w, b, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, want_pre=True):
    """bn_train_stats without reading the map: the chunk partials came out of the producing convolution's epilogue (p1, p2, rows per chunk)"""
    p1, p2, rows = partials
    C = p1.shape[-1]
    _chk(p1, p2, w, b, running_mean, running_var)
    sm, sr = torch.empty(C, device=p1.device), torch.empty(C, device=p1.device)
    ps, pb = (torch.empty(C, device=p1.device), torch.empty(C, device=p1.device)) if want_pre else (None, None)
    _capi.check(_capi.lib().dir_bn_train_stats_from_partials(_capi.ptr(p1), _capi.ptr(p2), rows, p1.shape[0], _capi.ptr(w), _capi.ptr(b), _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(ps),
                                                             _capi.ptr(pb), _capi.ptr(running_mean), _capi.ptr(running_var), R, C, float(eps), float(momentum),
                                                             _capi.stream_ptr()), 'dir_bn_train_stats_from_partials')
    written = [t for t in (running_mean, running_var) if t is not None]
    if written:
        torch._C._increment_version(written)
    return (sm, sr), (ps, pb)


def bn_train_fwd_from_partials(x, partials, w, b, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, relu=False, residual=None):
    """bn_train_fwd whose statistics pass over x is replaced by the producing convolution's chunk partials (one library call)"""
    p1, p2, rows = partials
    _chk(x, p1, p2, w, b, running_mean, running_var, residual)
    R, C = x.shape
    y, sm, sr = torch.empty_like(x), torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    _capi.check(_capi.lib().dir_bn_train_forward_from_partials(_capi.ptr(x), _capi.ptr(p1), _capi.ptr(p2), rows, p1.shape[0], _capi.ptr(w), _capi.ptr(b), _capi.ptr(y),
                                                               _capi.ptr(sm), _capi.ptr(sr), _capi.ptr(running_mean), _capi.ptr(running_var), R, C, C, float(eps),
                                                               float(momentum), int(relu), _capi.ptr(residual), _capi.stream_ptr()), 'dir_bn_train_forward_from_partials')
    written = [t for t in (running_mean, running_var) if t is not None]
    if written:
        torch._C._increment_version(written)
    return y, (sm, sr)


def bn_train_apply(x, w, b, stats, relu=False, residual=None):
    """y = act(BatchNorm(x) + residual) from statistics already formed (the third launch of bn_train_fwd)"""
    _chk(x, w, b, residual)
    R, C = x.shape
    y = torch.empty_like(x)
    _capi.check(_capi.lib().dir_bn_train_apply(_capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(y), R, C, C, int(relu),
                                               _capi.ptr(residual), _capi.stream_ptr()), 'dir_bn_train_apply')
    return y


def bn_bwd_spec(x2d, w, b, stats, relu):
    """what conv_dgrad(bn_bwd=...) needs to form this BatchNorm backward's sums in the data-gradient convolution's epilogue, or None when this
    BatchNorm does not run the plain large-map path (frozen / pooled statistics, small maps, odd widths)"""
    R, C = x2d.shape
    if (not FUSE_BN or BN_FROZEN or len(stats) != 2 or R < FUSE_BN_MIN_ROWS or C % 4 or w is None or b is None or x2d.data_ptr() % 16
            or _sync_active(C, x2d, w, b)):
        return None
    return dict(z=x2d, mean=stats[0], rstd=stats[1], w=w, b=b, relu=bool(relu), out=[])


def bn_train_bwd(gy, x, w, stats, need_gx=True, b=None, relu=False, partials=None):
    """partials: (p1, p2, chunks) formed by the data-gradient convolution that wrote gy (dir_conv2d_forward_ex): the pass over (gy, x) for the two column
    sums is skipped (dir_bn_train_backward_from_partials)"""
    _chk(gy, x, w, b)
    assert not relu or b is not None
    R, C = x.shape
    if len(stats) == 3:                                     # saved by sync_bn_fwd (pooled row count in stats[2])
        return sync_bn_bwd(gy, x, w, stats, need_gx, b, relu)
    gx = torch.empty_like(x) if need_gx else None
    gw, gb = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    if partials is not None and not BN_FROZEN and C % 4 == 0 and gy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0:
        p1, p2, chunks = partials
        ws = torch.empty(2 * C, device=x.device)
        _capi.check(_capi.lib().dir_bn_train_backward_from_partials(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]),
                                                                    _capi.ptr(p1), _capi.ptr(p2), chunks, _capi.ptr(gx), _capi.ptr(gw), _capi.ptr(gb), R, C, C, int(relu),
                                                                    _capi.ptr(ws), 2 * C * 4, _capi.stream_ptr()), 'dir_bn_train_backward_from_partials')
        return gx, gw, gb
    if BN_FROZEN:
        n = _capi.lib().dir_bn_frozen_workspace_bytes(R, C)
        ws = torch.empty(n // 4, device=x.device)
        _capi.check(_capi.lib().dir_bn_frozen_backward(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(gx),
                                                       _capi.ptr(gw), _capi.ptr(gb), R, C, C, int(relu), _capi.ptr(ws), n, _capi.stream_ptr()), 'dir_bn_frozen_backward')
        return gx, gw, gb
    ws, n = _bn_ws(R, C, x.device)
    _capi.check(_capi.lib().dir_bn_train_backward(_capi.ptr(gy), _capi.ptr(x), _capi.ptr(w), _capi.ptr(b), _capi.ptr(stats[0]), _capi.ptr(stats[1]), _capi.ptr(gx),
                                                  _capi.ptr(gw), _capi.ptr(gb), R, C, C, int(relu), _capi.ptr(ws), n, _capi.stream_ptr()), 'dir_bn_train_backward')
    return gx, gw, gb


def gemm_strided(A, B, C, M, N, K, lda, ldb, ldc, ta=False, tb=False, batch=1, sa=0, sb=0, sc=0, a_off=0, b_off=0, c_off=0, bias=None,
                 accumulate=False):
    """dir_gemm_f32 with explicit leading dimensions / batch strides / element offsets into the three buffers (views that torch would
    have to copy: node j of a [B,21,C] tensor is the matrix at offset j*C with row pitch 21*C)"""
    import ctypes as Ct
    _chk(A, B, C, bias)
    d = GemmDesc(M, N, K, lda, ldb, ldc, int(ta), int(tb), int(accumulate), batch, sa, sb, sc)
    p = lambda t, off: Ct.c_void_p(t.data_ptr() + 4 * off)  # noqa: E731
    if _capi.PROFILE is not None:
        _capi.annotate(family='gemm', flops=2.0 * batch * M * N * K, bytes=4.0 * batch * (M * K + K * N + M * N),
                       shape='gemm_strided M=%d N=%d K=%d batch=%d ta=%d tb=%d' % (M, N, K, batch, ta, tb))
    if batch == 1 and K >= 512 and ((M + 63) // 64) * ((N + 63) // 64) <= 32:
        # a long reduction into a small output (the regressors' Linear layers over 1344 / 2688 inputs): K chunks on separate workgroups
        n = _capi.lib().dir_gemm_f32_splitk_workspace_bytes(d)
        ws = torch.empty(n // 4, device=A.device)
        _capi.check(_capi.lib().dir_gemm_f32_splitk(d, p(A, a_off), p(B, b_off), _capi.ptr(bias), p(C, c_off), _capi.ptr(ws), n, _capi.stream_ptr()),
                    'dir_gemm_f32_splitk')
        return C
    _capi.check(_capi.lib().dir_gemm_f32(d, p(A, a_off), p(B, b_off), _capi.ptr(bias), p(C, c_off), _capi.stream_ptr()), 'dir_gemm_f32')
    return C


def gemm_grouped(A, B, C, M, N, K, lda, ldb, ldc, groups, disp_a, disp_b, disp_c=(0, 0), reduce=False, ta=False, tb=False, batch=1, sa=0, sb=0, sc=0,
                 a_off=0, b_off=0, c_off=0):
    """dir_gemm_f32_grouped: an ny x nx grid of products whose operands (and, unless reduce, results) are displaced by gy * disp[0] + gx * disp[1]
    elements; reduce: C = the sum of the groups' products"""
    import ctypes as Ct
    _chk(A, B, C)
    d = GemmDesc(M, N, K, lda, ldb, ldc, int(ta), int(tb), 0, batch, sa, sb, sc)
    g = GemmGroups(groups[0], groups[1], int(reduce), 0, disp_a[0], disp_a[1], disp_b[0], disp_b[1], disp_c[0], disp_c[1])
    p = lambda t, off: Ct.c_void_p(t.data_ptr() + 4 * off)  # noqa: E731
    _capi.check(_capi.lib().dir_gemm_f32_grouped(d, g, p(A, a_off), p(B, b_off), None, p(C, c_off), _capi.stream_ptr()), 'dir_gemm_f32_grouped')
    return C


def relu_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    _capi.check(_capi.lib().dir_relu_forward(_capi.ptr(x), _capi.ptr(y), x.numel(), _capi.stream_ptr()), 'dir_relu_forward')
    return y


def relu_bwd(gy, y):
    _chk(gy, y)
    gx = torch.empty_like(y)
    _capi.check(_capi.lib().dir_relu_backward(_capi.ptr(gy), _capi.ptr(y), _capi.ptr(gx), y.numel(), _capi.stream_ptr()), 'dir_relu_backward')
    return gx


def pgcn_adjacency(e1):
    _chk(e1)
    A = torch.empty(21, 21, device=e1.device)
    _capi.check(_capi.lib().dir_pgcn_adjacency_forward(_capi.ptr(e1), _capi.ptr(A), _capi.stream_ptr()), 'dir_pgcn_adjacency_forward')
    return A


def pgcn_adjacency_bwd(e1, gz, h1):
    _chk(e1, gz, h1)
    scratch, ge = torch.empty(40, device=e1.device), torch.empty_like(e1)
    _capi.check(_capi.lib().dir_pgcn_adjacency_backward(_capi.ptr(e1), _capi.ptr(gz), _capi.ptr(h1), _capi.ptr(scratch), _capi.ptr(ge), gz.shape[0],
                                                        _capi.stream_ptr()), 'dir_pgcn_adjacency_backward')
    return ge


def grid_rows_fwd(feat_nhwc, uv):
    """[B,S,S,C], uv [B,21,2] -> [B*21, C]"""
    _chk(feat_nhwc, uv)
    B, S, _, C = feat_nhwc.shape
    rows = torch.empty(B * 21, C, device=uv.device)
    _capi.check(_capi.lib().dir_grid_rows_forward(_capi.ptr(feat_nhwc), _capi.ptr(uv), _capi.ptr(rows), B, S, C, _capi.stream_ptr()), 'dir_grid_rows_forward')
    return rows


def grid_rows_bwd(g_rows_list, uv_list, B, S, C, out=None):
    """sum over the samplers (hands) -> g feat NHWC [B,S,S,C] (written; or added onto `out`)"""
    import ctypes as Ct
    _chk(*(list(g_rows_list) + list(uv_list)))
    n = len(g_rows_list)
    zero = out is None
    if out is None:
        out = torch.empty(B, S, S, C, device=uv_list[0].device)
    P = Ct.c_void_p * n
    _capi.check(_capi.lib().dir_grid_rows_backward(P(*[t.data_ptr() for t in g_rows_list]), P(*[t.data_ptr() for t in uv_list]), n, _capi.ptr(out), B, S, C,
                                                   int(zero), _capi.stream_ptr()), 'dir_grid_rows_backward')
    return out


def axpy(dst, src, alpha=1.0):
    """dst += alpha src (both contiguous, same size)"""
    _chk(dst, src)
    assert dst.numel() == src.numel()
    _capi.check(_capi.lib().dir_axpy_f32(_capi.ptr(dst), _capi.ptr(src), dst.numel(), float(alpha), _capi.stream_ptr()), 'dir_axpy_f32')
    return dst


def axpy_multi(dsts, srcs, alpha=1.0):
    """dst_t += alpha src_t for lists of tensors, ~len / 40 launches (dir_axpy_multi_f32)"""
    import ctypes as C
    n = len(dsts)
    if n == 0:
        return
    srcs = [s.contiguous() for s in srcs]
    _chk(*dsts)
    _chk(*srcs)
    assert all(d.numel() == s.numel() for d, s in zip(dsts, srcs))
    P, L = C.c_void_p * n, C.c_longlong * n
    _capi.check(_capi.lib().dir_axpy_multi_f32(P(*[d.data_ptr() for d in dsts]), P(*[s.data_ptr() for s in srcs]), L(*[d.numel() for d in dsts]), n,
                                               float(alpha), _capi.stream_ptr()), 'dir_axpy_multi_f32')


def stage_positions(xyz_l, xyz_r, offset):
    _chk(xyz_l, xyz_r, offset)
    B = xyz_l.shape[0]
    outs = [torch.empty(B * 21, 3, device=xyz_l.device) for _ in range(4)]
    _capi.check(_capi.lib().dir_stage_positions(_capi.ptr(xyz_l), _capi.ptr(xyz_r), _capi.ptr(offset), *[_capi.ptr(o) for o in outs], B, _capi.stream_ptr()),
                'dir_stage_positions')
    return outs
