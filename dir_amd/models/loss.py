"""Host-side mirror of the reference's training objective (forward values), on libdir_hip.so.

  DirLoss.__call__      models/dir.py:542-594   the loss block of DIR.forward: same inputs (the per-stage output dicts incl.
                                                pd_mesh_uv_*, the decoder's seg / dense maps, `target`, `meta_info`), same 42 keys
                                                ('seg', 'dense', 'lovasz', 'joint_left_uv_0', ..., 'offset_2'), same weights
                                                (coord_weight 10, dense_weight 1, class weights .1/.45/.45)
  stage_losses          models/dir.py:571-592   one dir_stage_losses_forward launch pair per stage (13 terms)
  dense_losses          models/dir.py:562-569   dir_dense_losses_forward (interpolate + CE + SmoothL1 + Lovasz)
The modules behind them in the reference: models/loss.py (SmoothL1Loss, EdgeLengthLoss, NormalVectorLoss),
models/lovasz_loss.py (lovasz_softmax).  Their gradients w.r.t. the predictions: stage_loss_grads / dense_loss_grads below (dir_stage_losses_backward,
dir_dense_losses_backward); the network behind the predictions: dir_amd/train/.
"""
import ctypes as C

import torch

from .. import _capi

STAGE_KEYS = ('joint_left_uv', 'joint_right_uv', 'mesh_left_uv', 'mesh_right_uv', 'joint_left_xyz', 'joint_right_xyz',
              'mesh_left_xyz', 'mesh_right_xyz', 'edge_left', 'edge_right', 'normal_left', 'normal_right', 'offset')
SIDES = ('left', 'right')


def _pair(d, prefix, dev=None):
    ts = [_capi.f32c(d[prefix + s] if dev is None else d[prefix + s].to(dev)) for s in SIDES]
    return ts, (C.c_void_p * 2)(*[_capi.ptr(t) for t in ts])


def _check_faces(faces, dev, n_vertices=778):
    """two [F,3] int tables of the same size with indices in [0, n_vertices): an index out of range would read out of bounds on the
    device.  The range check costs a host round trip, so it is done once per table OBJECT and version: the verdict is stored on the
    tensor itself (an address-keyed cache would be fooled by a freed-and-reused allocation, and would grow without bound)."""
    fs = [torch.as_tensor(f).to(device=dev, dtype=torch.int32).contiguous() for f in faces]          # (callers that step in a loop should pass the
    # same tensor objects every step -- train_step's `faces` tuple, the mirror's th_faces buffers do: a rebuilt tensor is a new object and is checked again)
    if fs[0].shape != fs[1].shape or fs[0].dim() != 2 or fs[0].shape[1] != 3:
        raise _capi.DirHipError('stage losses: faces must be two [F,3] tables of the same size')
    for f, src in zip(fs, faces):
        tag = (src._version, n_vertices) if torch.is_tensor(src) else None
        if tag is None or getattr(src, '_dir_faces_checked', None) != tag:
            if f.numel() and (int(f.min()) < 0 or int(f.max()) >= n_vertices):
                raise _capi.DirHipError('stage losses: face index outside [0, %d)' % n_vertices)
            if tag is not None:
                src._dir_faces_checked = tag
    return fs


def _marshal_stage(pred, target, meta_info, faces, mesh_uv_required):
    """Shared argument marshalling / validation of dir_stage_losses_forward and _backward.  Targets and meta_info are moved to the
    predictions' device (the reference calls .cuda() on them, models/dir.py:543-560).  Returns (LossPred, LossTarget, keep-alive list,
    pd_offset tensor or None when the stage has no offset, B, faces)."""
    keep = []
    p, g = _capi.LossPred(), _capi.LossTarget()
    dev = pred['pd_joint_uv_left'].device
    for field, prefix in (('joint_uv', 'pd_joint_uv_'), ('joint_xyz', 'pd_joint_xyz_'), ('mesh_xyz', 'pd_mesh_xyz_')):
        ts, arr = _pair(pred, prefix)
        keep += ts
        setattr(p, field, arr)
    # pd_mesh_uv_* is an output of the reference's regressors that only this loss reads (models/dir.py:278-280,574-575); the
    # engine's stage dicts carry pd_proj_* instead and the forward kernel projects the mesh itself
    if 'pd_mesh_uv_left' in pred:
        ts, arr = _pair(pred, 'pd_mesh_uv_')
        p.mesh_uv = arr
    elif mesh_uv_required:
        raise _capi.DirHipError('stage loss gradients: pd_mesh_uv_* must be in the stage dict (it is an independent input of the gradient)')
    else:
        ts, arr = _pair(pred, 'pd_proj_')
        p.proj = arr
    keep += ts
    B = keep[0].shape[0]
    has_off = pred.get('pd_offset') is not None                      # models/dir.py:589: the offset term exists only if predicted
    off = _capi.f32c(pred['pd_offset']) if has_off else torch.zeros(B, 3, device=dev)
    keep.append(off)
    p.offset = _capi.ptr(off)
    t2d = []
    for field, prefix in (('joint_2d', 'joint_2d_'), ('mesh_2d', 'mesh_2d_'), ('joint_3d', 'joint_3d_'), ('mesh_3d', 'mesh_3d_')):
        ts, arr = _pair(target, prefix, dev)
        keep += ts
        setattr(g, field, arr)
        if field.endswith('2d'):
            t2d += ts
    ts, arr = _pair(meta_info, 'center_', dev)
    keep += ts
    g.center = arr
    c2 = t2d[0].shape[-1]
    if any(t.shape[-1] != c2 for t in t2d) or c2 < 2:
        raise _capi.DirHipError('stage losses: the 2-D targets must share their last dimension (>= 2)')
    if any(t.shape[0] != B for t in keep):
        raise _capi.DirHipError('stage losses: batch size mismatch between predictions and targets')
    fs = _check_faces(faces, dev)
    keep += fs
    g.faces = (C.c_void_p * 2)(*[_capi.ptr(f) for f in fs])
    g.c2, g.n_faces = int(c2), int(fs[0].shape[0])
    _capi.require_cuda(*keep)
    return p, g, keep, (off if has_off else None), B, fs


def stage_losses(pred, target, meta_info, faces, coord_weight=10.0):
    """pred: one entry of iter_outs (pd_joint_uv_*, pd_mesh_uv_* or pd_proj_*, pd_joint_xyz_*, pd_mesh_xyz_*, pd_offset); target: joint_2d_*,
    mesh_2d_* [B,N,>=2], joint_3d_*, mesh_3d_*; meta_info: center_* [B,1,3]; faces: (left, right) int tensors [F,3].
    Returns a float32 tensor of the 13 terms in STAGE_KEYS order (on the GPU, no host synchronisation); the last one (offset) is
    meaningless when the stage predicts no offset (DirLoss drops the key, like the reference)."""
    p, g, keep, off, B, _ = _marshal_stage(pred, target, meta_info, faces, False)
    dev = keep[0].device
    scratch = torch.empty(B * 13, device=dev, dtype=torch.float64)
    out = torch.empty(13, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.check(_capi.lib().dir_stage_losses_forward(C.byref(p), C.byref(g), float(coord_weight), _capi.ptr(scratch),
                                                         _capi.ptr(out), B, _capi.stream_ptr()), 'dir_stage_losses_forward')
    return out


def dense_losses(seg_logits, dense_pred, gt_seg, gt_dense, class_weight=(0.1, 0.45, 0.45), dense_weight=1.0):
    """models/dir.py:562-569 -> float32 tensor (seg, dense, lovasz) on the GPU"""
    seg, dense = _capi.f32c(seg_logits), _capi.f32c(dense_pred)
    gs, gd = _capi.f32c(gt_seg.to(seg.device)), _capi.f32c(gt_dense.to(seg.device))      # the reference: target[...].cuda()
    _capi.require_cuda(seg, dense, gs, gd)
    B, Cc, S, S2 = seg.shape
    if Cc != 3 or S != S2 or dense.shape != seg.shape or gs.shape[:2] != (B, 1) or gd.shape[:2] != (B, 3) or gs.shape[2:] != gd.shape[2:]:
        raise _capi.DirHipError('dense_losses: expected seg / dense [B,3,S,S], gt_seg [B,1,H,W], gt_dense [B,3,H,W]')
    H, W = gs.shape[2:]
    L = _capi.lib()
    with torch.cuda.device(seg.device):
        nbytes = L.dir_dense_losses_workspace_bytes(B, S)
        if nbytes < 0:
            raise _capi.DirHipError('dir_dense_losses_workspace_bytes failed')
        ws = torch.empty(nbytes, device=seg.device, dtype=torch.uint8)
        out = torch.empty(3, device=seg.device, dtype=torch.float32)
        _capi.check(L.dir_dense_losses_forward(_capi.ptr(seg), _capi.ptr(dense), _capi.ptr(gs), _capi.ptr(gd),
                                               (C.c_float * 3)(*class_weight), float(dense_weight), _capi.ptr(ws), nbytes,
                                               _capi.ptr(out), B, S, H, W, _capi.stream_ptr()), 'dir_dense_losses_forward')
    return out


class DirLoss(object):
    """The loss block of DIR.forward (models/dir.py:505-511 for the weights, :542-594 for the terms)."""

    def __init__(self, faces_left, faces_right, coord_weight=10.0, dense_weight=1.0, class_weight=(0.1, 0.45, 0.45)):
        self.faces = (torch.as_tensor(faces_left), torch.as_tensor(faces_right))
        self.coord_weight, self.dense_weight, self.class_weight = coord_weight, dense_weight, tuple(class_weight)

    def __call__(self, iter_outs, decode_list, target, meta_info):
        """iter_outs: the per-stage dicts; decode_list: {'seg', 'dense'}.  Returns {name: 0-d tensor} with the reference's keys."""
        loss = {}
        d = dense_losses(decode_list['seg'], decode_list['dense'], target['seg'], target['dense'], self.class_weight,
                         self.dense_weight)
        loss['seg'], loss['dense'], loss['lovasz'] = d[0], d[1], d[2]
        for index, out in enumerate(iter_outs):
            t = stage_losses(out, target, meta_info, self.faces, self.coord_weight)
            for i, k in enumerate(STAGE_KEYS):
                if k == 'offset' and out.get('pd_offset') is None:
                    continue
                loss['%s_%d' % (k, index)] = t[i]
        return loss


# ----------------------------------------------------------------------------------------------- gradients w.r.t. the predictions
def vertex_face_csr(faces, n_vertices=778):
    """CSR lists vertex -> (face * 3 + corner) of a [F,3] triangle table (int32 tensors on the faces' device): the order in which
    dir_stage_losses_backward sums a vertex's triangle contributions."""
    f = faces.to(torch.int64).reshape(-1).cpu()
    order = torch.argsort(f, stable=True)
    counts = torch.bincount(f, minlength=n_vertices)
    off = torch.zeros(n_vertices + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(counts, 0)
    return off.to(torch.int32).to(faces.device), order.to(torch.int32).to(faces.device)


def stage_loss_grads(pred, target, meta_info, faces, coord_weight=10.0, grad_out=None, csr=None):
    """Gradients of sum_k grad_out[k] * term_k (grad_out None = ones) of one stage w.r.t. pd_joint_uv_*, pd_mesh_uv_*, pd_joint_xyz_*,
    pd_mesh_xyz_* and pd_offset -- what autograd returns through the reference's loss modules.  pd_mesh_uv_* must be in `pred` (it
    is an independent input of this gradient).  Returns a dict with the prediction keys."""
    p, g, keep, off, B, fs = _marshal_stage(pred, target, meta_info, faces, True)
    o = _capi.LossPredGrad()
    out = {}
    for field, prefix in (('joint_uv', 'pd_joint_uv_'), ('mesh_uv', 'pd_mesh_uv_'), ('joint_xyz', 'pd_joint_xyz_'), ('mesh_xyz', 'pd_mesh_xyz_')):
        gs = [torch.empty_like(_capi.f32c(pred[prefix + s_])) for s_ in SIDES]
        setattr(o, field, (C.c_void_p * 2)(*[_capi.ptr(t) for t in gs]))
        for s_, t in zip(SIDES, gs):
            out[prefix + s_] = t
    goff = torch.empty(B, 3, device=keep[0].device)
    o.offset = _capi.ptr(goff)
    if off is not None:
        out['pd_offset'] = goff
    off = keep[0]                     # device anchor below
    if csr is None:
        # vertex -> triangle lists: built on the host (a synchronisation and a CPU sort) ONCE per faces tensor object and version, kept on the
        # tensor itself like _check_faces' verdict -- round 4 found a training step rebuilding them six times (three stages x two hands), each a
        # host round trip that drained the GPU in the middle of the backward pass
        csr = []
        for f, src in zip(fs, faces):
            tag = (src._version, str(f.device)) if torch.is_tensor(src) else None
            hit = getattr(src, '_dir_faces_csr', None) if tag is not None else None
            if hit is None or hit[0] != tag:
                hit = (tag, vertex_face_csr(f))
                if tag is not None:
                    src._dir_faces_csr = hit
            csr.append(hit[1])
    offs = (C.c_void_p * 2)(*[_capi.ptr(c[0]) for c in csr])
    idxs = (C.c_void_p * 2)(*[_capi.ptr(c[1]) for c in csr])
    go = None if grad_out is None else _capi.f32c(grad_out)
    with torch.cuda.device(off.device):
        _capi.check(_capi.lib().dir_stage_losses_backward(C.byref(p), C.byref(g), float(coord_weight), None if go is None else _capi.ptr(go),
                                                          C.byref(offs), C.byref(idxs), C.byref(o), off.shape[0], _capi.stream_ptr()),
                    'dir_stage_losses_backward')
    return out


def dense_loss_grads(seg_logits, dense_pred, gt_seg, gt_dense, class_weight=(0.1, 0.45, 0.45), dense_weight=1.0, grad_out=None):
    """Gradients of grad_out . (seg, dense, lovasz) w.r.t. the seg logits and the dense prediction -> (grad_seg, grad_dense)"""
    seg, dense = _capi.f32c(seg_logits), _capi.f32c(dense_pred)
    gs, gd = _capi.f32c(gt_seg.to(seg.device)), _capi.f32c(gt_dense.to(seg.device))
    _capi.require_cuda(seg, dense, gs, gd)
    B, Cc, S, S2 = seg.shape
    if Cc != 3 or S != S2 or dense.shape != seg.shape or gs.shape[:2] != (B, 1) or gd.shape[:2] != (B, 3) or gs.shape[2:] != gd.shape[2:]:
        raise _capi.DirHipError('dense_loss_grads: expected seg / dense [B,3,S,S], gt_seg [B,1,H,W], gt_dense [B,3,H,W]')
    H, W = gs.shape[2:]
    L = _capi.lib()
    go = None if grad_out is None else _capi.f32c(grad_out)
    with torch.cuda.device(seg.device):
        nbytes = L.dir_dense_losses_backward_workspace_bytes(B, S)
        if nbytes < 0:
            raise _capi.DirHipError('dir_dense_losses_backward_workspace_bytes failed')
        ws = torch.empty(nbytes, device=seg.device, dtype=torch.uint8)
        gseg, gdense = torch.empty_like(seg), torch.empty_like(dense)
        _capi.check(L.dir_dense_losses_backward(_capi.ptr(seg), _capi.ptr(dense), _capi.ptr(gs), _capi.ptr(gd), (C.c_float * 3)(*class_weight),
                                                float(dense_weight), None if go is None else _capi.ptr(go), _capi.ptr(ws), nbytes,
                                                _capi.ptr(gseg), _capi.ptr(gdense), B, S, H, W, _capi.stream_ptr()), 'dir_dense_losses_backward')
    return gseg, gdense
