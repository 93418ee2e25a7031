"""Host-side mirror of the reference's apps/eval.py evaluation loop, on libdir_hip.so.

  Jr              apps/eval.py:22-44   same constructor / __call__; the matmul runs in dir_joint_regress_forward
  EvalMetrics     apps/eval.py:128-306 the per-batch maths (:151-241) is ONE dir_eval_metrics_forward launch per batch;
                                       per-sample errors stay on the GPU until summarize()/save_txt(), which reduce and
                                       format exactly like :246-306 (numpy float32 means, *1000 for mm, '%.3f' files)
  evaluate        apps/eval.py:137-241 the loop: network(...) -> metrics.update(...)
  evaluate_from_disk / main   apps/eval.py:88-136  the command line (`python -m dir_amd.apps.eval --model DIR.pth --data_path ... --bs 256
                                       --root_joint 0`): checkpoint -> DIR, the prepared split from disk through dir_amd.apps.dataset
                                       (decode ring -> uint8 frames -> two forwards in flight), GT on the GPU, the same report / files
The licensed MANO pickle is not needed: the checkpoint carries the MANO buffers (dataset.gt_layers_from_checkpoint).
"""
import os

import numpy as np
import torch

from .. import _capi

IMG_MEAN, IMG_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # apps/eval.py:49-50


def normalize_images(img_u8_bgr):
    """apps/eval.py:59-61 for a batch: uint8 BGR [B,H,W,3] (cv.imread / cv.resize output) -> float32 RGB NCHW, / 255,
    ImageNet-normalised; bit-identical to the reference's torch CPU result.  (DirEngine.forward / DIR accept the uint8 batch
    directly and fuse this into the stem staging.)"""
    import ctypes as C
    _capi.require_cuda(img_u8_bgr)
    if img_u8_bgr.dtype != torch.uint8 or img_u8_bgr.dim() != 4 or img_u8_bgr.shape[3] != 3:
        raise _capi.DirHipError('normalize_images: expected a uint8 [B,H,W,3] tensor')
    x = img_u8_bgr.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty(B, 3, H, W, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dir_image_normalize_forward(_capi.ptr(x), _capi.ptr(out), (C.c_float * 3)(*IMG_MEAN),
                                                            (C.c_float * 3)(*IMG_STD), B, H, W, _capi.stream_ptr()),
                    'dir_image_normalize_forward')
    return out


TIP_VERTS = (745, 317, 444, 556, 673)
NEW_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)


class Jr:
    """apps/eval.py:22-44"""

    def __init__(self, J_regressor, device='cuda'):
        self.device = device
        self.process_J_regressor(J_regressor)

    def process_J_regressor(self, J_regressor):
        J = J_regressor.clone().detach().float()
        tips = torch.zeros_like(J[:5])
        for r, v in enumerate(TIP_VERTS):
            tips[r, v] = 1.0
        self.J_regressor = torch.cat([J, tips], 0)[list(NEW_ORDER)].contiguous().to(self.device)

    def __call__(self, v):
        _capi.require_cuda(v, self.J_regressor)
        v = _capi.f32c(v)
        out = torch.empty(v.shape[0], 21, 3, device=v.device)
        with torch.cuda.device(v.device):
            _capi.check(_capi.lib().dir_joint_regress_forward(_capi.ptr(self.J_regressor), _capi.ptr(v), _capi.ptr(out),
                                                              v.shape[0], _capi.stream_ptr()), 'dir_joint_regress_forward')
        return out


_KEYS = ('joint_err', 'vert_err', 'joint2d_err', 'vert2d_err', 'joints_pd', 'joints_gt', 'root_err')
_SHAPES = {'joint_err': (2, 21), 'vert_err': (2, 778), 'joint2d_err': (2, 21), 'vert2d_err': (2, 778),
           'joints_pd': (2, 21, 3), 'joints_gt': (2, 21, 3), 'root_err': ()}


def eval_batch(J_regressor, verts_pd, pd_offset, verts_gt, verts2d_gt, cam, root_joint=0, scale=True):
    """apps/eval.py:151-241 for one batch.  J_regressor = {'left': Jr, 'right': Jr}; verts_* = {'left': [B,778,3], ...};
    verts2d_gt = {'left': [B,778,2], ...}; returns a dict of device tensors (hand axis: 0 = left, 1 = right)."""
    ts = [_capi.f32c(verts_pd[s]) for s in ('left', 'right')] + [_capi.f32c(pd_offset)] + \
         [_capi.f32c(verts_gt[s]) for s in ('left', 'right')] + [_capi.f32c(verts2d_gt[s]) for s in ('left', 'right')] + \
         [_capi.f32c(cam), J_regressor['left'].J_regressor, J_regressor['right'].J_regressor]
    _capi.require_cuda(*ts)
    B, dev = ts[0].shape[0], ts[0].device
    for t, shp in zip(ts, [(B, 778, 3)] * 2 + [(B, 3)] + [(B, 778, 3)] * 2 + [(B, 778, 2)] * 2 + [(B, 3, 3), (21, 778), (21, 778)]):
        if tuple(t.shape) != shp:
            raise _capi.DirHipError('eval_batch: tensor of shape %s where %s is expected' % (tuple(t.shape), shp))
    ins = _capi.EvalInputs()
    p = [t.data_ptr() for t in ts]
    ins.verts_pd[0], ins.verts_pd[1], ins.pd_offset = p[0], p[1], p[2]
    ins.verts_gt[0], ins.verts_gt[1], ins.verts2d_gt[0], ins.verts2d_gt[1] = p[3], p[4], p[5], p[6]
    ins.cam, ins.jr[0], ins.jr[1] = p[7], p[8], p[9]
    res = {k: torch.empty((B,) + _SHAPES[k], device=dev) for k in _KEYS}
    outs = _capi.EvalOutputs(*[res[k].data_ptr() for k in _KEYS])
    with torch.cuda.device(dev):
        _capi.check(_capi.lib().dir_eval_metrics_forward(ins, outs, B, int(root_joint), int(bool(scale)), _capi.stream_ptr()),
                    'dir_eval_metrics_forward')
    return res


class EvalMetrics:
    """The accumulators of apps/eval.py:128-136 and their reduction (:246-306)."""

    def __init__(self, J_regressor, root_joint=0, scale=True, stage_num=3):
        self.J_regressor, self.root_joint, self.scale, self.stage_num = J_regressor, root_joint, scale, stage_num
        self.batches = []

    def update(self, result, data):
        """`result` = network(...)[0]; `data` = the dataloader tuple of apps/eval.py:139-149."""
        r = result[self.stage_num - 1]
        out = eval_batch(self.J_regressor, {'left': r['pd_mesh_xyz_left'], 'right': r['pd_mesh_xyz_right']}, r['pd_offset'],
                         {'left': data[3].cuda(), 'right': data[5].cuda()},
                         {'left': data[7].cuda(), 'right': data[9].cuda()}, data[10].cuda(), self.root_joint, self.scale)
        self.batches.append(out)
        return out

    def arrays(self):
        """the concatenated numpy arrays of :246-270, named as the reference names them."""
        cat = {k: torch.cat([b[k] for b in self.batches], 0).cpu().numpy() for k in _KEYS}
        a = {}
        for h, side in enumerate(('left', 'right')):
            a['joints_loss_' + side] = cat['joint_err'][:, h]
            a['verts_loss_' + side] = cat['vert_err'][:, h]
            a['joints_2d_loss_' + side] = cat['joint2d_err'][:, h]
            a['verts_2d_loss_' + side] = cat['vert2d_err'][:, h]
            a['joints_xyz_%s_pd' % side] = cat['joints_pd'][:, h].reshape(-1, 63)
            a['joints_xyz_%s_gt' % side] = cat['joints_gt'][:, h].reshape(-1, 63)
        a['root_loss'] = cat['root_err'].reshape(-1)
        return a

    def summarize(self):
        """:285-306 (the printed numbers)."""
        a, s = self.arrays(), {}
        for nm, key, mul in (('joint_mm', 'joints_loss_', 1000), ('vert_mm', 'verts_loss_', 1000),
                             ('joint_px', 'joints_2d_loss_', 1), ('vert_px', 'verts_2d_loss_', 1)):
            l, r = a[key + 'left'].mean() * mul, a[key + 'right'].mean() * mul
            s[nm] = {'left': float(l), 'right': float(r), 'all': float((l + r) / 2)}
        s['root_mm'] = float(a['root_loss'].mean() * 1000)
        return s

    def save_txt(self, file_folder):
        """:272-283: the same twelve text files."""
        os.makedirs(file_folder, exist_ok=True)
        a = self.arrays()
        w = lambda n, x: np.savetxt(os.path.join(file_folder, n), x, fmt='%.3f')  # noqa: E731
        w('left_joint.txt', a['joints_xyz_left_pd'] * 1000)
        w('right_joint.txt', a['joints_xyz_right_pd'] * 1000)
        w('joint_left_error.txt', a['joints_loss_left'] * 1000)
        w('joint_right_error.txt', a['joints_loss_right'] * 1000)
        w('mesh_left_error.txt', a['verts_loss_left'].mean(-1) * 1000)
        w('mesh_right_error.txt', a['verts_loss_right'].mean(-1) * 1000)
        w('joint_2d_left_error.txt', a['joints_2d_loss_left'])
        w('joint_2d_right_error.txt', a['joints_2d_loss_right'])
        w('mesh_2d_left_error.txt', a['verts_2d_loss_left'].mean(-1))
        w('mesh_2d_right_error.txt', a['verts_2d_loss_right'].mean(-1))
        w('root_loss.txt', a['root_loss'] * 1000)
        w('volume.txt', a['joints_loss_right'] * 1000)

    def report(self):
        """the print block of :295-306"""
        s = self.summarize()
        lines = []
        for title, key, unit in (('joint mean error:', 'joint_mm', 'mm'), ('vert mean error:', 'vert_mm', 'mm'),
                                 ('pixel joint mean error:', 'joint_px', 'mm'), ('pixel vert mean error:', 'vert_px', 'mm')):
            lines += [title, '    left: {} {u}, right: {} {u}'.format(s[key]['left'], s[key]['right'], u=unit),
                      '    all: {} {u}'.format(s[key]['all'], u=unit)]
        lines.append('root error: {} mm'.format(s['root_mm']))
        return '\n'.join(lines)


def evaluate(network, dataloader, J_regressor, root_joint=0, scale=True, stage_num=3):
    """apps/eval.py:137-241."""
    m = EvalMetrics(J_regressor, root_joint, scale, stage_num)
    with torch.no_grad():
        for data in dataloader:
            result, _ = network({'img': data[0].cuda()}, None, None)
            m.update(result, data)
    return m


# tuning aid (tools/bench_fromdisk.py): DIR_EVAL_SKIP_DEVICE_DECODE=1 leaves the device half of the JPEG decode out of the loop (the forwards then see
# stale frames: rates only) -- what the two decode kernels cost the loop under contention
_SKIP_DEVICE_DECODE = __import__('os').environ.get('DIR_EVAL_SKIP_DEVICE_DECODE', '0') == '1'


def evaluate_from_disk(eng, data_path, J_regressor, mano_layer, bs=256, root_joint=0, scale=True, split='test', workers=8,
                       stage_num=3, indices=None, progress=None, source='jpeg', nslot=3):
    """apps/eval.py:121-241 from the prepared split on disk, at pipeline speed: decode processes (dataset.DecodeRing) -> pinned uint8
    batches -> two forwards in flight (engine.ForwardPipeline over uint8 input slots; the normalisation runs inside the stem kernel,
    proj_feat is not produced: the evaluation never reads it) -> GT MANO + metrics on the GPU.  `eng`: a DirEngine.
    Returns (EvalMetrics, {'images', 'seconds', 'images_per_sec'})."""
    import time
    from ..engine import ForwardPipeline
    from .dataset import IMG_SIZE, DecodeRing, ShardRing, gt_batch
    dev = eng.device
    # 'jpeg': the host decodes the Huffman stream only, the GPU does the rest of cv.imread (apps/jpeg.py, round 5); 'jpeg-host': the whole decode
    # on the host (PIL's libjpeg-turbo, rounds 2-4); 'u8': the prepared uint8 split (dataset.write_u8_shards), no JPEG decode in the loop
    assert source in ('jpeg', 'jpeg-host', 'u8')
    if source == 'u8':
        ring = ShardRing(data_path, split, bs, workers=workers, indices=indices)
    else:
        ring = DecodeRing(data_path, split, bs, workers=workers, indices=indices, records=(source == 'jpeg'))
    m = EvalMetrics(J_regressor, root_joint, scale, stage_num)
    # nslot forwards in flight (round 5: 3 -- with 2, only ONE forward is on the GPU while the host scores the batch that just finished and hands the
    # next one over, a third of the time)
    nslot = max(2, int(nslot))
    slots = [torch.zeros(bs, IMG_SIZE, IMG_SIZE, 3, device=dev, dtype=torch.uint8) for _ in range(nslot)]
    rec_dev, rec_dec = None, None
    if source == 'jpeg':
        from .jpeg import RecordDecoder
        rec_dev = [torch.zeros(bs, ring.record_bytes, device=dev, dtype=torch.uint8) for _ in range(nslot)]
        rec_dec = [RecordDecoder(bs, ring.record_bytes, IMG_SIZE, dev) for _ in range(nslot)]
    pending = [None] * nslot
    pipe = None

    def finish(slot):
        n, annos = pending[slot]
        outs = pipe.wait(slot)
        res = [{k: (v[:n] if torch.is_tensor(v) else v) for k, v in o.items()} for o in outs[:3]]
        gt = gt_batch(mano_layer, annos[:n])
        m.update(res, (None,) * 2 + gt)          # data[0] image / data[1] mask are not read by the metric maths (apps/eval.py:151-241)
        pending[slot] = None

    t0, seen = time.perf_counter(), 0
    try:                                                               # everything that can raise while the decode workers run: the ring is closed on the way out
        import itertools
        batches = iter(ring)
        first = next(batches, None)
        if first is not None and eng.arith is not None and not eng.calibrated:
            # f16 arithmetic modes: the per-layer power-of-two operand scales come from the first batch, BEFORE the slots' graphs are captured
            # (the scales are launch arguments: a graph captured earlier would replay the uncalibrated ones)
            f0 = first[0].to(dev)
            # (only the first[2] valid rows: in a partial first batch the rows beyond it are still the ring's zero-initialised records, which
            # the device decoder would report as unusable -- and evaluate would raise after every image had been scored)
            n0 = int(first[2])
            eng.calibrate(rec_dec[0](f0, slots[0].clone(), n0)[:n0].contiguous() if source == 'jpeg' else f0[:n0].contiguous())
        pipe = ForwardPipeline(eng, slots, want_proj_feat=False)
        t0 = time.perf_counter()
        for k, (frames, annos, n) in enumerate(itertools.chain([first] if first is not None else [], batches)):
            slot = k % nslot
            if pending[slot] is not None:
                finish(slot)
            with torch.cuda.stream(pipe.streams[slot]):
                if source == 'jpeg':
                    rec_dev[slot].copy_(frames, non_blocking=True)     # the coefficient records by DMA
                else:
                    pipe.imgs[slot].copy_(frames, non_blocking=True)   # async DMA from the ring's page-locked buffer, on the slot's stream (= pipe.refill)
                annos_dev = annos.to(dev, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record()                                        # both DMAs done = the ring's buffer is free: recorded BEFORE any kernel of this
                if source == 'jpeg' and not _SKIP_DEVICE_DECODE:       # slot, so the host never waits behind compute that queues with the other slots' forwards
                    rec_dec[slot](rec_dev[slot], pipe.imgs[slot], n)   # the rest of the JPEG decode, straight into the slot's input
            pipe.launch(slot)
            pending[slot] = (n, annos_dev)
            copied.synchronize()                                       # the ring may hand this buffer back to the decoders
            seen += n
            if progress:
                progress(seen)
        for slot in [(k + 1 + j) % nslot for j in range(nslot)] if seen else ():      # flush in launch order: the oldest slot first
            if pending[slot] is not None:
                finish(slot)
        torch.cuda.synchronize(dev)
        for d_ in rec_dec or ():
            d_.check()                                                 # a record that was not a 256x256 image would have left its frame stale
        dt = time.perf_counter() - t0                                  # the loop's time: every image scored (joining the decode processes below is
    finally:                                                           # 0.3-0.4 s of interpreter tear-down, which round 5 found counted into the JPEG rate)
        ring.close()
    return m, {'images': seen, 'seconds': dt, 'images_per_sec': seen / dt if dt > 0 else 0.0}


def main(argv=None):
    """python -m dir_amd.apps.eval: the reference's command line (apps/eval.py:88-94) and outputs (:272-306)"""
    import argparse
    from ..engine import DirEngine
    from .dataset import gt_layers_from_checkpoint
    ap = argparse.ArgumentParser(description='InterHand2.6M evaluation of a DIR checkpoint on MI355X (apps/eval.py of the reference)')
    ap.add_argument('--model', type=str, default='./DIR.pth')
    ap.add_argument('--data_path', type=str, default='./data/interhand2.6m/')
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--root_joint', type=int, default=0)              # 0 wrist, 9 middle MCP
    ap.add_argument('--scale', type=lambda v: str(v).lower() not in ('0', 'false', 'no'), default=True)
    ap.add_argument('--dtype', choices=['f16', 'bf16', 'f32'], default='f16', help='f16 feature maps (the throughput mode of round 5: every stage inside 0.01 mm of the '
                    'reference), bf16 feature maps (rounds 1-4: 0.04-0.05 mm at the init stage, 2 %% faster) or exact fp32 (parity mode)')
    ap.add_argument('--arith', choices=['f16x3', 'f16'], default=None, help="with --dtype f32: convolutions on the f16 matrix cores -- 'f16x3' split "
                    "precision (the 1e-4 mm parity mode at 10 k images/s), 'f16' one MFMA per product (every stage within 0.01 mm, 13 k images/s)")
    ap.add_argument('--source', choices=['jpeg', 'jpeg-host', 'u8'], default='jpeg', help="jpeg: <split>/img/<idx>.jpg as the reference prepares them (Huffman decode on the host, the rest of the decode on the GPU); jpeg-host: the whole decode on the host; u8: the prepared "
                    "uint8 shards of dir_amd.apps.dataset.write_u8_shards (no decode in the loop)")
    ap.add_argument('--workers', type=int, default=16, help='decode processes (jpeg) / copy threads (u8); more than the CPUs the process may use is slower')
    ap.add_argument('--result_dir', type=str, default='./result/DIR-PoseEmb-Wrist')
    opt = ap.parse_args(argv)
    state = torch.load(opt.model, map_location='cpu', weights_only=False)
    state = state['net'] if isinstance(state, dict) and 'net' in state else state
    if opt.arith is not None and opt.dtype != 'f32':
        ap.error('--arith needs --dtype f32 (fp32 feature maps, f16 matrix-core arithmetic)')
    eng = DirEngine(state, dtype={'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[opt.dtype], root_joint=0, arith=opt.arith)     # apps/eval.py:104: DIR(21, './misc/mano')
    mano_layer = gt_layers_from_checkpoint(state)
    J_regressor = {s: Jr(mano_layer[s].J_regressor) for s in ('left', 'right')}
    m, rate = evaluate_from_disk(eng, opt.data_path, J_regressor, mano_layer, bs=opt.bs, root_joint=opt.root_joint, scale=opt.scale,
                                 workers=opt.workers, source=opt.source)
    m.save_txt(opt.result_dir)
    print(m.report())
    print('%d images in %.1f s: %.0f images/s from files' % (rate['images'], rate['seconds'], rate['images_per_sec']))
    return m


if __name__ == '__main__':
    main()
