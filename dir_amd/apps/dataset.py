"""Input side of the evaluation loop: the prepared InterHand2.6M split on disk -> batches on the GPU.

  layout            dataset/prepare_data.py:123-166   <data_path>/<split>/img/<idx>.jpg   256x256 crop written by cv.imwrite
                                                      <data_path>/<split>/anno/<idx>.pkl  {'camera': {'R','t','camera'},
                                                                                           'mano_params': {'left'|'right': {'R','pose','shape','trans'}}}
  InterHandSplit    dataset/interhand.py:31-105       InterHand_dataset.__getitem__: image + annotation of one index
  frame decode      apps/eval.py:56-58                cv.imread -> (cv.resize to 256, the identity on the prepared crops) -> BGR uint8
  DecodeRing                                          what the reference leaves to DataLoader(num_workers=8): decode processes fill a
                                                      ring of pinned uint8 batches [B,256,256,3]; the uint8 batch goes to the GPU as it
                                                      is (4x fewer bytes than the float tensor) and the normalisation of apps/eval.py:
                                                      59-61 happens inside the stem kernel (dir_stem_pool_forward)
  gt_batch          dataset/interhand.py:62-95        ground-truth vertices / joints: the GT MANO layer on the whole batch on the GPU
                                                      (dir_gt_mano_forward) instead of per sample on a CPU worker, then the camera
                                                      transform and projection
JPEG decoding is libjpeg-turbo through PIL (cv.imread uses the same library with the same defaults: integer slow DCT, fancy
upsampling).  cv2 is not installed where this was written, so the equality of the two decoders' pixels is not pinned by a fixture
("parity unpinned" for the decode step; everything after the uint8 frame is pinned by G11 / G9 / G10).  `resize_bilinear_u8`
restates cv.resize(INTER_LINEAR) on uint8 (OpenCV's 11-bit fixed-point coefficients, and its INTER_AREA fast path for an exact 2x
decimation) for frames that are not already 256x256; it is held to a scalar restatement of the published OpenCV algorithm
(oracle/resize.py) and to hand-worked vectors (tests/test_dataset.py) -- residual risk: the library's SIMD paths are written to be
bit-equal to its scalar code, but that, like the decoder, is not checkable without the library.  On the prepared split the resize is
never taken.

  prepared uint8 split (`write_u8_shards` / `ShardRing`)   an explicit SECOND form of the prepared split, made once by
                                                      `python -m dir_amd.apps.dataset prepare-u8 <data_path> <split>` next to
                                                      dataset/prepare_data.py's output: <split>/u8/frames_%05d.npy (uint8
                                                      [n,256,256,3], the decoded BGR crops) + annos_%05d.npy (float32 [n,155]) +
                                                      index.json.  JPEG decode is the from-files bottleneck (9.5 k images/s on 12
                                                      cores against ~28 k on the GPU); the shards are read back at memory-copy speed.
"""
import glob
import os
import pickle
import queue

import numpy as np
import torch

IMG_SIZE = 256          # dataset/dataset_utils.py:5


def resize_bilinear_u8(img, wo, ho):
    """cv.resize(img, (wo, ho)) for uint8 HxWxC, INTER_LINEAR: half-pixel centres, coefficients quantised to 2^-11, the horizontal pass
    kept at full precision, one rounding at the end (OpenCV's HResizeLinear / VResizeLinear for 8-bit images)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    if (w, h) == (wo, ho):
        return img.copy()
    if w == 2 * wo and h == 2 * ho:
        # cv::resize switches INTER_LINEAR to the INTER_AREA fast path for an exact 2x decimation: 2x2 box average, (sum + 2) >> 2
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(n_in, n_out):
        scale = n_in / float(n_out)
        src = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)     # the library computes fx in float
        i0 = np.floor(src).astype(np.int64)
        f = (src - i0.astype(np.float32)).astype(np.float32)
        neg = i0 < 0
        i0[neg], f[neg] = 0, 0.0
        over = i0 >= n_in - 1
        i0[over], f[over] = n_in - 1, 0.0
        i1 = np.minimum(i0 + 1, n_in - 1)
        c1 = np.rint(f * 2048.0).astype(np.int64)            # saturate_cast<short>(f * INTER_RESIZE_COEF_SCALE)
        return i0, i1, 2048 - c1, c1
    x0, x1, a0, a1 = taps(w, wo)
    y0, y1, b0, b1 = taps(h, ho)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None] if img.ndim == 3 else src[:, x0] * a0[None] + src[:, x1] * a1[None]
    top, bot = rows[y0], rows[y1]
    bb0 = b0.reshape((-1, 1, 1) if img.ndim == 3 else (-1, 1))
    bb1 = b1.reshape((-1, 1, 1) if img.ndim == 3 else (-1, 1))
    out = (((bb0 * (top >> 4)) >> 16) + ((bb1 * (bot >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def decode_bgr(path, size=IMG_SIZE):
    """cv.imread(path) (+ cv.resize to size x size when needed): uint8 BGR [size, size, 3]"""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    return bgr if bgr.shape[:2] == (size, size) else resize_bilinear_u8(bgr, size, size)


class InterHandSplit(object):
    """the prepared split on disk (dataset/interhand.py:31-47): len() = number of annotation files; frame(i) / anno(i) read one index"""

    def __init__(self, data_path, split='test'):
        assert split in ('train', 'test', 'val')
        self.data_path, self.split = data_path, split
        self.size = len(glob.glob(os.path.join(data_path, split, 'anno', '*.pkl')))

    def __len__(self):
        return self.size

    def img_path(self, idx):
        return os.path.join(self.data_path, self.split, 'img', '%d.jpg' % idx)

    def frame(self, idx):
        return decode_bgr(self.img_path(idx))

    def anno(self, idx):
        """-> 31 + 2 x 67 float32 numbers: camera R (9) | t (3) | K (9) | per hand (left, right): root R (9) | pose (45) | shape (10) | trans (3)"""
        with open(os.path.join(self.data_path, self.split, 'anno', '%d.pkl' % idx), 'rb') as f:
            d = pickle.load(f, encoding='latin1')
        cam = d['camera']
        parts = [np.asarray(cam['R'], np.float32).reshape(9), np.asarray(cam['t'], np.float32).reshape(3),
                 np.asarray(cam['camera'], np.float32).reshape(9)]
        for side in ('left', 'right'):
            p = d['mano_params'][side]
            pose = np.asarray(p['pose'], np.float32).reshape(-1)
            assert pose.size == 45, 'the prepared split stores 45 PCA coefficients per hand (dataset/prepare_data.py:100)'
            parts += [np.asarray(p['R'], np.float32).reshape(9), pose, np.asarray(p['shape'], np.float32).reshape(10),
                      np.asarray(p['trans'], np.float32).reshape(3)]
        return np.concatenate(parts)


ANNO_FLOATS = 21 + 2 * 67


def _decode_worker(data_path, split, wid, workers, indices, bs, chunk, ready, frames, annos, flags, ctrl, records=False):
    """decode process `wid` of `workers`: owns the chunks t = wid, wid + workers, ... of the flat (batch, chunk) sequence; chunk t of batch b
    goes to rows [k * chunk, ...) of ring slot b % depth once batch b - depth has been released (ctrl[0] = batches released); flags[t] = 1 when
    its frames and annotations are in place (2: failed).  No queue in the steady state: shared-memory words only.
    Ordering: the frame bytes and then the flag are plain stores from this process, the consumer polls with plain loads -- correct on x86 (total
    store order: stores become visible in program order), which is what every MI355X host is; a weakly ordered host would need a release
    store for the flag."""
    import time
    torch.set_num_threads(1)
    nice = int(os.environ.get('DIR_RING_NICE', '0'))          # tuning aid: run the decoders below the consumer's priority (the consumer issues the GPU's work)
    if nice > 0:
        try:
            os.nice(nice)
        except OSError:
            pass
    ds = InterHandSplit(data_path, split)
    if records:
        from .jpeg import file_to_record
    fr, an = [f.numpy() for f in frames], [a.numpy() for a in annos]
    fl, ct = flags.numpy(), ctrl.numpy()
    depth, n, npb = len(fr), len(indices), -(-bs // chunk)
    nb = -(-n // bs)
    ready.put(wid)                      # the interpreter and its imports are up
    for t in range(wid, nb * npb, workers):
        b, k = divmod(t, npb)
        lo = b * bs + k * chunk
        hi = min(lo + chunk, (b + 1) * bs, n)
        if lo >= hi:
            fl[t] = 1
            continue
        while b - depth >= ct[0]:       # the slot still holds a batch the consumer has not released
            if ct[1]:
                return
            time.sleep(0.0002)
        if ct[1]:
            return
        try:
            slot, j0 = b % depth, k * chunk
            for j in range(hi - lo):
                if records:                 # entropy decode only (libdir_jpeg.so): the row becomes a coefficient record, decoded further on the GPU
                    file_to_record(ds.img_path(indices[lo + j]), fr[slot][j0 + j], IMG_SIZE)
                else:
                    fr[slot][j0 + j] = ds.frame(indices[lo + j])
                an[slot][j0 + j] = ds.anno(indices[lo + j])
            fl[t] = 1
        except Exception:               # noqa: BLE001  (the consumer raises when it sees the flag)
            fl[t] = 2
            return


class DecodeRing(object):
    """Batches of decoded frames from `workers` processes through a ring of `depth` shared, page-locked host buffers.

        ring = DecodeRing(data_path, 'test', batch_size=256)
        for frames_u8, annos, n in ring:          # pinned uint8 [B,256,256,3], float32 [B,155]; the first n rows are valid
            pipe.refill(slot, frames_u8[:n]) ...
    A batch is cut into chunks of `chunk` images; chunk t of the flat (batch, chunk) sequence belongs to worker t % workers, which writes it
    into the ring slot of its batch as soon as that slot's previous batch has been released, and raises a flag word in shared memory; the
    consumer polls the flags of the batch it wants.  There is no queue traffic after start-up: rounds 2-3 handed every batch to the workers
    through multiprocessing queues (one task per worker and batch, completions of later batches re-queued) and the rate FELL with the worker
    count (256-thread host: 12 workers 11.7 k images/s, 24: 8.5 k, 96: 3.4 k; with 32-image tasks still 17.1 k at 16 and 6.1 k at 96 --
    profiles/r04_fromdisk_sweep_*.txt).  A batch counts as released when the consumer asks for the next one (it must have finished copying out
    of the buffer by then, as evaluate_from_disk does).  One pass per ring.  The buffers are shared memory registered with the HIP runtime
    (cudaHostRegister), so the host -> device copy is an asynchronous DMA from where the decoders wrote."""

    def __init__(self, data_path, split='test', batch_size=256, workers=8, depth=None, indices=None, pin=True, chunk=32, records=False):
        import torch.multiprocessing as mp
        self.ds = InterHandSplit(data_path, split)
        self.indices = list(range(len(self.ds))) if indices is None else list(indices)
        self.bs, self.workers, self.chunk = batch_size, max(1, workers), max(1, min(chunk, batch_size))
        if depth is None:
            depth = max(3, -(-self.workers * self.chunk // batch_size) + 2)
        self.depth = depth
        self.npb = -(-batch_size // self.chunk)
        # records=True (round 5): the workers only decode the Huffman stream (apps/jpeg.py); a row is then a coefficient record of `record_bytes`
        # bytes (the size of the frame for 4:2:0) that dir_jpeg_decode_records turns into the frame on the GPU
        self.records = bool(records)
        if self.records:
            from .jpeg import record_bytes
            self.record_bytes = record_bytes(IMG_SIZE)
            self.frames = [torch.zeros(batch_size, self.record_bytes, dtype=torch.uint8).share_memory_() for _ in range(depth)]
        else:
            self.frames = [torch.zeros(batch_size, IMG_SIZE, IMG_SIZE, 3, dtype=torch.uint8).share_memory_() for _ in range(depth)]
        self.annos = [torch.zeros(batch_size, ANNO_FLOATS, dtype=torch.float32).share_memory_() for _ in range(depth)]
        self.flags = torch.zeros(max(1, len(self) * self.npb), dtype=torch.uint8).share_memory_()
        self.ctrl = torch.zeros(2, dtype=torch.int64).share_memory_()          # [batches released, stop]
        self.pinned = False
        if pin and torch.cuda.is_available():
            rt = torch.cuda.cudart()
            self.pinned = all(int(rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)) == 0 for t in self.frames + self.annos)
        ctx = mp.get_context('spawn')
        ready = ctx.Queue()
        self.procs = [ctx.Process(target=_decode_worker, args=(data_path, split, w, self.workers, self.indices, self.bs, self.chunk, ready, self.frames,
                                                               self.annos, self.flags, self.ctrl, self.records), daemon=True) for w in range(self.workers)]
        for p in self.procs:
            p.start()
        for _ in self.procs:             # wait until every decoder is up: a spawned interpreter takes seconds to import
            ready.get(timeout=600)
        self._used = False

    def __len__(self):
        return (len(self.indices) + self.bs - 1) // self.bs

    def __iter__(self):
        import time
        assert not self._used, 'DecodeRing: one pass per ring (the workers walk the index list once)'
        self._used = True
        fl, ct = self.flags.numpy(), self.ctrl.numpy()
        nb, n_all = len(self), len(self.indices)
        for b in range(nb):
            ct[0] = b                                    # batches < b are released: their slots may be overwritten
            mine = fl[b * self.npb:(b + 1) * self.npb]
            t0 = time.perf_counter()
            while not mine.all():
                if (mine == 2).any() or not all(p.is_alive() or p.exitcode == 0 for p in self.procs):
                    raise RuntimeError('DecodeRing: a decode worker failed')
                if time.perf_counter() - t0 > 600:
                    raise RuntimeError('DecodeRing: decode workers made no progress for 600 s')
                time.sleep(0.0001)
            if (mine == 2).any():
                raise RuntimeError('DecodeRing: a decode worker failed')
            slot = b % self.depth
            yield self.frames[slot], self.annos[slot], min(self.bs, n_all - b * self.bs)
        ct[0] = nb

    def close(self):
        self.ctrl[1] = 1
        for p in self.procs:
            p.join(timeout=10)
        if self.pinned:
            rt = torch.cuda.cudart()
            for t in self.frames + self.annos:
                rt.cudaHostUnregister(t.data_ptr())
            self.pinned = False


def gt_batch(mano_layer, annos):
    """dataset/interhand.py:62-95 for a batch: annos float32 [n,155] on the GPU (InterHandSplit.anno layout); mano_layer = {'left','right'}
    GT layers (dir_amd.models.manolayer.ManoLayer, center_idx=None).  Returns the dataloader tuple entries of apps/eval.py:63-78 after
    the image: (joints_left, verts_left, joints_right, verts_right, joints2d_left, verts2d_left, joints2d_right, verts2d_right, cam)."""
    n = annos.shape[0]
    R = annos[:, 0:9].reshape(n, 3, 3)
    T = annos[:, 9:12].reshape(n, 1, 3)
    K = annos[:, 12:21].reshape(n, 3, 3).contiguous()
    out = {}
    for h, side in enumerate(('left', 'right')):
        o = 21 + 67 * h
        root = annos[:, o:o + 9].reshape(n, 3, 3).contiguous()
        pose, shape, trans = annos[:, o + 9:o + 54].contiguous(), annos[:, o + 54:o + 64].contiguous(), annos[:, o + 64:o + 67].contiguous()
        v, j = mano_layer[side](root, pose, shape, trans=trans)
        v = torch.baddbmm(T, v, R.transpose(1, 2))            # handV @ R.T + T
        j = torch.baddbmm(T, j, R.transpose(1, 2))
        v2 = torch.bmm(v, K.transpose(1, 2))
        j2 = torch.bmm(j, K.transpose(1, 2))
        out[side] = (j, v, j2[..., :2] / j2[..., 2:], v2[..., :2] / v2[..., 2:])
    (jl, vl, j2l, v2l), (jr, vr, j2r, v2r) = out['left'], out['right']
    return jl, vl, jr, vr, j2l, v2l, j2r, v2r, K


def gt_layers_from_checkpoint(state, device='cuda'):
    """The two ground-truth MANO layers (apps/eval.py:110-113) built from the MANO buffers the published checkpoint already carries
    (init_regressor.mano_layer_{left,right}.th_*) -- no licensed pickle needed.  The checkpoint's left shapedirs are
    already the `fix_shape`-corrected ones (models/dir.py:306-309 ran before it was saved), i.e. what dataset/interhand.py:19-22
    applies to the GT layer."""
    from ..models.manolayer import ManoLayer
    layers = {}
    for side in ('left', 'right'):
        p = 'init_regressor.mano_layer_%s.' % side
        g = lambda k: state[p + k].detach().cpu().float().numpy()  # noqa: E731
        comps = g('th_selected_comps')
        t = dict(hands_components=comps, hands_mean=g('th_hands_mean').reshape(45), J_regressor=g('th_J_regressor'),
                 weights=g('th_weights'), posedirs=g('th_posedirs'), v_template=g('th_v_template').reshape(778, 3),
                 shapedirs=g('th_shapedirs'), f=state[p + 'th_faces'].cpu().numpy() if (p + 'th_faces') in state else np.zeros((1538, 3), np.int64),
                 kintree_table=np.array([[4294967295, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], list(range(16))], dtype=np.int64))
        t['J'] = t['J_regressor'] @ t['v_template']
        layers[side] = ManoLayer(None, center_idx=None, _tables=t).to(device)
    return layers


# ------------------------------------------------------------------------------------------------- prepared uint8 split
U8_DIR = 'u8'


def write_u8_shards(data_path, split='test', shard_size=2048, workers=8, progress=None):
    """One-off preparation step next to dataset/prepare_data.py:123-166: decode every <split>/img/<idx>.jpg exactly as `decode_bgr` does
    and store the uint8 BGR crops + the packed annotations in shards of `shard_size` images under <split>/u8/.  Returns the image count."""
    import json
    from concurrent.futures import ThreadPoolExecutor
    ds = InterHandSplit(data_path, split)
    out = os.path.join(data_path, split, U8_DIR)
    os.makedirs(out, exist_ok=True)
    n = len(ds)
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:      # libjpeg releases the GIL
        for s0 in range(0, n, shard_size):
            idx = list(range(s0, min(n, s0 + shard_size)))
            frames = np.stack(list(ex.map(ds.frame, idx)))
            annos = np.stack([ds.anno(i) for i in idx])
            np.save(os.path.join(out, 'frames_%05d.npy' % (s0 // shard_size)), frames)
            np.save(os.path.join(out, 'annos_%05d.npy' % (s0 // shard_size)), annos)
            if progress:
                progress(idx[-1] + 1)
    with open(os.path.join(out, 'index.json'), 'w') as f:
        json.dump({'count': n, 'shard_size': shard_size, 'image_size': IMG_SIZE, 'layout': 'uint8 BGR [n,256,256,3] / float32 [n,%d]' % ANNO_FLOATS}, f)
    return n


class ShardRing(object):
    """DecodeRing's interface over the prepared uint8 split: batches are gathered from memory-mapped shards into a ring of page-locked
    host buffers by a few copy threads (numpy copies release the GIL), `depth - 1` batches ahead of the consumer."""

    def __init__(self, data_path, split='test', batch_size=256, workers=4, depth=3, indices=None, pin=True):
        import json
        self.dir = os.path.join(data_path, split, U8_DIR)
        with open(os.path.join(self.dir, 'index.json')) as f:
            meta = json.load(f)
        self.count, self.ss = int(meta['count']), int(meta['shard_size'])
        nsh = (self.count + self.ss - 1) // self.ss
        self.fr = [np.load(os.path.join(self.dir, 'frames_%05d.npy' % i), mmap_mode='r') for i in range(nsh)]
        self.an = [np.load(os.path.join(self.dir, 'annos_%05d.npy' % i), mmap_mode='r') for i in range(nsh)]
        self.indices = list(range(self.count)) if indices is None else list(indices)
        self.bs, self.depth, self.workers = batch_size, depth, max(1, workers)
        pinned = pin and torch.cuda.is_available()
        self.frames = [torch.zeros(batch_size, IMG_SIZE, IMG_SIZE, 3, dtype=torch.uint8, pin_memory=pinned) for _ in range(depth)]
        self.annos = [torch.zeros(batch_size, ANNO_FLOATS, dtype=torch.float32, pin_memory=pinned) for _ in range(depth)]
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=self.workers)

    def __len__(self):
        return (len(self.indices) + self.bs - 1) // self.bs

    def _fill(self, slot, rows):
        fr, an = self.frames[slot].numpy(), self.annos[slot].numpy()
        for j, idx in rows:
            sh, r = divmod(idx, self.ss)
            fr[j] = self.fr[sh][r]
            an[j] = self.an[sh][r]

    def _submit(self, b, slot):
        rows = list(enumerate(self.indices[b * self.bs:(b + 1) * self.bs]))
        futs = [self.pool.submit(self._fill, slot, rows[w::self.workers]) for w in range(self.workers) if rows[w::self.workers]]
        return futs, len(rows)

    def __iter__(self):
        nb = len(self)
        pending = {b: self._submit(b, b % self.depth) for b in range(min(self.depth - 1, nb))}
        for b in range(nb):
            nxt = b + self.depth - 1
            if nxt < nb:
                pending[nxt] = self._submit(nxt, nxt % self.depth)
            futs, n = pending.pop(b)
            for f in futs:
                f.result()
            yield self.frames[b % self.depth], self.annos[b % self.depth], n

    def close(self):
        self.pool.shutdown(wait=True)


if __name__ == '__main__':
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == 'prepare-u8':
        split = sys.argv[3] if len(sys.argv) > 3 else 'test'
        n = write_u8_shards(sys.argv[2], split, progress=lambda k: print('\r%d' % k, end='', flush=True))
        print('\nwrote %d frames under %s' % (n, os.path.join(sys.argv[2], split, U8_DIR)))
    else:
        print('usage: python -m dir_amd.apps.dataset prepare-u8 <data_path> [split]')
