"""GPU: dir_jpeg_decode_records (csrc/jpeg.hip) -- everything of cv.imread (apps/eval.py:56) after the Huffman decode -- is bit-exact with
libjpeg-turbo: the frames it produces from the host library's coefficient records equal the committed libjpeg pixels (g22_jpeg.npz, BGR order),
for 4:2:0 / 4:2:2 / 4:4:4 / grayscale, odd sizes and restart intervals, alone and in batches; pixel records pass through; a record of another
size raises the error word; and the evaluation loop from <split>/img/*.jpg gives the metrics of the host-decode path bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from dir_amd.apps import jpeg as AJ

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, 'tests', 'golden', 'g22_jpeg.npz')


def record_of(data, nbytes):
    rec = np.zeros(nbytes, np.uint8)
    assert AJ.host_lib().dir_jpeg_decode_coefficients(data, len(data), rec.ctypes.data, rec.size) == 0
    return rec


def test_device_decode_is_bit_exact_with_libjpeg():
    g = np.load(GOLD)
    for n in sorted(k[:-4] for k in g.files if k.endswith('.rgb')):
        ref = g[n + '.rgb']
        H, W = ref.shape[:2]
        stride = (512 + 3 * ((H + 15) // 16 * 16) * ((W + 15) // 16 * 16) * 2 + 15) // 16 * 16      # room for 4:4:4 at this size
        rec = torch.from_numpy(record_of(g[n + '.jpg'].tobytes(), stride)).cuda().unsqueeze(0)
        dec = AJ.RecordDecoder(1, stride, (H, W))
        out = dec(rec, torch.zeros(1, H, W, 3, dtype=torch.uint8, device='cuda'))
        dec.check()
        assert np.array_equal(out[0].cpu().numpy()[..., ::-1], ref), n             # BGR like cv.imread


def test_batches_pixel_records_and_the_error_word():
    g = np.load(GOLD)
    data = g['full_256.jpg'].tobytes()
    ref = g['full_256.rgb'][..., ::-1]
    stride = AJ.record_bytes(256)
    assert stride == 512 + 256 * 256 * 3
    B = 37
    recs = np.zeros((B, stride), np.uint8)
    for i in range(B):
        recs[i] = record_of(data, stride)
    px = np.random.RandomState(0).randint(0, 256, (256, 256, 3)).astype(np.uint8)
    recs[5, :512] = 0
    recs[5, :16].view(np.int32)[:4] = (AJ.MAGIC_PIXELS, 256, 256, 3)                   # a frame the host decoded itself
    recs[5, 512:] = px.reshape(-1)
    dec = AJ.RecordDecoder(64, stride, 256)
    out = dec(torch.from_numpy(recs).cuda(), torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device='cuda'))
    dec.check()
    o = out.cpu().numpy()
    assert all(np.array_equal(o[i], ref) for i in range(B) if i != 5) and np.array_equal(o[5], px)
    # a record of another geometry is reported, the others still decode
    small = g['s420_q92.jpg'].tobytes()
    recs[9] = 0
    recs[9, :512 + 6 * 4 * 4 * 128] = record_of(small, 512 + 6 * 4 * 4 * 128)
    out2 = dec(torch.from_numpy(recs).cuda(), torch.zeros(B, 256, 256, 3, dtype=torch.uint8, device='cuda'))
    from dir_amd._capi import DirHipError
    with pytest.raises(DirHipError):
        dec.check()
    assert np.array_equal(out2[8].cpu().numpy(), ref) and int(out2[9].max()) == 0


def test_evaluation_from_jpeg_files_equals_the_host_decode_path(tmp_path):
    """the whole loop (decode ring of coefficient records -> DMA -> dir_jpeg_decode_records into the slot's input -> forward -> metrics) against the
    rounds 2-4 path (PIL's libjpeg-turbo on the host): identical frames, so identical metrics; a progressive file in the split rides along as a
    pixel record"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from fake_split import write_split
    from PIL import Image
    from dir_amd import synth
    from dir_amd.apps import dataset as DS
    from dir_amd.apps import eval as EV
    from dir_amd.engine import DirEngine
    d = str(tmp_path)
    write_split(d, 24, seed=3)
    p3 = os.path.join(d, 'test', 'img', '3.jpg')
    Image.open(p3).save(p3, quality=90, progressive=True)
    # frames through both rings are the same bytes
    ring = DS.DecodeRing(d, 'test', batch_size=8, workers=2, records=True)
    dec = AJ.RecordDecoder(8, ring.record_bytes, 256)
    try:
        for b, (recs, annos, n) in enumerate(ring):
            fr = dec(recs.cuda(), torch.zeros(8, 256, 256, 3, dtype=torch.uint8, device='cuda'), n).cpu().numpy()
            for j in range(n):
                assert np.array_equal(fr[j], DS.decode_bgr(os.path.join(d, 'test', 'img', '%d.jpg' % (8 * b + j)))), (b, j)
        dec.check()
    finally:
        ring.close()
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    eng = DirEngine(state, dtype=torch.bfloat16)
    mano = DS.gt_layers_from_checkpoint(state)
    jreg = {s: EV.Jr(mano[s].J_regressor) for s in ('left', 'right')}
    m1, r1 = EV.evaluate_from_disk(eng, d, jreg, mano, bs=8, workers=2, source='jpeg')
    m2, r2 = EV.evaluate_from_disk(eng, d, jreg, mano, bs=8, workers=2, source='jpeg-host')
    assert r1['images'] == r2['images'] == 24
    assert m1.summarize() == m2.summarize()
