"""CPU: the input side of the evaluation loop (dir_amd/apps/dataset.py): on-disk layout (dataset/prepare_data.py:123-166), frame decode
(apps/eval.py:56-58), the decode ring, cv.resize restatement."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers'))
from fake_split import write_split  # noqa: E402

from dir_amd.apps import dataset as DS  # noqa: E402


def test_resize_bilinear_u8():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (64, 48, 3)).astype(np.uint8)
    assert np.array_equal(DS.resize_bilinear_u8(img, 48, 64), img)            # same size: identity (what the prepared split hits)
    big = DS.resize_bilinear_u8(img, 96, 128)                                  # exact 2x: against the float formula, within the fixed point's 1 LSB
    src_y = np.clip((np.arange(128) + 0.5) / 2 - 0.5, 0, 63)
    src_x = np.clip((np.arange(96) + 0.5) / 2 - 0.5, 0, 47)
    y0, x0 = np.floor(src_y).astype(int), np.floor(src_x).astype(int)
    y1, x1 = np.minimum(y0 + 1, 63), np.minimum(x0 + 1, 47)
    fy, fx = (src_y - y0)[:, None, None], (src_x - x0)[None, :, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert big.shape == (128, 96, 3) and np.abs(big.astype(np.float64) - ref).max() <= 1.0
    flat = np.full((10, 10), 200, np.uint8)
    assert np.array_equal(DS.resize_bilinear_u8(flat, 256, 256), np.full((256, 256), 200, np.uint8))   # constants survive the fixed point
    small = DS.resize_bilinear_u8(img, 24, 32)                                 # 2x down: average of the two nearest (half-pixel centres)
    assert small.shape == (32, 24, 3)


def test_split_layout_and_decode(tmp_path):
    from PIL import Image
    write_split(str(tmp_path), 5)
    ds = DS.InterHandSplit(str(tmp_path), 'test')
    assert len(ds) == 5
    f = ds.frame(3)
    assert f.shape == (256, 256, 3) and f.dtype == np.uint8
    rgb = np.asarray(Image.open(ds.img_path(3)).convert('RGB'))
    assert np.array_equal(f[..., ::-1], rgb)                                   # BGR like cv.imread
    a = ds.anno(2)
    assert a.shape == (DS.ANNO_FLOATS,) and a.dtype == np.float32
    R = a[:9].reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5) and abs(a[11] - 0.8) < 1e-6 and a[12] == 1502.0
    assert np.allclose(a[21:30].reshape(3, 3) @ a[21:30].reshape(3, 3).T, np.eye(3), atol=1e-5)     # left root rotation


@pytest.mark.parametrize('bs,workers', [(4, 2), (3, 3), (16, 2)])
def test_decode_ring_yields_every_frame_in_order(tmp_path, bs, workers):
    write_split(str(tmp_path), 10, seed=bs)
    ds = DS.InterHandSplit(str(tmp_path))
    ring = DS.DecodeRing(str(tmp_path), 'test', batch_size=bs, workers=workers, depth=3, pin=False)
    try:
        seen = 0
        for frames, annos, n in ring:
            assert 1 <= n <= bs
            for j in range(n):
                assert np.array_equal(frames[j].numpy(), ds.frame(seen + j))
                assert np.array_equal(annos[j].numpy(), ds.anno(seen + j))
            seen += n
        assert seen == 10 and len(ring) == (10 + bs - 1) // bs
    finally:
        ring.close()


def test_resize_matches_the_scalar_restatement_of_opencv_and_hand_worked_vectors():
    """`resize_bilinear_u8` against oracle/resize.py (scalar loops following the published OpenCV 8-bit INTER_LINEAR:
    float32 source coordinates, cvRound(f * 2048) coefficients, the >> 4 / >> 16 / (+2) >> 2 fixed-point cast, the INTER_AREA fast path of an
    exact 2x decimation) and against vectors worked out by hand from those formulas."""
    from oracle.resize import cv_resize_linear_u8
    rng = np.random.RandomState(3)
    for (h, w, ho, wo) in [(7, 5, 11, 9), (16, 12, 8, 6), (9, 9, 20, 3), (10, 13, 4, 31), (5, 4, 5, 9), (33, 21, 16, 16), (3, 3, 7, 7)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(DS.resize_bilinear_u8(img, wo, ho), cv_resize_linear_u8(img, wo, ho)), (h, w, ho, wo)
    gray = rng.randint(0, 256, (6, 10)).astype(np.uint8)
    assert np.array_equal(DS.resize_bilinear_u8(gray, 7, 9), cv_resize_linear_u8(gray, 7, 9))
    # 1 x 4 -> 1 x 8 (vertical taps b0 = 2048, b1 = 0).  By hand: dx = 1: fx = 0.25 -> a = (1536, 512): 100 * 512 = 51200; >> 4 = 3200;
    # (2048 * 3200) >> 16 = 100; (100 + 2) >> 2 = 25.  dx = 5: 200 * 1536 + 255 * 512 = 437760; >> 4 = 27360; * 2048 >> 16 = 855; 857 >> 2 = 214
    # (the exact value is 213.75).  dx = 6: 200 * 512 + 255 * 1536 = 494080 -> 30880 -> 965 -> 967 >> 2 = 241 (241.25).  Ends clamp to the border pixel.
    row = np.array([[0, 100, 200, 255]], np.uint8)
    assert DS.resize_bilinear_u8(row, 8, 1).tolist() == [[0, 25, 75, 125, 175, 214, 241, 255]]
    # the truncation inside the fixed-point cast: 3 -> 5 columns of [1, 2, 4]: dx = 1: fx = (1.5 * 0.6 - 0.5) = 0.4 -> cvRound(819.2) = 819:
    # 1 * 1229 + 2 * 819 = 2867; >> 4 = 179; * 2048 >> 16 = 5; (5 + 2) >> 2 = 1 (exact 1.4); dx = 2: fx = 1.0 -> sx = 1, f = 0: 2.
    # dx = 3: fx = 1.6 -> 0.6 -> cvRound(1228.8) = 1229: 2 * 819 + 4 * 1229 = 6554; >> 4 = 409; * 2048 >> 16 = 12; 14 >> 2 = 3 (exact 3.2)
    assert DS.resize_bilinear_u8(np.array([[1, 2, 4]], np.uint8), 5, 1).tolist() == [[1, 1, 2, 3, 4]]
    # exact 2x decimation: cv::resize switches to the area fast path: (1 + 2 + 3 + 5 + 2) >> 2 = 3 (bilinear at the centre would give 2.75 -> 3 too;
    # (0 + 0 + 0 + 3 + 2) >> 2 = 1 where the 11-bit bilinear path gives (3 * 0.25 = 0.75 ->) 1; (255 + 255 + 254 + 255 + 2) >> 2 = 255)
    assert DS.resize_bilinear_u8(np.array([[1, 2], [3, 5]], np.uint8), 1, 1).tolist() == [[3]]
    assert DS.resize_bilinear_u8(np.array([[0, 0, 255, 255], [0, 3, 254, 255]], np.uint8), 2, 1).tolist() == [[1, 255]]


@pytest.mark.parametrize('bs', [4, 7])
def test_u8_shards_round_trip(tmp_path, bs):
    """the prepared uint8 split: `write_u8_shards` stores exactly what `decode_bgr` / `anno` return, ShardRing yields every frame in order
    (ragged last batch, shard boundary inside a batch, an index list that revisits frames)"""
    write_split(str(tmp_path), 10, seed=5)
    ds = DS.InterHandSplit(str(tmp_path))
    assert DS.write_u8_shards(str(tmp_path), 'test', shard_size=4, workers=2) == 10
    idx = list(range(10)) + [3, 9, 0]
    ring = DS.ShardRing(str(tmp_path), 'test', batch_size=bs, workers=2, depth=3, indices=idx, pin=False)
    try:
        seen = 0
        for frames, annos, n in ring:
            for j in range(n):
                assert np.array_equal(frames[j].numpy(), ds.frame(idx[seen + j])) and np.array_equal(annos[j].numpy(), ds.anno(idx[seen + j]))
            seen += n
        assert seen == len(idx) and len(ring) == (len(idx) + bs - 1) // bs
    finally:
        ring.close()


def test_decode_ring_reports_a_failing_worker_instead_of_hanging(tmp_path):
    """round 4: the ring has no queues in the steady state (workers raise flag bytes in shared memory); a frame that cannot be decoded must
    surface as an exception in the consumer, not as a flag that never comes"""
    write_split(str(tmp_path), 6, seed=3)
    ds = DS.InterHandSplit(str(tmp_path))
    with open(ds.img_path(4), 'wb') as f:
        f.write(b'not a jpeg')
    ring = DS.DecodeRing(str(tmp_path), 'test', batch_size=2, workers=2, pin=False, chunk=1)
    try:
        with pytest.raises(RuntimeError):
            for _ in ring:
                pass
    finally:
        ring.close()


def test_decode_ring_many_chunks_and_deep_ring(tmp_path):
    """more chunks than workers, a ring deeper than the number of batches, a ragged last batch"""
    write_split(str(tmp_path), 11, seed=5)
    ds = DS.InterHandSplit(str(tmp_path))
    ring = DS.DecodeRing(str(tmp_path), 'test', batch_size=4, workers=3, depth=6, pin=False, chunk=1)
    try:
        seen = 0
        for frames, annos, n in ring:
            for j in range(n):
                assert np.array_equal(frames[j].numpy(), ds.frame(seen + j))
            seen += n
        assert seen == 11
    finally:
        ring.close()
