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
