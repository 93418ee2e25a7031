"""TEST INFRASTRUCTURE: writes tests/golden/g22_jpeg.npz -- a handful of small JPEG files (their bytes) together with the pixels the REAL libjpeg-turbo
decodes from them (through Pillow, which links the library OpenCV's cv.imread uses with the same defaults: integer slow IDCT, fancy upsampling).
Run only in the authoring container (needs Pillow):  python oracle/gen_jpeg_golden.py
The fixtures pin oracle/jpeg.py (tests/test_jpeg_oracle.py), which in turn is the checker of the product path (csrc/jpeg_huff.c + csrc/jpeg.hip)."""
import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def image(rng, h, w, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 'smooth':
        a = np.stack([127 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 13.0 + c) for c in range(3)], -1) + rng.normal(0, 6, (h, w, 3))
    elif kind == 'noise':
        a = rng.uniform(0, 255, (h, w, 3))
    else:                                   # hard edges + saturated colours: exercises the range limits of the IDCT and of the colour conversion
        a = np.zeros((h, w, 3))
        a[h // 3:, w // 4:] = [250, 10, 30]
        a[:h // 2, :w // 2] += [0, 200, 255]
    return np.clip(a, 0, 255).astype(np.uint8)


CASES = [  # name, height, width, content, quality, PIL subsampling (2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4), grayscale, extra save options
    ('s420_q92', 64, 64, 'smooth', 92, 2, False, {}),
    ('n420_q75', 64, 64, 'noise', 75, 2, False, {}),
    ('odd420_q90', 33, 47, 'smooth', 90, 2, False, {}),
    ('e420_q30', 40, 24, 'edges', 30, 2, False, {}),
    ('s444_q95', 64, 64, 'smooth', 95, 0, False, {}),
    ('n444_q100', 31, 57, 'noise', 100, 0, False, {}),
    ('s422_q85', 48, 80, 'smooth', 85, 1, False, {}),
    ('e422_q60', 35, 35, 'edges', 60, 1, False, {}),
    ('gray_q90', 64, 48, 'smooth', 90, 2, True, {}),
    ('tiny_5x3', 5, 3, 'edges', 90, 2, False, {}),
    ('rst_blocks2', 64, 96, 'noise', 80, 2, False, dict(restart_marker_blocks=2)),
    ('rst_rows1', 48, 64, 'smooth', 88, 2, False, dict(restart_marker_rows=1)),
    ('full_256', 256, 256, 'smooth', 92, 2, False, {}),
]


def main():
    from PIL import Image, features
    rng = np.random.RandomState(22)
    out = {'meta': np.array('Pillow %s, libjpeg-turbo %s' % (__import__('PIL').__version__, features.version('jpg')))}
    for name, h, w, kind, q, ss, gray, extra in CASES:
        a = image(rng, h, w, kind)
        buf = io.BytesIO()
        Image.fromarray(a[..., 0] if gray else a).save(buf, format='JPEG', quality=q, subsampling=ss, **extra)
        data = buf.getvalue()
        out[name + '.jpg'] = np.frombuffer(data, np.uint8)
        out[name + '.rgb'] = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
    # a progressive file: the product's entropy decoder must refuse it (and the host then decodes it the ordinary way)
    buf = io.BytesIO()
    Image.fromarray(image(rng, 32, 32, 'smooth')).save(buf, format='JPEG', quality=90, progressive=True)
    out['progressive.jpg'] = np.frombuffer(buf.getvalue(), np.uint8)
    path = os.path.join(ROOT, 'tests', 'golden', 'g22_jpeg.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
