"""CPU: the JPEG pieces of the from-files path that need no GPU.

  * oracle/jpeg.py (numpy restatement of libjpeg's default decode: Huffman, islow IDCT, fancy upsampling, fixed-point colour conversion) is PINNED to
    the pixels the real libjpeg-turbo produced for the committed files (tests/golden/g22_jpeg.npz, written by oracle/gen_jpeg_golden.py through
    Pillow -- the library cv.imread uses, apps/eval.py:56): bit-exact for 4:2:0 / 4:2:2 / 4:4:4 / grayscale, odd sizes, restart intervals;
    and, where Pillow is importable, to Pillow itself on freshly generated files.
  * lib/libdir_jpeg.so (csrc/jpeg_huff.c, the product's host half) loads, exports what include/dir_jpeg.h declares, and its coefficient records
    equal the oracle's entropy decode bit for bit; it refuses what it does not decode."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import jpeg as J

GOLD = os.path.join(ROOT, 'tests', 'golden', 'g22_jpeg.npz')


def cases():
    g = np.load(GOLD)
    return g, sorted(k[:-4] for k in g.files if k.endswith('.rgb'))


def test_oracle_equals_libjpeg_turbo_on_the_committed_files():
    g, names = cases()
    assert len(names) >= 12
    for n in names:
        got = J.decode(g[n + '.jpg'].tobytes())
        assert got.shape == g[n + '.rgb'].shape and np.array_equal(got, g[n + '.rgb']), n
    with pytest.raises(J.JpegError):
        J.decode(g['progressive.jpg'].tobytes())


def test_oracle_equals_pillow_on_fresh_files():
    Image = pytest.importorskip('PIL.Image')
    import io
    rng = np.random.RandomState(7)
    for h, w, q, ss in ((24, 40, 55, 2), (17, 16, 98, 0), (16, 18, 70, 1), (72, 8, 88, 2)):
        a = np.clip(rng.normal(128, 70, (h, w, 3)), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format='JPEG', quality=q, subsampling=ss)
        ref = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert('RGB'))
        assert np.array_equal(J.decode(buf.getvalue()), ref), (h, w, q, ss)


def host():
    from dir_amd import build
    from dir_amd.apps import jpeg as AJ
    build.build_jpeg_host(verbose=False)
    return AJ.host_lib()


def test_host_library_exports_its_header():
    lib = host()
    src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'dir_jpeg.h')).read(), flags=re.S)
    syms = sorted(set(re.findall(r'\b(dir_jpeg_[a-z0-9_]+)\s*\(', src)))
    assert syms == ['dir_jpeg_abi_version', 'dir_jpeg_decode_coefficients', 'dir_jpeg_record_bytes']
    for s in syms:
        assert hasattr(lib, s)
    assert lib.dir_jpeg_abi_version() == int(re.search(r'#define\s+DIR_JPEG_ABI_VERSION\s+(\d+)', src).group(1))
    assert lib.dir_jpeg_record_bytes(256, 256, 2, 2, 3) == 512 + 1536 * 128 == 512 + 256 * 256 * 3       # a 4:2:0 record = header + the frame's size
    assert lib.dir_jpeg_record_bytes(256, 256, 1, 1, 3) == 512 + 3072 * 128 and lib.dir_jpeg_record_bytes(0, 256, 2, 2, 3) == 0


def test_host_entropy_decode_equals_the_oracle():
    lib = host()
    g, names = cases()
    for n in names:
        data = g[n + '.jpg'].tobytes()
        info, coef = J.decode_coefficients(data)
        want = np.concatenate([c.reshape(-1) for c in coef])
        rec = np.zeros(512 + want.size * 2, np.uint8)
        assert lib.dir_jpeg_decode_coefficients(data, len(data), rec.ctypes.data, rec.size) == 0, n
        hdr = rec[:96].view(np.int32)
        assert hdr[0] == 0x4a524944 and hdr[1] == info['width'] and hdr[2] == info['height'] and hdr[3] == len(coef) and hdr[23] == want.size, n
        assert np.array_equal(rec[512:].view(np.int16), want), n
        q = rec[96:480].view(np.uint16).reshape(3, 64)
        assert all(np.array_equal(q[i], info['comps'][i]['q']) for i in range(len(coef))), n
        # a record that is too small is refused, never overrun
        small = np.zeros(512 + want.size * 2 - 128, np.uint8)
        assert lib.dir_jpeg_decode_coefficients(data, len(data), small.ctypes.data, small.size) == -4
    rec = np.zeros(4096, np.uint8)
    prog = g['progressive.jpg'].tobytes()
    assert lib.dir_jpeg_decode_coefficients(prog, len(prog), rec.ctypes.data, rec.size) == -3
    assert lib.dir_jpeg_decode_coefficients(b'\x00' * 64, 64, rec.ctypes.data, rec.size) == -2
    trunc = g['s420_q92.jpg'].tobytes()[:300]
    assert lib.dir_jpeg_decode_coefficients(trunc, len(trunc), rec.ctypes.data, 512 + 6 * 4 * 128 * 4) in (0, -2)      # truncated scans decode as zeros or fail: no crash


def test_file_to_record_falls_back_to_pixel_records(tmp_path):
    """apps/jpeg.py: a baseline 256x256 file becomes a coefficient record; a progressive one, or one of another size, is decoded on the host
    (dataset.decode_bgr) and travels as a pixel record of the same size"""
    pytest.importorskip('PIL.Image')
    from PIL import Image
    from dir_amd.apps import dataset as DS
    from dir_amd.apps import jpeg as AJ
    host()
    rng = np.random.RandomState(1)
    a = np.clip(rng.normal(128, 60, (256, 256, 3)), 0, 255).astype(np.uint8)
    row = np.zeros(AJ.record_bytes(256), np.uint8)
    for name, kw, size, kind in (('base.jpg', {}, 256, 'coef'), ('prog.jpg', dict(progressive=True), 256, 'pixels'), ('c444.jpg', dict(subsampling=0), 256, 'pixels'),
                                 ('small.jpg', {}, 128, 'pixels')):
        p = str(tmp_path / name)
        Image.fromarray(a[:size, :size]).save(p, quality=90, **kw)
        assert AJ.file_to_record(p, row, 256) == kind, name
        if kind == 'pixels':
            assert row[:16].view(np.int32)[0] == AJ.MAGIC_PIXELS
            assert np.array_equal(row[512:512 + 256 * 256 * 3].reshape(256, 256, 3), DS.decode_bgr(p, 256)), name
    # 4:4:0 (luma sampled 1x2): no device upsampler and no oracle for it -> the host decoder must refuse it (it used to return a
    # coefficient record that the colour kernel then read with the h2v2 formulas).  PIL cannot write 4:4:0, but a 4:2:2 stream of a 256x256 frame
    # has the same MCU count and blocks per MCU, so its SOF0 luma sampling byte 0x21 -> 0x12 gives a legal 4:4:0 file (other picture, same syntax)
    p422 = str(tmp_path / 'c422.jpg')
    Image.fromarray(a).save(p422, quality=90, subsampling=1)
    raw = bytearray(open(p422, 'rb').read())
    i = raw.index(b'\xff\xc0')
    assert raw[i + 11] == 0x21                      # FFC0 Lf(2) P(1) Y(2) X(2) Nf(1) C1(1) H1V1(1)
    raw[i + 11] = 0x12
    p440 = str(tmp_path / 'c440.jpg')
    open(p440, 'wb').write(bytes(raw))
    assert host().dir_jpeg_decode_coefficients(bytes(raw), len(raw), row.ctypes.data, row.size) == -3
    assert AJ.file_to_record(p440, row, 256) == 'pixels'
    assert np.array_equal(row[512:512 + 256 * 256 * 3].reshape(256, 256, 3), DS.decode_bgr(p440, 256))
    with pytest.raises(ValueError):
        (tmp_path / 'bad.jpg').write_bytes(b'not a jpeg at all')
        AJ.file_to_record(str(tmp_path / 'bad.jpg'), row, 256)


def test_host_library_survives_corrupt_files():
    """the decode workers read files they did not write: truncated, bit-flipped and spliced streams must come back as an error code (or decode to
    something) -- never crash, never write past the record.  6 000 mutations of the committed files, in a child process (a segfault there is a
    failure here), with guard bytes behind every record."""
    import subprocess
    import sys
    host()
    code = r'''
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, %r)
from dir_amd.apps import jpeg as AJ
lib = AJ.host_lib()
lib.dir_jpeg_decode_coefficients.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
g = np.load(%r)
files = [g[k].tobytes() for k in g.files if k.endswith('.jpg')]
rng = np.random.RandomState(11)
cap = 1 << 20
buf = np.empty(cap + 64, np.uint8)
codes = {}
for it in range(6000):
    d = bytearray(files[it %% len(files)])
    kind = it %% 4
    if kind == 0:                                   # truncate
        d = d[:rng.randint(0, len(d))]
    elif kind == 1:                                 # flip a few bytes anywhere (markers, lengths, tables, entropy data)
        for _ in range(rng.randint(1, 6)):
            d[rng.randint(0, len(d))] = rng.randint(0, 256)
    elif kind == 2:                                 # corrupt the header region only: segment lengths, sampling factors, table ids
        for _ in range(rng.randint(1, 4)):
            d[rng.randint(2, min(len(d), 700))] = rng.randint(0, 256)
    else:                                           # splice two files
        o = bytearray(files[rng.randint(0, len(files))])
        c = rng.randint(2, len(d))
        d = d[:c] + o[rng.randint(2, len(o)):]
    n = [cap, 4096, 600, 512, 100][rng.randint(0, 5)]      # also records that are too small
    buf[n:n + 64] = 0xA5
    rc = lib.dir_jpeg_decode_coefficients(bytes(d), len(d), buf.ctypes.data, n)
    assert (buf[n:n + 64] == 0xA5).all(), ('wrote past the record', it, n, rc)
    assert rc in (0, -1, -2, -3, -4), rc
    codes[rc] = codes.get(rc, 0) + 1
print('codes', sorted(codes.items()))
''' % (ROOT, GOLD)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert 'codes' in r.stdout
