"""TEST INFRASTRUCTURE (oracle): a restatement of the baseline JPEG decode the reference's input pipeline performs through OpenCV
(`cv.imread`, apps/eval.py:56, dataset/interhand.py:223 over the files dataset/prepare_data.py:123-166 writes) -- i.e. of libjpeg's default
decode path, the IJG algorithms OpenCV and Pillow both link (libjpeg-turbo's SIMD kernels are bit-exact re-implementations of them):

    entropy decode          ITU T.81 F.2.2 (Huffman, baseline sequential, restart intervals)         jdhuff.c
    dequantise + IDCT       the "islow" integer IDCT (CONST_BITS 13, PASS1_BITS 2), range-limited      jidctint.c, jdmaster.c (range-limit table)
    chroma upsampling       "fancy" triangle filter h2v2 (3/4 - 1/4 in each direction), h2v1           jdsample.c
    colour conversion       YCbCr -> RGB with the 16-bit fixed-point tables                            jdcolor.c

The library itself is a third-party dependency that is not under /root/reference (OpenCV's bundled libjpeg-turbo); the algorithm is restated from its
published description and PINNED here against Pillow's decoder -- the same libjpeg-turbo -- on generated images (tests/test_jpeg_oracle.py: bit-exact
for 4:2:0 / 4:4:4 / 4:2:2 / grayscale, odd sizes, restart intervals, qualities 30..100).  Only tests/ and bench tools may import this module: the
product path is dir_amd/csrc/jpeg_huff.c (host entropy decode) + dir_amd/csrc/jpeg.hip (everything after it, on the GPU), checked against this file.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57,
                   50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], np.int64)


class JpegError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------------------------ parsing + entropy decode
def parse(data):
    """-> dict(width, height, comps=[dict(id, h, v, tq, td, ta)], qt={id: [64] natural order}, dht={(class, id): (counts[16], symbols)},
    restart_interval, scan=bytes of the entropy-coded segment (byte-stuffed, with RSTn markers))"""
    data = bytes(data)
    if data[:2] != b'\xff\xd8':
        raise JpegError('not a JPEG (no SOI)')
    i, out = 2, dict(qt={}, dht={}, restart_interval=0, comps=None)
    while i < len(data):
        if data[i] != 0xFF:
            raise JpegError('marker expected at %d' % i)
        while data[i] == 0xFF:
            i += 1
        m = data[i]
        i += 1
        if m == 0xD9:
            break
        L = (data[i] << 8) | data[i + 1]
        seg = data[i + 2:i + L]
        if m == 0xDB:                                      # DQT
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                j += 1
                if pq:
                    vals = [(seg[j + 2 * k] << 8) | seg[j + 2 * k + 1] for k in range(64)]
                    j += 128
                else:
                    vals = list(seg[j:j + 64])
                    j += 64
                q = np.zeros(64, np.int64)
                q[ZIGZAG] = vals
                out['qt'][tq] = q
        elif m == 0xC4:                                    # DHT
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                counts = list(seg[j + 1:j + 17])
                n = sum(counts)
                out['dht'][(tc, th)] = (counts, list(seg[j + 17:j + 17 + n]))
                j += 17 + n
        elif m == 0xC0 or m == 0xC1:                       # SOF0 / SOF1 (baseline / extended sequential, Huffman)
            if seg[0] != 8:
                raise JpegError('only 8-bit samples')
            out['height'], out['width'] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            out['comps'] = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(seg[5])]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError('progressive / lossless / arithmetic JPEG (SOF%d) is not baseline' % (m - 0xC0))
        elif m == 0xDD:
            out['restart_interval'] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:                                    # SOS: the scan follows
            ns = seg[0]
            if out['comps'] is None or ns != len(out['comps']):
                raise JpegError('one interleaved scan with every component expected')
            for c in range(ns):
                comp = next(k for k in out['comps'] if k['id'] == seg[1 + 2 * c])
                comp['td'], comp['ta'] = seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15
            j = i + L
            k = j
            while True:                                    # the scan ends at the first marker that is neither a stuffed 0xFF00 nor RSTn
                k = data.index(b'\xff', k)
                if data[k + 1] == 0 or 0xD0 <= data[k + 1] <= 0xD7:
                    k += 2
                    continue
                break
            out['scan'] = data[j:k]
            return out
        i += L
    raise JpegError('no scan found')


def _huff_table(counts, symbols):
    """(code length, code) -> symbol as a dict keyed by (length, code)"""
    table, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            table[(ln, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits(object):
    def __init__(self, scan):
        self.d, self.i, self.acc, self.n = scan, 0, 0, 0

    def _fill(self):
        while self.n <= 24:
            if self.i >= len(self.d):
                b = 0
            else:
                b = self.d[self.i]
                if b == 0xFF:
                    nxt = self.d[self.i + 1] if self.i + 1 < len(self.d) else 0xD9
                    if nxt == 0:
                        self.i += 2
                    else:                                   # a marker: feed zeros, stay on it
                        b = 0
                        self.acc = (self.acc << 8) | b
                        self.n += 8
                        continue
                else:
                    self.i += 1
            self.acc = (self.acc << 8) | b
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def symbol(self, table):
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | self.get(1)
            s = table.get((ln, code))
            if s is not None:
                return s
        raise JpegError('bad Huffman code')

    def restart(self):
        self.acc = self.n = 0                               # discard the padding bits, step over the RSTn marker
        while self.i < len(self.d) and not (self.d[self.i] == 0xFF and 0xD0 <= self.d[self.i + 1] <= 0xD7):
            self.i += 1
        self.i += 2


def _extend(v, t):
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def decode_coefficients(data):
    """entropy decode only -> (info, [per component int16 array [blocks_y, blocks_x, 64] of QUANTISED coefficients in natural order])"""
    p = parse(data)
    comps = p['comps']
    if len(comps) == 1:                 # a single-component scan is never interleaved: one block per MCU whatever the SOF's sampling factors say (T.81 A.2.2)
        comps[0]['h'] = comps[0]['v'] = 1
    hmax, vmax = max(c['h'] for c in comps), max(c['v'] for c in comps)
    mcux, mcuy = -(-p['width'] // (8 * hmax)), -(-p['height'] // (8 * vmax))
    tabs = {k: _huff_table(*v) for k, v in p['dht'].items()}
    coef = [np.zeros((mcuy * c['v'], mcux * c['h'], 64), np.int16) for c in comps]
    bits = _Bits(p['scan'])
    pred = [0] * len(comps)
    ri, count = p['restart_interval'], 0
    for my in range(mcuy):
        for mx in range(mcux):
            if ri and count and count % ri == 0:
                bits.restart()
                pred = [0] * len(comps)
            count += 1
            for ci, c in enumerate(comps):
                dc, ac = tabs[(0, c['td'])], tabs[(1, c['ta'])]
                for by in range(c['v']):
                    for bx in range(c['h']):
                        blk = coef[ci][my * c['v'] + by, mx * c['h'] + bx]
                        t = bits.symbol(dc)
                        pred[ci] += _extend(bits.get(t), t)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = bits.symbol(ac)
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                raise JpegError('coefficient index out of range')
                            blk[ZIGZAG[k]] = _extend(bits.get(s), s)
                            k += 1
    info = dict(width=p['width'], height=p['height'], hmax=hmax, vmax=vmax, mcux=mcux, mcuy=mcuy,
                comps=[dict(h=c['h'], v=c['v'], q=p['qt'][c['tq']].copy()) for c in comps])
    return info, coef


# ------------------------------------------------------------------------------------------------------------------ IDCT (jidctint.c, "islow")
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865, F_0_899976223, F_1_175875602 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501321110, F_1_847759065, F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 12299, 15137, 16069, 16819, 20995, 25172
CONST_BITS, PASS1_BITS = 13, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v0, v1, v2, v3, v4, v5, v6, v7, shift):
    """one pass over eight int64 vectors (the same butterfly for columns and rows) -> eight vectors, descaled by `shift`"""
    z1 = (v2 + v6) * F_0_541196100
    tmp2 = z1 + v6 * (-F_1_847759065)
    tmp3 = z1 + v2 * F_0_765366865
    tmp0 = (v0 + v4) << CONST_BITS
    tmp1 = (v0 - v4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = v7, v5, v3, v1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0, t1, t2, t3 = t0 * F_0_298631336, t1 * F_2_053119869, t2 * F_3_072711026, t3 * F_1_501321110
    z1, z2, z3, z4 = z1 * (-F_0_899976223), z2 * (-F_2_562915447), z3 * (-F_1_961570560) + z5, z4 * (-F_0_390180644) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    return [_descale(tmp10 + t3, shift), _descale(tmp11 + t2, shift), _descale(tmp12 + t1, shift), _descale(tmp13 + t0, shift),
            _descale(tmp13 - t0, shift), _descale(tmp12 - t1, shift), _descale(tmp11 - t2, shift), _descale(tmp10 - t3, shift)]


def range_limit_idct(v):
    """jdmaster.c prepare_range_limit_table as the IDCT uses it (index = value & 1023 into the table that starts at CENTERJSAMPLE):
    clamp(v + 128, 0, 255) for -512 <= v < 512, the table's wrap-around outside (corrupt coefficient data)"""
    idx = np.asarray(v).astype(np.int64) & 1023
    out = np.where(idx < 128, idx + 128, np.where(idx < 512, 255, np.where(idx < 896, 0, idx - 896)))
    return out.astype(np.uint8)


def idct_islow(coef, q):
    """coef [..., 64] quantised, natural order; q [64] -> samples uint8 [..., 8, 8]"""
    c = coef.astype(np.int64).reshape(coef.shape[:-1] + (8, 8)) * q.astype(np.int64).reshape(8, 8)        # [.., row, col]
    cols = _idct_1d(*[c[..., r, :] for r in range(8)], shift=CONST_BITS - PASS1_BITS)                       # pass 1: down the columns
    ws = np.stack(cols, -2)                                                                                 # [.., row, col]
    rows = _idct_1d(*[ws[..., :, k] for k in range(8)], shift=CONST_BITS + PASS1_BITS + 3)                  # pass 2: along the rows
    return range_limit_idct(np.stack(rows, -1))


def component_planes(info, coef):
    """-> per component the full decoded (MCU-padded) sample plane uint8 [blocks_y * 8, blocks_x * 8]"""
    planes = []
    for c, cf in zip(info['comps'], coef):
        s = idct_islow(cf, c['q'])                                              # [by, bx, 8, 8]
        planes.append(np.ascontiguousarray(s.transpose(0, 2, 1, 3).reshape(cf.shape[0] * 8, cf.shape[1] * 8)))
    return planes


# ------------------------------------------------------------------------------------------------------------------ upsampling (jdsample.c)
def _h2_fancy_rows(cur, oth):
    """horizontal 2x of `3 * cur + oth` column sums with the h2v2 rounding (8 for even output columns, 7 for odd ones)"""
    s = 3 * cur.astype(np.int64) + oth.astype(np.int64)                        # [rows, w]
    w = s.shape[1]
    out = np.empty((s.shape[0], 2 * w), np.int64)
    last = np.concatenate([s[:, :1], s[:, :-1]], 1)
    nxt = np.concatenate([s[:, 1:], s[:, -1:]], 1)
    out[:, 0::2] = (3 * s + last + 8) >> 4
    out[:, 1::2] = (3 * s + nxt + 7) >> 4
    out[:, 0] = (s[:, 0] * 4 + 8) >> 4
    out[:, -1] = (s[:, -1] * 4 + 7) >> 4
    return out


def upsample_h2v2_fancy(plane, dw, dh):
    """plane: the component's decoded samples; dw / dh: its real (downsampled) width / height -> [2 dh, 2 dw] uint8.  Rows above the first / below
    the last real row are that row again (jdmainct.c context rows); a single-column component is replicated (h2v2_fancy needs width > 2 ... libjpeg
    falls back to the box filter for downsampled_width <= 2)"""
    p = plane[:dh, :dw]
    if dw <= 2:
        return np.repeat(np.repeat(p, 2, 0), 2, 1)
    up = np.concatenate([p[:1], p[:-1]], 0)
    dn = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * dh, 2 * dw), np.int64)
    out[0::2] = _h2_fancy_rows(p, up)
    out[1::2] = _h2_fancy_rows(p, dn)
    return out.astype(np.uint8)


def upsample_h2v1_fancy(plane, dw, dh):
    p = plane[:dh, :dw].astype(np.int64)
    if dw <= 2:
        return np.repeat(plane[:dh, :dw], 2, 1)
    out = np.empty((dh, 2 * dw), np.int64)
    last = np.concatenate([p[:, :1], p[:, :-1]], 1)
    nxt = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + last + 1) >> 2
    out[:, 1::2] = (3 * p + nxt + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


# ------------------------------------------------------------------------------------------------------------------ colour (jdcolor.c)
def _fix(x):
    return int(x * 65536 + 0.5)


def ycc_to_rgb(y, cb, cr):
    y, cb, cr = y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((_fix(1.40200) * cr + 32768) >> 16)
    g = y + ((-_fix(0.34414) * cb + 32768 - _fix(0.71414) * cr) >> 16)
    b = y + ((_fix(1.77200) * cb + 32768) >> 16)
    return np.stack([np.clip(r, 0, 255), np.clip(g, 0, 255), np.clip(b, 0, 255)], -1).astype(np.uint8)


def pixels_from_coefficients(info, coef):
    """everything after the entropy decode -> RGB uint8 [H, W, 3] (grayscale: the plane in all three channels, as cv.imread's default does)"""
    W, H = info['width'], info['height']
    planes = component_planes(info, coef)
    if len(planes) == 1:
        g = planes[0][:H, :W]
        return np.stack([g, g, g], -1)
    full = []
    for c, pl in zip(info['comps'], planes):
        fh, fv = info['hmax'] // c['h'], info['vmax'] // c['v']
        dw, dh = -(-W * c['h'] // info['hmax']), -(-H * c['v'] // info['vmax'])
        if (fh, fv) == (1, 1):
            full.append(pl[:H, :W])
        elif (fh, fv) == (2, 2):
            full.append(upsample_h2v2_fancy(pl, dw, dh)[:H, :W])
        elif (fh, fv) == (2, 1):
            full.append(upsample_h2v1_fancy(pl, dw, dh)[:H, :W])
        else:
            raise JpegError('sampling factors %dx%d are not on the path' % (fh, fv))
    return ycc_to_rgb(*full)


def decode(data):
    """JPEG bytes -> RGB uint8 [H, W, 3]"""
    return pixels_from_coefficients(*decode_coefficients(data))
