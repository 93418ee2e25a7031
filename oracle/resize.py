"""TEST INFRASTRUCTURE (oracle): scalar restatement of OpenCV's 8-bit `cv.resize(..., interpolation=INTER_LINEAR)` -- what
apps/eval.py:57 and dataset/interhand.py:48-94 call -- following the structure of the published OpenCV implementation
(modules/imgproc/src/resize.cpp, 4.x; OpenCV is NOT under /root/reference and not installed here, so this is a restatement of the
published algorithm, "parity unpinned" against the library itself):

  * cv::resize():     scale = src / dst per axis; `if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2)
                      interpolation = INTER_AREA` -- an EXACT 2x decimation is a 2x2 box average, (a + b + c + d + 2) >> 2
                      (ResizeAreaFastVec / the scalar tail of resizeAreaFast_).
  * otherwise:        per destination index fx = (dx + 0.5) * scale - 0.5, sx = floor(fx), fx -= sx; sx < 0 -> (0, 0);
                      sx >= src - 1 -> (src - 1, 0); coefficients ialpha = saturate_cast<short>(cvRound(f * 2048)) for the
                      right / lower tap and 2048 - that for the left / upper one (INTER_RESIZE_COEF_BITS = 11).
  * HResizeLinear:    D[dx] = S[sx] * a0 + S[sx + 1] * a1                       (int32, no rounding: values up to 255 * 2048)
  * VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>:
                      dst = uchar(( ((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2)

Pure-Python loops: small cases only."""
import math

import numpy as np


def _cv_round(v):
    """cvRound: round half to even (lrint under the default rounding mode)"""
    f = math.floor(v)
    d = v - f
    if d > 0.5 or (d == 0.5 and f % 2 == 1):
        return int(f) + 1
    return int(f)


def _taps(n_in, n_out):
    scale = n_in / float(n_out)
    out = []
    for d in range(n_out):
        fx = np.float32((d + 0.5) * scale - 0.5)            # computed in float in the library
        sx = int(math.floor(fx))
        fx = float(np.float32(fx - sx))
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= n_in - 1:
            sx, fx = n_in - 1, 0.0
        c1 = max(-32768, min(32767, _cv_round(fx * 2048.0)))
        out.append((sx, min(sx + 1, n_in - 1), 2048 - c1, c1))
    return out


def cv_resize_linear_u8(img, wo, ho):
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    src = img.reshape(h, w, ch).astype(np.int64)
    out = np.zeros((ho, wo, ch), np.uint8)
    if w == 2 * wo and h == 2 * ho:                          # INTER_LINEAR with an exact 2x decimation -> INTER_AREA fast path
        for y in range(ho):
            for x in range(wo):
                for c in range(ch):
                    s = src[2 * y, 2 * x, c] + src[2 * y, 2 * x + 1, c] + src[2 * y + 1, 2 * x, c] + src[2 * y + 1, 2 * x + 1, c]
                    out[y, x, c] = (s + 2) >> 2
        return out.reshape((ho, wo) if img.ndim == 2 else (ho, wo, ch))
    xt, yt = _taps(w, wo), _taps(h, ho)
    rows = np.zeros((h, wo, ch), np.int64)
    for y in range(h):
        for dx, (x0, x1, a0, a1) in enumerate(xt):
            rows[y, dx] = src[y, x0] * a0 + src[y, x1] * a1
    for dy, (y0, y1, b0, b1) in enumerate(yt):
        for dx in range(wo):
            for c in range(ch):
                v = (((b0 * (int(rows[y0, dx, c]) >> 4)) >> 16) + ((b1 * (int(rows[y1, dx, c]) >> 4)) >> 16) + 2) >> 2
                out[dy, dx, c] = max(0, min(255, v))
    return out.reshape((ho, wo) if img.ndim == 2 else (ho, wo, ch))
