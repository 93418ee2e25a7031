/* Host half of the from-files input path: the ENTROPY DECODE of a baseline JPEG -- and nothing else.
 *
 *   reference   apps/eval.py:56  cv.imread(<split>/img/<idx>.jpg)  over the files dataset/prepare_data.py:123-166 wrote with cv.imwrite
 *               (OpenCV's bundled libjpeg-turbo: Huffman decode -> dequantise + islow IDCT -> fancy chroma upsampling -> YCbCr -> BGR).
 *
 * The Huffman stream is inherently serial, so it stays on the host: this file turns the bytes of one JPEG into a fixed-layout RECORD -- a 512-byte
 * header (geometry, sampling factors, the quantisation tables in natural order) followed by the QUANTISED DCT coefficients as int16 [component]
 * [block row][block column][64], de-zigzagged.  Everything after that -- dequantisation, the integer IDCT, chroma upsampling, colour conversion --
 * is integer arithmetic on independent blocks / pixels and runs on the GPU (csrc/jpeg.hip: dir_jpeg_decode_records), bit-exact with libjpeg
 * (oracle/jpeg.py restates it and is pinned to Pillow's libjpeg-turbo).  A record is the same size as the decoded BGR frame for 4:2:0 (1.5 int16
 * per pixel), so the host -> device traffic does not change; the host's work per image drops to the entropy decode.
 *
 * Plain C (gcc), no HIP, no dependencies: built as dir_amd/lib/libdir_jpeg.so by dir_amd/build.py and loaded by the decode worker PROCESSES
 * (dir_amd/apps/jpeg.py), which must not touch the GPU runtime.  ITU T.81: baseline / extended sequential DCT, Huffman, 8-bit, one
 * interleaved scan (what cv.imwrite / libjpeg / Pillow write by default); progressive, arithmetic-coded and multi-scan files are refused
 * (DIR_JPEG_E_UNSUPPORTED) and the caller decodes those through the ordinary path. */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/dir_jpeg.h"

_Static_assert(sizeof(dir_jpeg_header) == 512, "dir_jpeg_header must be 512 bytes");

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

#define LOOK_BITS 9
typedef struct {
    uint16_t look[1 << LOOK_BITS]; /* (length << 8) | symbol for codes of at most LOOK_BITS bits, 0 = longer */
    int32_t maxcode[18];           /* largest code of each length (-1: none); [17] = sentinel */
    int32_t valoff[17];            /* symbol index of the first code of each length minus that code */
    uint8_t sym[256];
    int16_t fast_ac[1 << LOOK_BITS]; /* AC tables: (value << 8) | (run << 4) | (code length + magnitude bits) when both fit in LOOK_BITS, else 0 */
    int present;
} huff_t;

static int build_huff(huff_t* h, const uint8_t* counts, const uint8_t* symbols, int nsym) {
    int code = 0, k = 0;
    memset(h->look, 0, sizeof(h->look));
    memcpy(h->sym, symbols, (size_t)nsym);
    for (int len = 1; len <= 16; ++len) {
        h->valoff[len] = k - code;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (k >= nsym || code >= (1 << len)) return -1;
            if (len <= LOOK_BITS) {
                const int base = code << (LOOK_BITS - len);
                for (int f = 0; f < (1 << (LOOK_BITS - len)); ++f) h->look[base + f] = (uint16_t)((len << 8) | symbols[k]);
            }
        }
        h->maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    h->present = 1;
    /* one lookup for a whole small AC coefficient (code + magnitude bits inside the LOOK_BITS window): most non-zero coefficients of a photograph */
    for (int i = 0; i < (1 << LOOK_BITS); ++i) {
        h->fast_ac[i] = 0;
        const unsigned e = h->look[i];
        if (!e) continue;
        const int len = (int)(e >> 8), rs = (int)(e & 0xff), run = rs >> 4, mag = rs & 15;
        if (mag && len + mag <= LOOK_BITS) {
            int v = (i >> (LOOK_BITS - len - mag)) & ((1 << mag) - 1);
            if (v < (1 << (mag - 1))) v -= (1 << mag) - 1;
            if (v >= -128 && v <= 127) h->fast_ac[i] = (int16_t)(v * 256 + run * 16 + len + mag);
        }
    }
    return 0;
}

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc; /* bits are consumed from the top of the low `n` bits */
    int n;
    int marker; /* a marker was reached: zero bits are fed from here on (T.81 F.2.2.5) */
} bits_t;

static inline void refill(bits_t* b) {
    while (b->n <= 56) {
        unsigned byte = 0;
        if (!b->marker && b->p < b->end) {
            byte = *b->p;
            if (byte == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0x00) b->p += 2; /* a stuffed 0xFF */
                else { b->marker = 1; byte = 0; }                    /* a marker: stay on it */
            } else {
                ++b->p;
            }
        }
        b->acc = (b->acc << 8) | byte;
        b->n += 8;
    }
}
static inline int get_bits(bits_t* b, int k) { /* 0 <= k <= 16 */
    if (k == 0) return 0;
    if (b->n < k) refill(b);
    b->n -= k;
    return (int)((b->acc >> b->n) & ((1u << k) - 1u));
}
static inline int decode_symbol(bits_t* b, const huff_t* h) {
    if (b->n < 16) refill(b);
    const unsigned peek = (unsigned)((b->acc >> (b->n - LOOK_BITS)) & ((1u << LOOK_BITS) - 1u));
    const unsigned e = h->look[peek];
    if (e) {
        b->n -= (int)(e >> 8);
        return (int)(e & 0xff);
    }
    int code = (int)((b->acc >> (b->n - LOOK_BITS)) & ((1u << LOOK_BITS) - 1u));
    int len = LOOK_BITS;
    b->n -= LOOK_BITS;
    while (len < 16) {
        code = (code << 1) | (int)((b->acc >> (b->n - 1)) & 1u);
        --b->n;
        ++len;
        if (h->maxcode[len] >= 0 && code <= h->maxcode[len]) return h->sym[(code + h->valoff[len]) & 0xff];
    }
    return -1;
}
static inline int extend(int v, int t) { return (t == 0 || v >= (1 << (t - 1))) ? v : v - (1 << t) + 1; }

static inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

int dir_jpeg_decode_coefficients(const uint8_t* data, size_t n, void* record, size_t record_bytes) {
    if (!data || !record || n < 4 || record_bytes < sizeof(dir_jpeg_header)) return DIR_JPEG_E_ARG;
    if (data[0] != 0xFF || data[1] != 0xD8) return DIR_JPEG_E_FORMAT;
    dir_jpeg_header* H = (dir_jpeg_header*)record;
    memset(H, 0, sizeof(*H));
    uint16_t qt[4][64];
    int qt_present[4] = {0, 0, 0, 0};
    huff_t dc[4], ac[4];
    for (int i = 0; i < 4; ++i) dc[i].present = ac[i].present = 0;
    int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0}, comp_td[3] = {0, 0, 0}, comp_ta[3] = {0, 0, 0};
    int restart_interval = 0, have_sof = 0;
    size_t i = 2;
    const uint8_t* scan = NULL;
    while (i + 4 <= n) {
        if (data[i] != 0xFF) return DIR_JPEG_E_FORMAT;
        while (i < n && data[i] == 0xFF) ++i;
        if (i >= n) return DIR_JPEG_E_FORMAT;
        const unsigned m = data[i++];
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; /* stand-alone markers */
        if (i + 2 > n) return DIR_JPEG_E_FORMAT;
        const size_t L = be16(data + i);
        if (L < 2 || i + L > n) return DIR_JPEG_E_FORMAT;
        const uint8_t* seg = data + i + 2;
        const size_t sl = L - 2;
        if (m == 0xDB) {
            size_t j = 0;
            while (j < sl) {
                const int pq = seg[j] >> 4, tq = seg[j] & 15;
                ++j;
                if (tq > 3 || j + (pq ? 128u : 64u) > sl) return DIR_JPEG_E_FORMAT;
                for (int k = 0; k < 64; ++k) qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(seg + j + 2 * k) : seg[j + k];
                j += pq ? 128 : 64;
                qt_present[tq] = 1;
            }
        } else if (m == 0xC4) {
            size_t j = 0;
            while (j + 17 <= sl) {
                const int tc = seg[j] >> 4, th = seg[j] & 15;
                int ns = 0;
                for (int k = 0; k < 16; ++k) ns += seg[j + 1 + k];
                if (tc > 1 || th > 3 || ns > 256 || j + 17 + (size_t)ns > sl) return DIR_JPEG_E_FORMAT;
                if (build_huff(tc ? &ac[th] : &dc[th], seg + j + 1, seg + j + 17, ns)) return DIR_JPEG_E_FORMAT;
                j += 17 + (size_t)ns;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || seg[0] != 8) return DIR_JPEG_E_UNSUPPORTED;
            H->height = (int32_t)be16(seg + 1);
            H->width = (int32_t)be16(seg + 3);
            H->ncomp = seg[5];
            if ((H->ncomp != 1 && H->ncomp != 3) || sl < 6 + 3u * (unsigned)H->ncomp || H->width <= 0 || H->height <= 0) return DIR_JPEG_E_UNSUPPORTED;
            for (int c = 0; c < H->ncomp; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                H->h[c] = seg[7 + 3 * c] >> 4;
                H->v[c] = seg[7 + 3 * c] & 15;
                comp_tq[c] = seg[8 + 3 * c];
                if (H->h[c] < 1 || H->h[c] > 2 || H->v[c] < 1 || H->v[c] > 2 || comp_tq[c] > 3) return DIR_JPEG_E_UNSUPPORTED;
            }
            have_sof = 1;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return DIR_JPEG_E_UNSUPPORTED; /* progressive, lossless, arithmetic */
        } else if (m == 0xDD) {
            if (sl < 2) return DIR_JPEG_E_FORMAT;
            restart_interval = (int)be16(seg);
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1 || seg[0] != H->ncomp || sl < 1 + 2u * (unsigned)H->ncomp + 3u) return DIR_JPEG_E_UNSUPPORTED; /* one interleaved scan */
            for (int s = 0; s < H->ncomp; ++s) {
                int c = -1;
                for (int k = 0; k < H->ncomp; ++k)
                    if (comp_id[k] == seg[1 + 2 * s]) c = k;
                if (c != s) return DIR_JPEG_E_UNSUPPORTED;
                comp_td[c] = seg[2 + 2 * s] >> 4;
                comp_ta[c] = seg[2 + 2 * s] & 15;
                if (comp_td[c] > 3 || comp_ta[c] > 3 || !dc[comp_td[c]].present || !ac[comp_ta[c]].present || !qt_present[comp_tq[c]]) return DIR_JPEG_E_FORMAT;
            }
            scan = data + i + L;
            break;
        }
        i += L;
    }
    if (!scan) return DIR_JPEG_E_FORMAT;
    if (H->ncomp == 1) H->h[0] = H->v[0] = 1; /* a single-component scan is not interleaved: one block per MCU (T.81 A.2.2) */
    int hmax = 1, vmax = 1;
    for (int c = 0; c < H->ncomp; ++c) {
        if (H->h[c] > hmax) hmax = H->h[c];
        if (H->v[c] > vmax) vmax = H->v[c];
    }
    if (H->ncomp == 3 && (H->h[1] != 1 || H->v[1] != 1 || H->h[2] != 1 || H->v[2] != 1)) return DIR_JPEG_E_UNSUPPORTED; /* chroma at the base rate only */
    /* luma 1x2 (4:4:0) has no device upsampler (jpeg.hip: h1v1, h2v1, h2v2 only) and no oracle (oracle/jpeg.py raises): such a file must travel as a
       pixel record, not decode to silently wrong chroma */
    if (H->ncomp == 3 && H->h[0] == 1 && H->v[0] == 2) return DIR_JPEG_E_UNSUPPORTED;
    H->hmax = hmax;
    H->vmax = vmax;
    H->mcux = (H->width + 8 * hmax - 1) / (8 * hmax);
    H->mcuy = (H->height + 8 * vmax - 1) / (8 * vmax);
    int64_t total = 0;
    for (int c = 0; c < H->ncomp; ++c) {
        H->blocks_x[c] = H->mcux * H->h[c];
        H->blocks_y[c] = H->mcuy * H->v[c];
        H->coef_offset[c] = (int32_t)total;
        total += (int64_t)H->blocks_x[c] * H->blocks_y[c] * 64;
        memcpy(H->quant[c], qt[comp_tq[c]], sizeof(H->quant[c]));
    }
    if (total > 0x7fffffffll) return DIR_JPEG_E_UNSUPPORTED; /* the header's 32-bit offsets could not hold it (65535 x 65535 frames) */
    H->total_coef = (int32_t)total;
    H->magic = DIR_JPEG_MAGIC;
    if (sizeof(dir_jpeg_header) + (size_t)total * 2 > record_bytes) return DIR_JPEG_E_SPACE;
    int16_t* coef = (int16_t*)((char*)record + sizeof(dir_jpeg_header));
    memset(coef, 0, (size_t)total * 2);

    bits_t b = {scan, data + n, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    int count = 0;
    for (int my = 0; my < H->mcuy; ++my) {
        for (int mx = 0; mx < H->mcux; ++mx) {
            if (restart_interval && count && count % restart_interval == 0) {
                /* discard the padding bits; the reader stopped ON the marker (or has not reached it yet: scan forward) */
                b.acc = 0; b.n = 0; b.marker = 0;
                while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) ++b.p;
                if (b.p + 1 >= b.end) return DIR_JPEG_E_FORMAT;
                b.p += 2;
                pred[0] = pred[1] = pred[2] = 0;
            }
            ++count;
            for (int c = 0; c < H->ncomp; ++c) {
                const huff_t* hd = &dc[comp_td[c]];
                const huff_t* ha = &ac[comp_ta[c]];
                for (int by = 0; by < H->v[c]; ++by) {
                    for (int bx = 0; bx < H->h[c]; ++bx) {
                        int16_t* blk = coef + H->coef_offset[c] + ((int64_t)(my * H->v[c] + by) * H->blocks_x[c] + (mx * H->h[c] + bx)) * 64;
                        const int t = decode_symbol(&b, hd);
                        if (t < 0 || t > 11) return DIR_JPEG_E_FORMAT; /* 8-bit baseline: DC difference categories 0..11 (T.81 F.1.2.1.1) */
                        pred[c] += extend(get_bits(&b, t), t);
                        if (pred[c] < -32768 || pred[c] > 32767) return DIR_JPEG_E_FORMAT; /* a crafted stream must not wrap the predictor */
                        blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            if (b.n < 16) refill(&b);
                            const int fa = ha->fast_ac[(b.acc >> (b.n - LOOK_BITS)) & ((1u << LOOK_BITS) - 1u)];
                            if (fa) {
                                k += (fa >> 4) & 15;
                                if (k > 63) return DIR_JPEG_E_FORMAT;
                                b.n -= fa & 15;
                                blk[kZigzag[k++]] = (int16_t)(fa >> 8);
                                continue;
                            }
                            const int rs = decode_symbol(&b, ha);
                            if (rs < 0) return DIR_JPEG_E_FORMAT;
                            const int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break; /* EOB */
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return DIR_JPEG_E_FORMAT;
                            blk[kZigzag[k]] = (int16_t)extend(get_bits(&b, s), s);
                            ++k;
                        }
                    }
                }
            }
        }
    }
    return DIR_JPEG_OK;
}

size_t dir_jpeg_record_bytes(int width, int height, int hsamp, int vsamp, int ncomp) {
    if (width <= 0 || height <= 0 || hsamp < 1 || hsamp > 2 || vsamp < 1 || vsamp > 2 || (ncomp != 1 && ncomp != 3)) return 0;
    if (ncomp == 1) hsamp = vsamp = 1;
    const size_t mcux = ((size_t)width + 8u * (size_t)hsamp - 1) / (8u * (size_t)hsamp), mcuy = ((size_t)height + 8u * (size_t)vsamp - 1) / (8u * (size_t)vsamp);
    const size_t blocks = mcux * mcuy * ((size_t)hsamp * (size_t)vsamp + (ncomp == 3 ? 2u : 0u));
    return sizeof(dir_jpeg_header) + blocks * 128;
}

int dir_jpeg_abi_version(void) { return DIR_JPEG_ABI_VERSION; }
