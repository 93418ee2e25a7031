// dir_jpeg_decode_records: the GPU half of the from-files input path.
//
//   reference   apps/eval.py:56  cv.imread(<split>/img/<idx>.jpg)   (dataset/interhand.py:223; files written by dataset/prepare_data.py:123-166)
//               = libjpeg's default decode: Huffman -> dequantise + "islow" IDCT -> "fancy" chroma upsampling -> YCbCr -> BGR uint8
//
// The host decodes the entropy-coded stream only (csrc/jpeg_huff.c -> one RECORD per image: a 512-byte dir_jpeg_header + the quantised int16
// coefficients); a batch of records arrives here by one DMA and two launches produce the uint8 BGR frames [B,H,W,3] the stem kernel reads
// (dir_stem_pool_forward_dt, img_dtype = DIR_DT_U8):
//   jpeg_idct_kernel    one thread per 8x8 block: dequantise, jidctint.c's two-pass integer IDCT (CONST_BITS 13, PASS1_BITS 2; 64-bit
//                       intermediates like the LP64 library), jdmaster.c's range limit (value & 1023 into the centred table) -> the component's
//                       sample plane (MCU-padded, pitch = 8 x blocks per row)
//   jpeg_color_kernel   one thread per 4 output pixels: jdsample.c's h2v2 / h2v1 triangle filters with their alternating rounding constants and
//                       edge cases (rows beyond the first / last real row repeat it: jdmainct.c; components at most two samples wide use the box
//                       filter), jdcolor.c's 16-bit fixed-point YCbCr -> RGB, stored B, G, R like cv.imread
// Integer arithmetic throughout: BIT-EXACT with libjpeg-turbo (tests/test_gpu_jpeg.py against oracle/jpeg.py, which tests/test_jpeg_oracle.py pins
// to Pillow's libjpeg-turbo on the same files).  HBM-bound in principle (~0.6 MB per 256x256 image through both launches); at the model's
// ~31 k images/s that is 19 GB/s -- the kernels are written for exactness and simplicity, not for the roofline.
#include "dir_common.h"

#include "../../include/dir_jpeg.h"

namespace {

__device__ __forceinline__ long long descale(long long x, int n) { return (x + (1ll << (n - 1))) >> n; }

// jidctint.c's butterfly on eight values (columns in pass 1, rows in pass 2)
__device__ __forceinline__ void idct8(const int (&v)[8], int shift, int (&o)[8]) {
    constexpr long long F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                        F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                        F_2_562915447 = 20995, F_3_072711026 = 25172;
    long long z1 = ((long long)v[2] + v[6]) * F_0_541196100;
    const long long tmp2 = z1 + (long long)v[6] * (-F_1_847759065);
    const long long tmp3 = z1 + (long long)v[2] * F_0_765366865;
    const long long tmp0 = ((long long)v[0] + v[4]) << 13;
    const long long tmp1 = ((long long)v[0] - v[4]) << 13;
    const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    long long t0 = v[7], t1 = v[5], t2 = v[3], t3 = v[1];
    z1 = t0 + t3;
    long long z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const long long z5 = (z3 + z4) * F_1_175875602;
    t0 *= F_0_298631336; t1 *= F_2_053119869; t2 *= F_3_072711026; t3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447;
    z3 = z3 * (-F_1_961570560) + z5;
    z4 = z4 * (-F_0_390180644) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    o[0] = (int)descale(tmp10 + t3, shift); o[7] = (int)descale(tmp10 - t3, shift);
    o[1] = (int)descale(tmp11 + t2, shift); o[6] = (int)descale(tmp11 - t2, shift);
    o[2] = (int)descale(tmp12 + t1, shift); o[5] = (int)descale(tmp12 - t1, shift);
    o[3] = (int)descale(tmp13 + t0, shift); o[4] = (int)descale(tmp13 - t0, shift);
}
// jdmaster.c prepare_range_limit_table as the IDCT indexes it: clamp(v + 128, 0, 255) for |v| < 512, the table's wrap-around beyond
__device__ __forceinline__ unsigned range_limit_idct(int v) {
    const int idx = v & 1023;
    return idx < 128 ? (unsigned)(idx + 128) : idx < 512 ? 255u : idx < 896 ? 0u : (unsigned)(idx - 896);
}

struct JpegArgs {
    const char* records; long long stride;      // B records, `stride` bytes apart
    unsigned char* planes; long long pstride;   // scratch: one byte per coefficient, per image
    unsigned char* out;                          // [B][H][W][3] BGR
    int B, H, W, max_blocks;
    int* err;                                    // set to 1 + image index when a record does not describe an H x W image this kernel handles
};

// A record is trusted for plane addressing and divisions only if its geometry is the one its own (W, H, sampling) imply -- a slot of the shared-memory
// ring that a crashed worker left half-written, or a stale one, must raise the error flag instead of writing outside the image's plane scratch or
// dividing by blocks_x == 0.  Supported sampling: luma (1,1), (2,1), (2,2), chroma (1,1) -- what jpeg_huff.c emits.
__device__ __forceinline__ bool record_ok(const dir_jpeg_header* h, int H, int W, long long pstride) {
    if (!(h->magic == DIR_JPEG_MAGIC && h->width == W && h->height == H && (h->ncomp == 1 || h->ncomp == 3) && h->total_coef > 0 && h->total_coef <= pstride))
        return false;
    const int hm = h->hmax, vm = h->vmax;
    if (!((hm == 1 && vm == 1) || (hm == 2 && vm == 1) || (hm == 2 && vm == 2))) return false;
    if (h->h[0] != hm || h->v[0] != vm || (h->ncomp == 1 && hm != 1)) return false;
    const int mcux = (W + 8 * hm - 1) / (8 * hm), mcuy = (H + 8 * vm - 1) / (8 * vm);
    if (h->mcux != mcux || h->mcuy != mcuy) return false;
    long long total = 0;
    for (int c = 0; c < h->ncomp; ++c) {
        if (c > 0 && (h->h[c] != 1 || h->v[c] != 1)) return false;
        if (h->blocks_x[c] != mcux * h->h[c] || h->blocks_y[c] != mcuy * h->v[c] || h->coef_offset[c] != total) return false;
        total += (long long)h->blocks_x[c] * h->blocks_y[c] * 64;
    }
    return total == h->total_coef;
}
__device__ __forceinline__ bool record_is_pixels(const dir_jpeg_header* h, int H, int W, long long stride) {
    return h->magic == DIR_JPEG_MAGIC_PIXELS && h->width == W && h->height == H && (long long)sizeof(dir_jpeg_header) + 3ll * H * W <= stride;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(JpegArgs a) {
    const int img = blockIdx.y;
    const dir_jpeg_header* h = reinterpret_cast<const dir_jpeg_header*>(a.records + (long long)img * a.stride);
    if (!record_ok(h, a.H, a.W, a.pstride)) {
        if (threadIdx.x == 0 && blockIdx.x == 0 && !record_is_pixels(h, a.H, a.W, a.stride)) atomicMax(a.err, img + 1);
        return;
    }
    const int bi = blockIdx.x * 256 + threadIdx.x;
    if (bi >= h->total_coef / 64) return;
    int c = 0, rel = bi;                              // component of this block and its index inside the component
    while (c + 1 < h->ncomp && rel >= h->blocks_x[c] * h->blocks_y[c]) { rel -= h->blocks_x[c] * h->blocks_y[c]; ++c; }
    const short* coef = reinterpret_cast<const short*>(a.records + (long long)img * a.stride + sizeof(dir_jpeg_header)) + (long long)bi * 64;
    const unsigned short* q = h->quant[c];
    int ws[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint4 v = *reinterpret_cast<const uint4*>(coef + 8 * r);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ws[8 * r + 2 * e] = (int)(short)(u[e] & 0xffffu) * (int)q[8 * r + 2 * e];
            ws[8 * r + 2 * e + 1] = (int)(short)(u[e] >> 16) * (int)q[8 * r + 2 * e + 1];
        }
    }
#pragma unroll
    for (int col = 0; col < 8; ++col) {                // pass 1: columns, descale by CONST_BITS - PASS1_BITS
        int v[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = ws[8 * r + col];
        idct8(v, 11, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[8 * r + col] = o[r];
    }
    const int by = rel / h->blocks_x[c], bx = rel - by * h->blocks_x[c];
    const int pitch = h->blocks_x[c] * 8;
    unsigned char* plane = a.planes + (long long)img * a.pstride + h->coef_offset[c] + (long long)(by * 8) * pitch + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {                      // pass 2: rows, descale by CONST_BITS + PASS1_BITS + 3, range limit
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ws[8 * r + k];
        idct8(v, 18, o);
        uint2 px;
        px.x = range_limit_idct(o[0]) | (range_limit_idct(o[1]) << 8) | (range_limit_idct(o[2]) << 16) | (range_limit_idct(o[3]) << 24);
        px.y = range_limit_idct(o[4]) | (range_limit_idct(o[5]) << 8) | (range_limit_idct(o[6]) << 16) | (range_limit_idct(o[7]) << 24);
        *reinterpret_cast<uint2*>(plane + (long long)r * pitch) = px;
    }
}

// one chroma sample at full resolution: jdsample.c fullsize / h2v1_fancy / h2v2_fancy (box filters for components at most two samples wide)
__device__ __forceinline__ int chroma_at(const unsigned char* p, int pitch, int dw, int dh, int fh, int fv, int x, int y) {
    if (fh == 1 && fv == 1) return p[(long long)y * pitch + x];
    const int cx = x >> 1;
    if (fv == 1) {                                      // h2v1
        const unsigned char* row = p + (long long)y * pitch;
        if (dw <= 2) return row[cx];
        const int cur = row[cx];
        if (x & 1) return cx == dw - 1 ? cur : (3 * cur + row[cx + 1] + 2) >> 2;
        return cx == 0 ? cur : (3 * cur + row[cx - 1] + 1) >> 2;
    }
    const int cy = y >> 1;
    if (dw <= 2) return p[(long long)cy * pitch + cx];
    int oy = (y & 1) ? cy + 1 : cy - 1;                 // the nearer neighbour row; beyond the real rows: the edge row again (jdmainct.c)
    oy = oy < 0 ? 0 : oy > dh - 1 ? dh - 1 : oy;
    const unsigned char* r0 = p + (long long)cy * pitch;
    const unsigned char* r1 = p + (long long)oy * pitch;
    const int cur = 3 * r0[cx] + r1[cx];
    if (x & 1) return cx == dw - 1 ? (cur * 4 + 7) >> 4 : (3 * cur + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
    return cx == 0 ? (cur * 4 + 8) >> 4 : (3 * cur + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegArgs a) {
    const int img = blockIdx.y;
    const dir_jpeg_header* h = reinterpret_cast<const dir_jpeg_header*>(a.records + (long long)img * a.stride);
    const int wq = (a.W + 3) >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wq * a.H) return;
    const int y = t / wq, x0 = (t - y * wq) * 4;
    if (!record_ok(h, a.H, a.W, a.pstride)) {            // (an unusable record was reported by the IDCT launch)
        if (record_is_pixels(h, a.H, a.W, a.stride)) {   // a frame the host decoded itself: copied through
            const unsigned char* src = reinterpret_cast<const unsigned char*>(h) + sizeof(dir_jpeg_header) + ((long long)y * a.W + x0) * 3;
            unsigned char* o = a.out + ((long long)img * a.H * a.W + (long long)y * a.W + x0) * 3;
            for (int k = 0; k < 3 * min(4, a.W - x0); ++k) o[k] = src[k];
        }
        return;
    }
    const unsigned char* base = a.planes + (long long)img * a.pstride;
    const unsigned char* py = base + h->coef_offset[0] + (long long)y * (h->blocks_x[0] * 8);
    unsigned char px[12];
    // ---- the files of the path (dataset/prepare_data.py:123-166: cv.imwrite's default 4:2:0) in 9 dword loads instead of 36 byte loads: a thread's
    //      four pixels need luma x0 .. x0 + 3 and, per chroma component and row, the samples cx0 - 1 .. cx0 + 2 (cx0 = x0 / 2); same formulas as
    //      chroma_at (h2v2_fancy), so the same bits.  W % 8 == 0, H even, at least 8 chroma columns.
    if (h->ncomp == 3 && h->hmax == 2 && h->vmax == 2 && h->h[0] == 2 && h->v[0] == 2 && h->h[1] == 1 && h->v[1] == 1 && h->h[2] == 1 && h->v[2] == 1 &&
        (a.W & 7) == 0 && (a.H & 1) == 0 && a.W >= 16) {
        const int dw = a.W >> 1, dh = a.H >> 1, cx0 = x0 >> 1, cy = y >> 1;
        int oy = (y & 1) ? cy + 1 : cy - 1;
        oy = oy < 0 ? 0 : oy > dh - 1 ? dh - 1 : oy;
        const unsigned yq = *reinterpret_cast<const unsigned*>(py + x0);
        const int s0 = cx0 - 1, a0 = s0 & ~3, sh = (s0 - a0) * 8;          // first sample wanted, its dword, its bit offset (cx0 is even: 8 or 24)
        int cur[2][4];                                                     // [component][k]: 3 * r0[cx0 - 1 + k] + r1[cx0 - 1 + k]
#pragma unroll
        for (int c = 1; c <= 2; ++c) {
            const int pitch = h->blocks_x[c] * 8;
            const unsigned char* pl = base + h->coef_offset[c];
            unsigned long long v[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned char* row = pl + (long long)(r == 0 ? cy : oy) * pitch;
                const unsigned lo = a0 >= 0 ? *reinterpret_cast<const unsigned*>(row + a0) : 0u;      // (cx0 = 0: sample -1 does not exist and is not used)
                const int a1 = min(a0 + 4, pitch - 4);                                                 // (the last quad: sample dw may lie past the row; not used)
                unsigned hi = *reinterpret_cast<const unsigned*>(row + a1);
                if (a1 != a0 + 4) hi = 0u;
                v[r] = ((unsigned long long)hi << 32 | lo) >> sh;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[c - 1][k] = 3 * (int)((v[0] >> (8 * k)) & 255u) + (int)((v[1] >> (8 * k)) & 255u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int Y = (int)((yq >> (8 * e)) & 255u), cx = cx0 + (e >> 1), k = 1 + (e >> 1);
            int cc[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int m = cur[c][k];
                int val;
                if (e & 1) val = cx == dw - 1 ? (m * 4 + 7) >> 4 : (3 * m + cur[c][k + 1] + 7) >> 4;
                else val = cx == 0 ? (m * 4 + 8) >> 4 : (3 * m + cur[c][k - 1] + 8) >> 4;
                cc[c] = val - 128;
            }
            px[3 * e] = (unsigned char)clamp255(Y + ((116130 * cc[0] + 32768) >> 16));
            px[3 * e + 1] = (unsigned char)clamp255(Y + ((-22554 * cc[0] + 32768 - 46802 * cc[1]) >> 16));
            px[3 * e + 2] = (unsigned char)clamp255(Y + ((91881 * cc[1] + 32768) >> 16));
        }
        unsigned* o32 = reinterpret_cast<unsigned*>(a.out + ((long long)img * a.H * a.W + (long long)y * a.W + x0) * 3);
#pragma unroll
        for (int k = 0; k < 3; ++k) o32[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int x = min(x0 + e, a.W - 1);
        const int Y = py[x];
        int r = Y, g = Y, b = Y;
        if (h->ncomp == 3) {
            int cc[2];
#pragma unroll
            for (int c = 1; c <= 2; ++c) {
                const int fh = h->hmax / h->h[c], fv = h->vmax / h->v[c];
                const int dw = (a.W * h->h[c] + h->hmax - 1) / h->hmax, dh = (a.H * h->v[c] + h->vmax - 1) / h->vmax;
                cc[c - 1] = chroma_at(base + h->coef_offset[c], h->blocks_x[c] * 8, dw, dh, fh, fv, x, y) - 128;
            }
            // jdcolor.c: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768, SCALEBITS = 16
            r = clamp255(Y + ((91881 * cc[1] + 32768) >> 16));
            g = clamp255(Y + ((-22554 * cc[0] + 32768 - 46802 * cc[1]) >> 16));
            b = clamp255(Y + ((116130 * cc[0] + 32768) >> 16));
        }
        px[3 * e] = (unsigned char)b; px[3 * e + 1] = (unsigned char)g; px[3 * e + 2] = (unsigned char)r;
    }
    unsigned char* o = a.out + ((long long)img * a.H * a.W + (long long)y * a.W + x0) * 3;
    if (x0 + 4 <= a.W && ((a.W * 3) & 3) == 0) {
        unsigned* o32 = reinterpret_cast<unsigned*>(o);
#pragma unroll
        for (int k = 0; k < 3; ++k) o32[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
    } else {
        for (int k = 0; k < 3 * min(4, a.W - x0); ++k) o[k] = px[k];
    }
}

}  // namespace

extern "C" long long dir_jpeg_planes_bytes(long long record_bytes) { return record_bytes > (long long)sizeof(dir_jpeg_header) ? (record_bytes - (long long)sizeof(dir_jpeg_header)) / 2 : -1; }

extern "C" int dir_jpeg_decode_records(const void* records, long long record_stride, int B, int H, int W, void* planes_scratch, long long scratch_bytes, void* out_bgr,
                                       int32_t* err_flag, void* stream) {
    DIR_REQUIRE(records && planes_scratch && out_bgr && err_flag && B >= 0 && H > 0 && W > 0, "dir_jpeg_decode_records: bad arguments");
    DIR_REQUIRE(record_stride >= (long long)sizeof(dir_jpeg_header) + 128 && record_stride % 16 == 0 && ((uintptr_t)records & 15) == 0,
                "dir_jpeg_decode_records: records must be 16-byte aligned and at least a header + one block apart");
    if (B == 0) return DIR_OK;
    const long long pstride = ((record_stride - (long long)sizeof(dir_jpeg_header)) / 2 + 15) / 16 * 16;
    DIR_REQUIRE(scratch_bytes >= pstride * B, "dir_jpeg_decode_records: scratch of %lld bytes needed (dir_jpeg_planes_bytes(record_stride) rounded up to 16, per image)", pstride * B);
    JpegArgs a;
    a.records = (const char*)records; a.stride = record_stride; a.planes = (unsigned char*)planes_scratch; a.pstride = pstride; a.out = (unsigned char*)out_bgr;
    a.B = B; a.H = H; a.W = W; a.max_blocks = (int)((record_stride - (long long)sizeof(dir_jpeg_header)) / 128); a.err = err_flag;
    hipStream_t s = (hipStream_t)stream;
    DIR_LAUNCH(jpeg_idct_kernel, dim3((a.max_blocks + 255) / 256, B), dim3(256), 0, s, a);
    const int nthr = ((W + 3) / 4) * H;
    DIR_LAUNCH(jpeg_color_kernel, dim3((nthr + 255) / 256, B), dim3(256), 0, s, a);
    return dir::check_launch("dir_jpeg_decode_records");
}
