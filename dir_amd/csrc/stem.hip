// dir_stem_pool_forward: the ResNet stem in ONE kernel (bf16 mode)
//   models/backbone/resnet.py:244-247   conv1 7x7/s2/p3 (3 -> 64) -> bn1 (eval) -> ReLU -> MaxPool2d(3, 2, 1)
//   apps/eval.py:59-61                  (uint8 input) BGR -> RGB, / 255, (t - mean) / std, fused into the patch staging
//
// HBM-bound by construction: the image is read once (12.6 MB as uint8 / 50 MB as fp32 at B = 64) and the pooled NHWC map is
// written once (33.5 MB); the 128x128x64 conv output (134 MB in bf16) never leaves the CU.  The staged-GEMM path it replaces
// (dir_stem_prep_s2d -> dir_conv2d_forward -> dir_maxpool3x3s2) moves 98 + 98 + 134 + 134 + 33 MB through HBM in three
// launches.
//
// One persistent workgroup (4 waves) walks 8x8 tiles of the pooled map.  Per tile:
//   1. the 39x39x3 input patch (loaded into registers one tile ahead, consumed after the previous tile's MFMA phase so that
//      neither its latency nor the previous tile's store acknowledgements are waited for) -> LDS as [row][col][4] bf16
//      (channel 3 and column 39 are zero), the same fp32 -> bf16 rounding as the staged path;
//   2. the 17x17 conv pixels the tile's pooling windows touch, as a GEMM  D[channel][pixel] += W[channel][k] * P[k][pixel]
//      on v_mfma_f32_16x16x32_bf16 with k = (ky, kx (8, kx = 7 has zero weights), c (4)) = 7 steps of 32: the B operand
//      of step ky is ONE 16-byte LDS read per lane (two adjacent patch pixels), the A operand (weights, 28 fragments) lives
//      in registers for the whole kernel; epilogue = BN scale / shift, ReLU, bf16 -> LDS [pixel][64];
//      conv pixels outside the image are written as 0 (legal: every window holds a valid pixel and ReLU output >= 0);
//   3. 3x3/s2 max over the LDS tile -> 16-byte stores of the pooled NHWC rows.
#include "conv_common.h"

namespace dir {
namespace {

using convk::bf16_t;
using convk::f16s_t;
using convk::Half;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

// The VALU work per conv output is what bounds this kernel (75.8 M outputs at B = 64), so the epilogue and the pooling run on
// packed instructions: v_cvt_pk_bf16_f32 (round to nearest even, = convk::f2bf on finite values),
// and ReLU / max as v_pk_max_i16 on the bf16 bit patterns (sign bit set -> negative int16 -> 0; non-negative bf16 order = int16 order).
// H = the 16-bit storage kind of the output, the LDS tiles and the weights (bf16_t | f16s_t); f16 clamps to +-65504 first.  The packed
// max works on f16 bit patterns for the same reason (non-negative values: integer order = float order).
template <typename H> __device__ __forceinline__ unsigned pack_h(f32x2 v) { return Half<H>::pack2(v[0], v[1]); }
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b)));
}
template <typename H> __device__ __forceinline__ bf16_t cvt_h(float v) { return (bf16_t)(pack_h<H>(f32x2{v, 0.f}) & 0xffffu); }

constexpr int TP = 8;                    // pooled tile edge
constexpr int CR = 2 * TP + 1;           // conv rows / cols per tile (17)
constexpr int NPIX = CR * CR;            // 289
constexpr int NTILE = (NPIX + 15) / 16;  // 19 MFMA column tiles
constexpr int PR = 4 * TP + 7;           // patch rows / used cols (39)
constexpr int PW = 40;                   // patch cols incl. the zero column read by the kx = 7 lane group
constexpr int PPITCH = PW * 8;           // bytes per patch row
constexpr int CPITCH = 144;              // bytes per conv pixel in LDS (64 bf16 + 16 pad)
constexpr int NTHR = 256;
constexpr int RH = 20;                   // patch rows per thread: thread = (row half, element of a patch row)

struct NormArgs { float mean[3], stdv[3]; };

__device__ __forceinline__ float norm_px(unsigned char v, float mean, float stdv) {
#pragma clang fp contract(off)
    return ((float)v / 255.f - mean) / stdv;
}

struct StemArgs {
    const void* img; const convk::u32x4* w; const float* scale; const float* shift; bf16_t* y;
    int B, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x, ntiles;
    NormArgs nm;
    long long* stamps;      // DIR_STAMPS=stem (tuning aid, else NULL): workgroup 0, one stamp per phase of its first tiles
};

template <bool U8, typename H = bf16_t>
__global__ __launch_bounds__(NTHR, 2) void stem_pool_kernel(StemArgs a) {
    convk::half_kernel_init<H>();
    __shared__ __attribute__((aligned(16))) char s_patch[PR * PPITCH];
    __shared__ __attribute__((aligned(16))) char s_conv[NPIX * CPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ __attribute__((aligned(16))) float s_ss[128];
    __shared__ bf16_t s_lut[U8 ? 3 * 256 : 1];          // uint8 input: bf16(norm_px(v, mean[c], std[c])) per (c, v)
    const int g = lane >> 4, li = lane & 15;
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && blockIdx.x == 0 && tid == 0 && nstamp < dir::MAX_STAMPS) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    if (tid < 64) { s_ss[tid] = a.scale[tid]; s_ss[64 + tid] = a.shift[tid]; }
    if constexpr (U8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s_lut[c * 256 + tid] = cvt_h<H>(norm_px((unsigned char)tid, a.nm.mean[c], a.nm.stdv[c]));
    }

    // weights: A fragments, row = channel 16*mt + li, k = ky*32 + 8*g .. +8   (a.w = [64][28] 16-byte vectors)
    convk::u32x4 wf[4][7];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) wf[mt][ky] = a.w[(16 * mt + li) * 28 + ky * 4 + g];

    for (int i = tid; i < PR * PPITCH / 4; i += NTHR) reinterpret_cast<unsigned*>(s_patch)[i] = 0u;

    // patch staging: thread = (row half h, element j of a patch row), rows r = 20 h + k.  fp32 NCHW: j = c * 40 + col (plane c,
    // 39 consecutive floats); uint8 BGR HWC: j = byte of the row's 117 (col = j / 3, c = 2 - j % 3)
    constexpr int EPR = U8 ? 3 * PR : 3 * PW;
    const int half = tid >= EPR ? 1 : 0, jj = tid - half * EPR;
    const int pcol = U8 ? jj / 3 : jj % PW;
    const int pc = U8 ? 2 - (jj - pcol * 3) : jj / PW;
    const bool pact = tid < 2 * EPR && pcol < PR;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): each XCD walks one contiguous range of tiles, so the halo rows /
    // columns neighbouring tiles share are fetched into one L2 instead of eight
    int t, tstep, tend;
    if ((gridDim.x & 7) == 0 && a.ntiles % 8 == 0) {
        const int per = a.ntiles >> 3;
        t = (blockIdx.x & 7) * per + (blockIdx.x >> 3); tstep = gridDim.x >> 3; tend = ((blockIdx.x & 7) + 1) * per;
    } else { t = blockIdx.x; tstep = gridDim.x; tend = a.ntiles; }
    // raw prefetch registers + validity bits: every load is unconditional (clamped address) so that all 20 are in flight
    // together, and nothing consumes them before the next tile's staging (no wait in front of the MFMA phase)
    typedef typename std::conditional<U8, unsigned char, float>::type raw_t;
    raw_t pre[RH];
    unsigned okmask = 0;
    auto tile_origin = [&](int tt, int& b, int& py0, int& px0) {
        const int tx = tt % a.tiles_x, r = tt / a.tiles_x;
        b = r / a.tiles_y; py0 = (r - b * a.tiles_y) * TP; px0 = tx * TP;
    };
    auto fetch = [&](int tt) {
        int b, py0, px0;
        tile_origin(tt, b, py0, px0);
        const int ix = 4 * px0 - 5 + pcol, iy0 = 4 * py0 - 5 + half * RH;
        const bool colok = pact && ix >= 0 && ix < a.W;
        const int ixc = min(max(ix, 0), a.W - 1), pcc = pc < 3 ? pc : 0;
        const raw_t* plane = U8 ? (const raw_t*)a.img + (long long)b * a.H * a.W * 3 + ixc * 3 + (2 - pcc)
                                : (const raw_t*)a.img + ((long long)b * 3 + pcc) * a.H * a.W + ixc;
        const int rowstride = U8 ? a.W * 3 : a.W;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            const int iy = iy0 + k;
            if (colok && iy >= 0 && iy < a.H && half * RH + k < PR) okmask |= 1u << k;
            pre[k] = plane[min(max(iy, 0), a.H - 1) * rowstride];
        }
    };
    auto stage = [&]() {                                      // pre[] (this tile's raw patch) -> bf16 LDS patch
        bf16_t* sp = reinterpret_cast<bf16_t*>(s_patch);
        if (pact) {
#pragma unroll
            for (int k = 0; k < RH; ++k)
                if (half * RH + k < PR) {
                    bf16_t v;
                    if constexpr (U8) v = s_lut[pc * 256 + pre[k]]; else v = cvt_h<H>(pre[k]);
                    sp[((half * RH + k) * PW + pcol) * 4 + pc] = (okmask >> k) & 1u ? v : (bf16_t)0;
                }
        }
    };
    // B fragments of one 16-pixel column tile: step ky = the two patch pixels (2 xc + 2 g, +1) of patch row 2 yc + ky
    auto load_b = [&](int nt, uint4 (&bv)[7]) {
        const int n = nt * 16 + li;
        const int pix = n < NPIX ? n : NPIX - 1;
        const int yc = pix / CR, xc = pix - yc * CR;
        const char* bp = s_patch + (2 * yc) * PPITCH + (2 * xc + 2 * g) * 8;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) bv[ky] = *reinterpret_cast<const uint4*>(bp + ky * PPITCH);
    };

    if (t < tend) fetch(t);
    // all weight fragments landed before the tile loop: no vmcnt wait inside the MFMA phase, where the next tile's patch
    // loads are in flight
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) asm volatile("" ::"v"(wf[mt][ky]));
    __syncthreads();
    if (t < tend) stage();
    __syncthreads();
    if (t + tstep < tend) fetch(t + tstep);
    stamp();

    for (; t < tend; t += tstep) {
        int b, py0, px0;
        tile_origin(t, b, py0, px0);
        // ---- conv pixels of this tile
        for (int nt = wave; nt < NTILE; nt += 4) {
            uint4 bcur[7];
            load_b(nt, bcur);                                     // all 7 reads in flight; the MFMAs consume them as they land
            __builtin_amdgcn_sched_barrier(0);
            const int n = nt * 16 + li;
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = Half<H>::mfma16(wf[mt][ky], bcur[ky], acc[mt]);
            {
                const int nn = n < NPIX ? n : NPIX - 1;              // lanes past the last pixel rewrite pixel 288 with its own value
                const int yc = nn / CR, xc = nn - yc * CR;
                const int ycg = 2 * py0 - 1 + yc, xcg = 2 * px0 - 1 + xc;
                const unsigned vm = (ycg >= 0 && ycg < a.Hc && xcg >= 0 && xcg < a.Wc && n < NPIX) ? 0xffffffffu : 0u;
                float4 sc[4], sh[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    sc[mt] = *reinterpret_cast<const float4*>(s_ss + 16 * mt + 4 * g);
                    sh[mt] = *reinterpret_cast<const float4*>(s_ss + 64 + 16 * mt + 4 * g);
                }
                if (n < NPIX) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const f32x2 v01 = f32x2{acc[mt][0], acc[mt][1]} * f32x2{sc[mt].x, sc[mt].y} + f32x2{sh[mt].x, sh[mt].y};
                        const f32x2 v23 = f32x2{acc[mt][2], acc[mt][3]} * f32x2{sc[mt].z, sc[mt].w} + f32x2{sh[mt].z, sh[mt].w};
                        uint2 o;
                        o.x = pk_max_i16(pack_h<H>(v01), 0u) & vm;
                        o.y = pk_max_i16(pack_h<H>(v23), 0u) & vm;
                        *reinterpret_cast<uint2*>(s_conv + n * CPITCH + (16 * mt + 4 * g) * 2) = o;
                    }
                }
            }
        }
        __syncthreads();                                          // s_conv complete, s_patch free
        stamp();

        // ---- next tile's patch -> LDS: its loads were issued a whole MFMA phase ago, the previous stores even earlier
        if (t + tstep < tend) stage();
        stamp();

        // ---- 3x3 / s2 max-pool of the tile, 8 channels (16 bytes) per item
        for (int it = tid; it < TP * TP * 8; it += NTHR) {
            const int cg = it & 7, p = it >> 3, pyl = p / TP, pxl = p - pyl * TP;
            const int py = py0 + pyl, px = px0 + pxl;
            if (py >= a.Hp || px >= a.Wp) continue;
            uint4 o = {0u, 0u, 0u, 0u};                        // post-ReLU values: bf16 order == int16 order
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 v = *reinterpret_cast<const uint4*>(s_conv + ((2 * pyl + dy) * CR + 2 * pxl + dx) * CPITCH + cg * 16);
                    o.x = pk_max_i16(o.x, v.x); o.y = pk_max_i16(o.y, v.y); o.z = pk_max_i16(o.z, v.z); o.w = pk_max_i16(o.w, v.w);
                }
            *reinterpret_cast<uint4*>(a.y + ((((long long)b * a.Hp + py) * a.Wp + px) * 64 + cg * 8)) = o;
        }
        stamp();
        if (t + 2 * tstep < tend) fetch(t + 2 * tstep);   // lands during the next tile's MFMA phase
        __syncthreads();                                          // next patch complete, s_conv free
        stamp();
    }
}

}  // namespace
}  // namespace dir

// w_packed: [64][7][8][4] bf16 = w[n, c, ky, kx] at [n][ky][kx][c], zero for kx = 7 and c = 3; scale / shift: folded bn1.
// img: fp32 NCHW [B,3,H,W] (img_dtype DIR_DT_F32, already normalised) or uint8 BGR HWC [B,H,W,3] (DIR_DT_U8; mean / std =
// host pointers to 3 floats).  y: bf16 NHWC [B, H/4, W/4, 64].
extern "C" int dir_stem_pool_forward_dt(const void* img, int img_dtype, int out_dtype, const float* mean_host, const float* std_host, const void* w_packed,
                                        const float* scale, const float* shift, void* y, int B, int H, int W, void* stream) {
    using namespace dir;
    DIR_REQUIRE(out_dtype == DIR_DT_BF16 || out_dtype == DIR_DT_F16, "dir_stem_pool_forward_dt: out_dtype must be DIR_DT_BF16 or DIR_DT_F16");
    DIR_REQUIRE(img && w_packed && scale && shift && y && B > 0, "dir_stem_pool_forward: bad args");
    DIR_REQUIRE(H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "dir_stem_pool_forward: H and W must be positive multiples of 4");
    DIR_REQUIRE(img_dtype == DIR_DT_F32 || img_dtype == DIR_DT_U8, "dir_stem_pool_forward: img_dtype must be DIR_DT_F32 or DIR_DT_U8");
    StemArgs a;
    a.img = img; a.w = (const convk::u32x4*)w_packed; a.scale = scale; a.shift = shift; a.y = (convk::bf16_t*)y;
    a.B = B; a.H = H; a.W = W; a.Hc = H / 2; a.Wc = W / 2; a.Hp = H / 4; a.Wp = W / 4;
    a.tiles_y = (a.Hp + TP - 1) / TP; a.tiles_x = (a.Wp + TP - 1) / TP;
    const long long nt = (long long)B * a.tiles_y * a.tiles_x;
    DIR_REQUIRE(nt < (1ll << 31), "dir_stem_pool_forward: too many tiles");
    a.ntiles = (int)nt;
    for (int c = 0; c < 3; ++c) { a.nm.mean[c] = 0.f; a.nm.stdv[c] = 1.f; }
    if (img_dtype == DIR_DT_U8) {
        DIR_REQUIRE(mean_host && std_host, "dir_stem_pool_forward: uint8 input needs mean / std");
        for (int c = 0; c < 3; ++c) {
            DIR_REQUIRE(std_host[c] != 0.f, "dir_stem_pool_forward: std must be non-zero");
            a.nm.mean[c] = mean_host[c]; a.nm.stdv[c] = std_host[c];
        }
    }
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0; hipDeviceProp_t p;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    const int grid = (int)(nt < 2ll * num_cu ? nt : 2ll * num_cu);
    hipStream_t s = (hipStream_t)stream;
    a.stamps = dir::stamps_begin("stem");
    if (out_dtype == DIR_DT_F16) {
        if (img_dtype == DIR_DT_U8) DIR_LAUNCH((stem_pool_kernel<true, f16s_t>), dim3(grid), dim3(NTHR), 0, s, a);
        else DIR_LAUNCH((stem_pool_kernel<false, f16s_t>), dim3(grid), dim3(NTHR), 0, s, a);
    } else if (img_dtype == DIR_DT_U8) DIR_LAUNCH((stem_pool_kernel<true, bf16_t>), dim3(grid), dim3(NTHR), 0, s, a);
    else DIR_LAUNCH((stem_pool_kernel<false, bf16_t>), dim3(grid), dim3(NTHR), 0, s, a);
    dir::stamps_end("stem", a.stamps, s);
    return check_launch("dir_stem_pool_forward");
}

extern "C" int dir_stem_pool_forward(const void* img, int img_dtype, const float* mean_host, const float* std_host, const void* w_packed,
                                     const float* scale, const float* shift, void* y, int B, int H, int W, void* stream) {
    return dir_stem_pool_forward_dt(img, img_dtype, DIR_DT_BF16, mean_host, std_host, w_packed, scale, shift, y, B, H, W, stream);
}
