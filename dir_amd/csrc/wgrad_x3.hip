// dir_conv2d_wgrad_f16x3: d loss / d weight of nn.Conv2d (train.py:68 runs autograd through every Conv2d of models/backbone/resnet.py,
// models/backbone/hourglass.py and models/dir.py) on the f16 matrix cores in split precision -- the arithmetic of DIR_DT_F16X3
// (conv_common.h): both fp32 operands are split into f16 hi + lo, every product is hi*hi + lo*hi + hi*lo with fp32 accumulation
// (~2^-22 per product, below the fp32 accumulation noise of the 10^3 .. 10^5-term pixel sums this kernel forms).
//
//   gw[co][tap][ci] = sum over output pixels m of gy[m][co] * x[pixel(m, tap)][ci]
//
// is a GEMM whose REDUCTION runs over pixels while both tensors are stored pixel-major (NHWC): the MFMA wants 8 consecutive pixels of
// one channel per lane.  The transpose happens on the way into LDS: a thread loads 4 pixels x 4 channels (four 16-byte loads, one step
// ahead of the MFMAs), splits them, and writes per channel the 4 hi halves (8 bytes) and the 4 lo halves (8 bytes) into a
// [channel][32 pixels hi | 32 pixels lo] image -- the 128-byte row the forward kernels use for a 32-channel slab, so the fragment
// addressing (16-byte chunk c of row r at c ^ ((r >> 1) & 7): conflict-free ds_read_b128) and mma_slab<f16x3_t> are shared with them.
// One workgroup = 4 waves = a (TM output channels) x (TN input channels) tile of one tap over one chunk of the pixels; LDS double
// buffered, one barrier per 32-pixel step.  Partial tiles of the pixel chunks are added in chunk order by wgrad_x3_reduce_kernel
// (deterministic, no atomics), like dir_conv2d_wgrad_f32 (train_ops.hip), whose result layout this reproduces.
#include "conv_common.h"

#include <stdlib.h>

namespace dir {
namespace {

using convk::f32x16;

constexpr int XK = 32;                 // pixels per step

struct WgradX3Args {
    const float* x; const float* gy; float* out;        // out: gw, or the workspace [chunks][Cout][taps][Cin]
    int B, H, W, Cin, in_cs, in_co, Cout, gy_cs, gy_co, kh, kw, stride, pad, Ho, Wo, M, chunk, tiles_ci;
    float sx, sg, inv;                                   // power-of-two operand scales and 1 / (sx * sg)
    const float* pre_s; const float* pre_b; int pre_relu; // optional pre-activation of the x operand: x <- act(x * pre_s[c] + pre_b[c]) per input channel
                                                         // (round 5: the BatchNorm + ReLU between two convolutions, applied where x is read -- the
                                                         // normalised map is never written; padding taps stay zero: the mask is applied after it)
    unsigned mg_hw, sh_hw, mg_w, sh_w;                   // m / (Ho * Wo), r / Wo (convk::magic_u31)
};

template <int TM, int TN, int ROW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(WgradX3Args a) {
    constexpr int MI = TM / 64, NJ = TN / 64;            // 32 x 32 blocks per wave (waves 2 x 2)
    constexpr int BUF = (TM + TN) * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l32 = lane & 31, h = lane >> 5;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci, tap = blockIdx.y, ch = blockIdx.z;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int co0 = tco * TM, ci0 = tci * TN;
    const int m_beg = ch * a.chunk, m_end = min(a.M, m_beg + a.chunk);
    const int hw = a.Ho * a.Wo;

    // loader role: pixels 4 pg .. 4 pg + 3 of the step, channels 4 cq .. 4 cq + 3 of either tile
    const int pg = tid & 7, cq = tid >> 3;
    // Every thread ALWAYS issues its 8 loads (threads outside a narrow or ragged tile re-read a valid quad and store nothing / zeros; steps
    // past the chunk re-read its last pixels): with loads under a branch hipcc cannot count what is in flight and waits vmcnt(0) before
    // every use, which serialises the two register sets below.
    const int cqa = cq % (TM / 4), cqb = cq % (TN / 4);
    const bool a_on = co0 + 4 * cqa < a.Cout, b_on = ci0 + 4 * cqb < a.Cin;
    const int cha = a.gy_co + (a_on ? co0 + 4 * cqa : co0), chb = a.in_co + (b_on ? ci0 + 4 * cqb : ci0);
    const float sga = a_on ? a.sg : 0.f, sxb = b_on ? a.sx : 0.f;
    const bool pre = a.pre_s != nullptr;
    float4 ps4 = make_float4(1.f, 1.f, 1.f, 1.f), pb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pre && b_on) {
        ps4 = *reinterpret_cast<const float4*>(a.pre_s + ci0 + 4 * cqb);
        pb4 = *reinterpret_cast<const float4*>(a.pre_b + ci0 + 4 * cqb);
    }
    // two register sets: the operands of step k + 2 are requested while step k runs and are written to LDS at the end of step k + 1 --
    // one workgroup has two steps of global loads in flight (two workgroups per CU: four), which covers an HBM round trip
    float4 ra[2][4], rb[2][4];
    unsigned oka[2] = {0, 0}, okb[2] = {0, 0};           // per pixel: inside the chunk / inside the image
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[p][e] = rb[p][e] = make_float4(0.f, 0.f, 0.f, 0.f);
    // per-lane constants of the ROW == 32 form: byte offsets of the lane's 4 pixels in the gy row piece, their column offsets, the x pixel pitch
    unsigned goff[4];
    int pixs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { goff[e] = (unsigned)((4 * pg + e) * a.gy_cs + cha) * 4u; pixs[e] = (4 * pg + e) * a.stride; }
    const unsigned xpitch = (unsigned)a.in_cs * 4u, xoff = (unsigned)chb * 4u;
    auto gload = [&](auto P, int k0) {
        constexpr int p = decltype(P)::value;
        const float* pa[4];
        const float* pb[4];
        oka[p] = okb[p] = 0;
        if constexpr (ROW == 32) {
            // Wo % 32 == 0 (the 64x64 and 32x32 maps: the large-M layers): the 32 pixels of a step are one piece of one output row, so the
            // pixel -> (image, row, column) division, the row's base addresses and the row mask are UNIFORM -- scalar-unit work -- and a lane
            // adds constant 32-bit offsets.  The per-lane form below spent ~45 quarter-rate 32 / 64-bit multiplies per lane and step on this, and
            // the kernel was VALU-bound (1900 VALU cycles per wave and step against 768 of MFMA).
            const int k0u = __builtin_amdgcn_readfirstlane(k0);
            const bool sv = k0u < m_end;                 // (chunks are whole steps and M % 32 == 0: the whole step or nothing)
            const int mc0 = sv ? k0u : m_end - XK;
            const int b = convk::div_magic(mc0, a.mg_hw, a.sh_hw), r = mc0 - b * hw;
            const int oy = convk::div_magic(r, a.mg_w, a.sh_w), ox0 = r - oy * a.Wo;
            const int iy = oy * a.stride - a.pad + ky, ixb = ox0 * a.stride - a.pad + kx;
            const bool rowin = sv && iy >= 0 && iy < a.H;
            const char* xrow = reinterpret_cast<const char*>(a.x + ((long long)b * a.H + min(max(iy, 0), a.H - 1)) * a.W * a.in_cs);
            const char* grow = reinterpret_cast<const char*>(a.gy + (long long)mc0 * a.gy_cs);
            oka[p] = sv ? 15u : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = ixb + pixs[e];
                okb[p] |= ((rowin && ix >= 0 && ix < a.W) ? 1u : 0u) << e;
                pa[e] = reinterpret_cast<const float*>(grow + goff[e]);
                pb[e] = reinterpret_cast<const float*>(xrow + (__umul24((unsigned)min(max(ix, 0), a.W - 1), xpitch) + xoff));
            }
        } else if constexpr (ROW == 4) {                 // Wo % 4 == 0: a thread's 4 pixels are neighbours in one output row
            const int m = k0 + 4 * pg;
            const bool mv = m < m_end;                   // (chunks are whole steps of 32 and M % 4 == 0: all four or none)
            const int mc = mv ? m : m_end - 4;
            const int b = convk::div_magic(mc, a.mg_hw, a.sh_hw), r = mc - b * hw;
            const int oy = convk::div_magic(r, a.mg_w, a.sh_w), ox = r - oy * a.Wo;
            const int iy = oy * a.stride - a.pad + ky, ix0 = ox * a.stride - a.pad + kx;
            const bool rowin = mv && iy >= 0 && iy < a.H;
            const float* xrow = a.x + ((long long)b * a.H + min(max(iy, 0), a.H - 1)) * a.W * a.in_cs + chb;
            const float* grow = a.gy + (long long)mc * a.gy_cs + cha;
            oka[p] = mv ? 15u : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = ix0 + e * a.stride;
                okb[p] |= ((rowin && ix >= 0 && ix < a.W) ? 1u : 0u) << e;
                pa[e] = grow + (long long)e * a.gy_cs;
                pb[e] = xrow + (long long)min(max(ix, 0), a.W - 1) * a.in_cs;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                // unconditional loads from clamped addresses, masked when stored
                const int m = k0 + 4 * pg + e;
                const bool mv = m < m_end;
                const int mc = mv ? m : m_end - 1;
                const int b = convk::div_magic(mc, a.mg_hw, a.sh_hw), r = mc - b * hw;
                const int oy = convk::div_magic(r, a.mg_w, a.sh_w), ox = r - oy * a.Wo;
                const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                const bool in = mv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1);
                oka[p] |= (mv ? 1u : 0u) << e;
                okb[p] |= (in ? 1u : 0u) << e;
                pa[e] = a.gy + (long long)mc * a.gy_cs + cha;
                pb[e] = a.x + (((long long)b * a.H + iyc) * a.W + ixc) * a.in_cs + chb;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[p][e] = *reinterpret_cast<const float4*>(pa[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[p][e] = *reinterpret_cast<const float4*>(pb[e]);
    };
    // 4 pixels x 4 channels -> per channel 4 hi halves + 4 lo halves (8 bytes each) of row 4 cq + c; masked pixels are scaled by 0
    auto lstore_tile = [&](char* base, const float4 (&r)[4], unsigned ok, float s, bool act) {
        float v[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w},
                         {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
        if (act) {                                       // (uniform per launch)
            const float sc[4] = {ps4.x, ps4.y, ps4.z, ps4.w}, sh[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float t = fmaf(v[e][c], sc[c], sh[c]);
                    v[e][c] = a.pre_relu ? fmaxf(t, 0.f) : t;
                }
        }
        const float se[4] = {(ok & 1u) ? s : 0.f, (ok & 2u) ? s : 0.f, (ok & 4u) ? s : 0.f, (ok & 8u) ? s : 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint4 t = make_uint4(__float_as_uint(v[0][c] * se[0]), __float_as_uint(v[1][c] * se[1]), __float_as_uint(v[2][c] * se[2]),
                                       __float_as_uint(v[3][c] * se[3]));
            const uint4 sp = convk::split_f16x3(t, 1.f);
            const int row = 4 * cq + c, swz = (row >> 1) & 7;
            char* rp = base + row * 128 + (pg & 1) * 8;
            *reinterpret_cast<uint2*>(rp + (((pg >> 1) ^ swz) << 4)) = make_uint2(sp.x, sp.y);
            *reinterpret_cast<uint2*>(rp + (((4 + (pg >> 1)) ^ swz) << 4)) = make_uint2(sp.z, sp.w);
        }
    };
    auto lstore = [&](auto P, int buf) {
        constexpr int p = decltype(P)::value;
        char* as = smem + buf * BUF;
        if (4 * cq < TM) lstore_tile(as, ra[p], oka[p], sga, false);
        if (4 * cq < TN) lstore_tile(as + TM * 128, rb[p], okb[p], sxb, pre);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment q = 2 s + part of row r: k16-step s, part 0 = hi (chunk 2 s + h), 1 = lo (chunk 4 + 2 s + h) -- mma_slab<f16x3_t>'s order
    auto frag = [&](const char* base, int row, uint4 (&f)[4]) {
        const int swz = (row >> 1) & 7;
        const char* rp = base + row * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f[2 * s] = *reinterpret_cast<const uint4*>(rp + (((2 * s + h) ^ swz) << 4));
            f[2 * s + 1] = *reinterpret_cast<const uint4*>(rp + (((4 + 2 * s + h) ^ swz) << 4));
        }
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    gload(P0{}, m_beg);
    gload(P1{}, m_beg + XK);
    lstore(P0{}, 0);
    __syncthreads();
    // step k (LDS buffer k & 1, register set k & 1 free again): request step k + 2, MFMAs of step k, step k + 1's registers -> other buffer
    auto step = [&](auto P, int k0, int buf) {
        constexpr int p = decltype(P)::value;
        using Q = std::integral_constant<int, p ^ 1>;
        gload(P, k0 + 2 * XK);                           // (past the chunk: clamped re-reads, never stored)
        const char* as = smem + buf * BUF;
        uint4 af[MI][4], bf[NJ][4];
#pragma unroll
        for (int i = 0; i < MI; ++i) frag(as, wm * (TM / 2) + 32 * i + l32, af[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) frag(as + TM * 128, wn * (TN / 2) + 32 * j + l32, bf[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) convk::mma_slab<convk::f16x3_t>(af[i], bf[j], acc[i][j]);
        lstore(Q{}, buf ^ 1);                            // that buffer was last read before the previous step's barrier (past the chunk: zeros)
        __syncthreads();
    };
    for (int k0 = m_beg; k0 < m_end; k0 += 2 * XK) {
        step(P0{}, k0, 0);
        step(P1{}, k0 + XK, 1);                          // (a chunk of an odd number of steps: one step of zeros)
    }

    const int taps = a.kh * a.kw;
    float* out = a.out + (long long)ch * a.Cout * taps * a.Cin;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = ci0 + wn * (TN / 2) + 32 * j + l32;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * (TM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < a.Cout) out[((long long)co * taps + tap) * a.Cin + ci] = acc[i][j][r] * a.inv;
            }
        }
}

__global__ __launch_bounds__(256) void wgrad_x3_reduce_kernel(const float* part, float* gw, long long n, int chunks, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? gw[i] : 0.f;
    s = dir::sum_in_order(part + i, n, chunks, s);
    gw[i] = s;
}

int x3_tile(int c) { return c > 64 ? 128 : 64; }
// workgroups of a tile shape the whole GPU holds at once (occupancy x CUs), asked from the runtime once per shape
// (once per PROCESS, for the device current at the first call: one process drives one GPU here.  The chunk count -- hence the summation order and the
// last bits of a weight gradient -- follows CU count and occupancy: gradients reproduce run to run on one device model, not across models.)
template <int TM, int TN>
int x3_slots_of() {
    static const int slots = []() {
        int dev = 0, cus = 256, occ = 2;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv_wgrad_x3_kernel<TM, TN, 4>, 256, 0) != hipSuccess || occ < 1) occ = 2;
        return cus * occ;
    }();
    return slots;
}
int x3_slots(int tm, int tn) {
    return tm == 128 ? (tn == 128 ? x3_slots_of<128, 128>() : x3_slots_of<128, 64>()) : (tn == 128 ? x3_slots_of<64, 128>() : x3_slots_of<64, 64>());
}
// Number of pixel chunks.  Round 3 took ceil(768 / workgroups per chunk) -- "about three workgroups per CU" -- which ignores how the grid
// quantises: the fusion convolution's weight gradient (360 workgroups per chunk) ran 1080 workgroups on 512 slots = three rounds for 2.1 rounds
// of work, the 128-channel 3x3 layers 576 on 512.  Now: the c that minimises rounds(c) x (steps per chunk + a fixed prologue / epilogue) plus
// the reduce pass over c partial copies, in units of one 32-pixel step (~1.8 us).  Deterministic for a given shape and device.
int x3_chunks(const dir_conv_desc* d, long long M) {
    const int tm = x3_tile(d->Cout), tn = x3_tile(d->Cin);
    const long long per = (long long)((d->Cin + tn - 1) / tn) * ((d->Cout + tm - 1) / tm) * d->kh * d->kw;
    const long long cmax = (M + 511) / 512;               // at least 512 pixels (16 steps) per chunk
    const long long slots = x3_slots(tm, tn), steps_total = (M + XK - 1) / XK;
    const double n = (double)d->Cout * d->kh * d->kw * d->Cin;
    long long best = 1;
    double best_cost = 1e30;
    for (long long c = 1; c <= cmax && c <= 1024; ++c) {
        const long long rounds = (per * c + slots - 1) / slots, steps = (steps_total + c - 1) / c;
        const double cost = (double)rounds * (double)(steps + 24) + (c > 1 ? 3.0 + (double)(c + 1) * n * 7.4e-7 : 0.0);
        if (cost < best_cost * 0.999) { best_cost = cost; best = c; }
    }
    return (int)best;
}
bool pow2(float s) { int e; return s > 0.f && frexpf(s, &e) == 0.5f; }

}  // namespace
}  // namespace dir

extern "C" long long dir_conv2d_wgrad_f16x3_workspace_bytes(const dir_conv_desc* d) {
    if (!d) return -1;
    const int Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    const int Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    const int c = dir::x3_chunks(d, (long long)d->B * Ho * Wo);
    return c > 1 ? (long long)c * d->Cout * d->kh * d->kw * d->Cin * 4 : 0;
}

static int wgrad_f16x3(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                      long long workspace_bytes, float x_scale, float gy_scale, const float* pre_scale, const float* pre_shift, int pre_relu, void* stream) {
    using namespace dir;
    DIR_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr) && (((uintptr_t)pre_scale | (uintptr_t)pre_shift) & 15) == 0,
                "dir_conv2d_wgrad_f16x3_pre: pre_scale / pre_shift go together, 16-byte aligned");
    DIR_REQUIRE(d && x && gy && gw, "dir_conv2d_wgrad_f16x3: null pointer");
    DIR_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0,
                "dir_conv2d_wgrad_f16x3: bad geometry");
    DIR_REQUIRE(pow2(x_scale) && pow2(gy_scale), "dir_conv2d_wgrad_f16x3: the operand scales must be powers of two");
    WgradX3Args a;
    a.x = x; a.gy = gy; a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cs = d->in_cstride ? d->in_cstride : d->Cin; a.in_co = d->in_coff;
    a.Cout = d->Cout; a.gy_cs = d->out_cstride ? d->out_cstride : d->Cout; a.gy_co = d->out_coff;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    a.Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DIR_REQUIRE(a.Ho > 0 && a.Wo > 0, "dir_conv2d_wgrad_f16x3: empty output");
    DIR_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0 && a.in_cs % 4 == 0 && a.in_co % 4 == 0 && a.gy_cs % 4 == 0 && a.gy_co % 4 == 0 &&
                    ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0,
                "dir_conv2d_wgrad_f16x3: channel counts / strides / offsets must be multiples of 4 and the tensors 16-byte aligned (use dir_conv2d_wgrad_f32)");
    const long long M = (long long)d->B * a.Ho * a.Wo;
    DIR_REQUIRE(M < (1ll << 31), "dir_conv2d_wgrad_f16x3: too many output pixels");
    a.M = (int)M;
    const int chunks = x3_chunks(d, M);
    const long long n = (long long)d->Cout * d->kh * d->kw * d->Cin;
    const bool direct = chunks == 1 && !accumulate;
    DIR_REQUIRE(direct || (workspace && workspace_bytes >= (long long)chunks * n * 4),
                "dir_conv2d_wgrad_f16x3: workspace too small (dir_conv2d_wgrad_f16x3_workspace_bytes; accumulate needs at least the weight size)");
    a.chunk = (int)(((M + chunks - 1) / chunks + XK - 1) / XK * XK);
    a.out = direct ? gw : workspace;
    a.sx = x_scale; a.sg = gy_scale; a.inv = 1.f / (x_scale * gy_scale);
    a.pre_s = pre_scale; a.pre_b = pre_shift; a.pre_relu = pre_relu;
    convk::magic_u31((unsigned)(a.Ho * a.Wo), &a.mg_hw, &a.sh_hw);
    convk::magic_u31((unsigned)a.Wo, &a.mg_w, &a.sh_w);
    const int tm = x3_tile(d->Cout), tn = x3_tile(d->Cin);
    a.tiles_ci = (d->Cin + tn - 1) / tn;
    const dim3 grid(a.tiles_ci * ((d->Cout + tm - 1) / tm), d->kh * d->kw, chunks);
    hipStream_t s = (hipStream_t)stream;
    const bool row4 = a.Wo % 4 == 0;                      // then M % 4 == 0 too and a thread's 4 pixels share an output row
    static const bool row32_on = []() { const char* e = getenv("DIR_WGRAD_ROW32"); return !(e && e[0] == '0'); }();
    // a step's 32 pixels are one piece of one output row: uniform row arithmetic (needs 24-bit pitches and 32-bit byte offsets inside a row)
    const bool row32 = row32_on && a.Wo % 32 == 0 && a.W < (1 << 20) && (long long)a.in_cs * 4 < (1 << 24) && (long long)a.W * a.in_cs * 4 < (1ll << 31) &&
                       (long long)XK * a.gy_cs * 4 < (1ll << 31);
    auto launch = [&](auto TMc, auto TNc) {
        constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value;
        if (row32) DIR_LAUNCH((conv_wgrad_x3_kernel<TM, TN, 32>), grid, dim3(256), 0, s, a);
        else if (row4) DIR_LAUNCH((conv_wgrad_x3_kernel<TM, TN, 4>), grid, dim3(256), 0, s, a);
        else DIR_LAUNCH((conv_wgrad_x3_kernel<TM, TN, 0>), grid, dim3(256), 0, s, a);
    };
    using C64 = std::integral_constant<int, 64>;
    using C128 = std::integral_constant<int, 128>;
    if (tm == 128 && tn == 128) launch(C128{}, C128{});
    else if (tm == 128) launch(C128{}, C64{});
    else if (tn == 128) launch(C64{}, C128{});
    else launch(C64{}, C64{});
    if (!direct) DIR_LAUNCH(wgrad_x3_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace, gw, n, chunks, accumulate);
    return check_launch("dir_conv2d_wgrad_f16x3");
}

extern "C" int dir_conv2d_wgrad_f16x3(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                                      long long workspace_bytes, float x_scale, float gy_scale, void* stream) {
    return wgrad_f16x3(d, x, gy, gw, accumulate, workspace, workspace_bytes, x_scale, gy_scale, nullptr, nullptr, 0, stream);
}
extern "C" int dir_conv2d_wgrad_f16x3_pre(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                                          long long workspace_bytes, float x_scale, float gy_scale, const float* pre_scale, const float* pre_shift,
                                          int pre_relu, void* stream) {
    return wgrad_f16x3(d, x, gy, gw, accumulate, workspace, workspace_bytes, x_scale, gy_scale, pre_scale, pre_shift, pre_relu, stream);
}
