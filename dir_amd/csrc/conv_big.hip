// The 256 x 256 block tile (DIR_CONV_VARIANT 11): the lever DESIGN.md 7 names for the MFMA-bound layers whose grid still fills the chip at
// this tile (at B = 64: conv_final and the merged seg / dense heads, 3x3 256 -> 256 at 32 x 32 -- 256 tiles).  Same maths, operand layout,
// K order and fp32 accumulation order as conv.hip / conv_pipe.hip (outputs are bit-identical); what changes is the shape of the work:
//
//   * 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32: a k16-step reads 6 fragments for 8 MFMAs (0.75 KB of LDS
//     per MFMA against 1 KB for the 64 x 64 wave tiles of conv_pipe.hip) and a K-slab moves (256 + 256) rows through L2 -> LDS for 2 x the
//     MFMAs of a 256 x 128 slab (32 B/clk/CU at full MFMA rate against 48);
//   * 128 accumulator registers per lane leave room for ONE step's fragments twice (48 registers: the next k16-step's six fragments are read
//     while the current step's eight MFMAs run), not for a second slab's: fragments are read from LDS as they are needed;
//   * a slab is 64 KB, so the ring is two deep: slab k + 1 is requested (LDS-DMA, untracked asm + hand-counted vmcnt as in conv_pipe.hip)
//     right after the barrier that ends slab k - 1, lands behind slab k's 32 MFMAs per wave and is published by the next barrier -- the
//     two-buffer structure the CDNA GEMM notes rate equal to deeper register pipelines at this tile size;
//   * the 256 x 256 fp32 tile does not fit the LDS: the epilogue leaves in two passes of 128 columns (j = 0, 1 of every wave's two column tiles).
//
// Replaces the same reference calls as conv.hip for the layers launch_conv_big() accepts (dense, no pre-activation, bf16 operands).
#include "conv_common.h"

namespace dir {
namespace convk {
namespace {

struct BSlab { int tap, toff, k0; };

template <typename TO, typename TIN = bf16_t>
__global__ __launch_bounds__(512, 1) void conv_big_kernel(ConvArgs a) {
    typedef TIN TI;
    half_kernel_init<TO>();
    constexpr int MI = 4, NJ = 2, WM = 2, WN = 4, NT = 512, BM = 256, BN = 256, ROW = 128;
    constexpr int RPP = NT / 8, ACH = BM / RPP, BCH = BN / RPP, NP = ACH + BCH;      // 64 rows per DMA pass; 4 + 4 pieces per thread and slab
    constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int SMEM = 2 * BUF_BYTES;                                              // = the 256 x 128 fp32 epilogue pass
    constexpr int ES = 2, BK = 64;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (see conv.hip)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    tile_of(a, bid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    const TI* __restrict__ x = (const TI*)a.x;
    const TI* __restrict__ w = (const TI*)a.w;
    const i32x4 xd = {(int)(unsigned)(unsigned long long)x, (int)(unsigned)((unsigned long long)x >> 32), (int)a.x_bytes, 0x00020000};
    const i32x4 wd = {(int)(unsigned)(unsigned long long)w, (int)(unsigned)((unsigned long long)w >> 32), (int)a.w_bytes, 0x00020000};
    constexpr unsigned OOB = 0x80000000u;

    // per-thread DMA source state: row (tid >> 3) + 64 i of the A / B tile, 16-byte chunk (tid & 7) ^ swizzle (conv_pipe.hip)
    int avoff[ACH];
    unsigned amask[ACH];
    unsigned bvoff[BCH];
    const int col = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int m = m0 + (tid >> 3) + RPP * i;
        avoff[i] = 0;
        amask[i] = 0;
        if (m < a.M) {
            int b, oy, ox;
            pixel_setup(a, m, col * 8 * ES, ES, avoff[i], amask[i], b, oy, ox);
        }
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = n0 + (tid >> 3) + RPP * i;
        bvoff[i] = n < a.Cout ? (unsigned)((n * a.K + col * 8) * ES) : OOB;
    }
    const int ntaps = a.kh * a.kw;
    int d_tap = 0, d_ky = 0, d_kx = 0, d_c0 = 0;          // (tap, channel slab) of the next slab to request: channel slab outer, taps inner
    auto next_slab = [&]() -> BSlab {
        BSlab s;
        s.tap = d_tap;
        s.toff = ((d_ky * a.W + d_kx) * a.in_cs + d_c0) * ES;
        s.k0 = d_tap * a.Cin + d_c0;
        ++d_tap;
        if (++d_kx == a.kw) { d_kx = 0; ++d_ky; }
        if (d_tap == ntaps) { d_tap = 0; d_ky = 0; d_kx = 0; d_c0 += BK; }
        return s;
    };
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    auto dma_slab = [&](const BSlab& s, int buf) {
        const unsigned ba = lds_base + buf * BUF_BYTES + wave * 1024;
#pragma unroll
        for (int p = 0; p < ACH; ++p) {
            const bool ok = (amask[p] >> s.tap) & 1u;
            lds_dma16_m0(xd, ba + p * (RPP * ROW), ok ? (unsigned)(avoff[p] + s.toff) : OOB, 0);
        }
#pragma unroll
        for (int p = 0; p < BCH; ++p) lds_dma16_m0(wd, ba + A_BYTES + p * (RPP * ROW), bvoff[p], (unsigned)(s.k0 * ES));
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing (conv.hip): lane (i = lane & 31, h = lane >> 5) reads chunk h * 4 + q of row i, un-swizzled
    const int frag_a = (wm * MI * 32 + (lane & 31)) * ROW;
    const int frag_b = A_BYTES + (wn * NJ * 32 + (lane & 31)) * ROW;
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = ((((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) << 4);

    uint4 fa[2][MI], fb[2][NJ];                            // the fragments of two k16-steps
    auto frag_read = [&](auto Set, int q, const char* bufp) {
        constexpr int set = decltype(Set)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const uint4*>(bufp + frag_a + i * 32 * ROW + qoff[q]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[set][j] = *reinterpret_cast<const uint4*>(bufp + frag_b + j * 32 * ROW + qoff[q]);
    };
    auto mma_step = [&](auto Set) {
        constexpr int set = decltype(Set)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = Half<TI>::mfma32(fa[set][i], fb[set][j], acc[i][j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    {   // prologue: slab 0 -> buffer 0
        const BSlab s0 = next_slab();
        dma_slab(s0, 0);
        wait_vmcnt<0>();
        __syncthreads();
    }
    for (int ks = 0; ks < a.nk; ++ks) {
        const int buf = ks & 1;
        // buffer buf ^ 1 was last read in iteration ks - 1, which every wave has left (barrier below): refill it behind this slab's MFMAs
        if (ks + 1 < a.nk) {
            const BSlab s1 = next_slab();
            dma_slab(s1, buf ^ 1);
        }
        const char* bufp = smem + buf * BUF_BYTES;
        frag_read(S0{}, 0, bufp);
        frag_read(S1{}, 1, bufp);
        mma_step(S0{});
        frag_read(S0{}, 2, bufp);
        mma_step(S1{});
        frag_read(S1{}, 3, bufp);
        mma_step(S0{});
        mma_step(S1{});
        wait_vmcnt<0>();                                   // own pieces of slab ks + 1 have landed; everyone's after the barrier
        __syncthreads();
    }

    // ---- epilogue in two passes of 128 columns: pass p takes every wave's column tile j = p (scale / shift in registers -> fp32 tile in LDS ->
    //      16-byte row segments (+ residual, ReLU) to HBM: the values of epilogue_tile)
    TO* __restrict__ y = (TO*)a.y;
    const TO* __restrict__ res = (const TO*)a.res;
    const bool relu = (a.flags & 1) != 0;
    float* st = reinterpret_cast<float*>(smem);
    constexpr int SBN = 128;
#pragma unroll
    for (int p = 0; p < NJ; ++p) {
        const int cl = wn * 32 + (lane & 31);                                   // column inside the pass: (wave column, lane)
        const int n = n0 + wn * NJ * 32 + p * 32 + (lane & 31);
        const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        const float sh = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                st[rl * SBN + cl] = fmaf(acc[i][p][r], sc, sh);
            }
        __syncthreads();
        constexpr int VN = OutVec<TO>::N, CPR = SBN / VN;
        for (int c = tid; c < BM * CPR; c += NT) {
            const int rl = c / CPR, cc = (c - rl * CPR) * VN;                   // cc: column inside the pass = 32 wn' + offset
            const int m = m0 + rl, nn = n0 + (cc >> 5) * (NJ * 32) + p * 32 + (cc & 31);
            if (m >= a.M || nn >= a.Cout) continue;
            float v[VN];
            const float4* sp = reinterpret_cast<const float4*>(st + rl * SBN + cc);
#pragma unroll
            for (int q = 0; q < VN / 4; ++q) { const float4 t = sp[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
            if (res) {
                float rv[VN];
                OutVec<TO>::load(res + (long long)m * a.res_cs + a.res_co + nn, rv);
#pragma unroll
                for (int e = 0; e < VN; ++e) v[e] += rv[e];
            }
            OutVec<TO>::store_act(y + (long long)m * a.out_cs + a.out_co + nn, v, relu);
        }
        if (p + 1 < NJ) __syncthreads();
    }
}

}  // namespace

// DIR_CONV_VARIANT 11: returns true if it took the launch (bf16 operands, dense, no pre-activation / second source / split output, vector epilogue)
bool launch_conv_big(const ConvArgs& a0, bool out_f32, hipStream_t s, bool f16) {
    if (a0.pre_scale || a0.bbox || a0.x2 || !(a0.flags & 4) || a0.nk < 1 || a0.out_split_scale > 0.f || a0.st_p1 || a0.mask || a0.bs_p1) return false;
    if (a0.Cin % 64 != 0 || a0.splits > 1) return false;
    if (a0.Cout <= 128 || a0.M <= 128) return false;                            // (a half-empty tile: the other variants serve these)
    ConvArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.Cout + 255) / 256;
    choose_tile_order(a, 2);
    const dim3 grid(a.tiles_m * a.tiles_n), block(512);
    if (f16 && out_f32) DIR_LAUNCH((conv_big_kernel<float, f16s_t>), grid, block, 0, s, a);
    else if (f16) DIR_LAUNCH((conv_big_kernel<f16s_t, f16s_t>), grid, block, 0, s, a);
    else if (out_f32) DIR_LAUNCH((conv_big_kernel<float>), grid, block, 0, s, a);
    else DIR_LAUNCH((conv_big_kernel<bf16_t>), grid, block, 0, s, a);
    return true;
}

}  // namespace convk
}  // namespace dir
