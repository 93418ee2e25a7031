// Implicit-GEMM 2-D convolution on the gfx950 matrix cores (a1/a2/a3/a11: every Conv2d on the DIR path).
//
//   y[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] ),   m = (b,oy,ox) output pixel, n = output channel,
//   k = (ky,kx,c) with c fastest  ->  activations NHWC, weights [Cout][kh][kw][Cin]: both operands are
//   K-contiguous, so the A gather (im2col) is a 128-byte coalesced row segment per pixel and tap.
//
// One K-slab = 128 bytes per row for both precisions (32 fp32 / 64 bf16 channels).  A 32x32 MFMA lane
// (i = lane&31, h = lane>>5) owns the contiguous 64-byte half h of row i of the slab: 4 x ds_read_b128
// feed 16 x v_mfma_f32_32x32x2_f32 (fp32: exact fmaf chain) or 4 x v_mfma_f32_32x32x16_bf16.  Any
// bijection of k is legal as long as A and W use the same one.
//
// This file: the general 4-wave kernel.  Block = 256 threads (2x2 waves); tile (64|128)(M) x (64|128)(N) picked per layer
// (launch_conv, or forced through DIR_CONV_VARIANT); each wave owns MI x NJ MFMA tiles of 32x32 (up to 64 acc VGPRs).
//   * K-slabs travel global -> LDS by DMA (buffer_load ... lds; hardware bounds check = zero padding and M / N tails), rows
//     unpadded with an XOR swizzle applied on the SOURCE address so ds_read_b128 is conflict free; 2 LDS buffers, or a
//     3-buffer ring with two slabs in flight (untracked asm DMA + counted vmcnt) for long reductions.
//   * The pre-activation variant (hourglass.Residual's BatchNorm+ReLU on the input, models/backbone/hourglass.py:55-70) must
//     touch the data in registers: global -> registers -> LDS (rows padded 128 -> 144 B), two register stages.
//   * Short reductions (K <= 512, the HBM-bound 1x1 bottleneck layers) prefetch their residual tile at kernel start.
//   * Epilogue: per-channel scale/shift (folded BatchNorm / bias), optional residual add, optional ReLU, optional channel
//     offset/stride so a conv writes straight into a slice of a concat buffer; the fp32 tile is staged through LDS so HBM
//     sees 16-byte coalesced row segments for output and residual.
// Layers with K >= 512 and enough tiles go to the 8-wave pipelined kernel of conv_pipe.hip instead (same results).
//
// Replaces the ATen/MKL-DNN (cuDNN in the original) calls under models/backbone/resnet.py:120-140,243-255,
// models/backbone/hourglass.py:10-30,55-70 and models/dir.py:57-62,227-241,404-420.
#include "conv_common.h"

#include <stdlib.h>

using namespace dir::convk;

namespace dir { namespace convk { thread_local int stats_rows_launched = 0; } }

namespace {

constexpr long long SPLITK_COUNTER_BYTES = 16384;      // head of a split-K workspace: one arrival counter per output tile (<= 4096 tiles)

// MI x NJ = 32x32 MFMA tiles per wave; block tile (2*MI*32) x (2*NJ*32), 2x2 waves.
template <typename TI, typename TO, int MI, int NJ, bool PRE, bool RING, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    half_kernel_init<TO>();
    // DMA: K-slabs go global -> LDS directly (buffer_load ... lds): no VGPR round trip, no ds_write.  The LDS image of a
    // wave-level DMA is lane-linear (8 rows x 128 B per instruction), so rows are unpadded and the bank-conflict-free
    // layout is obtained by XOR-swizzling the 16-byte chunk index on the SOURCE address: chunk c of row r lives at
    // position c ^ ((r >> 1) & 7).  The prologue variant must touch the data in registers and keeps the staged path.
    // f16x3 (fp32 tensors, split-precision arithmetic): the activations are split into f16 hi / lo parts on their way into LDS, so that
    // mode always takes the register-staged path, with or without the pre-activation
    constexpr bool X1 = std::is_same<TI, f16x1_t>::value;            // hi parts only: same staging and LDS layout
    constexpr bool X3 = std::is_same<TI, f16x3_t>::value || X1;
    // pre-split activations (dir_split_f16_forward): the hi | lo row layout of X3, but delivered by DMA like bf16 rows
    constexpr bool XP = std::is_same<TI, f16x3p_t>::value || std::is_same<TI, f16x1p_t>::value;
    constexpr bool DMA = !PRE && !X3;
    constexpr int ROW = DMA ? 128 : LDS_STRIDE;
    constexpr int BM = 64 * MI, BN = 64 * NJ;
    constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW, BUF_BYTES = A_BYTES + B_BYTES;
    // RING3: three LDS buffers, two K-slabs in flight (untracked asm DMA + counted vmcnt); used when three buffers still
    // leave >= 2 workgroups per CU (every tile shape except 128x128, which keeps the 2-buffer compiler-tracked path)
    constexpr bool RING3 = DMA && RING && !(MI == 2 && NJ == 2);
    constexpr int NBUF = RING3 ? 3 : 2;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = (NBUF * BUF_BYTES > STAGE_BYTES) ? NBUF * BUF_BYTES : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    constexpr int EPC = Tr<TI>::EPC, BK = Tr<TI>::BK;
    constexpr int ACH = BM / 32, BCH = BN / 32;      // 16-byte chunks per thread per K-slab (A, B)

    // XCD-aware tile order: contiguous tile ranges share one XCD's L2 (block b runs on XCD b % 8)
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    tile_of(a, bid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const TI* __restrict__ x = (const TI*)a.x;
    const TI* __restrict__ w = (const TI*)a.w;

    // ---- per-thread loader state.  Loads are hardware-bounds-checked buffer loads: an out-of-range byte offset
    //      returns zeros, so zero padding, the M tail and the Cout tail cost one v_cndmask instead of branches.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, a.w_bytes, 0x00020000);
    const i32x4 xd = {(int)(unsigned)(unsigned long long)x, (int)(unsigned)((unsigned long long)x >> 32), (int)a.x_bytes, 0x00020000};
    const i32x4 wd = {(int)(unsigned)(unsigned long long)w, (int)(unsigned)((unsigned long long)w >> 32), (int)a.w_bytes, 0x00020000};
    constexpr unsigned OOB = 0x80000000u;            // > any buffer size accepted by the host wrapper (< 2 GiB)
    constexpr int ES = (int)sizeof(TI);
    int avoff[ACH];                                  // byte offset of (b, iy0, ix0, chunk) -- may be negative
    unsigned amask[ACH];                             // bit t: tap t of this row is inside the image (and m < M)
    unsigned avoff2[ACH];                            // second source (1x1, stride2): byte offset of the row's pixel, OOB if m >= M
    const bool dual = a.x2 != nullptr;
    unsigned bvoff[BCH];
    // row handled by this thread in chunk-group i is (tid >> 3) + 32*i: ((row >> 1) & 7) == (tid >> 4) & 7 for every i
    const int col = DMA ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7);
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        avoff[i] = 0;
        amask[i] = 0;
        avoff2[i] = 0x80000000u;
        if (m < a.M) {
            int b, oy, ox;
            pixel_setup(a, m, col * EPC * ES, ES, avoff[i], amask[i], b, oy, ox);
            if (dual) avoff2[i] = (unsigned)((((b * a.H2 + oy * a.stride2) * a.W2 + ox * a.stride2) * a.in_cs2 + a.in_co2 + col * EPC) * ES);
        } else if (dual) avoff2[i] = 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        bvoff[i] = n < a.Cout ? (unsigned)((n * a.K + col * EPC) * ES) : OOB;
    }
    const TI* __restrict__ x2 = (const TI*)a.x2;
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void*)x2, 0, a.x2_bytes, 0x00020000);
    const i32x4 x2d = {(int)(unsigned)(unsigned long long)x2, (int)(unsigned)((unsigned long long)x2 >> 32), (int)a.x2_bytes, 0x00020000};

    const int ntaps = a.kh * a.kw;

    // Short reductions (the 1x1 bottleneck expansions, K <= 512) are HBM-bound and a workgroup lives for little more than
    // two memory round trips: operands, then the residual in the epilogue.  Fetch the residual tile NOW, in the chunk
    // layout the coalesced epilogue uses, so both trips overlap (16-32 VGPRs held across a loop of at most 8 slabs).
    constexpr int RVN = OutVec<TO>::N, RCPR = BN / RVN, RCH = BM * RCPR / 256;
    uint4 rpre[sizeof(TO) == 2 ? RCH : 1];
    const bool respre = !SPLIT && a.res != nullptr && (a.flags & 4) && a.nk <= 8 && sizeof(TO) == 2;
    if (respre) {
        const TO* __restrict__ resp = (const TO*)a.res;
#pragma unroll
        for (int k = 0; k < (sizeof(TO) == 2 ? RCH : 1); ++k) {
            const int c = tid + 256 * k, rl = c / RCPR, cc = (c - rl * RCPR) * RVN;
            const int m = m0 + rl, n = n0 + cc;
            rpre[k] = make_uint4(0, 0, 0, 0);
            if (m < a.M && n < a.Cout) rpre[k] = *reinterpret_cast<const uint4*>(resp + (long long)m * a.res_cs + a.res_co + n);
        }
    }
    u32x4 ra[2][ACH], rb[2][BCH];                    // two register stages: prefetch distance 2
    const bool pre_relu = (a.flags & 2) != 0;

    auto gload = [&](auto P, int ks) {
        constexpr int p = decltype(P)::value;
        // ONE straight-line load sequence for both sources (descriptor / offsets chosen by scalar selects): loads on two sides of a branch
        // make hipcc wait for all of them (vmcnt(0)) at the first use.  Second source (ks >= nk1): slab (ks - nk1) of x2's channels, 1x1.
        // K order of the first source: channel slab outer, taps inner -- consecutive slabs touch the same pixels' cache lines (shifted
        // by one tap), so the im2col re-reads hit L1/L2 instead of re-streaming the feature map once per tap
        const bool second = ks >= a.nk1;
        const int c2 = (ks - a.nk1) * BK;
        const int cs = ks / ntaps, tap = second ? 0 : ks - cs * ntaps;
        const int c0 = cs * BK, k0 = second ? a.nk1 * BK + c2 : tap * a.Cin + c0;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int toff = ((ky * a.W + kx) * a.in_cs + c0) * ES;
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? x2 : x), 0, second ? a.x2_bytes : a.x_bytes, 0x00020000);
        const int asoff = second ? c2 * ES : 0;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const bool ok = (amask[i] >> tap) & 1u;
            const unsigned vo = second ? avoff2[i] : ok ? (unsigned)(avoff[i] + toff) : OOB;
            ra[p][i] = __builtin_amdgcn_raw_buffer_load_b128(ar, vo, asoff, 0);     // RAW: transformed when written to LDS (stage_value), so
        }                                                                           // that nothing waits for the load inside this function
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            rb[p][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, bvoff[i], k0 * ES, 0));
        }
    };
    // the pre-activation and the f16 hi / lo split of a staged A chunk of slab ks, applied on its way from the register stage to LDS
    auto stage_value = [&](u32x4 r, int i, int ks) -> uint4 {
        uint4 v = __builtin_bit_cast(uint4, r);
        if constexpr (PRE) {
            if (ks < a.nk1) {
                const int cs = ks / ntaps, tap = ks - cs * ntaps;
                if ((amask[i] >> tap) & 1u) v = prologue<TI>(v, a.pre_scale, a.pre_shift, cs * BK + col * EPC, pre_relu);
            }
        }
        if constexpr (X1) v = split_f16x1(v, a.a_scale); else if constexpr (X3) v = split_f16x3(v, a.a_scale);                        // {hi01, hi23, lo01, lo23}
        return v;
    };
    auto lstore = [&](auto P, int buf, int ks) {
        constexpr int p = decltype(P)::value;
        char* sa = smem + buf * BUF_BYTES;
        char* sb = sa + A_BYTES;
        const int off0 = (tid >> 3) * LDS_STRIDE + col * 16;
        if constexpr (PRE || X3) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) ra[p][i] = __builtin_bit_cast(u32x4, stage_value(ra[p][i], i, ks));
        }
        if constexpr (X3) {
            // LDS row of a 32-channel slab = [hi: 32 f16 | lo: 32 f16] (the layout the host packs the weights in): this thread's four
            // channels are 8 bytes of each half
            const int offa = (tid >> 3) * LDS_STRIDE + col * 8;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                *reinterpret_cast<uint2*>(sa + offa + 32 * i * LDS_STRIDE) = make_uint2(ra[p][i].x, ra[p][i].y);
                *reinterpret_cast<uint2*>(sa + offa + 64 + 32 * i * LDS_STRIDE) = make_uint2(ra[p][i].z, ra[p][i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ACH; ++i) *reinterpret_cast<u32x4*>(sa + off0 + 32 * i * LDS_STRIDE) = ra[p][i];
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) *reinterpret_cast<u32x4*>(sb + off0 + 32 * i * LDS_STRIDE) = rb[p][i];
    };

    // LDS-DMA of one K-slab into buffer `buf`: wave w writes row groups (w + 4*i) * 8 .. +7 (1 KiB each)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma = [&](int buf, int ks) {
        if (ks >= a.nk1) {                               // second source: slab (ks - nk1) of x2's channels, 1x1
            const int c2 = (ks - a.nk1) * BK;
            char* sa2 = smem + buf * BUF_BYTES + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < ACH; ++i) lds_dma16(x2r, sa2 + i * 4096, avoff2[i], (unsigned)(c2 * ES));
#pragma unroll
            for (int i = 0; i < BCH; ++i) lds_dma16(wr, sa2 + A_BYTES + i * 4096, bvoff[i], (unsigned)((a.nk1 * BK + c2) * ES));
            return;
        }
        const int cs = ks / ntaps, tap = ks - cs * ntaps;
        const int c0 = cs * BK, k0 = tap * a.Cin + c0;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int toff = ((ky * a.W + kx) * a.in_cs + c0) * ES;
        char* sa = smem + buf * BUF_BYTES + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const bool ok = (amask[i] >> tap) & 1u;
            const unsigned vo = ok ? (unsigned)(avoff[i] + toff) : OOB;       // out of range -> zeros land in LDS
            lds_dma16(xr, sa + i * 4096, vo, 0);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i)
            lds_dma16(wr, sa + A_BYTES + i * 4096, bvoff[i], (unsigned)(k0 * ES));
    };

    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    auto dma_ring = [&](int buf, int ks, bool live) {      // !live: every lane out of range -> zeros, branch-free code
        if (ks >= a.nk1) {                               // second source (never reached when !live: ks is then 0)
            const int c2 = (ks - a.nk1) * BK;
            const unsigned sa2 = lds_base + buf * BUF_BYTES + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < ACH; ++i) lds_dma16_untracked(x2d, sa2 + i * 4096, live ? avoff2[i] : OOB, (unsigned)(c2 * ES));
#pragma unroll
            for (int i = 0; i < BCH; ++i)
                lds_dma16_untracked(wd, sa2 + A_BYTES + i * 4096, live ? bvoff[i] : OOB, (unsigned)((a.nk1 * BK + c2) * ES));
            return;
        }
        const int cs = ks / ntaps, tap = ks - cs * ntaps;
        const int c0 = cs * BK, k0 = tap * a.Cin + c0;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        const int toff = ((ky * a.W + kx) * a.in_cs + c0) * ES;
        const unsigned sa = lds_base + buf * BUF_BYTES + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const bool ok = ((amask[i] >> tap) & 1u) && live;
            lds_dma16_untracked(xd, sa + i * 4096, ok ? (unsigned)(avoff[i] + toff) : OOB, 0);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i)
            lds_dma16_untracked(wd, sa + A_BYTES + i * 4096, live ? bvoff[i] : OOB, (unsigned)(k0 * ES));
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: lane (i = lane & 31, h = lane >> 5) reads the 64-byte half h of row i
    // (f16x3: fragment q = 2*s + part reads the 16 bytes of k16-step s, half h, of the row's hi (part 0) or lo (part 1) half)
    const int frag_off = (lane & 31) * ROW + (DMA || X3 ? 0 : (lane >> 5) * 64);
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        qoff[q] = X3 ? 32 * (q >> 1) + 16 * (lane >> 5) + 64 * (q & 1)
                  : XP ? ((((2 * (q >> 1) + (lane >> 5)) + 4 * (q & 1)) ^ ((lane >> 1) & 7)) << 4)      // chunk (2s + h) of the hi half, + 4 for the lo half, swizzled
                  : DMA ? ((((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) << 4) : q * 16;

    // ---- optional sparse-K: compact, ordered list of the K-slabs whose input group can be non-zero for this tile
    __shared__ short s_list[MAX_SLABS];
    __shared__ unsigned char s_flag[MAX_SLABS];
    __shared__ int s_nact;
    int nact = a.nk;
    const bool sparse = a.bbox != nullptr;
    if (sparse) {
        const int hw = a.Ho * a.Wo;
        const int b = m0 / hw, rem0 = m0 - b * hw;                 // host guarantees hw % BM == 0: one image per tile
        const int oy0 = rem0 / a.Wo, oy1 = (rem0 + BM - 1) / a.Wo;
        const bool fullw = BM >= a.Wo;
        const int ox0 = fullw ? 0 : rem0 - oy0 * a.Wo, ox1 = fullw ? a.Wo - 1 : ox0 + BM - 1;
        for (int ks = tid; ks < a.nk; ks += 256) {
            const int cs = ks / ntaps, tap = ks - cs * ntaps, c0 = cs * BK;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            const int* bb = a.bbox + ((long long)b * a.bbox_groups + c0 / 64) * 4;
            const int y0 = oy0 * a.stride - a.pad + ky, y1 = oy1 * a.stride - a.pad + ky;
            const int x0 = ox0 * a.stride - a.pad + kx, x1 = ox1 * a.stride - a.pad + kx;
            s_flag[ks] = (y1 >= bb[0] && y0 <= bb[1] && x1 >= bb[2] && x0 <= bb[3]) ? 1 : 0;
        }
        __syncthreads();
        if (wave == 0) {
            int cnt = 0;
            for (int base = 0; base < a.nk; base += 64) {
                const int ks = base + lane;
                const bool f = ks < a.nk && s_flag[ks];
                const unsigned long long mask = __ballot(f);
                if (f) s_list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (short)ks;
                cnt += __popcll(mask);
            }
            if (lane == 0) s_nact = cnt;
        }
        __syncthreads();
        nact = s_nact;
    }
    int kbeg = 0;
    if constexpr (SPLIT) {                   // this workgroup's contiguous share of the K-slabs
        const int z = blockIdx.y;
        kbeg = (int)((long long)a.nk * z / a.splits);
        nact = (int)((long long)a.nk * (z + 1) / a.splits) - kbeg;
    }
    auto slab = [&](int i) { return sparse ? (int)s_list[i] : kbeg + i; };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // slab i travels: global --(iteration i-2)--> register stage i&1 --(end of iteration i-1)--> LDS buffer i&1
    auto step = [&](auto P, int ks) {
        constexpr int p = decltype(P)::value;
        using Q = std::integral_constant<int, p ^ 1>;
        if constexpr (!DMA) {
            gload(P, slab(min(ks + 2, nact - 1)));     // ALWAYS issued (past the end: a re-read that is never stored): under a branch hipcc cannot
                                                       // count the loads in flight and drains them all (vmcnt(0)) before the register stage is used
        }
        const char* sa = smem + p * BUF_BYTES + (wm * MI * 32) * ROW + frag_off;
        const char* sb = smem + p * BUF_BYTES + A_BYTES + (wn * NJ * 32) * ROW + frag_off;
        uint4 af[MI][4], bfr[NJ][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) af[i][q] = *reinterpret_cast<const uint4*>(sa + i * 32 * ROW + qoff[q]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bfr[j][q] = *reinterpret_cast<const uint4*>(sb + j * 32 * ROW + qoff[q]);
        if constexpr (DMA) {
            // the other buffer was last read in the previous iteration (all waves have passed its closing barrier):
            // refill it now so the transfer overlaps this slab's MFMAs; __syncthreads() below drains it (vmcnt(0))
            if (ks + 1 < nact) dma(p ^ 1, slab(ks + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) mma_slab<TI>(af[i], bfr[j], acc[i][j]);
        if constexpr (!DMA) {
            if (ks + 1 < nact) lstore(Q{}, p ^ 1, slab(ks + 1));
        } else {
            __builtin_amdgcn_sched_barrier(0);      // keep the DMA drain + barrier BELOW the MFMAs it overlaps with
        }
        __syncthreads();
    };

    if constexpr (RING3) {
        // slab i lives in buffer i % 3.  Iteration ks: (1) wait until this wave's pieces of slab ks have landed -- the
        // only younger DMAs are slab ks+1's, NP pieces -- (2) barrier: every wave's pieces have landed AND every wave is
        // done reading buffer (ks-1) % 3, (3) refill that buffer with slab ks+2, (4) MFMAs on slab ks.
        constexpr int NP = ACH + BCH;
        dma_ring(0, nact > 0 ? slab(0) : 0, nact > 0);
        dma_ring(1, nact > 1 ? slab(1) : 0, nact > 1);
        int buf = 0;
        for (int ks = 0; ks < nact; ++ks) {
            wait_vmcnt<NP>();
            __syncthreads();
            {
                const int nb = buf == 0 ? 2 : buf - 1;          // (ks + 2) % 3
                const bool more = ks + 2 < nact;
                dma_ring(nb, more ? slab(ks + 2) : 0, more);
            }
            const char* sa = smem + buf * BUF_BYTES + (wm * MI * 32) * ROW + frag_off;
            const char* sb = smem + buf * BUF_BYTES + A_BYTES + (wn * NJ * 32) * ROW + frag_off;
            uint4 af[MI][4], bfr[NJ][4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) af[i][q] = *reinterpret_cast<const uint4*>(sa + i * 32 * ROW + qoff[q]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bfr[j][q] = *reinterpret_cast<const uint4*>(sb + j * 32 * ROW + qoff[q]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma_slab<TI>(af[i], bfr[j], acc[i][j]);
            buf = buf == 2 ? 0 : buf + 1;
        }
        wait_vmcnt<0>();                 // the trailing (out-of-range) refills must land before the LDS is reused
        __syncthreads();
    } else if constexpr (DMA) {
        if (nact > 0) dma(0, slab(0));
    } else {
        if (nact > 0) {
            gload(P0{}, slab(0));
            lstore(P0{}, 0, slab(0));
            gload(P1{}, slab(nact > 1 ? 1 : 0));
        }
    }
    if constexpr (!RING3) {
        __syncthreads();
        for (int ks = 0; ks < nact; ks += 2) {
            step(P0{}, ks);
            if (ks + 1 < nact) step(P1{}, ks + 1);
        }
    }

    if constexpr (SPLIT) {
        // Partial tile -> workspace ([element][thread]: every store instruction of a wave is one contiguous 256-byte row), then one
        // counter per tile decides who finishes: the last workgroup to arrive re-reads ALL partials in split order, so the sum does
        // not depend on the arrival order.  The partials cross XCDs (one L2 each): they are written and read as relaxed AGENT-scope
        // atomics (sc1 accesses that go through to the device-coherent level) and ordered by hand -- every wave drains its stores
        // (vmcnt(0)), the workgroup meets at a barrier, then one lane bumps the counter.  (Agent-scope FENCES are correct too but
        // write back / invalidate the whole L2 per workgroup: measured 12 us per extra split on the 128-tile layers.)
        constexpr int NE = MI * NJ * 16;
        float* part = a.ws_part + ((long long)blockIdx.y * nwg + bid) * (NE * 256);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __hip_atomic_store(part + ((i * NJ + j) * 16 + e) * 256 + tid, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) s_last = __hip_atomic_fetch_add(&a.ws_cnt[bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.splits - 1);
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int z = 0; z < a.splits; ++z) {
            const float* pz = a.ws_part + ((long long)z * nwg + bid) * (NE * 256);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        acc[i][j][e] += __hip_atomic_load(pz + ((i * NJ + j) * 16 + e) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) __hip_atomic_store(&a.ws_cnt[bid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    TO* __restrict__ y = (TO*)a.y;
    const TO* __restrict__ res = (const TO*)a.res;
    const bool relu = (a.flags & 1) != 0;
    if (a.flags & 4) {
        // coalesced path: scale/shift in registers -> fp32 tile in LDS -> 16-byte row segments (+ residual, ReLU) to HBM
        float* st = reinterpret_cast<float*>(smem);
        if constexpr (std::is_same<TO, float>::value && !SPLIT) {
            if (a.st_p1) tile_col_stats<MI, NJ, 2, 2>(a, acc, st, m0, n0, wm, wn, lane);        // (round 5: the following BatchNorm's chunk partials)
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cl = wn * NJ * 32 + j * 32 + (lane & 31);
            const int n = n0 + cl;
            const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
            const float sh = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    st[rl * BN + cl] = fmaf(acc[i][j][r], sc, sh);
                }
        }
        __syncthreads();
        constexpr int VN = OutVec<TO>::N, CPR = BN / VN;          // chunks per tile row
        constexpr int EPI_UNROLL = sizeof(TO) == 2 ? RCH : 1;      // rpre[] needs static indices (bf16 outputs only)
        BnBwdAcc bs;
        const bool bs_on = std::is_same<TO, float>::value && !SPLIT && a.bs_p1 != nullptr;
        if constexpr (std::is_same<TO, float>::value) {
            if (bs_on) bn_bwd_acc_init(a, n0 + (tid % CPR) * VN, bs);
        }
#pragma unroll EPI_UNROLL
        for (int k = 0; k < RCH; ++k) {
            const int c = tid + 256 * k;
            const int rl = c / CPR, cc = (c - rl * CPR) * VN;
            const int m = m0 + rl, n = n0 + cc;
            if (m >= a.M || n >= a.Cout) continue;
            float v[VN];
            const float4* sp = reinterpret_cast<const float4*>(st + rl * BN + cc);
#pragma unroll
            for (int q = 0; q < VN / 4; ++q) { const float4 t = sp[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
            if (res) {
                float rv[VN];
                if (respre) OutVec<TO>::unpack(rpre[sizeof(TO) == 2 ? k : 0], rv);
                else OutVec<TO>::load(res + (long long)m * a.res_cs + a.res_co + n, rv);
#pragma unroll
                for (int e = 0; e < VN; ++e) v[e] += rv[e];
            }
            if constexpr (std::is_same<TO, float>::value) {
                if (a.mask) {                  // (round 5: the producing block's ReLU backward, see ConvArgs::mask)
                    const float4 mk = *reinterpret_cast<const float4*>(a.mask + (long long)m * a.out_cs + a.out_co + n);
                    v[0] = mk.x > 0.f ? v[0] : 0.f; v[1] = mk.y > 0.f ? v[1] : 0.f; v[2] = mk.z > 0.f ? v[2] : 0.f; v[3] = mk.w > 0.f ? v[3] : 0.f;
                }
                if (bs_on) bn_bwd_acc_add(a, m, n, v, bs);
                if (a.out_split_scale > 0.f) { store_split4(y, m, n, a.Cout, v, relu, a.out_split_scale, a.out_split_hi_only != 0); continue; }
            }
            OutVec<TO>::store_act(y + (long long)m * a.out_cs + a.out_co + n, v, relu);
        }
        if constexpr (std::is_same<TO, float>::value) {
            if (bs_on) bn_bwd_acc_finish<256, BN>(a, st, tid, tm, n0, bs);         // (round 5: the BatchNorm backward's chunk partials, ConvArgs::bs_*)
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * NJ * 32 + j * 32 + (lane & 31);
        if (n >= a.Cout) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                float v = fmaf(acc[i][j][r], sc, sh);
                if (res) v += load_res<TO>(res + (long long)m * a.res_cs + a.res_co + n);
                if (relu) v = fmaxf(v, 0.f);
                store_out<TO>(y + (long long)m * a.out_cs + a.out_co + n, v);
            }
        }
    }
}

template <typename TI, typename TO>
void launch_conv(const ConvArgs& a0, int num_cu, hipStream_t s) {
    if (a0.splits > 1) {                                   // dir_conv2d_splitk_forward: 128x128 tiles, 2-buffer loop
        if constexpr (is_half<TI>::value && std::is_same<TO, TI>::value) {
            ConvArgs a = a0;
            a.tiles_m = (a.M + 127) / 128;
            a.tiles_n = (a.Cout + 127) / 128;
            dim3 grid(a.tiles_m * a.tiles_n, a.splits), block(256);
            if (a.pre_scale) DIR_LAUNCH((conv_igemm_kernel<TI, TO, 2, 2, true, false, true>), grid, block, 0, s, a);
            else DIR_LAUNCH((conv_igemm_kernel<TI, TO, 2, 2, false, false, true>), grid, block, 0, s, a);
        }
        return;
    }
    constexpr int XM = std::is_same<TI, f16x3p_t>::value ? 3 : std::is_same<TI, f16x1p_t>::value ? 1 : 0;
    if constexpr (is_half<TI>::value || XM != 0) {
        // MFMA-bound layers (long reduction, enough tiles for 8-wave workgroups): deep-pipelined kernel of conv_pipe.hip (bf16, or both operands
        // pre-split f16)
        if constexpr (XM == 0) {
            if (a0.variant == 11 && launch_conv_big(a0, std::is_same<TO, float>::value, s, std::is_same<TI, f16s_t>::value)) return;      // DIR_CONV_VARIANT 11: 256 x 256 block tile
        }
        const bool four_wave = ((a0.variant & 15) >= 1 && (a0.variant & 15) <= 4) || a0.x2;   // explicit DIR_CONV_VARIANT 1..4 (+16), or a second source
        if (!four_wave) {
            ConvArgs ap = a0;
            ap.stamps = dir::stamps_begin("conv_pipe");
            const bool taken = launch_conv_pipe(ap, std::is_same<TO, float>::value, num_cu, s, XM, std::is_same<TI, f16s_t>::value);
            if (ap.stamps) {
                fprintf(stderr, "conv M=%d N=%d K=%d %s: ", ap.M, ap.Cout, ap.K, taken ? "pipe" : "(not taken)");
                dir::stamps_end("conv_pipe", ap.stamps, s);
            }
            if (taken) return;
        }
    }
    ConvArgs a = a0;
    // tile shape: 64-wide N tile for Cout <= 64 (no half-empty MFMA tiles); 64-tall M tile when the 128-tall grid
    // would not even put two workgroups on every CU (8x8 / 16x16 feature maps)
    bool n64 = a.Cout <= 64;
    // if even the 64x128 grid leaves at most one workgroup per CU, halve the N tile too (64x64: 4 WGs/CU by LDS)
    static const int force_n64 = getenv("DIR_FORCE_N64") ? atoi(getenv("DIR_FORCE_N64")) : -1;   // tuning aid
    if (!n64 && force_n64 != 0 && ((long long)((a.M + 63) / 64) * ((a.Cout + 127) / 128) <= (long long)num_cu || force_n64 == 1)) n64 = true;
    const int bn = n64 ? 64 : 128;
    int tiles_n = (a.Cout + bn - 1) / bn;
    // ... or when the reduction is so short (<= 4 slabs) that the layer is HBM-bound: smaller tiles = more workgroups
    // per CU = more bytes in flight
    const bool m64_auto = (long long)((a.M + 127) / 128) * tiles_n <= (long long)num_cu || a.nk <= 4;
    static const int force_m64 = getenv("DIR_FORCE_M64") ? atoi(getenv("DIR_FORCE_M64")) : -1;   // tuning aid
    bool m64 = force_m64 >= 0 ? (force_m64 != 0) : m64_auto;
    int ring_sel = -1;
    const int var = a.variant & 15;
    if (var >= 1 && var <= 4) {                            // explicit tile (DIR_CONV_VARIANT): 128x128 | 128x64 | 64x128 | 64x64
        m64 = var >= 3;
        n64 = (var == 2 || var == 4);
        ring_sel = (a.variant & 16) ? 1 : 0;
    }
    const int bn2 = n64 ? 64 : 128;
    tiles_n = (a.Cout + bn2 - 1) / bn2;
    const int bm = m64 ? 64 : 128;
    a.tiles_m = (a.M + bm - 1) / bm;
    a.tiles_n = tiles_n;
    if ((a.st_p1 || a.bs_p1) && std::is_same<TO, float>::value && (a.flags & 4)) stats_rows_launched = bm;      // this kernel's epilogue forms the statistics (tile_col_stats / bn_bwd_acc_*)
    choose_tile_order(a, is_half<TI>::value ? 2 : 4);
    dim3 grid(a.tiles_m * a.tiles_n), block(256);
    const bool pre = a.pre_scale != nullptr;
    // 3-buffer ring (two slabs in flight) pays once the reduction is long enough to amortise its two-slab prologue
    static const int ring_min = getenv("DIR_RING_MIN_NK") ? atoi(getenv("DIR_RING_MIN_NK")) : 12;   // tuning aid
    const bool ring = ring_sel >= 0 ? (ring_sel == 1 && a.nk >= 3) : a.nk >= ring_min;
    // The 64x128 tile on the 3-buffer ring is NOT part of the product library any more (round 4).  It was the launch beside which other kernels'
    // `v_pk_fma_f32 ... op_sel:[0,1,0]` (packed FMA whose LOW result takes src1's HIGH dword) computed wrong low halves -- an interaction now
    // reproduced with two synthetic kernels and no library code (tools/pkfp32_repro.hip: victim 9 beside aggressor -5 / -8 / -9, 100 of 100
    // launches; DESIGN.md 10 "erratum").  The library itself is built without packed-FP32 instructions; foreign kernels on other streams
    // (torch elementwise ops, RCCL) are not, and of all the library's tiles this one triggered it in 100 of 100 launches (the same tile
    // without the ring: 9 of 100; every other tile: 0).  Investigation builds (-DDIR_INVESTIGATE_RING_64x128=1, dir_amd/build.py with
    // DIR_PACKED_FP32=1) still carry it, behind DIR_RING_64x128=1.
#ifdef DIR_INVESTIGATE_RING_64x128
    static const int ring_64x128 = getenv("DIR_RING_64x128") ? atoi(getenv("DIR_RING_64x128")) : 0;
    constexpr bool kHasRing64x128 = true;
#else
    const int ring_64x128 = 0;
    constexpr bool kHasRing64x128 = false;
#endif
#define DIR_IGEMM_LAUNCH(MI_, NJ_)                                                                             \
    do {                                                                                                       \
        constexpr bool ring_built = !(MI_ == 2 && NJ_ == 2) && (kHasRing64x128 || !(MI_ == 1 && NJ_ == 2));    \
        if (pre) DIR_LAUNCH((conv_igemm_kernel<TI, TO, MI_, NJ_, true, false>), grid, block, 0, s, a);         \
        else if (ring && ring_built && !(MI_ == 1 && NJ_ == 2 && !ring_64x128)) {                              \
            if constexpr (ring_built) DIR_LAUNCH((conv_igemm_kernel<TI, TO, MI_, NJ_, false, true>), grid, block, 0, s, a); \
        } else DIR_LAUNCH((conv_igemm_kernel<TI, TO, MI_, NJ_, false, false>), grid, block, 0, s, a);          \
    } while (0)
    if (!m64 && !n64) DIR_IGEMM_LAUNCH(2, 2);
    else if (!m64 && n64) DIR_IGEMM_LAUNCH(2, 1);
    else if (m64 && !n64) DIR_IGEMM_LAUNCH(1, 2);
    else DIR_IGEMM_LAUNCH(1, 1);
#undef DIR_IGEMM_LAUNCH
}

}  // namespace

static int conv_forward(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                        const float* pre_scale, const float* pre_shift, const void* residual, void* y,
                        const int32_t* bbox, void* stream, const dir_conv_src2* d2 = nullptr, const void* x2 = nullptr,
                        int splits = 0, void* workspace = nullptr, long long workspace_bytes = 0, float* st_p1 = nullptr, float* st_p2 = nullptr,
                        const float* mask = nullptr, const dir_conv_bn_bwd* bs = nullptr) {
    DIR_REQUIRE(d && x && w && y, "dir_conv2d_forward: null pointer");
    DIR_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "dir_conv2d_forward: bad shape");
    DIR_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "dir_conv2d_forward: bad kernel geometry");
    const bool xp = d->in_dtype == DIR_DT_F16X3P || d->in_dtype == DIR_DT_F16X1P;      // activations pre-split by dir_split_f16_forward
    const bool x1 = d->in_dtype == DIR_DT_F16X1 || d->in_dtype == DIR_DT_F16X1P;
    const bool x3 = d->in_dtype == DIR_DT_F16X3 || d->in_dtype == DIR_DT_F16X3P || x1;   // fp32-sized tensors, f16 arithmetic (split precision / hi only)
    DIR_REQUIRE(!(xp && pre_scale), "dir_conv2d_forward: pre-split activations carry their pre-activation already (dir_split_f16_forward)");
    const bool f32 = d->in_dtype == DIR_DT_F32 || x3;
    const bool h16 = d->in_dtype == DIR_DT_F16;                 // f16 STORAGE (round 5): the bf16 data path on the f16 matrix cores
    DIR_REQUIRE(f32 || d->in_dtype == DIR_DT_BF16 || h16, "dir_conv2d_forward: in_dtype must be f32, bf16, f16 or f16x3");
    DIR_REQUIRE(d->out_dtype == DIR_DT_F32 || d->out_dtype == DIR_DT_BF16 || d->out_dtype == DIR_DT_F16, "dir_conv2d_forward: bad out_dtype");
    DIR_REQUIRE(!(f32 && d->out_dtype != DIR_DT_F32), "dir_conv2d_forward: f32 in / 16-bit out not built");
    DIR_REQUIRE(d->out_dtype == DIR_DT_F32 || d->out_dtype == d->in_dtype, "dir_conv2d_forward: a 16-bit output must have the input's 16-bit type");
    const int BK = f32 ? 32 : 64, EPC = f32 ? 4 : 8;
    DIR_REQUIRE(d->Cin % BK == 0, "dir_conv2d_forward: Cin=%d must be a multiple of %d", d->Cin, BK);
    const int in_cs = d->in_cstride ? d->in_cstride : d->Cin;
    const int out_cs = d->out_cstride ? d->out_cstride : d->Cout;
    const int res_cs = d->res_cstride ? d->res_cstride : d->Cout;
    // every 16-byte A chunk must be aligned: either whole pixels are (in_cs % EPC == 0), or -- the pre-padded NHWC4
    // stem image -- rows and the horizontal stride are (kw == 1, pad == 0)
    DIR_REQUIRE(d->in_coff % EPC == 0 && (in_cs % EPC == 0 || (d->kw == 1 && d->pad == 0 && (in_cs * d->stride) % EPC == 0 &&
                                                             (in_cs * d->W) % EPC == 0)),
                "dir_conv2d_forward: input channel slice must be 16-byte aligned");
    DIR_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "dir_conv2d_forward: pre_scale/pre_shift go together");
    ConvArgs a;
    a.stamps = nullptr;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.pre_scale = pre_scale; a.pre_shift = pre_shift;
    a.res = residual; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cs = in_cs; a.in_co = d->in_coff;
    a.Cout = d->Cout; a.out_cs = out_cs; a.out_co = d->out_coff; a.res_cs = res_cs; a.res_co = d->res_coff;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    a.Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DIR_REQUIRE(a.Ho > 0 && a.Wo > 0, "dir_conv2d_forward: empty output");
    const long long M = (long long)d->B * a.Ho * a.Wo;
    DIR_REQUIRE(M < (1ll << 31), "dir_conv2d_forward: too many output pixels");
    a.M = (int)M; a.K = d->kh * d->kw * d->Cin; a.nk = a.K / BK;
    a.x2 = nullptr; a.x2_bytes = 0; a.H2 = a.W2 = a.in_cs2 = a.in_co2 = a.stride2 = 0; a.nk1 = a.nk;
    if (d2) {        // second source: 1x1 (strided) convolution over x2 accumulated into the same output tile
        DIR_REQUIRE(x2 && d2->H > 0 && d2->W > 0 && d2->Cin > 0 && d2->stride > 0, "dir_conv2d_dual_forward: bad second source");
        DIR_REQUIRE(d2->Cin % BK == 0, "dir_conv2d_dual_forward: Cin2=%d must be a multiple of %d", d2->Cin, BK);
        DIR_REQUIRE((d2->H - 1) / d2->stride + 1 == a.Ho && (d2->W - 1) / d2->stride + 1 == a.Wo,
                    "dir_conv2d_dual_forward: the second source does not produce a %dx%d output", a.Ho, a.Wo);
        const int cs2 = d2->in_cstride ? d2->in_cstride : d2->Cin;
        DIR_REQUIRE(cs2 % EPC == 0 && d2->in_coff % EPC == 0, "dir_conv2d_dual_forward: second source channel slice must be 16-byte aligned");
        DIR_REQUIRE(pre_scale == nullptr && bbox == nullptr,
                    "dir_conv2d_dual_forward: pre-activation / sparse-K do not combine with a second source");
        const long long x2b = (long long)d->B * d2->H * d2->W * cs2 * (f32 ? 4 : 2);
        DIR_REQUIRE(x2b < (1ll << 31), "dir_conv2d_dual_forward: second source must be < 2 GiB");
        a.x2 = x2; a.x2_bytes = (unsigned)x2b; a.H2 = d2->H; a.W2 = d2->W; a.in_cs2 = cs2; a.in_co2 = d2->in_coff;
        a.stride2 = d2->stride;
        a.K += d2->Cin; a.nk += d2->Cin / BK;
    }
    a.tiles_m = a.tiles_n = 0;
    set_magic(a);
    a.flags = d->flags & 3;
    a.variant = (d->flags >> 8) & 0xff;
    a.a_scale = (x3 && !xp && d->in_scale > 0.f) ? d->in_scale : 1.f;
    a.out_split_scale = 0.f; a.out_split_hi_only = 0;
    if (d->out_split_scale != 0.f) {        // > 0: hi | lo; < 0: hi only, scale = |value|
        DIR_REQUIRE(d->out_dtype == DIR_DT_F32 && d->Cout % 32 == 0 && out_cs == d->Cout && d->out_coff == 0,
                    "dir_conv2d_forward: out_split_scale needs an fp32 output that is a whole tensor with Cout %% 32 == 0");
        a.out_split_scale = fabsf(d->out_split_scale); a.out_split_hi_only = d->out_split_scale < 0.f;
    }
    DIR_REQUIRE(d->kh * d->kw <= 32, "dir_conv2d_forward: at most 32 taps");
    const long long xb = (long long)d->B * d->H * d->W * in_cs * (f32 ? 4 : 2);
    const long long wb = (long long)d->Cout * a.K * (f32 ? 4 : 2);
    DIR_REQUIRE(xb < (1ll << 31) && wb < (1ll << 31), "dir_conv2d_forward: tensors must be < 2 GiB (32-bit buffer offsets)");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    // sparse-K needs whole tiles inside one image and a slab to sit inside one 64-channel group
    a.bbox = nullptr; a.bbox_groups = d->Cin / 64;
    if (bbox && d->Cin % 64 == 0 && (a.Ho * a.Wo) % 128 == 0 && a.nk <= MAX_SLABS && pre_scale == nullptr) a.bbox = bbox;
    // coalesced (LDS-staged, 16-byte) epilogue whenever every output / residual row segment is 16-byte aligned
    const int epo = d->out_dtype == DIR_DT_F32 ? 4 : 8;
    const bool vec = d->Cout % epo == 0 && out_cs % epo == 0 && d->out_coff % epo == 0 &&
                     (residual == nullptr || (res_cs % epo == 0 && d->res_coff % epo == 0));
    if (vec) a.flags |= 4;
    DIR_REQUIRE(a.out_split_scale == 0.f || vec, "dir_conv2d_forward: out_split_scale needs 16-byte aligned output / residual rows");
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    a.splits = 0; a.ws_part = nullptr; a.ws_cnt = nullptr;
    a.st_p1 = st_p1; a.st_p2 = st_p2;
    stats_rows_launched = 0;
    a.mask = mask;
    DIR_REQUIRE(!mask || (vec && d->out_dtype == DIR_DT_F32 && ((uintptr_t)mask & 15) == 0),
                "dir_conv2d_forward_masked: an fp32 output with 16-byte aligned rows (and mask) only");
    if (bs) {
        DIR_REQUIRE(bs->z && bs->mean && bs->rstd && bs->p1 && bs->p2, "dir_conv2d_forward_ex: incomplete dir_conv_bn_bwd");
        DIR_REQUIRE(vec && d->out_dtype == DIR_DT_F32 && out_cs == d->Cout && d->out_coff == 0 && d->Cout % 4 == 0 && a.out_split_scale == 0.f && splits <= 1 &&
                        (((uintptr_t)bs->z | (uintptr_t)bs->mean | (uintptr_t)bs->rstd | (uintptr_t)bs->w | (uintptr_t)bs->b) & 15) == 0,
                    "dir_conv2d_forward_ex: the BatchNorm-backward sums need a whole fp32 output tensor, Cout %% 4 == 0, 16-byte aligned vectors");
        a.bs_z = bs->z; a.bs_mu = bs->mean; a.bs_rs = bs->rstd; a.bs_w = bs->w; a.bs_b = bs->b; a.bs_relu = bs->relu; a.bs_p1 = bs->p1; a.bs_p2 = bs->p2;
    }
    if (splits > 1) {
        DIR_REQUIRE(!f32 && d->out_dtype != DIR_DT_F32 && vec && bbox == nullptr, "dir_conv2d_splitk_forward: 16-bit -> 16-bit layers with 16-byte aligned output rows only");
        DIR_REQUIRE(splits <= a.nk && splits <= 16, "dir_conv2d_splitk_forward: splits must be <= min(16, K / 64)");
        const long long tiles = (long long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
        const long long need = SPLITK_COUNTER_BYTES + (long long)splits * tiles * 128 * 128 * 4;
        DIR_REQUIRE(workspace && workspace_bytes >= need && tiles * 4 <= SPLITK_COUNTER_BYTES && ((uintptr_t)workspace & 15) == 0,
                    "dir_conv2d_splitk_forward: workspace of %lld bytes needed (dir_conv2d_splitk_workspace_bytes), 16-byte aligned", need);
        a.splits = splits; a.ws_cnt = (unsigned*)workspace; a.ws_part = (float*)((char*)workspace + SPLITK_COUNTER_BYTES);
    }
    hipStream_t s = (hipStream_t)stream;
    if (xp && x1) launch_conv<f16x1p_t, float>(a, num_cu, s);
    else if (xp) launch_conv<f16x3p_t, float>(a, num_cu, s);
    else if (x1) launch_conv<f16x1_t, float>(a, num_cu, s);
    else if (x3) launch_conv<f16x3_t, float>(a, num_cu, s);
    else if (f32) launch_conv<float, float>(a, num_cu, s);
    else if (h16 && d->out_dtype == DIR_DT_F16) launch_conv<f16s_t, f16s_t>(a, num_cu, s);
    else if (h16) launch_conv<f16s_t, float>(a, num_cu, s);
    else if (d->out_dtype == DIR_DT_BF16) launch_conv<bf16_t, bf16_t>(a, num_cu, s);
    else launch_conv<bf16_t, float>(a, num_cu, s);
    return dir::check_launch("dir_conv2d_forward");
}


// ---- dir_split_f16_forward: fp32 NHWC channel slice -> [pixel][C/32][hi 32 | lo 32] f16, times in_scale, optional pre-activation
namespace {
struct SplitArgs {
    const float* x; uint4* y; const float* ps; const float* pb;
    long long nchunk;          // pixels * C / 4
    int C4, cs, co, relu, hi_only;
    float s;
};
__global__ __launch_bounds__(256) void split_f16_kernel(SplitArgs a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.nchunk; i += (long long)gridDim.x * 256) {
        const long long pix = i / a.C4;
        const int c4 = (int)(i - pix * a.C4);                          // 4-channel chunk inside the pixel
        uint4 v = *reinterpret_cast<const uint4*>(a.x + pix * a.cs + a.co + 4 * c4);
        if (a.ps) v = prologue<float>(v, a.ps, a.pb, 4 * c4, a.relu != 0);
        const uint4 sp = a.hi_only ? split_f16x1(v, a.s) : split_f16x3(v, a.s);
        // slab = 8 chunks (32 channels) = 128 bytes: hi parts at bytes [0, 64), lo parts at [64, 128); this chunk's 8 bytes at 8 * (c4 & 7)
        char* row = reinterpret_cast<char*>(a.y) + (pix * a.C4 + (c4 & ~7)) * 16;
        *reinterpret_cast<uint2*>(row + 8 * (c4 & 7)) = make_uint2(sp.x, sp.y);
        *reinterpret_cast<uint2*>(row + 64 + 8 * (c4 & 7)) = make_uint2(sp.z, sp.w);
    }
}
}  // namespace

// ---- dir_pack_f16x3_weights: the host packing of dir_amd/functional.py::pack_f16x3_weights as one launch (weights that change every
//      optimiser step): one workgroup per output channel: max |w|, the power of two p with max |w| p in [2^12, 2^13), hi | lo slabs, 1 / p
namespace {
__global__ __launch_bounds__(256) void pack_f16x3_kernel(const float* __restrict__ w, uint4* __restrict__ out, float* __restrict__ scale_out,
                                                         const float* __restrict__ scale_in, int K) {
    __shared__ float s_max[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* row = w + (long long)n * K;
    float m = 0.f;
    for (int k = tid; k < K; k += 256) m = fmaxf(m, fabsf(row[k]));
    m = dir::wave_max(m);
    if ((tid & 63) == 0) s_max[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    // m = f * 2^(E - 127), f in [1, 2)  ->  p = 2^(12 - (E - 127)): exponent field 266 - E (clamped; an all-zero row takes p = 1)
    const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
    const float p = (m > 0.f && E >= 13u && E <= 253u) ? __uint_as_float((266u - E) << 23) : 1.f;
    if (tid == 0) scale_out[n] = (scale_in ? scale_in[n] : 1.f) / p;
    for (int c4 = tid; c4 < K / 4; c4 += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + 4 * c4);
        const uint4 sp = split_f16x3(v, p);
        char* slab = reinterpret_cast<char*>(out) + ((long long)n * K + 4 * (c4 & ~7)) * 4;
        *reinterpret_cast<uint2*>(slab + 8 * (c4 & 7)) = make_uint2(sp.x, sp.y);
        *reinterpret_cast<uint2*>(slab + 64 + 8 * (c4 & 7)) = make_uint2(sp.z, sp.w);
    }
}
}  // namespace

extern "C" int dir_pack_f16x3_weights(const float* w, void* packed, float* scale_out, const float* scale_in, int N, int K, void* stream) {
    DIR_REQUIRE(w && packed && scale_out && N > 0 && K > 0 && K % 32 == 0, "dir_pack_f16x3_weights: bad arguments (K must be a multiple of 32)");
    DIR_LAUNCH(pack_f16x3_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, w, (uint4*)packed, scale_out, scale_in, K);
    return dir::check_launch("dir_pack_f16x3_weights");
}

// ---- dir_train_pack_conv_weights: EVERY convolution weight of a training step, in both operand forms, from the OIHW parameters in ONE launch.
//      A step used to spend ~600 launches on this (per convolution: OIHW -> OHWI copy, pack, scale division for the forward; flip, transpose
//      copy, channel padding, pack, division for the data gradient).  The tensor is [O][C][T] (T = kh * kw taps, contiguous):
//        forward row o:        k = t * Cin + c      <-  w[o][c][t]            one workgroup per row; its source is ONE contiguous chunk
//        data-gradient row c:  k = t * Cout32 + o   <-  w[o][c][T - 1 - t]    one workgroup per FOUR rows c .. c + 3 (16 T contiguous bytes per o)
//      A thread always reads 16-byte pieces of which every byte is used by its own workgroup (a first version gathered single floats at tap
//      stride and re-fetched every line once per tap: 2.3 ms for the network's 77 M weights), keeps 4 (inner) x T or 4 x 4 x T values in
//      registers and emits 8-byte hi / lo stores into the [32 hi | 32 lo] slabs.  Same p, split and layout as pack_f16x3_kernel: bit-identical.
namespace {
__device__ __forceinline__ float pack_pow2(float m) {
    const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
    return (m > 0.f && E >= 13u && E <= 253u) ? __uint_as_float((266u - E) << 23) : 1.f;
}
__device__ __forceinline__ void pack_store(char* row_out, int c4, float v0, float v1, float v2, float v3, float p) {
    const uint4 sp = split_f16x3(make_uint4(__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)), p);
    char* slab = row_out + (long long)(c4 & ~7) * 16;
    *reinterpret_cast<uint2*>(slab + 8 * (c4 & 7)) = make_uint2(sp.x, sp.y);
    *reinterpret_cast<uint2*>(slab + 64 + 8 * (c4 & 7)) = make_uint2(sp.z, sp.w);
}
template <int T>
__device__ __forceinline__ void train_pack_body(const dir_train_weight& e, int n, bool fwd, float (&s_max)[4][4]) {
    const int tid = threadIdx.x;
    if (fwd) {
        const int Cin = e.Cin, K4 = T * Cin / 4;
        const float4* src = reinterpret_cast<const float4*>(e.w + (long long)n * Cin * T);
        float m = 0.f;
        for (int i = tid; i < K4; i += 256) { const float4 v = src[i]; m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
        m = dir::wave_max(m);
        if ((tid & 63) == 0) s_max[0][tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(s_max[0][0], s_max[0][1]), fmaxf(s_max[0][2], s_max[0][3]));
        const float p = pack_pow2(m);
        if (tid == 0) e.fwd_scale[n] = (1.f / p) * e.fwd_inv_in;                       // powers of two: exact
        char* out = reinterpret_cast<char*>(e.fwd) + (long long)n * K4 * 16;
        for (int j4 = tid; j4 < Cin / 4; j4 += 256) {                                 // channels 4 j4 .. + 3, all taps: 4 T contiguous floats
            float v[4 * T];
#pragma unroll
            for (int q = 0; q < T; ++q) {
                const float4 f = src[j4 * T + q];
                v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) pack_store(out, t * (Cin / 4) + j4, v[t], v[T + t], v[2 * T + t], v[3 * T + t], p);
        }
        return;
    }
    // data gradient: rows c0 .. c0 + 3
    const int c0 = 4 * n, Cin = e.Cin, Cout = e.Cout, Co32 = (Cout + 31) & ~31, K4 = T * Co32 / 4;
    const float* src = e.w + (long long)c0 * T;
    const long long ostride = (long long)Cin * T;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < Cout * T; i += 256) {                                        // (o, float4 q of the 4 T floats): float 4 q + x is (c0 + (4 q + x) / T, tap)
        const int o = i / T, q = i - o * T;
        const float4 f = *reinterpret_cast<const float4*>(src + o * ostride + 4 * q);
        const float a[4] = {fabsf(f.x), fabsf(f.y), fabsf(f.z), fabsf(f.w)};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int cl = (4 * q + x) / T;
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], cl == r ? a[x] : 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m[r] = dir::wave_max(m[r]);
        if ((tid & 63) == 0) s_max[r][tid >> 6] = m[r];
    }
    __syncthreads();
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p[r] = pack_pow2(fmaxf(fmaxf(s_max[r][0], s_max[r][1]), fmaxf(s_max[r][2], s_max[r][3])));
        if (tid == r) e.dgrad_scale[c0 + r] = (1.f / p[r]) * e.dgrad_inv_in;
    }
    char* out = reinterpret_cast<char*>(e.dgrad) + (long long)c0 * K4 * 16;
    for (int o4 = tid; o4 < Co32 / 4; o4 += 256) {
        float v[4][4 * T];                                                             // [o of the quad][c local * T + tap]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 4 * o4 + i;
#pragma unroll
            for (int q = 0; q < T; ++q) {
                const float4 f = o < Cout ? *reinterpret_cast<const float4*>(src + o * ostride + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[i][4 * q] = f.x; v[i][4 * q + 1] = f.y; v[i][4 * q + 2] = f.z; v[i][4 * q + 3] = f.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int u = r * T + (T - 1 - t);                                     // flipped tap
                pack_store(out + (long long)r * K4 * 16, t * (Co32 / 4) + o4, v[0][u], v[1][u], v[2][u], v[3][u], p[r]);
            }
    }
}
__global__ __launch_bounds__(256) void train_pack_kernel(const dir_train_weight* __restrict__ table, const int* __restrict__ wg_start, int entries) {
    __shared__ float s_max[4][4];
    const int r = blockIdx.x;
    int lo = 0, hi = entries;                                           // last entry with wg_start[e] <= r
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wg_start[mid] <= r) lo = mid; else hi = mid; }
    const dir_train_weight e = table[lo];
    int n = r - wg_start[lo];
    const bool fwd = e.fwd != nullptr && n < e.Cout;
    if (!fwd && e.fwd != nullptr) n -= e.Cout;
    if (e.kh * e.kw == 1) train_pack_body<1>(e, n, fwd, s_max);
    else train_pack_body<9>(e, n, fwd, s_max);
}
}  // namespace

extern "C" int dir_train_pack_conv_weights(const dir_train_weight* table, const int* wg_start, int entries, int total_workgroups, void* stream) {
    DIR_REQUIRE(table && wg_start && entries > 0 && total_workgroups > 0, "dir_train_pack_conv_weights: bad arguments");
    DIR_LAUNCH(train_pack_kernel, dim3(total_workgroups), dim3(256), 0, (hipStream_t)stream, table, wg_start, entries);
    return dir::check_launch("dir_train_pack_conv_weights");
}

extern "C" int dir_split_f16_forward(const float* x, void* y, long long pixels, int C, int in_cstride, int in_coff, const float* pre_scale,
                                     const float* pre_shift, int pre_relu, float in_scale, int hi_only, void* stream) {
    DIR_REQUIRE(x && y && pixels >= 0 && C > 0 && C % 32 == 0, "dir_split_f16_forward: bad arguments (C must be a multiple of 32)");
    const int cs = in_cstride ? in_cstride : C;
    DIR_REQUIRE(cs % 4 == 0 && in_coff % 4 == 0 && in_coff + C <= cs, "dir_split_f16_forward: channel slice must be 16-byte aligned");
    DIR_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "dir_split_f16_forward: pre_scale / pre_shift go together");
    if (pixels == 0) return DIR_OK;
    SplitArgs a{x, (uint4*)y, pre_scale, pre_shift, pixels * (C / 4), C / 4, cs, in_coff, pre_relu, hi_only, in_scale > 0.f ? in_scale : 1.f};
    const long long blocks = (a.nchunk + 255) / 256;
    DIR_LAUNCH(split_f16_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, a);
    return dir::check_launch("dir_split_f16_forward");
}

extern "C" int dir_conv2d_forward(const dir_conv_desc* d, const void* x, const void* w, const float* scale,
                                  const float* shift, const float* pre_scale, const float* pre_shift,
                                  const void* residual, void* y, void* stream) {
    return conv_forward(d, x, w, scale, shift, pre_scale, pre_shift, residual, y, nullptr, stream);
}

// round 5: dir_conv2d_forward + the chunk partials of the training-mode BatchNorm that follows it (tile_col_stats in the epilogue).  *chunk_rows
// receives the rows per chunk (the M tile of the kernel that took the launch), or 0 when that kernel does not form them (the caller then runs
// dir_bn_train_stats on the stored map).  p1 / p2: room for ceil(M / 64) x Cout floats each.
extern "C" int dir_conv2d_forward_stats(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                                        const float* pre_shift, void* y, float* p1, float* p2, int* chunk_rows, void* stream) {
    DIR_REQUIRE(d && p1 && p2 && chunk_rows, "dir_conv2d_forward_stats: null pointer");
    DIR_REQUIRE(d->out_dtype == DIR_DT_F32 && (d->out_cstride == 0 || d->out_cstride == d->Cout) && d->out_coff == 0 && !(d->flags & 1) && d->out_split_scale == 0.f,
                "dir_conv2d_forward_stats: the statistics are those of a whole fp32 output tensor without activation");
    const int rc = conv_forward(d, x, w, scale, shift, pre_scale, pre_shift, nullptr, y, nullptr, stream, nullptr, nullptr, 0, nullptr, 0, p1, p2);
    *chunk_rows = rc == 0 ? stats_rows_launched : 0;
    return rc;
}

// round 5: y = mask > 0 ? conv(x) (+ residual) : 0, mask [.., out_cstride] fp32 laid out like y (see ConvArgs::mask)
extern "C" int dir_conv2d_forward_masked(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                                         const float* pre_shift, const void* residual, const float* mask, void* y, void* stream) {
    DIR_REQUIRE(mask, "dir_conv2d_forward_masked: null mask");
    return conv_forward(d, x, w, scale, shift, pre_scale, pre_shift, residual, y, nullptr, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, mask);
}

// round 5: everything the fp32 epilogue can do for the training step's data-gradient convolutions in one entry point: residual, output mask, and the
// chunk partials of the backward pass of the BatchNorm whose output's gradient this convolution writes (*chunk_rows as in dir_conv2d_forward_stats)
extern "C" int dir_conv2d_forward_ex(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                                     const float* pre_shift, const void* residual, const float* mask, void* y, const dir_conv_bn_bwd* bn, int* chunk_rows,
                                     void* stream) {
    DIR_REQUIRE(!bn || chunk_rows, "dir_conv2d_forward_ex: chunk_rows is needed with bn");
    const int rc = conv_forward(d, x, w, scale, shift, pre_scale, pre_shift, residual, y, nullptr, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, mask, bn);
    if (chunk_rows) *chunk_rows = (rc == 0 && bn) ? stats_rows_launched : 0;
    return rc;
}

extern "C" long long dir_conv2d_splitk_workspace_bytes(const dir_conv_desc* d, int splits) {
    if (!d || splits < 1) return -1;
    const int Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    const int Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    const long long M = (long long)d->B * Ho * Wo;
    const long long tiles = ((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (tiles * 4 > SPLITK_COUNTER_BYTES) return -1;
    return SPLITK_COUNTER_BYTES + (long long)splits * tiles * 128 * 128 * 4;
}

extern "C" int dir_conv2d_splitk_forward(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                                         const float* pre_scale, const float* pre_shift, const void* residual, void* y, int splits,
                                         void* workspace, long long workspace_bytes, void* stream) {
    DIR_REQUIRE(splits >= 1, "dir_conv2d_splitk_forward: splits must be >= 1");
    return conv_forward(d, x, w, scale, shift, pre_scale, pre_shift, residual, y, nullptr, stream, nullptr, nullptr, splits, workspace,
                        workspace_bytes);
}

extern "C" int dir_conv2d_dual_forward(const dir_conv_desc* d, const void* x, const dir_conv_src2* d2, const void* x2, const void* w,
                                       const float* shift, void* y, void* stream) {
    DIR_REQUIRE(d2, "dir_conv2d_dual_forward: null second-source descriptor");
    return conv_forward(d, x, w, nullptr, shift, nullptr, nullptr, nullptr, y, nullptr, stream, d2, x2);
}

extern "C" int dir_conv2d_dual_scaled_forward(const dir_conv_desc* d, const void* x, const dir_conv_src2* d2, const void* x2, const void* w,
                                              const float* scale, const float* shift, void* y, void* stream) {
    DIR_REQUIRE(d2, "dir_conv2d_dual_scaled_forward: null second-source descriptor");
    return conv_forward(d, x, w, scale, shift, nullptr, nullptr, nullptr, y, nullptr, stream, d2, x2);
}

extern "C" int dir_conv2d_sparse_forward(const dir_conv_desc* d, const void* x, const void* w, const float* scale,
                                         const float* shift, const void* residual, void* y, const int32_t* group_bbox,
                                         void* stream) {
    DIR_REQUIRE(group_bbox, "dir_conv2d_sparse_forward: null bbox");
    return conv_forward(d, x, w, scale, shift, nullptr, nullptr, residual, y, group_bbox, stream);
}
