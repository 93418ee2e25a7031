// Deep-pipelined implicit-GEMM convolution for the MFMA-bound layers (bf16 operands, fp32 accumulate): the same
// maths, operand layout and epilogue as conv.hip, restructured so that the matrix cores never wait inside a K-slab.
//
//   * 8 waves (512 threads) per workgroup, one workgroup per CU, block tile 256x128 | 128x128 | 256x64:
//     twice the rows of conv.hip's largest tile per byte fetched from L2 (the 128x128 2-barrier structure is capped
//     near 900 TFLOP/s on this chip by L2 -> LDS traffic and by its exposed DMA / ds_read latencies).
//   * K-slabs (64 bf16 channels = 128 B per row) travel global -> LDS by DMA into a 3-buffer ring, issued THREE
//     slabs ahead through untracked inline asm and retired with counted s_waitcnt vmcnt(N) -- never vmcnt(0).
//   * MFMA operands are double-buffered in registers: while the 16 MFMAs of slab k run, the same instruction stream
//     issues the ds_read_b128 of slab k+1's fragments (first half of the slots) and the DMA pieces of slab k+3
//     (second half), so one barrier per slab is the only synchronisation and nothing in a slab's MFMA chain
//     depends on a load of the same iteration.
//   * K order, XOR swizzle on the DMA source address, hardware-bounds-checked zero padding, sparse-K slab list and
//     LDS-staged coalesced epilogue are those of conv.hip (bit-identical results: same fp32 accumulation order).
//
// Replaces the same reference calls as conv.hip (models/backbone/resnet.py:120-140, models/backbone/hourglass.py:55-70,
// models/dir.py:57-62,227-241,404-420) for the layers launch_conv_pipe() accepts.
#include "conv_common.h"

#include <stdlib.h>

#ifndef DIR_P15_NBUF
#define DIR_P15_NBUF 3
#endif
#ifndef DIR_PATCH_PLAIN_KEY
#define DIR_PATCH_PLAIN_KEY 0
#endif

namespace dir {
namespace convk {
namespace {

// one K-slab of the reduction: tap (ky,kx) of channel slab c0
struct Slab {
    int tap, toff, k0;     // tap index, byte offset of (ky,kx,c0) inside the input, element offset inside a weight row
};

constexpr int PRE_MAX_CIN = 2304;      // pre-activation parameters staged in LDS (fusion_layer4: 2048 + 256 channels)

// XM: 0 = bf16 operands; 3 / 1 = both operands pre-split f16 hi | lo slabs (DIR_DT_F16X3P / F16X1P: 32 channels per 128-byte row, three / one
// v_mfma_f32_32x32x16_f16 per product and k16-step) -- the same ring, the same slot plan with 6 / 2 instead of 4 MFMAs per tile and slab
// TIN: the 16-bit storage kind of the operands when XM == 0 (bf16_t | f16s_t: same bytes and data path, the other matrix-core instruction)
template <typename TO, int MI, int NJ, int WM, int WN, bool SPARSE, bool PRE = false, int NBUF = 3, int XM = 0, typename TIN = bf16_t>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_pipe_kernel(ConvArgs a) {
    typedef TIN TI;
    half_kernel_init<TO>();
    static_assert(XM == 0 || (!SPARSE && !PRE && std::is_same<TO, float>::value), "pre-split operands: dense, no pre-activation, fp32 output");
    constexpr int NT = 64 * WM * WN;               // threads
    constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    constexpr int RPP = NT / 8;                    // rows covered by one DMA pass of the whole workgroup
    constexpr int ACH = BM / RPP, BCH = BN / RPP, NP = ACH + BCH;   // DMA pieces per thread per slab
    constexpr int ROW = 128;
    constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = NBUF * BUF_BYTES > STAGE_BYTES ? NBUF * BUF_BYTES : STAGE_BYTES;
    constexpr int EPC = XM ? 4 : 8, BK = XM ? 32 : 64, ES = XM ? 4 : 2;      // (pre-split rows are addressed like the fp32 tensor they replace)
    constexpr int NM = XM == 3 ? 6 : XM == 1 ? 2 : 4;                        // MFMAs per 32x32 tile and slab
    constexpr int NSLOT = MI * NJ * NM;            // MFMAs per slab per wave
    constexpr int NREAD = (MI + NJ) * 4;           // ds_read_b128 per slab per wave
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the DMA pass");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    // PRE: hourglass.Residual's pre-activation BatchNorm + ReLU (models/backbone/hourglass.py:55-70) on the A operand.  The
    // raw slab still arrives by LDS-DMA; the thread that issued a piece rewrites its own 16-byte chunks in LDS once they have
    // landed (its counted vmcnt) and before the slab's barrier, so the deep ring is kept.  Parameters live in LDS: a
    // compiler-tracked global load inside the loop would drain the untracked DMA ring (vmcnt is shared and in order).
    __shared__ float s_pre[PRE ? 2 * PRE_MAX_CIN : 1];
    if constexpr (PRE) {
        for (int i = threadIdx.x; i < a.Cin; i += NT) {
            s_pre[i] = a.pre_scale[i];
            s_pre[PRE_MAX_CIN + i] = a.pre_shift[i];
        }
        __syncthreads();
    }

    // XCD-aware tile order (see conv.hip)
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    tile_of(a, bid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    int nstamp = blockIdx.x == 0 ? 0 : 4;
    auto stamp = [&]() { if (a.stamps && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && tid == 0) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    const TI* __restrict__ x = (const TI*)a.x;
    const TI* __restrict__ w = (const TI*)a.w;
    const i32x4 xd = {(int)(unsigned)(unsigned long long)x, (int)(unsigned)((unsigned long long)x >> 32), (int)a.x_bytes, 0x00020000};
    const i32x4 wd = {(int)(unsigned)(unsigned long long)w, (int)(unsigned)((unsigned long long)w >> 32), (int)a.w_bytes, 0x00020000};
    constexpr unsigned OOB = 0x80000000u;

    // ---- per-thread DMA source state: row (tid >> 3) + RPP * i of the A / B tile, 16-byte chunk (tid & 7) ^ swizzle
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
            pixel_setup(a, m, col * EPC * ES, ES, avoff[i], amask[i], b, oy, ox);
        }
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = n0 + (tid >> 3) + RPP * i;
        bvoff[i] = n < a.Cout ? (unsigned)((n * a.K + col * EPC) * ES) : OOB;
    }
    const int ntaps = a.kh * a.kw;

    // ---- optional sparse-K: ordered list of the slabs whose 64-channel input group can be non-zero for this tile
    __shared__ int s_list[SPARSE ? MAX_SLABS : 1];
    __shared__ unsigned char s_flag[SPARSE ? MAX_SLABS : 1];
    __shared__ int s_nact;
    int nact = a.nk;
    if constexpr (SPARSE) {
        const int hw = a.Ho * a.Wo;
        const int b = m0 / hw, rem0 = m0 - b * hw;                 // host guarantees hw % BM == 0: one image per tile
        const int oy0 = rem0 / a.Wo, oy1 = (rem0 + BM - 1) / a.Wo;
        const bool fullw = BM >= a.Wo;
        const int ox0 = fullw ? 0 : rem0 - oy0 * a.Wo, ox1 = fullw ? a.Wo - 1 : ox0 + BM - 1;
        for (int ks = tid; ks < a.nk; ks += NT) {
            const int cs = ks / ntaps, tap = ks - cs * ntaps, c0 = cs * BK;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            const int* bb = a.bbox + ((long long)b * a.bbox_groups + c0 / 64) * 4;
            const int y0 = oy0 * a.stride - a.pad + ky, y1 = oy1 * a.stride - a.pad + ky;
            const int x0 = ox0 * a.stride - a.pad + kx, x1 = ox1 * a.stride - a.pad + kx;
            s_flag[ks] = (y1 >= bb[0] && y0 <= bb[1] && x1 >= bb[2] && x0 <= bb[3]) ? 1 : 0;
        }
        __syncthreads();
        if (tid < 64) {
            int cnt = 0;
            for (int base = 0; base < a.nk; base += 64) {
                const int ks = base + lane;
                const bool f = ks < a.nk && s_flag[ks];
                const unsigned long long mask = __ballot(f);
                if (f) {      // packed descriptor: tap | ky << 8 | kx << 12 | channel slab << 16 (no division in the K loop)
                    const int cs = ks / ntaps, tap = ks - cs * ntaps, ky = tap / a.kw, kx = tap - ky * a.kw;
                    s_list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = tap | (ky << 8) | (kx << 12) | (cs << 16);
                }
                cnt += __popcll(mask);
            }
            if (lane == 0) s_nact = cnt;
        }
        __syncthreads();
        nact = s_nact;
    }

    // slab descriptor of the i-th ACTIVE slab.  Dense: an incremental (tap, c0) counter -- no division in the loop.
    int d_tap = 0, d_ky = 0, d_kx = 0, d_c0 = 0;      // state of the DMA stream (runs 3 slabs ahead of the MFMAs)
    auto next_slab = [&](int i) -> Slab {
        Slab s;
        if constexpr (SPARSE) {
            const int e = __builtin_amdgcn_readfirstlane(s_list[i < nact ? i : 0]);
            const int tap = e & 0xff, ky = (e >> 8) & 0xf, kx = (e >> 12) & 0xf, cs = e >> 16;
            s.tap = tap;
            s.toff = ((ky * a.W + kx) * a.in_cs + cs * BK) * ES;
            s.k0 = tap * a.Cin + cs * BK;
        } else {
            s.tap = d_tap;
            s.toff = ((d_ky * a.W + d_kx) * a.in_cs + d_c0) * ES;
            s.k0 = d_tap * a.Cin + d_c0;
            ++d_tap;
            if (++d_kx == a.kw) { d_kx = 0; ++d_ky; }
            if (d_tap == ntaps) { d_tap = 0; d_ky = 0; d_kx = 0; d_c0 += BK; }
        }
        return s;
    };

    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    // DMA piece p of slab s into ring buffer `buf`: wave w fills rows 8w..8w+7 of every RPP-row pass (1 KiB each)
    auto dma_piece = [&](auto PIdx, const Slab& s, unsigned bufaddr, bool live) {
        constexpr int p = decltype(PIdx)::value;
        if constexpr (p < ACH) {
            const bool ok = ((amask[p] >> s.tap) & 1u) && live;
            lds_dma16_m0(xd, bufaddr + p * (RPP * ROW), ok ? (unsigned)(avoff[p] + s.toff) : OOB, 0);
        } else {
            constexpr int i = p - ACH;
            lds_dma16_m0(wd, bufaddr + A_BYTES + i * (RPP * ROW), live ? bvoff[i] : OOB, (unsigned)(s.k0 * ES));
        }
    };
    auto dma_slab = [&](const Slab& s, int buf, bool live) {
        const unsigned ba = lds_base + buf * BUF_BYTES + wave * 1024;
        [&]<int... P>(std::integer_sequence<int, P...>) {
            (dma_piece(std::integral_constant<int, P>{}, s, ba, live), ...);
        }(std::make_integer_sequence<int, NP>{});
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing (conv.hip): lane (i = lane & 31, h = lane >> 5) reads chunks h*4+q of row i, un-swizzled
    const int frag_a = (wm * MI * 32 + (lane & 31)) * ROW;
    const int frag_b = A_BYTES + (wn * NJ * 32 + (lane & 31)) * ROW;
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        qoff[q] = XM ? ((((2 * (q >> 1) + (lane >> 5)) + 4 * (q & 1)) ^ ((lane >> 1) & 7)) << 4)      // q = 2s + part: chunk 2s + h of the hi (lo: + 4) half
                     : ((((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) << 4);

    uint4 fa[2][MI][4], fb[2][NJ][4];              // two register sets of MFMA operands
    // read r (0 .. NREAD-1) of a slab's fragments, q-major so the first MFMAs' operands arrive first
    auto frag_read = [&](auto Set, auto RIdx, const char* bufp) {
        constexpr int set = decltype(Set)::value, r = decltype(RIdx)::value;
        constexpr int q = r / (MI + NJ), t = r - q * (MI + NJ);
        if constexpr (t < MI) fa[set][t][q] = *reinterpret_cast<const uint4*>(bufp + frag_a + t * 32 * ROW + qoff[q]);
        else fb[set][t - MI][q] = *reinterpret_cast<const uint4*>(bufp + frag_b + (t - MI) * 32 * ROW + qoff[q]);
    };

    // PRE: (tap, first channel) of the slab whose fragments are read next; dense order (PRE is never sparse)
    int r_tap = 0, r_c0 = 0;
    const bool pre_relu = (a.flags & 2) != 0;
    auto pre_transform = [&](int buf) {
        if constexpr (PRE) {
            char* base = smem + buf * BUF_BYTES + wave * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                if ((amask[i] >> r_tap) & 1u) {                       // zero padding / M tail stay zero: conv pads the ACTIVATED input
                    uint4* cp = reinterpret_cast<uint4*>(base + i * (RPP * ROW));
                    *cp = prologue<TI>(*cp, s_pre, s_pre + PRE_MAX_CIN, r_c0 + col * EPC, pre_relu);
                }
            }
            if (++r_tap == ntaps) { r_tap = 0; r_c0 += BK; }
        }
    };

    // ---- prologue: three slabs in flight, the first one's fragments in register set 0
    {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const Slab si = next_slab(i);
            dma_slab(si, i, nact > i);
        }
        wait_vmcnt<(NBUF - 1) * NP>();
        pre_transform(0);
        __syncthreads();
        [&]<int... R>(std::integer_sequence<int, R...>) {
            (frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, R>{}, smem), ...);
        }(std::make_integer_sequence<int, NREAD>{});
    }

    // iteration ks (fragments of slab ks are in register set P; DMAs of slabs ks+1, ks+2 are in flight):
    //   wait until only slab ks+2's pieces are outstanding, barrier (everyone's pieces of slab ks+1 have landed and
    //   everyone's reads of buffer ks % 3 have returned), then MFMA(ks) || ds_read(ks+1) || DMA(ks+3 -> buffer ks % 3)
    int buf = 0;                                   // ks % 3
    auto iteration = [&](auto Pc, int ks) {
        constexpr int P = decltype(Pc)::value;
        wait_vmcnt<(NBUF - 2) * NP>();             // slabs ks+2 .. ks+NBUF-1 may still be in flight
        const int nb = buf == NBUF - 1 ? 0 : buf + 1;     // (ks + 1) % NBUF
        if (ks + 1 < nact) pre_transform(nb);      // own pieces of slab ks+1 have landed; everyone else's after the barrier
        __syncthreads();
        const char* rbuf = smem + nb * BUF_BYTES;
        const bool live = ks + NBUF < nact;
        const Slab sd = next_slab(ks + NBUF);
        const unsigned ba = lds_base + buf * BUF_BYTES + wave * 1024;
        // slot plan: MFMA t is followed by RPS fragment reads (slots [0, H)) or DPS DMA pieces (slots [H, NSLOT))
        constexpr int RPS = (NREAD + NSLOT / 2 - 1) / (NSLOT / 2);
        constexpr int H = (NREAD + RPS - 1) / RPS;
        constexpr int DPS = (NP + NSLOT - H - 1) / (NSLOT - H);
        static_assert(H < NSLOT, "slot plan");
        [&]<int... T>(std::integer_sequence<int, T...>) {
            (([&] {
                 constexpr int t = T;
                 constexpr int q = t / (MI * NJ), ij = t - q * (MI * NJ), i = ij / NJ, j = ij - i * NJ;
                 if constexpr (XM == 0) {
                     acc[i][j] = Half<TI>::mfma32(fa[P][i][q], fb[P][j][q], acc[i][j]);
                 } else {           // q counts (k16-step, product): hi*hi, lo*hi, hi*lo (XM = 3) or hi*hi only (XM = 1)
                     constexpr int ks16 = XM == 3 ? q / 3 : q, part = XM == 3 ? q % 3 : 0;
                     acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[P][i][2 * ks16 + (part == 1)]),
                                                                        __builtin_bit_cast(f16x8, fb[P][j][2 * ks16 + (part == 2)]), acc[i][j], 0, 0, 0);
                 }
                 if constexpr (t < H) {
                     [&]<int... R>(std::integer_sequence<int, R...>) {
                         (([&] {
                              if constexpr (t * RPS + R < NREAD)
                                  frag_read(std::integral_constant<int, P ^ 1>{}, std::integral_constant<int, t * RPS + R>{}, rbuf);
                          }()),
                          ...);
                     }(std::make_integer_sequence<int, RPS>{});
                 } else {
                     [&]<int... D>(std::integer_sequence<int, D...>) {
                         (([&] {
                              if constexpr ((t - H) * DPS + D < NP)
                                  dma_piece(std::integral_constant<int, (t - H) * DPS + D>{}, sd, ba, live);
                          }()),
                          ...);
                     }(std::make_integer_sequence<int, DPS>{});
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, NSLOT>{});
        buf = nb;
    };
    stamp();
    for (int ks = 0; ks < nact; ks += 2) {
        iteration(std::integral_constant<int, 0>{}, ks);
        if (ks + 1 < nact) iteration(std::integral_constant<int, 1>{}, ks + 1);
    }
    wait_vmcnt<0>();                               // trailing (out-of-range) refills must land before the LDS is reused
    __syncthreads();
    stamp();

    epilogue_tile<TO, MI, NJ, WM, WN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
    stamp();
}


// ------------------------------------------------------------------------------------------------------------------
// Halo-reuse variant for stride-1 kh x kw convolutions (every 3x3 on the path).  The pipelined kernel above is bound by
// the CU's global -> LDS path (measured: 48 KB per slab at ~43 B/clk/CU = 1130 cycles against 1024 MFMA cycles), and 2/3
// of those bytes are im2col re-reads: the kh*kw taps of one channel slab gather the SAME input pixels, shifted.  Here a
// tile of BM output pixels is a rectangle of whole image rows (or whole small images); its input halo patch
// ((rows + kh - 1) x (Wo + kw - 1) pixels x 64 channels, <= 400 rows of 128 B) is DMA'd ONCE per channel slab into a
// double-buffered LDS patch and every tap reads its MFMA A-operand from the patch at a shifted row.  Per channel slab the
// CU now fetches ~npr + kh*kw*BN rows instead of kh*kw*(BM + BN): 2.2x fewer bytes for 256x128 tiles.  Zero padding is
// the hardware bounds check on the patch DMA (no per-tap masks).  The weight slabs keep the 3-deep ring; the next
// channel slab's patch pieces ride along, one per tap iteration.  fp32 accumulation order is unchanged (bit-identical).
struct PatchGeom {
    int PW, PH;      // patch width / height in input pixels (one segment)
    int nseg;        // segments per tile: 1 (rows of one image) or BM / (Ho*Wo) whole images
    int seg_px;      // output pixels per segment
    int npr;         // patch rows in total = nseg * PH * PW  (<= PATCH_MAX_ROWS)
};
constexpr int PATCH_MAX_ROWS = 400;

struct KPos {        // position in the (channel slab, tap) stream
    int ci, tap, ky, kx;
};

// PRELOAD (single channel slab, e.g. ResNet layer1's 3x3 64 -> 64): the patch AND every tap's weight slab are fetched in one
// burst, one barrier, then all kh*kw taps run back to back with no ring, no counted waits and no per-tap barrier.
template <typename TO, int MI, int NJ, int WM, int WN, bool SPARSE, bool PRELOAD = false, int PMAX = PATCH_MAX_ROWS, typename TIN = bf16_t>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_patch_kernel(ConvArgs a, PatchGeom g) {
    typedef TIN TI;
    half_kernel_init<TO>();
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    constexpr int RPP = NT / 8;                        // rows per DMA pass of the workgroup
    constexpr int BCH = BN / RPP;                      // weight DMA pieces per thread per slab
    constexpr int MAXPP = (PMAX + RPP - 1) / RPP;             // patch DMA passes per channel slab
    constexpr int ROW = 128;
    constexpr int P_BYTES = PMAX * ROW, B_BYTES = BN * ROW;
    constexpr int NBUF = PRELOAD ? 9 : 3;              // PRELOAD: one weight buffer per tap (kh*kw <= 9), one patch buffer
    constexpr int NPATCH = PRELOAD ? 1 : 2;
    constexpr int RING_BYTES = NPATCH * P_BYTES + NBUF * B_BYTES;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = RING_BYTES > STAGE_BYTES ? RING_BYTES : STAGE_BYTES;
    constexpr int BK = 64, ES = 2;
    static_assert(BN % RPP == 0, "weight tile rows must be a multiple of the DMA pass");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    __shared__ int s_list[SPARSE ? 64 : 1];            // active channel slabs (sparse-K at (hand, bone) granularity)
    __shared__ int s_nact;

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
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
    const int ntaps = a.kh * a.kw;
    const int hw = a.Ho * a.Wo;
    const int b0 = m0 / hw, y0 = (m0 - b0 * hw) / a.Wo;        // the tile starts at the beginning of an image row

    // ---- patch DMA source of this thread: patch row RPP * i + (tid >> 3), chunk (tid & 7) ^ swizzle, channel slab 0
    // Swizzle of the PATCH rows (round 5, profiles/r05_a_sq_counters.txt: SQ_LDS_BANK_CONFLICT was 25 % of this kernel's LDS cycles): chunk c of
    // patch row r sits at position c ^ key(r), key(r) = ((r - (kw - 1) * (r / PW)) >> 1) & 7 -- the row index with the halo columns of the
    // patch lines above it taken out.  A wave's 32 output pixels read rows that are consecutive EXCEPT for a jump of kw - 1 at every image-row
    // end (Wo = 16: two image rows per 32 lanes, Wo = 8: four); with the plain (r >> 1) & 7 key those jumps put two rows of a 16-lane ds_read_b128
    // group on the same 16-byte slot.  In t = r - (kw - 1) * line the lanes of a 32-pixel block are linear again (t = t0 + lane for every tap),
    // row parity = parity of t, and a 16-lane group covers 16 distinct (parity, key) pairs for any t0: conflict free at every tap.
    unsigned pvoff[MAXPP];
    const int col = (tid & 7) ^ ((tid >> 4) & 7);           // (weight rows: the plain key)
    const int pseg = g.PH * g.PW;
#pragma unroll
    for (int i = 0; i < MAXPP; ++i) {
        const int prow = RPP * i + (tid >> 3);
        pvoff[i] = OOB;
        if (prow < g.npr) {
            const int seg = prow / pseg, rem = prow - seg * pseg;
            const int py = rem / g.PW, px = rem - py * g.PW;
            const int b = g.nseg > 1 ? b0 + seg : b0;
            const int iy = (g.nseg > 1 ? 0 : y0) + py - a.pad, ix = px - a.pad;
#if DIR_PATCH_PLAIN_KEY            // (A/B aid, -DDIR_PATCH_PLAIN_KEY=1: the pre-round-5 key (row >> 1) & 7)
            const int pcol = (tid & 7) ^ ((prow >> 1) & 7);
#else
            const int pcol = (tid & 7) ^ (((prow - (a.kw - 1) * (seg * g.PH + py)) >> 1) & 7);
#endif
            if (b < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                pvoff[i] = (unsigned)((((b * a.H + iy) * a.W + ix) * a.in_cs + a.in_co + pcol * 8) * ES);
        }
    }
    // passes this wave takes part in (a piece always writes the wave's 8 rows; rows >= npr are zero-filled)
    const int npr8 = (g.npr + 7) & ~7;
    const int npw = npr8 > 8 * wave ? (npr8 - 8 * wave + RPP - 1) / RPP : 0;
    unsigned bvoff[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int n = n0 + (tid >> 3) + RPP * i;
        bvoff[i] = n < a.Cout ? (unsigned)((n * a.K + col * 8) * ES) : OOB;
    }

    // ---- optional sparse-K: channel slabs (= 64-channel input groups) whose support can touch the tile's rows
    int ncs = a.Cin / BK;
    if constexpr (SPARSE) {
        const int oy1 = y0 + BM / a.Wo - 1;                        // host guarantees one image per tile when sparse
        if (tid < 64) {
            int cnt = 0;
            for (int base = 0; base < ncs; base += 64) {
                const int cs = base + lane;
                bool f = false;
                if (cs < ncs) {
                    const int* bb = a.bbox + ((long long)b0 * a.bbox_groups + cs) * 4;
                    f = oy1 + (a.kh - 1 - a.pad) >= bb[0] && y0 - a.pad <= bb[1] && bb[2] <= bb[3];
                }
                const unsigned long long mask = __ballot(f);
                if (f) s_list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = cs;
                cnt += __popcll(mask);
            }
            if (lane == 0) s_nact = cnt;
        }
        __syncthreads();
        ncs = s_nact;
    }
    const int nslab = ncs * ntaps;
    auto c0_of = [&](int ci) -> int {                  // first channel of the ci-th active channel slab
        if constexpr (SPARSE) return __builtin_amdgcn_readfirstlane(s_list[ci < ncs ? ci : 0]) * BK;
        else return ci * BK;
    };
    auto advance = [&](KPos& p) {
        ++p.tap;
        if (++p.kx == a.kw) { p.kx = 0; ++p.ky; }
        if (p.tap == ntaps) { p.tap = 0; p.ky = 0; p.kx = 0; ++p.ci; }
    };

    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned bring = lds_base + NPATCH * P_BYTES + wave * 1024;
    // weight piece i of the slab at position p into ring buffer `buf`
    auto dma_b = [&](auto I, const KPos& p, int c0, int buf, bool live) {
        constexpr int i = decltype(I)::value;
        lds_dma16_m0(wd, bring + buf * B_BYTES + i * (RPP * ROW), live ? bvoff[i] : OOB, (unsigned)((p.tap * a.Cin + c0) * ES));
    };
    // patch piece t (wave-uniform) of the channel slab starting at channel c0 into patch buffer pb
    // (an if-chain over static indices: a dynamically indexed pvoff[] would be demoted to scratch memory, and the
    //  scratch load's vmcnt(0) would drain the DMA ring)
    auto dma_patch = [&](int t, int c0, int pb, bool live) {
        const unsigned dst = lds_base + pb * P_BYTES + (RPP * t + 8 * wave) * ROW;
#pragma unroll
        for (int i = 0; i < MAXPP; ++i)
            if (t == i) lds_dma16_m0(xd, dst, live ? pvoff[i] : OOB, (unsigned)(c0 * ES));
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing.  A: lane (i = lane & 31, h = lane >> 5) reads chunks h*4+q of the patch row of its output
    //      pixel shifted by the tap; B: as in conv.hip.  Chunk c of LDS row r sits at position c ^ ((r >> 1) & 7).
    int pr0[MI], pt0[MI];                             // patch row of the pixel's first tap, and its key index t0 = row - (kw - 1) * line
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * MI * 32 + i * 32 + (lane & 31);
        const int seg = r / g.seg_px, rr = r - seg * g.seg_px;
        const int y = rr / a.Wo, xx = rr - y * a.Wo;
        pr0[i] = seg * pseg + y * g.PW + xx;
#if DIR_PATCH_PLAIN_KEY
        pt0[i] = pr0[i];
#else
        pt0[i] = pr0[i] - (a.kw - 1) * (seg * g.PH + y);
#endif
    }
    int hq4[4], qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hq4[q] = ((lane >> 5) * 4 + q) << 4;
        qoff[q] = (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) << 4;
    }
    const int frag_b = (wn * NJ * 32 + (lane & 31)) * ROW;

    uint4 fa[MI][4], fb[NJ][4];
    auto frag_load = [&](const KPos& p, const char* bbuf) {
#if DIR_PATCH_PLAIN_KEY
        const int shift = p.ky * g.PW + p.kx, tshift = shift;
#else
        const int shift = p.ky * g.PW + p.kx, tshift = p.ky * (g.PW - (a.kw - 1)) + p.kx;      // (row, key index) of tap (ky, kx) relative to tap 0
#endif
        const int pbase = (p.ci & 1) * P_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = pr0[i] + shift;
                fa[i][q] = *reinterpret_cast<const uint4*>(smem + pbase + row * ROW + (hq4[q] ^ (((pt0[i] + tshift) << 3) & 0x70)));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j][q] = *reinterpret_cast<const uint4*>(bbuf + frag_b + j * 32 * ROW + qoff[q]);
        }
    };

    if constexpr (PRELOAD) {
        KPos cur = {0, 0, 0, 0}, dm = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < MAXPP; ++t)
            if (t < npw) dma_patch(t, 0, 0, true);
        for (int t = 0; t < ntaps; ++t) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (dma_b(std::integral_constant<int, I>{}, dm, 0, t, true), ...);
            }(std::make_integer_sequence<int, BCH>{});
            advance(dm);
        }
        wait_vmcnt<0>();
        __syncthreads();
        for (int t = 0; t < ntaps; ++t) {
            frag_load(cur, smem + NPATCH * P_BYTES + t * B_BYTES);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = Half<TI>::mfma32(fa[i][q], fb[j][q], acc[i][j]);
            advance(cur);
        }
        __syncthreads();
        epilogue_tile<TO, MI, NJ, WM, WN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
        return;
    }

    // ---- prologue: patch of channel slab 0 and weight slabs 0, 1 (the loop issues slab s+2 during slab s)
    KPos cur = {0, 0, 0, 0}, dm = {0, 0, 0, 0};
    {
        const int c0 = c0_of(0);
#pragma unroll
        for (int t = 0; t < MAXPP; ++t)
            if (t < npw) dma_patch(t, c0, 0, nslab > 0);
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const int cb = c0_of(dm.ci);
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (dma_b(std::integral_constant<int, I>{}, dm, cb, sidx, sidx < nslab), ...);
            }(std::make_integer_sequence<int, BCH>{});
            advance(dm);
        }
        wait_vmcnt<0>();
        __syncthreads();
    }

    // ---- main loop: two wave groups in ping-pong.  Waves w and w + 4 share a SIMD; group 0 = waves 0..3, group 1 = 4..7.
    // Each slab has a memory phase M (16 ds_read_b128 of the slab's fragments + this wave's DMA pieces of slab s+2) and a
    // compute phase C (16 MFMAs, nothing else); group 1 runs one phase behind group 0, so on every SIMD one wave is in C
    // while the other is in M and neither wave's memory instructions ever sit in front of its own MFMAs.
    //   slot:      2s          2s+1        2s+2
    //   group 0:   M(s)        C(s)        M(s+1)
    //   group 1:   C(s-1)      M(s)        C(s)
    // Slab s must have landed before slot 2s: every wave waits for its own pieces of slab s (everything but the pieces of
    // slab s+1 it issued last) at the end of slot 2s-1, which is the end of C for group 0 and the end of M for group 1.
    // Slab s+2 goes into the ring buffer of slab s-1, last read by group 1 in slot 2s-1.
    const bool grp = wave >= (WM * WN) / 2;
    if (grp) __builtin_amdgcn_s_barrier();
    int buf = 0;                                   // s % 3
    for (int sidx = 0; sidx < nslab; ++sidx) {
        // ---- M(s)
        frag_load(cur, smem + NPATCH * P_BYTES + buf * B_BYTES);
        const int nb2 = buf == 0 ? 2 : buf - 1;    // (s + 2) % 3
        const bool pp = cur.tap < npw;
        if (pp) dma_patch(cur.tap, c0_of(cur.ci + 1), (cur.ci + 1) & 1, cur.ci + 1 < ncs);
        {
            const int c0d = c0_of(dm.ci);
            const bool blive = sidx + 2 < nslab;
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (dma_b(std::integral_constant<int, I>{}, dm, c0d, nb2, blive), ...);
            }(std::make_integer_sequence<int, BCH>{});
        }
        if (grp) {
            if (pp) wait_vmcnt<BCH + 1>();
            else wait_vmcnt<BCH>();
        }
        __syncthreads();                            // (drains the ds_reads: lgkmcnt(0))
        // ---- C(s)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = Half<TI>::mfma32(fa[i][q], fb[j][q], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (!grp) {
            if (pp) wait_vmcnt<BCH + 1>();
            else wait_vmcnt<BCH>();
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the MFMAs above the phase-closing barrier
        __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
        advance(cur);
        advance(dm);
    }
    if (!grp) __builtin_amdgcn_s_barrier();        // group 1 executed one more barrier up front
    wait_vmcnt<0>();
    __syncthreads();

    epilogue_tile<TO, MI, NJ, WM, WN>(a, acc, smem, m0, n0, wm, wn, tid, lane);
}

// geometry of the halo patch for BM-pixel tiles; false if the layer does not fit the patch kernel
static bool patch_geometry(const ConvArgs& a, int bm, PatchGeom* g) {
    const int ntaps = a.kh * a.kw;
    if (a.stride != 1 || ntaps < 4 || bm % a.Wo != 0) return false;
    const int rows = bm / a.Wo;
    int rows_seg;
    if (rows <= a.Ho) {
        if (a.Ho % rows != 0) return false;
        g->nseg = 1;
        rows_seg = rows;
    } else {
        if (rows % a.Ho != 0) return false;
        g->nseg = rows / a.Ho;
        rows_seg = a.Ho;
    }
    g->PW = a.Wo + a.kw - 1;
    g->PH = rows_seg + a.kh - 1;
    g->seg_px = rows_seg * a.Wo;
    g->npr = g->nseg * g->PH * g->PW;
    if (g->npr > PATCH_MAX_ROWS) return false;
    // one patch piece per tap iteration, the last one at least one iteration before the next channel slab starts
    if ((g->npr + 63) / 64 > ntaps - 1) return false;
    return true;
}

template <typename TO, int MI, int NJ, int WM, int WN, int XM = 0, typename TIN = bf16_t>
void launch_tile(ConvArgs a, hipStream_t s) {
    constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    choose_tile_order(a, XM ? 4 : 2);
    dim3 grid(a.tiles_m * a.tiles_n), block(64 * WM * WN);
    if ((a.st_p1 || a.bs_p1) && std::is_same<TO, float>::value) stats_rows_launched = BM;          // every kernel below ends in epilogue_tile, which forms the statistics
    if constexpr (XM != 0) {
        DIR_LAUNCH((conv_pipe_kernel<float, MI, NJ, WM, WN, false, false, 3, XM>), grid, block, 0, s, a);
        return;
    }
    static const int use_patch = getenv("DIR_PATCH") ? atoi(getenv("DIR_PATCH")) : 0;           // halo-reuse kernel: opt-in
    PatchGeom g;
    const bool want_patch = use_patch || (a.variant >= 12 && a.variant <= 14);      // DIR_CONV_VARIANT 12..14: halo reuse
    if (want_patch && !a.pre_scale && patch_geometry(a, BM, &g) && (!a.bbox || a.Cin / 64 <= 64)) {
        if constexpr (MI == 2 && NJ == 1 && WM == 4 && WN == 2) {      // 256x64: patch + 9 weight slabs = 125 KB
            if (!a.bbox && a.Cin == 64 && a.kh * a.kw <= 9) {
                DIR_LAUNCH((conv_patch_kernel<TO, MI, NJ, WM, WN, false, true, PATCH_MAX_ROWS, TIN>), grid, block, 0, s, a, g);
                return;
            }
        }
        if (a.bbox) DIR_LAUNCH((conv_patch_kernel<TO, MI, NJ, WM, WN, true, false, PATCH_MAX_ROWS, TIN>), grid, block, 0, s, a, g);
        else DIR_LAUNCH((conv_patch_kernel<TO, MI, NJ, WM, WN, false, false, PATCH_MAX_ROWS, TIN>), grid, block, 0, s, a, g);
        return;
    }
    if constexpr (!(MI == 2 && NJ == 2)) {                // pre-activation variant: the tiles whose ring leaves room for s_pre
        if (a.pre_scale) {
            DIR_LAUNCH((conv_pipe_kernel<TO, MI, NJ, WM, WN, false, true, 3, 0, TIN>), grid, block, 0, s, a);
            return;
        }
    }
    if (a.bbox) DIR_LAUNCH((conv_pipe_kernel<TO, MI, NJ, WM, WN, true, false, 3, 0, TIN>), grid, block, 0, s, a);
    else DIR_LAUNCH((conv_pipe_kernel<TO, MI, NJ, WM, WN, false, false, 3, 0, TIN>), grid, block, 0, s, a);
}

}  // namespace

bool launch_conv_pipe(const ConvArgs& a, bool out_f32, int num_cu, hipStream_t s, int xm, bool f16) {
    // DIR_PIPE: 0 = never, 1 = 256x128, 2 = 128x128, 3 = 256x64, unset = automatic (tuning aid)
    static const int force = getenv("DIR_PIPE") ? atoi(getenv("DIR_PIPE")) : -1;
    static const int min_nk = getenv("DIR_PIPE_MIN_NK") ? atoi(getenv("DIR_PIPE_MIN_NK")) : 8;
    const bool explicit_variant = (a.variant >= 8 && a.variant <= 10) || (!xm && a.variant >= 12 && a.variant <= 15);
    if (xm && (a.pre_scale || a.bbox || !out_f32)) return false;
    if (force == 0 || !(a.flags & 4) || (a.nk < min_nk && !explicit_variant)) return false;
    const bool pre = a.pre_scale != nullptr;
    static const int pre_pipe = getenv("DIR_PIPE_PRE") ? atoi(getenv("DIR_PIPE_PRE")) : 1;     // tuning aid
    if (pre && (!pre_pipe || a.Cin > PRE_MAX_CIN || a.bbox)) return false;
    const long long hw = (long long)a.Ho * a.Wo;
    auto tiles = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn); };
    int shape = 0;
    if (explicit_variant) {                                // explicit tile (DIR_CONV_VARIANT 8..10, 12..14 = with halo reuse)
        shape = a.variant >= 12 ? a.variant - 11 : a.variant - 7;
        if (a.nk < 1 || (pre && shape == 1)) return false;
        if (a.variant == 15) {
            // DIR_CONV_VARIANT 15: 128 x 64 tile on EIGHT waves (32 x 32 wave tiles) -- the small-M layers (8 x 8 / 16 x 16 maps: 256 such
            // tiles at B = 64) otherwise run on four waves, one per SIMD, with nothing to hide a wave's fragment-read latency behind.
            // Ring depth DIR_P15_NBUF: 3 .. 6 measured equal (layer4 3x3: 33-35 us), so the shallow one, which lets two workgroups share a CU
            if (a.bbox || xm) return false;
            ConvArgs b = a;
            if ((b.st_p1 || b.bs_p1) && out_f32) stats_rows_launched = 128;
            b.tiles_m = (b.M + 127) / 128;
            b.tiles_n = (b.Cout + 63) / 64;
            choose_tile_order(b, 2);
            const dim3 grid(b.tiles_m * b.tiles_n), block(512);
            if (f16) {
                if (pre) {
                    if (out_f32) DIR_LAUNCH((conv_pipe_kernel<float, 1, 1, 4, 2, false, true, DIR_P15_NBUF, 0, f16s_t>), grid, block, 0, s, b);
                    else DIR_LAUNCH((conv_pipe_kernel<f16s_t, 1, 1, 4, 2, false, true, DIR_P15_NBUF, 0, f16s_t>), grid, block, 0, s, b);
                } else if (out_f32) DIR_LAUNCH((conv_pipe_kernel<float, 1, 1, 4, 2, false, false, DIR_P15_NBUF, 0, f16s_t>), grid, block, 0, s, b);
                else DIR_LAUNCH((conv_pipe_kernel<f16s_t, 1, 1, 4, 2, false, false, DIR_P15_NBUF, 0, f16s_t>), grid, block, 0, s, b);
                return true;
            }
            if (pre) {
                if (out_f32) DIR_LAUNCH((conv_pipe_kernel<float, 1, 1, 4, 2, false, true, DIR_P15_NBUF>), grid, block, 0, s, b);
                else DIR_LAUNCH((conv_pipe_kernel<bf16_t, 1, 1, 4, 2, false, true, DIR_P15_NBUF>), grid, block, 0, s, b);
            } else if (out_f32) DIR_LAUNCH((conv_pipe_kernel<float, 1, 1, 4, 2, false, false, DIR_P15_NBUF>), grid, block, 0, s, b);
            else DIR_LAUNCH((conv_pipe_kernel<bf16_t, 1, 1, 4, 2, false, false, DIR_P15_NBUF>), grid, block, 0, s, b);
            return true;
        }
    } else if (force > 0) shape = force;
    else {
        // largest tile that still gives (nearly) every CU a workgroup; 64-wide N tile only for Cout <= 64
        const long long need = (long long)num_cu * 3 / 4;
        if (a.Cout <= 64) shape = tiles(256, 64) >= need ? 3 : 0;
        else if (tiles(256, 128) >= need && !pre) shape = 1;
        else if (tiles(128, 128) >= (pre ? need / 2 : need)) shape = 2;   // the register-staged pre-activation path is slow: accept half-full grids
    }
    if (shape == 0 || (pre && shape == 1)) return false;
    const int bm = shape == 2 ? 128 : 256;
    ConvArgs b = a;
    if (b.bbox && hw % bm != 0) b.bbox = nullptr;          // sparse-K needs whole tiles inside one image
#define DIR_PIPE_LAUNCH(MI_, NJ_, WM_, WN_)                                       \
    do {                                                                          \
        if (xm == 3) launch_tile<float, MI_, NJ_, WM_, WN_, 3>(b, s);             \
        else if (xm == 1) launch_tile<float, MI_, NJ_, WM_, WN_, 1>(b, s);        \
        else if (f16 && out_f32) launch_tile<float, MI_, NJ_, WM_, WN_, 0, f16s_t>(b, s);   \
        else if (f16) launch_tile<f16s_t, MI_, NJ_, WM_, WN_, 0, f16s_t>(b, s);   \
        else if (out_f32) launch_tile<float, MI_, NJ_, WM_, WN_>(b, s);           \
        else launch_tile<bf16_t, MI_, NJ_, WM_, WN_>(b, s);                       \
    } while (0)
    if (shape == 1) DIR_PIPE_LAUNCH(2, 2, 4, 2);
    else if (shape == 2) DIR_PIPE_LAUNCH(2, 1, 2, 4);
    else DIR_PIPE_LAUNCH(2, 1, 4, 2);
#undef DIR_PIPE_LAUNCH
    return true;
}

}  // namespace convk
}  // namespace dir
