// dir_conv1x1_stream_forward: the HBM-bound 1x1 convolutions of the path (bf16) as a STREAMING kernel.
//   models/backbone/hourglass.py:55-70   Residual: conv1 (pre-activation BN + ReLU on the input, 1x1, Cin -> Cout/2) and
//                                        conv3 + skip_layer (two 1x1 convolutions summed: a second K range over the block input)
//   models/backbone/resnet.py:117-140    Bottleneck conv1 / conv3 (+ the strided projection shortcut as a second K range)
// These layers move 80 - 150 MB for a few GFLOP: what counts is bytes in flight.  A plain copy kernel reaches 6.4 - 7 TB/s on this
// chip (tools/ubench_hbm.hip) with ~32 KB outstanding per CU; the tiled implicit-GEMM kernels (conv.hip, conv_pipe.hip), built for
// the MFMA-bound layers, put one or two fat workgroups on a CU and stream these layers at 2.1 - 3.7 TB/s: every workgroup
// alternates between waiting for a K-slab and computing on it, and their weight tiles travel L2 -> LDS with the activations.
//
// Here a workgroup is small and synchronous (4 waves, 128 output pixels x up to 256 output channels, <= 40 KB of LDS), TWO to
// three of them share a CU and are in different phases, so the memory system always has a workgroup's loads to serve:
//   * activations: 128 px x 64 channels (16 KB) per K-chunk, global -> registers two chunks ahead -> LDS (double buffer).  All
//     loads are ordinary compiler-tracked loads: vmcnt is in order on gfx9, and with every load visible to the compiler its
//     counted waits are exact -- an (invisible) LDS-DMA in the same wave would make every weight wait cover the DMAs behind it;
//   * weights never touch LDS: the GEMM is computed as D[channel][pixel] (weights = MFMA A operand) and a wave owns 32 or 64
//     output channels, so a weight fragment is used by exactly one wave -- the host packs them in consumption order
//     (dir_amd/engine.py::pack_stream_weights) and a fragment is one coalesced 1 KB load a chunk ahead, L2-resident;
//   * the pre-activation BatchNorm + ReLU is applied in registers on the way to LDS; the epilogue (scale, shift, ReLU) writes
//     8-byte pieces -- a lane holds 4 consecutive channels of a pixel (bneck.hip).
// fp32 accumulation in the K order AND the k-slot assignment of the tiled kernels (MFMA q of a 64-channel chunk multiplies channels
// 8q..8q+7 in lanes 0-31 and 32+8q..32+8q+7 in lanes 32-63, conv.hip): the sums are the same fp32 chains, the outputs bit-identical.
#include "conv_common.h"

namespace dir {
namespace {

using convk::bf16_t;
using convk::f16s_t;
using convk::f32x16;
using convk::Half;

constexpr int SBM = 128;               // pixels per workgroup (NPB = 4 blocks of 32; the NPB = 2 variant: 64)
constexpr int SKC = 64;                // channels per K-chunk
constexpr int STHR = 256;
constexpr int PRE_MAX = 2304;          // pre-activation parameters staged in LDS

struct StreamArgs {
    const bf16_t* x; const bf16_t* x2; bf16_t* y;
    const uint4* w;                    // [N chunks][4 waves][nk chunks][4 k-steps][NCB][64 lanes] x 16 bytes
    const float* scale; const float* shift; const float* pre_scale; const float* pre_shift;
    int M, Cin, in_cs, in_co, Cout, out_cs, out_co, nk1, nk, relu, pre_relu;
    int HoWo, Wo, H2, W2, in_cs2, in_co2, stride2;        // second source geometry (nk > nk1)
};

// NCB: 32-channel blocks per wave (1: 128 output channels per workgroup, 2: 256); NPB: 32-pixel blocks per workgroup (4: 128 pixels; 2: 64 pixels --
// twice the workgroups, each half as long: more of them per CU in different phases for the layers whose 128-pixel grid is only two tiles per CU)
// H: the 16-bit storage kind of activations, weights and outputs (bf16_t | f16s_t: DIR_DT_BF16 | DIR_DT_F16)
template <int NCB, bool PRE, int NPB, typename H = bf16_t>
__global__ __launch_bounds__(STHR, NCB == 1 ? 3 : 2) void stream1x1_kernel(StreamArgs a) {
    convk::half_kernel_init<H>();
    constexpr int SBM = 32 * NPB;
    constexpr int NWG = 128 * NCB;                         // output channels per workgroup
    constexpr int OPITCH = 256 + 16;                       // bytes per pixel of the staged output half (128 channels), 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) char s_raw[SBM * OPITCH];      // two 16 KB activation chunks during the K loop, then the output stage
    char (*s_a)[SBM * 128] = reinterpret_cast<char (*)[SBM * 128]>(s_raw);
    __shared__ float s_pre[PRE ? 2 * PRE_MAX : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * SBM, nchunk = blockIdx.y;
    if constexpr (PRE) {
        for (int i = tid; i < a.Cin; i += STHR) { s_pre[i] = a.pre_scale[i]; s_pre[PRE_MAX + i] = a.pre_shift[i]; }
    }

    // ---- activation loads: thread = (row (tid >> 3) + 32 i, 16-byte column tid & 7) of the 128 x 64 chunk
    const int col = tid & 7;
    unsigned off1[NPB], off2[NPB];                             // element offsets (tensors < 2^31 elements, checked by the launcher)
    bool ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        ok[i] = m < a.M;
        const int mc = ok[i] ? m : a.M - 1;
        off1[i] = (unsigned)((long long)mc * a.in_cs + a.in_co + col * 8);
        off2[i] = 0;
        if (a.nk > a.nk1) {
            const int b = mc / a.HoWo, rem = mc - b * a.HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            off2[i] = (unsigned)((((long long)b * a.H2 + oy * a.stride2) * a.W2 + ox * a.stride2) * a.in_cs2 + a.in_co2 + col * 8);
        }
    }
    // Every load below is UNCONDITIONAL (chunk indices clamped, the source picked with a select): a load inside a branch makes
    // hipcc's wait-count pass assume the worst at the join -- vmcnt(0) at the top of every iteration, i.e. no load ever overlaps
    // a computation (first version of this kernel: 2 TB/s).
    auto a_load = [&](int c, uint4 (&r)[NPB]) {
        const int cc = min(c, a.nk - 1);
        const bool first = cc < a.nk1;
        const bf16_t* base = first ? a.x : a.x2;
        const unsigned koff = (unsigned)((first ? cc : cc - a.nk1) * SKC);
#pragma unroll
        for (int i = 0; i < NPB; ++i) r[i] = *reinterpret_cast<const uint4*>(base + (first ? off1[i] : off2[i]) + koff);
    };
    auto a_store = [&](int c, int buf, uint4 (&r)[NPB]) {
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int row = (tid >> 3) + 32 * i;
            uint4 v = r[i];
            if constexpr (PRE) {
                if (c < a.nk1) v = convk::prologue<H>(v, s_pre, s_pre + PRE_MAX, c * SKC + col * 8, a.pre_relu != 0);
            }
            if (!ok[i]) v = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(s_a[buf] + row * 128 + ((col ^ ((row >> 1) & 7)) << 4)) = v;
        }
    };
    // ---- weight fragments of a chunk: [4 k-steps][NCB]
    const uint4* wbase = a.w + ((long long)(nchunk * 4 + wave) * a.nk) * (4 * NCB * 64) + lane;
    auto w_load = [&](int c, uint4 (&wr)[4][NCB]) {
        const uint4* p = wbase + (long long)min(c, a.nk - 1) * (4 * NCB * 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) wr[ks][cb] = p[(ks * NCB + cb) * 64];
    };

    f32x16 acc[NCB][NPB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;

    uint4 ar[2][NPB];
    uint4 wr[2][4][NCB];
    w_load(0, wr[0]);
    a_load(0, ar[0]);
    a_load(1, ar[1]);
    __syncthreads();                                       // s_pre
    a_store(0, 0, ar[0]);
    __syncthreads();

    // iteration c (chunk c is in LDS buffer c & 1, its weights in wr[c & 1], chunk c + 1 on its way into ar[(c + 1) & 1]):
    //   request the weights of chunk c + 1 (L2) and then the activations of chunk c + 2 (HBM) -- in this order, so that no wait
    //   on a weight fragment ever covers a younger HBM load --, compute chunk c, park chunk c + 1 in LDS
    auto iteration = [&](auto Par, int c) {
        constexpr int par = decltype(Par)::value;
        w_load(c + 1, wr[par ^ 1]);
        a_load(c + 2, ar[par]);                              // (ar[par] held chunk c: in LDS since the previous iteration)
        __builtin_amdgcn_sched_barrier(0);
        const char* sa = s_a[par];
        if (c < a.nk) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 bv[NPB];
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const int row = 32 * pb + l32;
                bv[pb] = *reinterpret_cast<const uint4*>(sa + row * 128 + (((ks + 4 * h) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
                    acc[cb][pb] = Half<H>::mfma32(wr[par][ks][cb], bv[pb], acc[cb][pb]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        a_store(min(c + 1, a.nk - 1), par ^ 1, ar[par ^ 1]);   // (past the end: a copy of the last chunk that nobody reads)
        __syncthreads();
    };
    for (int c = 0; c < a.nk; c += 2) {                      // an odd chunk count runs one idle half-iteration (loads clamped, no MFMA)
        iteration(std::integral_constant<int, 0>{}, c);
        iteration(std::integral_constant<int, 1>{}, c + 1);
    }

    // ---- epilogue: lane = pixel 32 pb + l32, channels n0 + 32 cb + 8 q + 4 h .. +4.  The tile leaves through LDS, 128 channels at a
    //      time: 8-byte pieces in (a lane holds 4 consecutive channels of a pixel), coalesced 16-byte row segments out -- written
    //      straight from the MFMA layout every store instruction would touch 64 different 64-byte sectors for 8 bytes each.
    const bool relu = a.relu != 0;
#pragma unroll
    for (int half = 0; half < NCB; ++half) {
        // this half = the workgroup's channels [128 half, 128 half + 128): wave w contributes its block cb with (w * NCB + cb) / 4 == half
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int blk = wave * NCB + cb;               // 32-channel block inside the workgroup's NWG channels
            if ((blk >> 2) != half) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = blk * 32 + 8 * q + 4 * h, n = nchunk * NWG + nl;
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + n);
                if (a.shift) sh = *reinterpret_cast<const float4*>(a.shift + n);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    uint2 o;
                    const float v0 = fmaf(acc[cb][pb][4 * q], sc.x, sh.x), v1 = fmaf(acc[cb][pb][4 * q + 1], sc.y, sh.y);
                    const float v2 = fmaf(acc[cb][pb][4 * q + 2], sc.z, sh.z), v3 = fmaf(acc[cb][pb][4 * q + 3], sc.w, sh.w);
                    if (relu) { o.x = Half<H>::pack2_relu(v0, v1); o.y = Half<H>::pack2_relu(v2, v3); }
                    else { o.x = Half<H>::pack2(v0, v1); o.y = Half<H>::pack2(v2, v3); }
                    *reinterpret_cast<uint2*>(s_raw + (32 * pb + l32) * OPITCH + (nl - 128 * half) * 2) = o;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SBM * 16 / STHR; ++i) {        // 16-byte chunk c = tid + 256 i = pixel c >> 4, channels 8 (c & 15) .. +8 of the half
            const int c = tid + STHR * i, row = c >> 4, cc = c & 15;
            const int m = m0 + row;
            if (m < a.M)
                *reinterpret_cast<uint4*>(a.y + (long long)m * a.out_cs + a.out_co + nchunk * NWG + 128 * half + cc * 8) =
                    *reinterpret_cast<const uint4*>(s_raw + row * OPITCH + cc * 16);
        }
        if (half + 1 < NCB) __syncthreads();
    }
}


// ---- the PIPELINED form (round 5, DIR_CONV_VARIANT 24; VERDICT r4 item 1b: "break the load -> compute -> store lock-step").  The kernel above keeps an
// activation chunk two K-iterations ahead in registers -- at 0.25 us of MFMAs per iteration that is 0.5 us of lead against 1 - 2 us of HBM latency, so
// every iteration waits (profiles/r05_a_sq_counters.txt: 58 % of its wave cycles in s_waitcnt, MFMA pipe 24 % busy).  Here a FIFTH wave does nothing but
// move activation chunks global -> LDS by DMA into a ring of D buffers, D - 1 chunks (48 KB) ahead; the four MFMA waves never issue an activation
// load, so their only vector-memory traffic is the compiler-tracked weight stream (the in-order vmcnt problem of mixing invisible DMAs with tracked
// loads in ONE wave -- the reason the kernel above stages through registers -- does not arise across waves).  One barrier per chunk: it tells the
// consumers that chunk c has landed (the producer waited for its own DMAs first) and the producer that buffer (c - 1) % D is free again.
// Same k-slots in the same order, same epilogue: bit-identical to the kernel above and to the tiled kernels.  No pre-activation form (that one
// must touch the data in registers).
constexpr int PD = 4;                  // ring depth: PD - 1 chunks in flight
template <int NCB, typename H>
__global__ __launch_bounds__(320, 2) void stream1x1p_kernel(StreamArgs a, unsigned x_bytes, unsigned x2_bytes) {
    convk::half_kernel_init<H>();
    constexpr int NPB = 4, SBM = 128, NWG = 128 * NCB, OPITCH = 256 + 16, CH = SBM * 128;       // CH: bytes of one activation chunk
    static_assert(PD * CH >= SBM * OPITCH, "the output stage re-uses the ring");
    __shared__ __attribute__((aligned(16))) char s_raw[PD * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * SBM, nchunk = blockIdx.y;
    constexpr unsigned OOB = 0x80000000u;

    if (wave == 4) {
        // ---- producer: DMA instruction j of a chunk fills rows 8 j .. 8 j + 7 (1 KB); lane = (row 8 j + (lane >> 3), position lane & 7) and fetches the
        //      16-byte column that belongs at that position under the XOR swizzle the consumers read with
        const convk::i32x4 d1 = {(int)(unsigned)(unsigned long long)a.x, (int)(unsigned)((unsigned long long)a.x >> 32), (int)x_bytes, 0x00020000};
        const convk::i32x4 d2 = {(int)(unsigned)(unsigned long long)a.x2, (int)(unsigned)((unsigned long long)a.x2 >> 32), (int)x2_bytes, 0x00020000};
        unsigned o1[16], o2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 8 * j + (lane >> 3), col = (lane & 7) ^ ((row >> 1) & 7), m = m0 + row;
            o1[j] = o2[j] = OOB;
            if (m < a.M) {
                o1[j] = (unsigned)(((long long)m * a.in_cs + a.in_co + col * 8) * 2);
                if (a.nk > a.nk1) {
                    const int b = m / a.HoWo, rem = m - b * a.HoWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
                    o2[j] = (unsigned)(((((long long)b * a.H2 + oy * a.stride2) * a.W2 + ox * a.stride2) * a.in_cs2 + a.in_co2 + col * 8) * 2);
                }
            }
        }
        const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)s_raw;
        auto issue = [&](int c) {                              // chunk c (past the end: all lanes out of range -> zeros nobody reads) into buffer c % PD
            const bool live = c < a.nk, first = c < a.nk1;
            const unsigned soff = (unsigned)((first ? c : c - a.nk1) * SKC * 2);
            const unsigned dst = lds0 + (unsigned)(c % PD) * CH;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned vo = !live ? OOB : first ? o1[j] : o2[j];
                if (first) convk::lds_dma16_m0(d1, dst + j * 1024, vo, soff);
                else convk::lds_dma16_m0(d2, dst + j * 1024, vo, soff);
            }
        };
#pragma unroll
        for (int c = 0; c < PD - 1; ++c) issue(c);
        for (int c = 0; c < a.nk; ++c) {
            convk::wait_vmcnt<(PD - 2) * 16>();                 // chunk c has landed: only the PD - 2 chunks behind it may still be in flight
            __syncthreads();                                     // consumers: chunk c is there; producer: they are done with chunk c - 1
            issue(c + PD - 1);                                   // -> buffer (c - 1) % PD
        }
        convk::wait_vmcnt<0>();                                  // the trailing (out-of-range) transfers land before the ring becomes the output stage
        __syncthreads();
        for (int half = 0; half < NCB; ++half) {                 // (the output stage's barriers)
            __syncthreads();
            if (half + 1 < NCB) __syncthreads();
        }
        return;
    }

    // ---- consumers: four waves, 32 * NCB output channels each, weights as MFMA A operands straight from the packed stream (L2), one chunk ahead
    const uint4* wbase = a.w + ((long long)(nchunk * 4 + wave) * a.nk) * (4 * NCB * 64) + lane;
    auto w_load = [&](int c, uint4 (&wr)[4][NCB]) {
        const uint4* p = wbase + (long long)min(c, a.nk - 1) * (4 * NCB * 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) wr[ks][cb] = p[(ks * NCB + cb) * 64];
    };
    f32x16 acc[NCB][NPB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;
    uint4 wr[2][4][NCB];
    w_load(0, wr[0]);
    auto iteration = [&](auto Par, int c) {
        constexpr int par = decltype(Par)::value;
        w_load(c + 1, wr[par ^ 1]);
        __syncthreads();                                         // chunk c is in buffer c % PD
        const char* sa = s_raw + (c % PD) * CH;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 bv[NPB];
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                const int row = 32 * pb + l32;
                bv[pb] = *reinterpret_cast<const uint4*>(sa + row * 128 + (((ks + 4 * h) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) acc[cb][pb] = Half<H>::mfma32(wr[par][ks][cb], bv[pb], acc[cb][pb]);
        }
    };
    int c = 0;
    for (; c + 1 < a.nk; c += 2) {
        iteration(std::integral_constant<int, 0>{}, c);
        iteration(std::integral_constant<int, 1>{}, c + 1);
    }
    if (c < a.nk) {                                              // odd chunk count: the last chunk's weights are in wr[0]
        iteration(std::integral_constant<int, 0>{}, c);
    }
    __syncthreads();                                             // every consumer is done with the ring (and the producer's last transfers have landed)

    // ---- epilogue: as in the kernel above (the tile leaves through LDS, 128 channels at a time, coalesced 16-byte row segments out)
    const bool relu = a.relu != 0;
#pragma unroll
    for (int half = 0; half < NCB; ++half) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int blk = wave * NCB + cb;
            if ((blk >> 2) != half) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = blk * 32 + 8 * q + 4 * h, n = nchunk * NWG + nl;
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + n);
                if (a.shift) sh = *reinterpret_cast<const float4*>(a.shift + n);
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) {
                    uint2 o;
                    const float v0 = fmaf(acc[cb][pb][4 * q], sc.x, sh.x), v1 = fmaf(acc[cb][pb][4 * q + 1], sc.y, sh.y);
                    const float v2 = fmaf(acc[cb][pb][4 * q + 2], sc.z, sh.z), v3 = fmaf(acc[cb][pb][4 * q + 3], sc.w, sh.w);
                    if (relu) { o.x = Half<H>::pack2_relu(v0, v1); o.y = Half<H>::pack2_relu(v2, v3); }
                    else { o.x = Half<H>::pack2(v0, v1); o.y = Half<H>::pack2(v2, v3); }
                    *reinterpret_cast<uint2*>(s_raw + (32 * pb + l32) * OPITCH + (nl - 128 * half) * 2) = o;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SBM * 16 / 256; ++i) {
            const int cc_ = tid + 256 * i, row = cc_ >> 4, cc = cc_ & 15;
            const int m = m0 + row;
            if (m < a.M)
                *reinterpret_cast<uint4*>(a.y + (long long)m * a.out_cs + a.out_co + nchunk * NWG + 128 * half + cc * 8) =
                    *reinterpret_cast<const uint4*>(s_raw + row * OPITCH + cc * 16);
        }
        if (half + 1 < NCB) __syncthreads();
    }
}

}  // namespace
}  // namespace dir

extern "C" int dir_conv1x1_stream_forward(const dir_conv_desc* d, const void* x, const dir_conv_src2* d2, const void* x2,
                                          const void* w_stream, const float* scale, const float* shift, const float* pre_scale,
                                          const float* pre_shift, void* y, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && x && w_stream && y, "dir_conv1x1_stream_forward: null pointer");
    DIR_REQUIRE(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0, "dir_conv1x1_stream_forward: 1x1 stride-1 convolutions only");
    DIR_REQUIRE((d->in_dtype == DIR_DT_BF16 || d->in_dtype == DIR_DT_F16) && d->out_dtype == d->in_dtype, "dir_conv1x1_stream_forward: bf16 -> bf16 or f16 -> f16 only");
    const bool f16 = d->in_dtype == DIR_DT_F16;
    DIR_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % SKC == 0 && d->Cout > 0 && d->Cout % 128 == 0,
                "dir_conv1x1_stream_forward: Cin must be a multiple of 64, Cout of 128");
    DIR_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr) && (!pre_scale || d->Cin <= PRE_MAX), "dir_conv1x1_stream_forward: bad pre-activation");
    StreamArgs a;
    a.x = (const convk::bf16_t*)x; a.x2 = (const convk::bf16_t*)x2; a.y = (convk::bf16_t*)y; a.w = (const uint4*)w_stream;
    a.scale = scale; a.shift = shift; a.pre_scale = pre_scale; a.pre_shift = pre_shift;
    const long long M = (long long)d->B * d->H * d->W;
    DIR_REQUIRE(M < (1ll << 31), "dir_conv1x1_stream_forward: too many pixels");
    DIR_REQUIRE(M * (d->in_cstride ? d->in_cstride : d->Cin) < (1ll << 31) && (!d2 || (long long)d->B * d2->H * d2->W * (d2->in_cstride ? d2->in_cstride : d2->Cin) < (1ll << 31)),
                "dir_conv1x1_stream_forward: sources must have < 2^31 elements");
    a.M = (int)M; a.Cin = d->Cin; a.in_cs = d->in_cstride ? d->in_cstride : d->Cin; a.in_co = d->in_coff;
    a.Cout = d->Cout; a.out_cs = d->out_cstride ? d->out_cstride : d->Cout; a.out_co = d->out_coff;
    DIR_REQUIRE(a.in_cs % 8 == 0 && a.in_co % 8 == 0 && a.out_cs % 8 == 0 && a.out_co % 8 == 0, "dir_conv1x1_stream_forward: channel slices must be 16-byte aligned");
    a.nk1 = d->Cin / SKC; a.nk = a.nk1; a.relu = (d->flags & DIR_CONV_RELU) != 0; a.pre_relu = (d->flags & DIR_CONV_PRE_RELU) != 0;
    a.HoWo = d->H * d->W; a.Wo = d->W; a.H2 = a.W2 = a.in_cs2 = a.in_co2 = 0; a.stride2 = 1;
    if (d2) {
        DIR_REQUIRE(x2 && d2->Cin > 0 && d2->Cin % SKC == 0 && d2->stride > 0, "dir_conv1x1_stream_forward: bad second source");
        DIR_REQUIRE((d2->H - 1) / d2->stride + 1 == d->H && (d2->W - 1) / d2->stride + 1 == d->W,
                    "dir_conv1x1_stream_forward: the second source does not produce a %dx%d output", d->H, d->W);
        a.H2 = d2->H; a.W2 = d2->W; a.in_cs2 = d2->in_cstride ? d2->in_cstride : d2->Cin; a.in_co2 = d2->in_coff; a.stride2 = d2->stride;
        DIR_REQUIRE(a.in_cs2 % 8 == 0 && a.in_co2 % 8 == 0, "dir_conv1x1_stream_forward: second source channel slice must be 16-byte aligned");
        a.nk += d2->Cin / SKC;
    }
    hipStream_t s = (hipStream_t)stream;
    // DIR_CONV_VARIANT 22 / 23 in d->flags: 64- / 32-pixel workgroups (same sums, same bits); anything else: 128-pixel workgroups
    const int var = (d->flags >> 8) & 0xff;                 // 22: 64-pixel workgroups, 23: 32-pixel workgroups
    const int sbm = var == 22 ? 64 : var == 23 ? 32 : 128;
    const int tiles = (a.M + sbm - 1) / sbm;
#define DIR_STREAM_H(NCB_, PRE_, H_) do { if (var == 22) DIR_LAUNCH((stream1x1_kernel<NCB_, PRE_, 2, H_>), grid, dim3(STHR), 0, s, a); \
                                    else if (var == 23) DIR_LAUNCH((stream1x1_kernel<NCB_, PRE_, 1, H_>), grid, dim3(STHR), 0, s, a); \
                                    else DIR_LAUNCH((stream1x1_kernel<NCB_, PRE_, 4, H_>), grid, dim3(STHR), 0, s, a); } while (0)
#define DIR_STREAM(NCB_, PRE_) do { if (f16) DIR_STREAM_H(NCB_, PRE_, convk::f16s_t); else DIR_STREAM_H(NCB_, PRE_, convk::bf16_t); } while (0)
    // DIR_CONV_VARIANT 24: the pipelined form (a fifth wave feeds the activation ring by DMA); no pre-activation; sources < 2 GiB (32-bit buffer offsets)
    const long long xb = M * a.in_cs * 2, x2b = d2 ? (long long)d->B * d2->H * d2->W * a.in_cs2 * 2 : 0;
    if (var == 24 && !pre_scale && xb < (1ll << 31) && x2b < (1ll << 31)) {
        const dim3 gridp((a.M + 127) / 128, d->Cout / (d->Cout % 256 == 0 ? 256 : 128));
        if (d->Cout % 256 == 0) {
            if (f16) DIR_LAUNCH((stream1x1p_kernel<2, convk::f16s_t>), gridp, dim3(320), 0, s, a, (unsigned)xb, (unsigned)x2b);
            else DIR_LAUNCH((stream1x1p_kernel<2, convk::bf16_t>), gridp, dim3(320), 0, s, a, (unsigned)xb, (unsigned)x2b);
        } else {
            if (f16) DIR_LAUNCH((stream1x1p_kernel<1, convk::f16s_t>), gridp, dim3(320), 0, s, a, (unsigned)xb, (unsigned)x2b);
            else DIR_LAUNCH((stream1x1p_kernel<1, convk::bf16_t>), gridp, dim3(320), 0, s, a, (unsigned)xb, (unsigned)x2b);
        }
        return check_launch("dir_conv1x1_stream_forward");
    }
    if (d->Cout % 256 == 0) {
        dim3 grid(tiles, d->Cout / 256);
        if (pre_scale) DIR_STREAM(2, true); else DIR_STREAM(2, false);
    } else {
        dim3 grid(tiles, d->Cout / 128);
        if (pre_scale) DIR_STREAM(1, true); else DIR_STREAM(1, false);
    }
#undef DIR_STREAM
#undef DIR_STREAM_H
    return check_launch("dir_conv1x1_stream_forward");
}
