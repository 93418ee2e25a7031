// dir_bottleneck_tail_forward: the 1x1 tail of one ResNet bottleneck and the 1x1 head of the next in ONE kernel (bf16 mode), for the
// layer2 / layer3 geometry (planes P = 128 / 256, block width C4 = 4 P = 512 / 1024):
//   models/backbone/resnet.py:132-140   block i  : conv3 1x1 (P -> 4P) -> bn3 -> += identity -> ReLU            -> out   (HBM)
//   models/backbone/resnet.py:122-124   block i+1: conv1 1x1 (4P -> N2) -> bn1 -> ReLU                           -> y1n   (HBM)
// Unfused, the block output (the widest tensor of the layer: 67 MB at 32x32x512, B = 64) is written by conv3 and read back by the
// next conv1; both launches stream (K <= 1024), so only bytes count: y2 + identity + out + (out again) + y1n = 235 MB per block at
// layer2 against 168 MB here (layer3: 117 -> 84 MB), and one launch (ramp, first-load latency, drain) instead of two -- at layer3's
// M = 16 384 the two launches are latency-bound at 2.1 - 2.6 TB/s.
//
// One persistent 8-wave workgroup per CU walks 64-pixel tiles (1x1 convolutions: a tile is any 64 consecutive NHW pixels).  Both
// GEMMs are computed as D[channel][pixel] (weights = MFMA A operand): a lane holds 4 consecutive channels of a pixel -- what the
// bf16 NHWC stores, the residual loads and the LDS hand-over between the GEMMs want (bneck.hip).  The weights do not fit on the
// CU (w3 + w1n = 256 KB at layer2, 1 MB at layer3): every wave STREAMS its own slice from L2 into a register ring, fragment by
// fragment, in exactly the order its MFMAs consume them -- the host packs them in that order (dir_amd/engine.py::pack_tail_stream),
// so a fragment is one coalesced 1 KB load per wave and no weight byte goes through LDS.  The block output is processed in
// 512-channel halves (NH = C4 / 512) so that the T tile ([64 px][512 ch] bf16, 65 KB) fits in LDS at layer3 too; the next conv1
// accumulates over the halves in registers.  Per unit (tile, half):
//   B. conv3:  y2 tile (LDS) x this wave's 64 output channels (stream), bn3 + residual + ReLU -> T (LDS, bf16: the rounding point
//      of the unfused path); the residual sits in registers since the previous unit's epilogue (a unit of lead time)
//   -  T -> HBM with coalesced 16-byte stores (block output)
//   C. next conv1: T (LDS) x this wave's N2 / 8 output channels (stream), accumulated over the halves; after the last half
//      bn1 + ReLU -> y1n (HBM)
// The next tile's y2 rows are requested at the top of a tile and parked in LDS at its end.
#include "conv_common.h"

namespace dir {
namespace {

using convk::bf16_t;
using convk::f16s_t;
using convk::f32x16;
using convk::Half;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// The activations are touched once per launch; the weight stream is re-read by every workgroup.  DIR_TAIL_NT (compile-time A/B
// switch) marks the residual loads and block-output stores non-temporal so that they do not push the weights out of L2.
// MEASURED AND REJECTED (r02): 46.6 -> 71.0 us at layer2, 34.7 -> 44.6 us at layer3 -- the tensors a layer reads were written by the
// previous launch and sit in the 256 MB Infinity Cache; a non-temporal access gives that up.  Off.
#ifndef DIR_TAIL_RES16
#define DIR_TAIL_RES16 1
#endif
#ifndef DIR_TAIL_NT
#define DIR_TAIL_NT 0
#endif
#if DIR_TAIL_NT
typedef unsigned __attribute__((ext_vector_type(2))) u32x2_t;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;
__device__ __forceinline__ uint2 nt_load2(const uint2* p) { const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p)); return make_uint2(v.x, v.y); }
__device__ __forceinline__ void nt_store4(uint4 v, uint4* p) { __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p)); }
#define NT_LOAD(p) nt_load2(p)
#define NT_STORE(v, p) nt_store4(v, p)
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif

constexpr int TM = 64;                 // pixels per tile
constexpr int HC = 512;                // channels per half of the block output
constexpr int NTHR = 512;
constexpr int TPITCH = HC * 2 + 16;    // bytes per T pixel

struct TailArgs {
    const bf16_t* y2; const bf16_t* res; bf16_t* out; bf16_t* y1n;
    const uint4* wstream;              // [NH][8 waves][NBF + NCF fragments][64 lanes] x 16 bytes
    const float* sc3; const float* sh3; const float* sc1n; const float* sh1n;
    int ntiles; unsigned y2_bytes;
};

template <typename H> __device__ __forceinline__ void unpack4(uint2 v, float (&f)[4]) {
    convk::unpack2<H>(v.x, f[0], f[1]);
    convk::unpack2<H>(v.y, f[2], f[3]);
}

// P: planes (K of conv3); N2: output channels of the next conv1 (128: 16 per wave on v_mfma_f32_16x16x32_bf16; 256: 32 per wave
// on v_mfma_f32_32x32x16_bf16)
// H: the 16-bit storage kind (bf16_t | f16s_t = DIR_DT_BF16 | DIR_DT_F16): tensors, weight stream and the T tile
template <int P, int N2, typename H = bf16_t>
__global__ __launch_bounds__(NTHR, 1) void tail_chain_kernel(TailArgs a) {
    convk::half_kernel_init<H>();
    constexpr int C4 = 4 * P, NH = C4 / HC;
    constexpr int YROW = P * 2;                         // bytes per y2 pixel (unpadded: 16-byte chunk c of row r sits at c ^ (r & 15))
    constexpr int GRP = P == 256 ? 4 : 8;               // weight fragments per ring group (1 KB each per wave): register budget
    constexpr int KBS = P / 16;                         // k-steps of conv3
    constexpr int NBF = 2 * KBS;                        // phase-B fragments per unit: (channel block of 32, k-step), channel-block-major
    constexpr bool C16 = N2 == 128;                     // phase C on 16x16x32 (16 channels per wave) or 32x32x16 (32 per wave)
    constexpr int NCF = C16 ? HC / 32 : HC / 16;        // phase-C fragments per unit (K = 512 per half)
    constexpr int NF = NBF + NCF, NG = NF / GRP;        // fragments / ring groups per unit
    static_assert(NF % GRP == 0 && C4 % HC == 0 && (N2 == 128 || N2 == 256), "tail_chain_kernel: unsupported geometry");
    constexpr int YCH = TM * (P / 8) / NTHR;            // 16-byte y2 chunks per thread (2 | 4) = 1 KB LDS-DMA pieces per wave
    __shared__ __attribute__((aligned(16))) char s_t[TM * TPITCH];
    __shared__ __attribute__((aligned(16))) char s_y2[2][TM * YROW];
    __shared__ __attribute__((aligned(16))) float s_ss[2 * C4 + 2 * N2];     // sc3 | sh3 | sc1n | sh1n
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5, l16 = lane & 15, g16 = lane >> 4;

    for (int i = tid; i < C4; i += NTHR) { s_ss[i] = a.sc3[i]; s_ss[C4 + i] = a.sh3[i]; }
    for (int i = tid; i < N2; i += NTHR) { s_ss[2 * C4 + i] = a.sc1n[i]; s_ss[2 * C4 + N2 + i] = a.sh1n[i]; }

    const int tstep = gridDim.x;
    int t = blockIdx.x;
    if (t >= a.ntiles) return;

    // ---- y2 rows of a tile: global -> LDS by DMA (no registers: the prefetch of the next tile is in flight for a whole tile).  Piece i of
    //      wave w fills LDS bytes [(8 i + w) KB, +1 KB) of the buffer: lane l lands on 16-byte slot s = (8 i + w) * 64 + l = (pixel
    //      s / (P/8), position s % (P/8)) and fetches the chunk that belongs there under the XOR swizzle (conflict-free ds_read_b128)
    const convk::i32x4 yd = {(int)(unsigned)(unsigned long long)a.y2, (int)(unsigned)((unsigned long long)a.y2 >> 32), (int)a.y2_bytes, 0x00020000};
    const unsigned y2_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)&s_y2[0][0];
    auto y2_dma = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < YCH; ++i) {
            const int slot = (8 * i + wave) * 64 + lane, px = slot / (P / 8), pos = slot % (P / 8);
            const unsigned voff = (unsigned)(((long long)tile * TM + px) * YROW) + (unsigned)((pos ^ (px & 15)) * 16);
            convk::lds_dma16_m0(yd, y2_lds + buf * (TM * YROW) + (8 * i + wave) * 1024, voff, 0);
        }
    };
    // ---- residual of a unit in the epilogue-B layout: pixel 32 pb + l32, channels half*512 + 64 wave + 32 cb + 8 q + 4 h .. +4
#if DIR_TAIL_RES16
    // fetched as 16-byte chunks -- lane half h takes channels 16 j + 8 h .. + 8 of each 32-channel block -- and turned into the accumulator
    // layout at the point of use by swapping 8 bytes per chunk with the partner lane (v_permlane32_swap): half the load instructions,
    // 32 contiguous bytes per pixel row and instruction instead of 16
    uint4 xr[2][2][2];
    auto res_fetch = [&](int tile, int half) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const bf16_t* rp = a.res + ((long long)tile * TM + 32 * pb + l32) * C4 + half * HC + 64 * wave + 8 * h;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int j = 0; j < 2; ++j) xr[cb][pb][j] = *reinterpret_cast<const uint4*>(rp + 32 * cb + 16 * j);
        }
#else
    uint2 xr[2][2][4];
    auto res_fetch = [&](int tile, int half) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const bf16_t* rp = a.res + ((long long)tile * TM + 32 * pb + l32) * C4 + half * HC + 64 * wave + 4 * h;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) xr[cb][pb][q] = NT_LOAD(reinterpret_cast<const uint2*>(rp + 32 * cb + 8 * q));
        }
#endif
    };
    // ---- weight stream of this wave: fragment f of half hf at wstream[((hf * 8 + wave) * NF + f) * 64 + lane]
    uint4 ring[2][GRP];
    auto ring_load = [&](auto Slot, int hf, int grp) {
        constexpr int slot = decltype(Slot)::value;
        const uint4* p = a.wstream + ((long long)(hf * 8 + wave) * NF + grp * GRP) * 64 + lane;
#pragma unroll
        for (int i = 0; i < GRP; ++i) ring[slot][i] = p[i * 64];
    };

    y2_dma(t, 0);
    ring_load(std::integral_constant<int, 0>{}, 0, 0);
    res_fetch(t, 0);
    convk::wait_vmcnt<0>();
    __syncthreads();
    int ybuf = 0;

    f32x16 accb[2];                                      // conv3: [pixel block] of the current channel block
    f32x16 accc32[C16 ? 1 : 2];                          // next conv1, 32x32 path: [pixel block]
    f32x4 accc16[C16 ? 4 : 1];                           // next conv1, 16x16 path: [16-pixel group]
    // activation (MFMA B) operands of a fragment step, read from LDS one step ahead of the MFMAs that use them
    constexpr int NOP = C16 ? 4 : 2;
    uint4 bop[2][NOP];
    const char* ycur = s_y2[0];
    auto act_read = [&](auto F, auto Set) {
        constexpr int f = decltype(F)::value, set = decltype(Set)::value;
        if constexpr (f < NBF) {                         // conv3: y2[pixel 32 pb + l32][16 ks + 8 h ..]
            constexpr int ks = f % KBS;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
                bop[set][pb] = *reinterpret_cast<const uint4*>(ycur + (32 * pb + l32) * YROW + (((2 * ks + h) ^ (l32 & 15)) << 4));
        } else {
            constexpr int fc = f - NBF;
            if constexpr (C16) {                         // T[pixel 16 u + l16][32 fc + 8 g ..]
#pragma unroll
                for (int u = 0; u < 4; ++u) bop[set][u] = *reinterpret_cast<const uint4*>(s_t + (16 * u + l16) * TPITCH + 64 * fc + 16 * g16);
            } else {                                     // T[pixel 32 pb + l32][16 fc + 8 h ..]
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) bop[set][pb] = *reinterpret_cast<const uint4*>(s_t + (32 * pb + l32) * TPITCH + 32 * fc + 16 * h);
            }
        }
    };

    for (; t < a.ntiles; t += tstep) {
        const bool more = t + tstep < a.ntiles;
        if (more) y2_dma(t + tstep, ybuf ^ 1);           // lands during this tile; published by the tile's last barrier
        ycur = s_y2[ybuf];
#pragma nounroll
        for (int hf = 0; hf < NH; ++hf) {
            const bool last_half = hf == NH - 1;
            if (hf == 0) {
                if constexpr (C16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) accc16[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accc32[pb][r] = 0.f;
                }
            }
            act_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            // the unit's fragments, group by group; the next group (of this unit, or group 0 of the next unit) is requested first
            [&]<int... G>(std::integer_sequence<int, G...>) {
                (([&] {
                     constexpr int grp = G, slot = G & 1;
                     if constexpr (grp + 1 < NG) ring_load(std::integral_constant<int, slot ^ 1>{}, hf, grp + 1);
                     else ring_load(std::integral_constant<int, slot ^ 1>{}, last_half ? 0 : hf + 1, 0);     // NG is even: slot 0 again
                     [&]<int... I>(std::integer_sequence<int, I...>) {
                         (([&] {
                              constexpr int f = grp * GRP + I, set = f & 1;
                              // operands of the next step (not across the phase boundary: T does not exist yet; not across the unit)
                              if constexpr (f + 1 < NF && f + 1 != NBF) act_read(std::integral_constant<int, f + 1>{}, std::integral_constant<int, set ^ 1>{});
                              if constexpr (f < NBF) {
                                  // ---- B. conv3: fragment (channel block cb, k-step ks), channel-block-major: two accumulators live
                                  constexpr int cb = f / KBS, ks = f - cb * KBS;
                                  if constexpr (ks == 0) {
#pragma unroll
                                      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                                          for (int r = 0; r < 16; ++r) accb[pb][r] = 0.f;
                                  }
#pragma unroll
                                  for (int pb = 0; pb < 2; ++pb)
                                      accb[pb] = Half<H>::mfma32(ring[slot][I], bop[set][pb], accb[pb]);
                                  if constexpr (ks == KBS - 1) {
                                      // ---- epilogue B of this channel block -> T: bn3 + residual + ReLU, bf16
#pragma unroll
                                      for (int q = 0; q < 4; ++q) {
                                          const int cl = 64 * wave + 32 * cb + 8 * q + 4 * h;                  // channel inside the half
                                          const float4 sc = *reinterpret_cast<const float4*>(s_ss + hf * HC + cl);
                                          const float4 sh = *reinterpret_cast<const float4*>(s_ss + C4 + hf * HC + cl);
#pragma unroll
                                          for (int pb = 0; pb < 2; ++pb) {
                                              float v[4] = {fmaf(accb[pb][4 * q], sc.x, sh.x), fmaf(accb[pb][4 * q + 1], sc.y, sh.y),
                                                            fmaf(accb[pb][4 * q + 2], sc.z, sh.z), fmaf(accb[pb][4 * q + 3], sc.w, sh.w)};
                                              float rv[4];
#if DIR_TAIL_RES16
                                              {   // (q even, q odd) = swap(lower 8 bytes, upper 8 bytes) of chunk q / 2, dword by dword
                                                  const uint4 c = xr[cb][pb][q >> 1];
                                                  const auto sx = __builtin_amdgcn_permlane32_swap(c.x, c.z, false, false);
                                                  const auto sy = __builtin_amdgcn_permlane32_swap(c.y, c.w, false, false);
                                                  unpack4<H>(make_uint2(sx[q & 1], sy[q & 1]), rv);
                                              }
#else
                                              unpack4<H>(xr[cb][pb][q], rv);
#endif
#pragma unroll
                                              for (int e = 0; e < 4; ++e) v[e] += rv[e];
                                              uint2 o;
                                              o.x = Half<H>::pack2_relu(v[0], v[1]);
                                              o.y = Half<H>::pack2_relu(v[2], v[3]);
                                              *reinterpret_cast<uint2*>(s_t + (32 * pb + l32) * TPITCH + cl * 2) = o;
                                          }
                                          __builtin_amdgcn_sched_barrier(0);          // keep the (sc, sh) reads of later q's out of this one's registers
                                      }
                                  }
                                  if constexpr (f == NBF - 1) {
                                      __syncthreads();                               // T = this half of the block output; every wave is done with y2
                                      act_read(std::integral_constant<int, NBF>{}, std::integral_constant<int, set ^ 1>{});
                                      // residual of the NEXT unit into the registers just consumed (a unit of lead time)
                                      if (!last_half) res_fetch(t, hf + 1);
                                      else if (more) res_fetch(t + tstep, 0);
                                      // block output: coalesced 16-byte stores from T (chunk c = pixel c >> 6, 16-byte chunk c & 63)
#pragma unroll
                                      for (int i = 0; i < TM * (HC / 8) / NTHR; ++i) {
                                          const int c = tid + NTHR * i;
                                          NT_STORE(*reinterpret_cast<const uint4*>(s_t + (c >> 6) * TPITCH + (c & 63) * 16),
                                                   reinterpret_cast<uint4*>(a.out + ((long long)t * TM + (c >> 6)) * C4 + hf * HC + (c & 63) * 8));
                                          if (i & 1) __builtin_amdgcn_sched_barrier(0);
                                      }
                                  }
                              } else {
                                  // ---- C. next conv1 over this half's 512 input channels
                                  if constexpr (C16) {
#pragma unroll
                                      for (int u = 0; u < 4; ++u)
                                          accc16[u] = Half<H>::mfma16(ring[slot][I], bop[set][u], accc16[u]);
                                  } else {
#pragma unroll
                                      for (int pb = 0; pb < 2; ++pb)
                                          accc32[pb] = Half<H>::mfma32(ring[slot][I], bop[set][pb], accc32[pb]);
                                  }
                              }
                              // the next tile's y2 DMA is the oldest thing this wave can still have in flight: everything but the ring
                              // group requested at the top of this (last) group has to be back before the closing barrier publishes it
                              if constexpr (f == NF - 1) convk::wait_vmcnt<GRP>();
                              __builtin_amdgcn_sched_barrier(0);
                          }()),
                          ...);
                     }(std::make_integer_sequence<int, GRP>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, NG>{});
            if (last_half) {
                // ---- epilogue C: bn1 + ReLU -> next y1
                if constexpr (C16) {
                    const int c0 = 16 * wave + 4 * g16;
                    const float4 sc = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + c0);
                    const float4 sh = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + N2 + c0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        uint2 o;
                        o.x = Half<H>::pack2_relu(fmaf(accc16[u][0], sc.x, sh.x), fmaf(accc16[u][1], sc.y, sh.y));
                        o.y = Half<H>::pack2_relu(fmaf(accc16[u][2], sc.z, sh.z), fmaf(accc16[u][3], sc.w, sh.w));
                        *reinterpret_cast<uint2*>(a.y1n + ((long long)t * TM + 16 * u + l16) * N2 + c0) = o;
                    }
                } else {
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        uint2 o[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c0 = 32 * wave + 8 * q + 4 * h;
                            const float4 sc = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + c0);
                            const float4 sh = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + N2 + c0);
                            o[q].x = Half<H>::pack2_relu(fmaf(accc32[pb][4 * q], sc.x, sh.x), fmaf(accc32[pb][4 * q + 1], sc.y, sh.y));
                            o[q].y = Half<H>::pack2_relu(fmaf(accc32[pb][4 * q + 2], sc.z, sh.z), fmaf(accc32[pb][4 * q + 3], sc.w, sh.w));
                        }
#if DIR_TAIL_RES16
                        // the residual fetch's swap in reverse: lane half h ends up with channels 16 j + 8 h .. + 8 -- one 16-byte store per j
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const auto dx = __builtin_amdgcn_permlane32_swap(o[2 * j].x, o[2 * j + 1].x, false, false);
                            const auto dy = __builtin_amdgcn_permlane32_swap(o[2 * j].y, o[2 * j + 1].y, false, false);
                            *reinterpret_cast<uint4*>(a.y1n + ((long long)t * TM + 32 * pb + l32) * N2 + 32 * wave + 16 * j + 8 * h) = make_uint4(dx[0], dy[0], dx[1], dy[1]);
                        }
#else
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<uint2*>(a.y1n + ((long long)t * TM + 32 * pb + l32) * N2 + 32 * wave + 8 * q + 4 * h) = o[q];
#endif
                    }
                }
            }
            __syncthreads();                             // T is free for the next unit; the next tile's y2 rows are visible
        }
        ybuf ^= 1;
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Thin variant (layer3 geometry at moderate batch: M = 16 384 pixels = one 64-pixel tile per CU, so the persistent fat workgroup
// above has nothing to overlap its own latencies with): 4 waves, 32-pixel tiles, <= 76 KB of LDS -- TWO workgroups per CU, each
// in a different phase.  Same phases and arithmetic; a wave owns 128 conv3 channels per half (4 blocks, one accumulator live at a
// time) and N2 / 4 channels of the next conv1 on v_mfma_f32_32x32x16_bf16.  The weight stream is packed for 4 waves
// (dir_amd/engine.py::pack_tail_stream(..., waves=4)).
constexpr int TTM = 32, TTHR = 256;

template <int P, int N2, typename H = bf16_t>
__global__ __launch_bounds__(TTHR, 2) void tail_thin_kernel(TailArgs a) {
    convk::half_kernel_init<H>();
    constexpr int C4 = 4 * P, NH = C4 / HC;
    constexpr int YROW = P * 2;
    constexpr int GRP = 8;
    constexpr int KBS = P / 16;
    constexpr int NBF = 4 * KBS;                        // (channel block of 32, k-step), channel-block-major
    constexpr int NCC = N2 / 128;                       // 32-channel blocks of the next conv1 per wave
    constexpr int NCF = 32 * NCC;                       // (k-step, channel block), k-step-major
    constexpr int NF = NBF + NCF, NG = NF / GRP;
    static_assert(NF % GRP == 0 && NG % 2 == 0 && C4 % HC == 0 && (N2 == 128 || N2 == 256), "tail_thin_kernel: unsupported geometry");
    constexpr int YCH = TTM * (P / 8) / TTHR;           // 1 KB LDS-DMA pieces per wave (2 | 4)
    __shared__ __attribute__((aligned(16))) char s_t[TTM * TPITCH];
    __shared__ __attribute__((aligned(16))) char s_y2[2][TTM * YROW];
    __shared__ __attribute__((aligned(16))) float s_ss[2 * C4 + 2 * N2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;

    for (int i = tid; i < C4; i += TTHR) { s_ss[i] = a.sc3[i]; s_ss[C4 + i] = a.sh3[i]; }
    for (int i = tid; i < N2; i += TTHR) { s_ss[2 * C4 + i] = a.sc1n[i]; s_ss[2 * C4 + N2 + i] = a.sh1n[i]; }

    const int tstep = gridDim.x;
    int t = blockIdx.x;
    if (t >= a.ntiles) return;

    const convk::i32x4 yd = {(int)(unsigned)(unsigned long long)a.y2, (int)(unsigned)((unsigned long long)a.y2 >> 32), (int)a.y2_bytes, 0x00020000};
    const unsigned y2_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)&s_y2[0][0];
    auto y2_dma = [&](int tile, int buf) {
#pragma unroll
        for (int i = 0; i < YCH; ++i) {
            const int slot = (4 * i + wave) * 64 + lane, px = slot / (P / 8), pos = slot % (P / 8);
            const unsigned voff = (unsigned)(((long long)tile * TTM + px) * YROW) + (unsigned)((pos ^ (px & 15)) * 16);
            convk::lds_dma16_m0(yd, y2_lds + buf * (TTM * YROW) + (4 * i + wave) * 1024, voff, 0);
        }
    };
    // residual of a unit: pixel l32, channels half*512 + 128 wave + 32 cb + 8 q + 4 h .. +4
    uint2 xr[4][4];
    auto res_fetch = [&](int tile, int half) {
        const bf16_t* rp = a.res + ((long long)tile * TTM + l32) * C4 + half * HC + 128 * wave + 4 * h;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[cb][q] = NT_LOAD(reinterpret_cast<const uint2*>(rp + 32 * cb + 8 * q));
    };
    uint4 ring[2][GRP];
    auto ring_load = [&](auto Slot, int hf, int grp) {
        constexpr int slot = decltype(Slot)::value;
        const uint4* p = a.wstream + ((long long)(hf * 4 + wave) * NF + grp * GRP) * 64 + lane;
#pragma unroll
        for (int i = 0; i < GRP; ++i) ring[slot][i] = p[i * 64];
    };

    y2_dma(t, 0);
    ring_load(std::integral_constant<int, 0>{}, 0, 0);
    res_fetch(t, 0);
    convk::wait_vmcnt<0>();
    __syncthreads();
    int ybuf = 0;

    f32x16 accb;
    f32x16 accc[NCC];
    uint4 bop[2];
    const char* ycur = s_y2[0];
    auto act_read = [&](auto F, auto Set) {
        constexpr int f = decltype(F)::value, set = decltype(Set)::value;
        if constexpr (f < NBF) {
            constexpr int ks = f % KBS;
            bop[set] = *reinterpret_cast<const uint4*>(ycur + l32 * YROW + (((2 * ks + h) ^ (l32 & 15)) << 4));
        } else {
            constexpr int ks = (f - NBF) / NCC;
            bop[set] = *reinterpret_cast<const uint4*>(s_t + l32 * TPITCH + 32 * ks + 16 * h);
        }
    };

    for (; t < a.ntiles; t += tstep) {
        const bool more = t + tstep < a.ntiles;
        if (more) y2_dma(t + tstep, ybuf ^ 1);
        ycur = s_y2[ybuf];
#pragma nounroll
        for (int hf = 0; hf < NH; ++hf) {
            const bool last_half = hf == NH - 1;
            if (hf == 0) {
#pragma unroll
                for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accc[cc][r] = 0.f;
            }
            act_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            [&]<int... G>(std::integer_sequence<int, G...>) {
                (([&] {
                     constexpr int grp = G, slot = G & 1;
                     if constexpr (grp + 1 < NG) ring_load(std::integral_constant<int, slot ^ 1>{}, hf, grp + 1);
                     else ring_load(std::integral_constant<int, slot ^ 1>{}, last_half ? 0 : hf + 1, 0);
                     [&]<int... I>(std::integer_sequence<int, I...>) {
                         (([&] {
                              constexpr int f = grp * GRP + I;
                              // operand of the next step, unless it is the same LDS vector (phase C: NCC fragments share a k-step) or
                              // would cross the phase boundary (T does not exist yet) / the unit
                              constexpr bool same_next = f >= NBF && f + 1 < NF && (f - NBF) / NCC == (f + 1 - NBF) / NCC;
                              constexpr int set = f < NBF ? (f & 1) : ((((f - NBF) / NCC) + NBF) & 1);
                              if constexpr (f + 1 < NF && f + 1 != NBF && !same_next)
                                  act_read(std::integral_constant<int, f + 1>{}, std::integral_constant<int, set ^ 1>{});
                              if constexpr (f < NBF) {
                                  constexpr int cb = f / KBS, ks = f - cb * KBS;
                                  if constexpr (ks == 0) {
#pragma unroll
                                      for (int r = 0; r < 16; ++r) accb[r] = 0.f;
                                  }
                                  accb = Half<H>::mfma32(ring[slot][I], bop[set], accb);
                                  if constexpr (ks == KBS - 1) {
#pragma unroll
                                      for (int q = 0; q < 4; ++q) {
                                          const int cl = 128 * wave + 32 * cb + 8 * q + 4 * h;
                                          const float4 sc = *reinterpret_cast<const float4*>(s_ss + hf * HC + cl);
                                          const float4 sh = *reinterpret_cast<const float4*>(s_ss + C4 + hf * HC + cl);
                                          float v[4] = {fmaf(accb[4 * q], sc.x, sh.x), fmaf(accb[4 * q + 1], sc.y, sh.y),
                                                        fmaf(accb[4 * q + 2], sc.z, sh.z), fmaf(accb[4 * q + 3], sc.w, sh.w)};
                                          float rv[4];
                                          unpack4<H>(xr[cb][q], rv);
#pragma unroll
                                          for (int e = 0; e < 4; ++e) v[e] += rv[e];
                                          uint2 o;
                                          o.x = Half<H>::pack2_relu(v[0], v[1]);
                                          o.y = Half<H>::pack2_relu(v[2], v[3]);
                                          *reinterpret_cast<uint2*>(s_t + l32 * TPITCH + cl * 2) = o;
                                      }
                                  }
                                  if constexpr (f == NBF - 1) {
                                      __syncthreads();
                                      act_read(std::integral_constant<int, NBF>{}, std::integral_constant<int, (NBF & 1)>{});
                                      if (!last_half) res_fetch(t, hf + 1);
                                      else if (more) res_fetch(t + tstep, 0);
#pragma unroll
                                      for (int i = 0; i < TTM * (HC / 8) / TTHR; ++i) {
                                          const int c = tid + TTHR * i;
                                          NT_STORE(*reinterpret_cast<const uint4*>(s_t + (c >> 6) * TPITCH + (c & 63) * 16),
                                                   reinterpret_cast<uint4*>(a.out + ((long long)t * TTM + (c >> 6)) * C4 + hf * HC + (c & 63) * 8));
                                          if (i & 1) __builtin_amdgcn_sched_barrier(0);
                                      }
                                  }
                              } else {
                                  constexpr int cc = (f - NBF) % NCC;
                                  accc[cc] = Half<H>::mfma32(ring[slot][I], bop[set], accc[cc]);
                              }
                              if constexpr (f == NF - 1) convk::wait_vmcnt<GRP>();
                              __builtin_amdgcn_sched_barrier(0);
                          }()),
                          ...);
                     }(std::make_integer_sequence<int, GRP>{});
                 }()),
                 ...);
            }(std::make_integer_sequence<int, NG>{});
            if (last_half) {
#pragma unroll
                for (int cc = 0; cc < NCC; ++cc)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c0 = (N2 / 4) * wave + 32 * cc + 8 * q + 4 * h;
                        const float4 sc = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + c0);
                        const float4 sh = *reinterpret_cast<const float4*>(s_ss + 2 * C4 + N2 + c0);
                        uint2 o;
                        o.x = Half<H>::pack2_relu(fmaf(accc[cc][4 * q], sc.x, sh.x), fmaf(accc[cc][4 * q + 1], sc.y, sh.y));
                        o.y = Half<H>::pack2_relu(fmaf(accc[cc][4 * q + 2], sc.z, sh.z), fmaf(accc[cc][4 * q + 3], sc.w, sh.w));
                        *reinterpret_cast<uint2*>(a.y1n + ((long long)t * TTM + l32) * N2 + c0) = o;
                    }
            }
            __syncthreads();
        }
        ybuf ^= 1;
    }
}

}  // namespace
}  // namespace dir

extern "C" int dir_bottleneck_tail_forward(const dir_bneck_tail_params* p, const void* y2, const void* residual, void* out, void* y1_next,
                                           long long M, void* stream) {
    using namespace dir;
    DIR_REQUIRE(p && y2 && residual && out && y1_next, "dir_bottleneck_tail_forward: null pointer");
    DIR_REQUIRE(p->wstream && p->scale3 && p->shift3 && p->scale1n && p->shift1n, "dir_bottleneck_tail_forward: missing parameters");
    DIR_REQUIRE(M > 0 && M % TM == 0 && M / TM < (1ll << 31), "dir_bottleneck_tail_forward: M must be a positive multiple of 64");
    DIR_REQUIRE(p->waves == 8 || p->waves == 4, "dir_bottleneck_tail_forward: the stream must be packed for 8 or 4 waves");
    DIR_REQUIRE(p->dtype == 0 || p->dtype == DIR_DT_BF16 || p->dtype == DIR_DT_F16, "dir_bottleneck_tail_forward: dtype must be bf16 (or 0) or f16");
    const bool f16 = p->dtype == DIR_DT_F16;
    TailArgs a;
    a.y2 = (const convk::bf16_t*)y2; a.res = (const convk::bf16_t*)residual; a.out = (convk::bf16_t*)out; a.y1n = (convk::bf16_t*)y1_next;
    a.wstream = (const uint4*)p->wstream; a.sc3 = p->scale3; a.sh3 = p->shift3; a.sc1n = p->scale1n; a.sh1n = p->shift1n;
    a.ntiles = (int)(M / TM);
    DIR_REQUIRE(M * p->planes * 2 < (1ll << 31), "dir_bottleneck_tail_forward: y2 must be < 2 GiB (32-bit buffer offsets)");
    a.y2_bytes = (unsigned)(M * p->planes * 2);
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0; hipDeviceProp_t pr;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    }
    hipStream_t s = (hipStream_t)stream;
    if (p->waves == 4) {               // thin variant: 32-pixel tiles, two workgroups per CU
        a.ntiles = (int)(M / TTM);
        const int grid4 = a.ntiles < 2 * num_cu ? a.ntiles : 2 * num_cu;
#define DIR_TAIL4(P_, N2_) do { if (f16) DIR_LAUNCH((tail_thin_kernel<P_, N2_, f16s_t>), dim3(grid4), dim3(TTHR), 0, s, a); \
                                 else DIR_LAUNCH((tail_thin_kernel<P_, N2_, bf16_t>), dim3(grid4), dim3(TTHR), 0, s, a); } while (0)
        if (p->planes == 128 && p->n_next == 128) DIR_TAIL4(128, 128);
        else if (p->planes == 128 && p->n_next == 256) DIR_TAIL4(128, 256);
        else if (p->planes == 256 && p->n_next == 256) DIR_TAIL4(256, 256);
        else DIR_REQUIRE(false, "dir_bottleneck_tail_forward: (planes, n_next) must be (128,128), (128,256) or (256,256)");
#undef DIR_TAIL4
        return check_launch("dir_bottleneck_tail_forward");
    }
    const int grid = a.ntiles < num_cu ? a.ntiles : num_cu;
#define DIR_TAIL(P_, N2_) do { if (f16) DIR_LAUNCH((tail_chain_kernel<P_, N2_, f16s_t>), dim3(grid), dim3(NTHR), 0, s, a); \
                                else DIR_LAUNCH((tail_chain_kernel<P_, N2_, bf16_t>), dim3(grid), dim3(NTHR), 0, s, a); } while (0)
    if (p->planes == 128 && p->n_next == 128) DIR_TAIL(128, 128);
    else if (p->planes == 128 && p->n_next == 256) DIR_TAIL(128, 256);
    else if (p->planes == 256 && p->n_next == 256) DIR_TAIL(256, 256);
    else DIR_REQUIRE(false, "dir_bottleneck_tail_forward: (planes, n_next) must be (128,128), (128,256) or (256,256)");
#undef DIR_TAIL
    return check_launch("dir_bottleneck_tail_forward");
}
