// dir_bottleneck_chain_forward: the HBM-bound tail of one ResNet bottleneck and the head of the next in ONE kernel (bf16 mode)
//   models/backbone/resnet.py:126-140   block i  : conv2 3x3 (64->64) -> bn2 -> ReLU -> conv3 1x1 (64->256) -> bn3 -> += identity -> ReLU
//   models/backbone/resnet.py:122-124   block i+1: conv1 1x1 (256->64) -> bn1 -> ReLU                                   (optional)
//   models/backbone/resnet.py:117-119   block i with a projection shortcut: downsample conv 1x1 (64->256) + BN as 64 more K of conv3's GEMM
// for the layer1 geometry (planes = 64, stride 1).  Unfused, block i moves y1, y2 (64 ch) twice each and the 256-channel map
// three times (residual read, output write, next conv1 read) = 534 MB at B = 64; these launches run at the practical HBM rate
// (4.0-4.8 TB/s), so only traffic counts.  Here y2 and the conv1 input never leave the CU: read y1 (+ halo) and the residual,
// write the block output and the next block's y1 = 334 MB.
//
// One persistent 8-wave workgroup per CU walks 8x16-pixel tiles.  conv2's weights (73 KB) sit in LDS for the whole kernel,
// conv3's and the next conv1's are MFMA A-fragments in registers (each wave owns 32 / 16 output channels), so no weight byte
// is re-read per tile.  Every GEMM is computed as D[channel][pixel] (weights = A operand): a lane then holds 4 consecutive
// channels of one pixel, which is what the bf16 NHWC stores, the residual loads and the LDS hand-over to the next GEMM want --
// no transposition through LDS in any epilogue.  Per tile:
//   A. conv2: 10x18 halo patch of y1 (LDS, loaded one tile ahead into registers) x w2 (LDS), 36 x v_mfma_f32_32x32x16_bf16 per
//      wave (wave = 32 pixels x 32 channels), bn2 + ReLU -> y2 (LDS, bf16: same rounding point as the unfused path)
//   B. conv3 per 64-pixel half: y2 (LDS) x w3 (registers); bn3 + residual + ReLU in place on the T tile (LDS, bf16), which the
//      residual entered and the block output leaves with coalesced 16-byte accesses (two whole pixels per wave instruction)
//   C. next conv1 per half: T (LDS) x w1' (registers) on v_mfma_f32_16x16x32_bf16, bn1' + ReLU -> next y1 (HBM)
// A half's residual registers are re-requested for the next tile as soon as they have been copied to T (a tile of lead time);
// all global accesses are unconditional (clamped addresses, zero-select at the LDS write) so that the compiler's in-order vmcnt
// accounting stays exact and no wait covers more than it needs.
#include "conv_common.h"

#ifndef DIR_BNECK_RES16
#define DIR_BNECK_RES16 1      // A/B switch: the residual as 16-byte chunks + v_permlane32_swap (1) or as 8-byte pieces in the accumulator layout (0)
#endif

namespace dir {
namespace {

using convk::bf16_t;
using convk::f16s_t;
using convk::Half;
using convk::f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TH = 8, TW = 16, NPX = TH * TW;            // tile: 8 rows x 16 cols = 128 pixels, pixel P = row * 16 + col
constexpr int PH = TH + 2, PWD = TW + 2, NPP = PH * PWD;  // halo patch 10 x 18
constexpr int PPITCH = 144;                               // bytes per patch / y2 pixel: 64 bf16 + 16 (conflict-free 16-lane reads)
constexpr int W2PITCH = 1168;                             // bytes per w2 row: 576 bf16 + 16
constexpr int TPITCH = 528;                               // bytes per T pixel: 256 bf16 + 16
constexpr int NTHR = 512;
constexpr int NPRE = (NPP * 8 + NTHR - 1) / NTHR;         // 16-byte patch chunks per thread (3)

struct ChainArgs {
    const bf16_t* y1; const bf16_t* res; bf16_t* out; bf16_t* y1n;
    const bf16_t* w2; const float* sc2; const float* sh2;
    const bf16_t* w3; const float* sc3; const float* sh3;
    const bf16_t* w1n; const float* sc1n; const float* sh1n;
    const bf16_t* x2; const bf16_t* wd;                   // projection shortcut: second 64-channel source of conv3's GEMM
    int B, H, W, tiles_x, tiles_y, ntiles;
    int decim;                                            // 1: only the pixels with even y and even x leave, as out [B][H/2][W/2][256]
};

template <typename H> __device__ __forceinline__ void unpack4(uint2 v, float (&f)[4]) {
    convk::unpack2<H>(v.x, f[0], f[1]);
    convk::unpack2<H>(v.y, f[2], f[3]);
}

// N2: output channels of the fused next conv1 (0 = none, 64 = next layer1 block, 128 = first block of layer2)
// H: the 16-bit storage kind (bf16_t | f16s_t = DIR_DT_BF16 | DIR_DT_F16) of every tensor, weight and LDS tile
template <bool HAS_RES, int N2, bool HAS_DUAL, typename H = bf16_t>
__global__ __launch_bounds__(NTHR, 1) void bneck_chain_kernel(ChainArgs a) {
    convk::half_kernel_init<H>();
    constexpr bool HAS_NEXT = N2 > 0;
    __shared__ __attribute__((aligned(16))) char s_w2[64 * W2PITCH];
    __shared__ __attribute__((aligned(16))) char s_patch[NPP * PPITCH];
    __shared__ __attribute__((aligned(16))) char s_y2[NPX * PPITCH];
    __shared__ __attribute__((aligned(16))) char s_t[64 * TPITCH];
    __shared__ __attribute__((aligned(16))) float s_ss[896];          // sc2 sh2 (64 each) | sc3 sh3 (256 each) | sc1n sh1n (128 each)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;                         // 32x32 MFMA: row / col index, 8-element k group
    const int l16 = lane & 15, g = lane >> 4;                         // 16x16 MFMA

    // ---- once per workgroup: conv2 weights and the folded BN vectors -> LDS; conv3 / next conv1 weights -> registers
    for (int c = tid; c < 64 * 72; c += NTHR) {
        const int row = c / 72, col = c - row * 72;
        *reinterpret_cast<uint4*>(s_w2 + row * W2PITCH + col * 16) = *reinterpret_cast<const uint4*>(a.w2 + row * 576 + col * 8);
    }
    if (tid < 64) { s_ss[tid] = a.sc2[tid]; s_ss[64 + tid] = a.sh2[tid]; }
    if (tid < 256) { s_ss[128 + tid] = a.sc3[tid]; s_ss[384 + tid] = a.sh3[tid]; }
    if (HAS_NEXT && tid < N2) { s_ss[640 + tid] = a.sc1n[tid]; s_ss[768 + tid] = a.sh1n[tid]; }
    convk::u32x4 w3f[4];                                                    // channel 32 wave + l32, k = 16 s + 8 h
#pragma unroll
    for (int s = 0; s < 4; ++s) w3f[s] = *reinterpret_cast<const convk::u32x4*>(a.w3 + (32 * wave + l32) * 64 + 16 * s + 8 * h);
    convk::u32x4 wdf[HAS_DUAL ? 4 : 1];                                     // projection weights, same fragment layout as w3f
    if constexpr (HAS_DUAL) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wdf[s] = *reinterpret_cast<const convk::u32x4*>(a.wd + (32 * wave + l32) * 64 + 16 * s + 8 * h);
    }
    // phase-C roles: 16 output channels per wave (N2 = 128: wave; 64: wave >> 1) x NPT 16-pixel groups of the half
    constexpr int NPT = N2 == 128 ? 4 : 2;
    const int ct = N2 == 128 ? wave : (wave >> 1), pt0 = N2 == 128 ? 0 : 2 * (wave & 1);
    convk::u32x4 w1f[HAS_NEXT ? 8 : 1];                                     // channel 16 ct + l16, k = 32 s + 8 g
    if constexpr (HAS_NEXT) {
#pragma unroll
        for (int s = 0; s < 8; ++s) w1f[s] = *reinterpret_cast<const convk::u32x4*>(a.w1n + (16 * ct + l16) * 256 + 32 * s + 8 * g);
    }

    // XCD-aware tile order: the 32 workgroups of one XCD walk one contiguous range of tiles (halo rows hit the same L2)
    const int nblk = gridDim.x;
    int t0, tstep, tend;
    if ((nblk & 7) == 0 && a.ntiles % 8 == 0) {
        const int per = a.ntiles >> 3;
        t0 = (blockIdx.x & 7) * per + (blockIdx.x >> 3); tstep = nblk >> 3; tend = ((blockIdx.x & 7) + 1) * per;
    } else { t0 = blockIdx.x; tstep = nblk; tend = a.ntiles; }

    // ---- halo patch prefetch: chunk e = tid + 512 i -> patch pixel e >> 3, 16-byte channel chunk e & 7
    uint4 pre[NPRE];
    unsigned okmask = 0;
    auto origin = [&](int t, int& b, int& y0, int& x0) {
        const int tx = t % a.tiles_x, r = t / a.tiles_x;
        b = r / a.tiles_y; y0 = (r - b * a.tiles_y) * TH; x0 = tx * TW;
    };
    auto fetch = [&](int t) {
        int b, y0, x0;
        origin(t, b, y0, x0);
        okmask = 0;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + NTHR * i, pp = e >> 3, cc = e & 7;
            const int pr = pp / PWD, pc = pp - pr * PWD;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            if (e < NPP * 8 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) okmask |= 1u << i;
            const int iyc = min(max(iy, 0), a.H - 1), ixc = min(max(ix, 0), a.W - 1);
            pre[i] = *reinterpret_cast<const uint4*>(a.y1 + (((long long)b * a.H + iyc) * a.W + ixc) * 64 + cc * 8);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + NTHR * i, pp = e >> 3, cc = e & 7;
            if (e < NPP * 8)
                *reinterpret_cast<uint4*>(s_patch + pp * PPITCH + cc * 16) = (okmask >> i) & 1u ? pre[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    int t = t0;
    if (t < tend) fetch(t);
    // weight fragments landed before the tile loop (no vmcnt wait on them inside it)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(w3f[s]));
    if constexpr (HAS_DUAL) {
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(wdf[s]));
    }
    if constexpr (HAS_NEXT) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(w1f[s]));
    }
    if (t < tend) stage();
    __syncthreads();

    // phase-A roles: wave = (pixel group mt: tile rows 2 mt, 2 mt + 1) x (channel half nt)
    const int mt = wave >> 1, nt = wave & 1;
    const int prow = 2 * mt + (l32 >> 4), pcol = l32 & 15;
    const char* pa_b = s_patch + (prow * PWD + pcol) * PPITCH + 16 * h;       // + ((ky * 18 + kx) * 144 + 32 s)
    const char* pa_w = s_w2 + (32 * nt + l32) * W2PITCH + 16 * h;             // + (tap * 128 + 32 s)

    // Block output leaves through the T tile ([64 pixels][256 channels] bf16 per half) with COALESCED 16-byte stores: chunk
    // c = tid + 512 i of a half = pixel c >> 5, channel chunk c & 31, i.e. two whole 512-byte pixels per wave instruction.  The
    // residual is read straight into the epilogue-B register layout (pixel 64 mg + 32 j + l32, channels 32 wave + 8 q + 4 h .. +4;
    // 8-byte pieces): measured, the extra LDS round trip + barrier of a coalesced residual costs more than it saves.
    // Since round 3 as 16-byte chunks (lane half h: channels 16 jj + 8 h .. + 8), swapped into the accumulator layout with
    // v_permlane32_swap where they are used: half the load instructions (tail.hip, same change).
#if DIR_BNECK_RES16
    uint4 xr[2][2][2];
#else
    uint2 xr[2][2][4];
#endif
    auto out_ptr = [&](int mg, int i, int rb, int ry0, int rx0) {
        const int c = tid + NTHR * i, P = 64 * mg + (c >> 5);
        return a.out + (((long long)rb * a.H + ry0 + (P >> 4)) * a.W + rx0 + (P & 15)) * 256 + (c & 31) * 8;
    };

    auto tile = [&](int t) {
        int b, y0, x0;
        origin(t, b, y0, x0);
        const bool more = t + tstep < tend;
        fetch(more ? t + tstep : t);                                          // consumed at the end of this tile
        // this tile's residual: in flight during phase A, consumed in the epilogues B.  (Requesting it a tile ahead does not pay on
        // gfx9: loads and stores share the in-order vmcnt, so waiting for old loads also waits for the wave's recent stores.)
        if constexpr (HAS_RES) {
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int P = 64 * mg + 32 * j + l32;
#if DIR_BNECK_RES16
                    const bf16_t* rp = a.res + (((long long)b * a.H + y0 + (P >> 4)) * a.W + x0 + (P & 15)) * 256 + 32 * wave + 8 * h;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) xr[mg][j][jj] = *reinterpret_cast<const uint4*>(rp + 16 * jj);
#else
                    const bf16_t* rp = a.res + (((long long)b * a.H + y0 + (P >> 4)) * a.W + x0 + (P & 15)) * 256 + 32 * wave + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr[mg][j][q] = *reinterpret_cast<const uint2*>(rp + 8 * q);
#endif
                }
        }

        // projection shortcut input (the block's own input, 64 channels) for this tile: chunk c = tid + 512 i = pixel c >> 3,
        // 16-byte chunk c & 7; parked in the patch buffer once phase A is done with it
        uint4 xq0 = make_uint4(0u, 0u, 0u, 0u), xq1 = xq0;                   // (two scalars: as an array the pair went through scratch)
        if constexpr (HAS_DUAL) {
            const int P0 = tid >> 3, P1 = (tid + NTHR) >> 3;
            xq0 = *reinterpret_cast<const uint4*>(a.x2 + (((long long)b * a.H + y0 + (P0 >> 4)) * a.W + x0 + (P0 & 15)) * 64 + (tid & 7) * 8);
            xq1 = *reinterpret_cast<const uint4*>(a.x2 + (((long long)b * a.H + y0 + (P1 >> 4)) * a.W + x0 + (P1 & 15)) * 64 + (tid & 7) * 8);
        }

        // ---- A. conv2 (3x3): D[channel 32][pixel 32] per wave, K = 9 taps x 64 channels
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // operand fragments are read two k-steps ahead of the MFMA that uses them (LDS latency ~ one MFMA)
            uint4 wv[3], pv[3];
            auto ldfrag = [&](int i, int slot) {
                const int tap = i >> 2, s = i & 3, ky = tap / 3, kx = tap - 3 * ky;
                wv[slot] = *reinterpret_cast<const uint4*>(pa_w + tap * 128 + 32 * s);
                pv[slot] = *reinterpret_cast<const uint4*>(pa_b + (ky * PWD + kx) * PPITCH + 32 * s);
            };
            ldfrag(0, 0);
            ldfrag(1, 1);
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                if (i + 2 < 36) ldfrag(i + 2, (i + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                acc = Half<H>::mfma32(wv[i % 3], pv[i % 3], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            // rows (channels) of the 32x32 tile held by this lane: 8 q + 4 h + {0..3}
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * nt + 8 * q + 4 * h;
                const float4 sc = *reinterpret_cast<const float4*>(s_ss + c0);
                const float4 sh = *reinterpret_cast<const float4*>(s_ss + 64 + c0);
                uint2 o;
                o.x = Half<H>::pack2_relu(fmaf(acc[4 * q], sc.x, sh.x), fmaf(acc[4 * q + 1], sc.y, sh.y));
                o.y = Half<H>::pack2_relu(fmaf(acc[4 * q + 2], sc.z, sh.z), fmaf(acc[4 * q + 3], sc.w, sh.w));
                *reinterpret_cast<uint2*>(s_y2 + (32 * mt + l32) * PPITCH + c0 * 2) = o;
            }
        }
        __syncthreads();                                                      // y2 complete; the patch is free
        if constexpr (HAS_DUAL) {
            *reinterpret_cast<uint4*>(s_patch + (tid >> 3) * PPITCH + (tid & 7) * 16) = xq0;
            *reinterpret_cast<uint4*>(s_patch + ((tid + NTHR) >> 3) * PPITCH + (tid & 7) * 16) = xq1;
            __syncthreads();
        }

#pragma unroll
        for (int mg = 0; mg < 2; ++mg) {
            // ---- B. conv3 (1x1, K = 64): this wave's 32 output channels x the half's 64 pixels
            f32x16 accb[2];
            uint4 pvb[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    pvb[j][s] = *reinterpret_cast<const uint4*>(s_y2 + (64 * mg + 32 * j + l32) * PPITCH + 32 * s + 16 * h);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) accb[j] = Half<H>::mfma32(w3f[s], pvb[j][s], accb[j]);
            if constexpr (HAS_DUAL) {                                         // + projection shortcut: K = 64 more, from the parked input tile
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        pvb[j][s] = *reinterpret_cast<const uint4*>(s_patch + (64 * mg + 32 * j + l32) * PPITCH + 32 * s + 16 * h);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) accb[j] = Half<H>::mfma32(wdf[s], pvb[j][s], accb[j]);
            }
            // epilogue B -> T: lane = pixel 32 j + l32, channels 32 wave + 8 q + 4 h .. +4
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * wave + 8 * q + 4 * h;
                    char* tp = s_t + (32 * j + l32) * TPITCH + c0 * 2;
                    const float4 sc = *reinterpret_cast<const float4*>(s_ss + 128 + c0);
                    const float4 sh = *reinterpret_cast<const float4*>(s_ss + 384 + c0);
                    float v[4] = {fmaf(accb[j][4 * q], sc.x, sh.x), fmaf(accb[j][4 * q + 1], sc.y, sh.y),
                                  fmaf(accb[j][4 * q + 2], sc.z, sh.z), fmaf(accb[j][4 * q + 3], sc.w, sh.w)};
                    if constexpr (HAS_RES) {
                        float rv[4];
#if DIR_BNECK_RES16
                        const uint4 c = xr[mg][j][q >> 1];            // (q even, q odd) = swap(lower 8 bytes, upper 8 bytes) with the partner lane
                        const auto sx = __builtin_amdgcn_permlane32_swap(c.x, c.z, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(c.y, c.w, false, false);
                        unpack4<H>(make_uint2(sx[q & 1], sy[q & 1]), rv);
#else
                        unpack4<H>(xr[mg][j][q], rv);
#endif
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    uint2 o;
                    o.x = Half<H>::pack2_relu(v[0], v[1]);
                    o.y = Half<H>::pack2_relu(v[2], v[3]);
                    *reinterpret_cast<uint2*>(tp) = o;
                }
            }
            __syncthreads();                                                  // T = this half of the block output
            // block output: coalesced 16-byte stores from T
            if (!a.decim) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = tid + NTHR * i;
                    *reinterpret_cast<uint4*>(out_ptr(mg, i, b, y0, x0)) = *reinterpret_cast<const uint4*>(s_t + (c >> 5) * TPITCH + (c & 31) * 16);
                }
            } else {
                // the only reader of this block output is a stride-2 1x1 convolution (layer2's projection shortcut, models/backbone/resnet.py:117-119,
                // the next conv1 being computed right here): a quarter of the pixels ever leave the CU
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = tid + NTHR * i, P = 64 * mg + (c >> 5), py = P >> 4, px = P & 15;
                    if (((py | px) & 1) == 0)
                        *reinterpret_cast<uint4*>(a.out + (((long long)b * (a.H >> 1) + ((y0 + py) >> 1)) * (a.W >> 1) + ((x0 + px) >> 1)) * 256 + (c & 31) * 8) =
                            *reinterpret_cast<const uint4*>(s_t + (c >> 5) * TPITCH + (c & 31) * 16);
                }
            }
            if constexpr (HAS_NEXT) {
                // ---- C. next block's conv1 (1x1, K = 256): 16 channels x NPT 16-pixel groups per wave
                f32x4 accc[NPT];
#pragma unroll
                for (int u = 0; u < NPT; ++u) accc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        const uint4 pv = *reinterpret_cast<const uint4*>(s_t + (16 * (pt0 + u) + l16) * TPITCH + 64 * s + 16 * g);
                        accc[u] = Half<H>::mfma16(w1f[s], pv, accc[u]);
                    }
                const int c0 = 16 * ct + 4 * g;
                const float4 sc = *reinterpret_cast<const float4*>(s_ss + 640 + c0);
                const float4 sh = *reinterpret_cast<const float4*>(s_ss + 768 + c0);
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int P = 64 * mg + 16 * (pt0 + u) + l16;
                    uint2 o;
                    o.x = Half<H>::pack2_relu(fmaf(accc[u][0], sc.x, sh.x), fmaf(accc[u][1], sc.y, sh.y));
                    o.y = Half<H>::pack2_relu(fmaf(accc[u][2], sc.z, sh.z), fmaf(accc[u][3], sc.w, sh.w));
                    *reinterpret_cast<uint2*>(a.y1n + (((long long)b * a.H + y0 + (P >> 4)) * a.W + x0 + (P & 15)) * N2 + c0) = o;
                }
            }
            if (mg == 0) __syncthreads();                                     // T free for the second half
        }
        stage();                                                              // next tile's patch (loads issued a whole tile ago)
        __syncthreads();                                                      // patch visible; y2 free
    };
    for (; t < tend; t += tstep) tile(t);
}

}  // namespace
}  // namespace dir

extern "C" int dir_bottleneck_chain_forward(const dir_bneck_chain_params* p, const void* y1, const void* residual, const void* x2, void* out,
                                            void* y1_next, int B, int H, int W, void* stream) {
    using namespace dir;
    DIR_REQUIRE(p && y1 && out && B > 0, "dir_bottleneck_chain_forward: bad args");
    DIR_REQUIRE(p->w2 && p->scale2 && p->shift2 && p->w3 && p->scale3 && p->shift3, "dir_bottleneck_chain_forward: missing conv2 / conv3 parameters");
    DIR_REQUIRE(H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "dir_bottleneck_chain_forward: H must be a multiple of 8 and W of 16");
    const bool next = y1_next != nullptr;
    const bool dual = x2 != nullptr;
    DIR_REQUIRE(!dual || (p->wd && !residual), "dir_bottleneck_chain_forward: a second source needs wd and excludes the identity residual");
    DIR_REQUIRE(!next || (p->w1n && p->scale1n && p->shift1n), "dir_bottleneck_chain_forward: y1_next needs the next conv1 parameters");
    ChainArgs a;
    a.y1 = (const convk::bf16_t*)y1; a.res = (const convk::bf16_t*)residual; a.out = (convk::bf16_t*)out; a.y1n = (convk::bf16_t*)y1_next;
    a.w2 = (const convk::bf16_t*)p->w2; a.sc2 = p->scale2; a.sh2 = p->shift2;
    a.w3 = (const convk::bf16_t*)p->w3; a.sc3 = p->scale3; a.sh3 = p->shift3;
    a.w1n = (const convk::bf16_t*)p->w1n; a.sc1n = p->scale1n; a.sh1n = p->shift1n;
    a.x2 = (const convk::bf16_t*)x2; a.wd = (const convk::bf16_t*)p->wd;
    a.B = B; a.H = H; a.W = W; a.tiles_x = W / TW; a.tiles_y = H / TH; a.decim = p->out_decimate ? 1 : 0;
    const long long nt = (long long)B * a.tiles_x * a.tiles_y;
    DIR_REQUIRE(nt < (1ll << 31) && (long long)B * H * W * 256 < (1ll << 40), "dir_bottleneck_chain_forward: too large");
    a.ntiles = (int)nt;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0; hipDeviceProp_t pr;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    }
    const int grid = (int)(nt < num_cu ? nt : num_cu);
    hipStream_t s = (hipStream_t)stream;
    const bool res = residual != nullptr;
    const int n2 = next ? p->n_next : 0;
    DIR_REQUIRE(n2 == 0 || n2 == 64 || n2 == 128, "dir_bottleneck_chain_forward: n_next must be 64 or 128");
    DIR_REQUIRE(p->dtype == 0 || p->dtype == DIR_DT_BF16 || p->dtype == DIR_DT_F16, "dir_bottleneck_chain_forward: dtype must be bf16 (or 0) or f16");
    const bool f16 = p->dtype == DIR_DT_F16;
#define DIR_CHAIN(RES_, N2_, DUAL_) do { if (f16) DIR_LAUNCH((bneck_chain_kernel<RES_, N2_, DUAL_, f16s_t>), dim3(grid), dim3(NTHR), 0, s, a); \
                                         else DIR_LAUNCH((bneck_chain_kernel<RES_, N2_, DUAL_, bf16_t>), dim3(grid), dim3(NTHR), 0, s, a); } while (0)
    if (dual) { if (n2 == 128) DIR_CHAIN(false, 128, true); else if (n2) DIR_CHAIN(false, 64, true); else DIR_CHAIN(false, 0, true); }
    else if (res) { if (n2 == 128) DIR_CHAIN(true, 128, false); else if (n2) DIR_CHAIN(true, 64, false); else DIR_CHAIN(true, 0, false); }
    else { if (n2 == 128) DIR_CHAIN(false, 128, false); else if (n2) DIR_CHAIN(false, 64, false); else DIR_CHAIN(false, 0, false); }
#undef DIR_CHAIN
    return check_launch("dir_bottleneck_chain_forward");
}
