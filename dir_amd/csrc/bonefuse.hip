// Factorised bone fusion: Joint2BoneFeature.bone_proj (models/dir.py:132-174) + fusion[0..2] (3x3 conv 2560 -> 256,
// BatchNorm, ReLU; models/dir.py:57-62) WITHOUT the [B,S,S,2560] bone map.
//
// The rasterised operand is rank 2 per bone:  img[p, hb, c] = m(p,hb) * (wa(p,hb) * fa[hb][c] + wb(p,hb) * fb[hb][c]),
// fa / fb = the 64-channel token features of the bone's two end joints, (m, wa, wb) per-pixel scalars.  Hence
//
//   conv(img)[p, n] = sum_tap sum_e  Wgt[p + tap, e] * G[tap, e, n],      e = (hand, bone, end) in [0, 80)
//   G[tap, (hb, end), n] = sum_c f_end[hb][c] * W[n, tap, hb*64 + c]      (per sample: 80 x 9 x 256 numbers)
//   Wgt[p, (hb, 0)] = m * wa,  Wgt[p, (hb, 1)] = m * wb
//
// i.e. a K = 9 * 80 = 720 reduction instead of K = 23040: 32x fewer MACs than the dense convolution, no 335 MB bone map
// written and re-read, and no sparsity bookkeeping.  Mathematically identical to the reference; the summation is
// re-associated, so this path is used in the bf16 throughput mode only (the fp32 parity mode keeps bone_proj + conv).
//
//   bone_g_kernel    : G for every sample, exact fp32 MFMA (v_mfma_f32_16x16x4_f32) over the bf16-rounded weights;
//                      one workgroup per (tap, hand-bone): [B*2 ends, 64] x [64, 256]; written as bf16 pairs (end 0, end 1)
//   bone_fuse_kernel : one workgroup per (sample, 256-pixel strip, 128 output channels): the strip's halo patch of Wgt
//                      (<= 400 pixels x 80) is computed ONCE into LDS with bone_proj's own distance / weight formulas
//                      (bit-identical mask), every tap reads its MFMA A operand from the patch at a shifted row (as
//                      conv_pipe.hip's halo-reuse kernel), G_tap streams through a double-buffered LDS tile;
//                      v_mfma_f32_32x32x16_bf16, fp32 accumulate, BN + ReLU + coalesced bf16 NHWC epilogue.
#include "bone_common.h"
#include "conv_common.h"
#include "dir_mfma.h"

namespace dir {
namespace {

using namespace dir::convk;
using dir::bone::bone_weights;
using dir::bone::kChild;
using dir::bone::kParent;

constexpr int NE = 80;            // bone ends: 2 hands x 20 bones x 2
constexpr int EP = 88;            // LDS row length in bf16: 176-byte pitch = 11 x 16 B, odd -> conflict-free ds_read_b128
constexpr int NCOUT = 256;        // fusion.0 output channels
constexpr int NTAP = 9;

// ---------------------------------------------------------------------------------------------------------------- G
struct GArgs {
    const float* w_g;     // [9][40][64][256]
    const float* emb;     // [B][42][64]
    unsigned* g;          // [B][9][40][256] words = (bf16 end 0) | (bf16 end 1) << 16;  F32: float2 (end 0, end 1) instead
    int B;
    long long* stamps;    // DIR_STAMPS=bone_g (tuning aid, else NULL)
};

constexpr int G_ROWS = 128, G_LD = 66;      // (sample, end) rows per pass; lda % 32 == 2 (dir_mfma.h)

// H (F32 = false only): the 16-bit kind G is rounded to (bf16_t | f16s_t = the throughput modes DIR_DT_BF16 | DIR_DT_F16)
template <bool F32, typename H = bf16_t>
__global__ __launch_bounds__(256) void bone_g_kernel(GArgs a) {
    half_kernel_init<H>();
    __shared__ float s_f[G_ROWS * G_LD];
    const int tap = blockIdx.x / 40, hb = blockIdx.x - tap * 40, hand = hb / 20, bone = hb - hand * 20;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jpar = hand * 21 + kParent[bone], jchi = hand * 21 + kChild[bone];
    const float* wt = a.w_g + ((long long)(tap * 40 + hb) * 64) * NCOUT;
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nstamp < MAX_STAMPS) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    for (int b0 = 0; b0 < a.B; b0 += G_ROWS / 2) {
        __syncthreads();
        // thread = (column c, row quarter): 32 independent loads from clamped addresses, all in flight together (a conditional load
        // per iteration is a chain of 32 memory round trips: 10 us of this kernel); rows past the batch are zeroed at the LDS write
        {
            const int c = tid & 63, r0 = tid >> 6;
            float v[G_ROWS / 4];
#pragma unroll
            for (int k = 0; k < G_ROWS / 4; ++k) {
                const int r = r0 + 4 * k, b = min(b0 + (r >> 1), a.B - 1);
                v[k] = a.emb[((long long)b * 42 + ((r & 1) ? jchi : jpar)) * 64 + c];
            }
#pragma unroll
            for (int k = 0; k < G_ROWS / 4; ++k) {
                const int r = r0 + 4 * k;
                s_f[r * G_LD + c] = b0 + (r >> 1) < a.B ? v[k] : 0.f;
            }
        }
        __syncthreads(); stamp();
        for (int nt = wave + 4 * blockIdx.y; nt < NCOUT / 16; nt += 4 * gridDim.y) {   // grid.y = 2: 720 workgroups of two column tiles
                                                                                     // per wave balance better over 256 CUs than 360 of four
            f32x4 acc[G_ROWS / 16];
#pragma unroll
            for (int m = 0; m < G_ROWS / 16; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            mfma_tile_f32<64, G_ROWS / 16>(s_f, G_LD, wt, NCOUT, nt * 16, lane, acc);
            asm volatile("" :: "v"(acc[0]), "v"(acc[7])); stamp();
            // lane holds column n = nt*16 + (lane & 15), rows m*16 + 4*(lane >> 4) + r: two samples x two ends
            const int n = nt * 16 + (lane & 15);
#pragma unroll
            for (int m = 0; m < G_ROWS / 16; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int b = b0 + m * 8 + 2 * (lane >> 4) + s;
                    if (b < a.B) {
                        const long long o = (((long long)b * NTAP + tap) * 40 + hb) * NCOUT + n;
                        if constexpr (F32) reinterpret_cast<float2*>(a.g)[o] = make_float2(acc[m][2 * s], acc[m][2 * s + 1]);
                        else a.g[o] = Half<H>::pack2(acc[m][2 * s], acc[m][2 * s + 1]);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------- fuse
struct FuseArgs {
    ConvArgs c;           // epilogue description: scale, shift, y, M, Cout, out_cs, out_co, flags (res = nullptr)
    const float* uv[2];   // [B][21][2] normalised joint uv, left / right
    const unsigned* g;
    int S;
    float distance;
    int PW, PH, npr;      // halo patch geometry of one 256-pixel strip
    long long* stamps;    // DIR_STAMPS=bone_fuse (tuning aid, else NULL)
    unsigned mg_npr, sh_npr, mg_pw, sh_pw;   // magic dividers (convk::magic_u31) for npr and PW
    float g_scale;        // split-precision variant: power of two applied to G before the f16 hi / lo split
    unsigned mg_pr8, sh_pr8;                 // and the divider for ceil(npr / 8)
};

constexpr int FUSE_MAX_ROWS = 400;

template <typename H>
__global__ __launch_bounds__(512, 1) void bone_fuse_kernel(FuseArgs a) {
    half_kernel_init<H>();
    constexpr int MI = 2, NJ = 2, WM = 4, WN = 2, NT = 512, BM = 256, BN = 128;
    constexpr int PITCH = EP * 2;                                   // 176 B
    constexpr int P_BYTES = FUSE_MAX_ROWS * PITCH, G_BYTES = BN * PITCH;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = P_BYTES + 2 * G_BYTES > STAGE_BYTES ? P_BYTES + 2 * G_BYTES : STAGE_BYTES;
    constexpr int GW = 40 * BN / NT;                                 // G words per thread per tap (10)
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    __shared__ float s_uv[84];
    __shared__ float s_bone[40 * 6];                                // per (hand, bone): ax, ay, bx, by (pixel units), unit direction dx, dy

    const int S = a.S, hw = S * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int tn = blockIdx.x & 1, tm = blockIdx.x >> 1;            // the two N halves of a strip are neighbours (same XCD pair)
    const int m0 = tm * BM, n0 = tn * BN;
    const int b = m0 / hw, y0 = (m0 - b * hw) / S;
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();

    // ---- G_tap loader: word (hb, n) -> LDS [n][hb*2 .. hb*2+1].  A wave covers 16 consecutive n x 2 hb x 2 (64-byte global segments); in
    //      LDS (rows 44 dwords apart) that is at most two lanes per bank for a ds_write_b32 -- free -- where one row per lane was 8-way
    const int gn = 16 * ((tid >> 5) & 7) + (tid & 15), ghb = ((tid >> 8) << 1) | ((tid >> 4) & 1);      // word k: n = gn, hb = 4 k + ghb
    const unsigned* gsrc = a.g + ((long long)b * NTAP * 40) * NCOUT + n0 + gn;
    unsigned greg[GW];
    auto g_load = [&](int tap) {
#pragma unroll
        for (int k = 0; k < GW; ++k) greg[k] = gsrc[((long long)tap * 40 + 4 * k + ghb) * NCOUT];
    };
    auto g_store = [&](int buf) {
        char* gb = smem + P_BYTES + buf * G_BYTES + gn * PITCH + ghb * 4;
#pragma unroll
        for (int k = 0; k < GW; ++k) *reinterpret_cast<unsigned*>(gb + 16 * k) = greg[k];
    };
    g_load(0);

    // ---- the strip's halo patch of Wgt: rows = (PH x PW) input pixels, 80 bf16 per row
    if (tid < 84) {
#pragma clang fp contract(off)
        const int hand = tid / 42, r = tid - hand * 42;
        const float v = a.uv[hand][(long long)b * 42 + r];
        s_uv[tid] = (v + 1.f) / 2.f * (float)S;                      // models/dir.py:150
    }
    __syncthreads(); stamp();
    // per-bone unit directions once per workgroup, then item i = hb * npr + prow: a wave covers 64 consecutive patch pixels of ONE
    // bone, so most waves take the early-out of bone_weights_fast as a whole (the capsule touches ~10 % of the patch)
    if (tid < 40) {
        const int hand = tid / 20, bone = tid - hand * 20;
        const float* uv = s_uv + hand * 42;
        const int pa = kParent[bone], ch = kChild[bone];
        float dx, dy;
        dir::bone::bone_dir(uv[2 * pa], uv[2 * pa + 1], uv[2 * ch], uv[2 * ch + 1], dx, dy);
        float* sb = s_bone + 6 * tid;
        sb[0] = uv[2 * pa]; sb[1] = uv[2 * pa + 1]; sb[2] = uv[2 * ch]; sb[3] = uv[2 * ch + 1]; sb[4] = dx; sb[5] = dy;
    }
    __syncthreads();
    const int pr8 = (a.npr + 7) >> 3;
    for (int i = tid; i < pr8 * 320; i += NT) {                      // i = ((hq * pr8 + rg) * 4 + c) * 8 + r: patch row 8 rg + r, hand-bone 4 hq + c:
        const int rest = i >> 5, hq = convk::div_magic(rest, a.mg_pr8, a.sh_pr8), rg = rest - hq * pr8;      // 32 distinct LDS banks per 32 lanes
        const int prow = 8 * rg + (i & 7), hb = 4 * hq + ((i >> 3) & 3);
        if (prow >= a.npr) continue;
        const int py = convk::div_magic(prow, a.mg_pw, a.sh_pw), px = prow - py * a.PW;
        const int iy = y0 + py - 1, ix = px - 1;                      // 3x3, pad 1
        unsigned word = 0;
        if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
            const float* sb = s_bone + 6 * hb;                        // (the parent / child tables are __constant__: indexed per lane
            float wa, wb;                                             //  they would be two global loads per item)
            if (dir::bone::bone_weights_fast((float)ix + 0.5f, (float)iy + 0.5f, sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], a.distance, wa, wb))
                word = Half<H>::pack2(wa, wb);   // torch.where(mask, v, 0), models/dir.py:172
        }
        *reinterpret_cast<unsigned*>(smem + prow * PITCH + hb * 4) = word;
    }
    g_store(0);
    __syncthreads(); stamp();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // lane (i = lane & 31, h = lane >> 5): k16-step s reads the 16 bytes at e = s*16 + h*8 of its A / B row
    int pr0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * MI * 32 + i * 32 + (lane & 31);
        const int y = r / S, x = r - y * S;
        pr0[i] = y * a.PW + x;
    }
    const int hoff = (lane >> 5) * 16;
    const int frag_b = (wn * NJ * 32 + (lane & 31)) * PITCH + hoff;

    for (int tap = 0; tap < NTAP; ++tap) {
        if (tap + 1 < NTAP) g_load(tap + 1);                          // in flight during this tap's MFMAs
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* gb = smem + P_BYTES + (tap & 1) * G_BYTES + frag_b;
        const int shift = (ky * a.PW + kx) * PITCH + hoff;
#pragma unroll
        for (int s = 0; s < NE / 16; ++s) {
            uint4 fa[MI], fb[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + pr0[i] * PITCH + shift + s * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const uint4*>(gb + j * 32 * PITCH + s * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = Half<H>::mfma32(fa[i], fb[j], acc[i][j]);
        }
        if (tap + 1 < NTAP) g_store((tap + 1) & 1);                   // buffer (tap+1)&1 was last read in tap-1
        __syncthreads();
    }

    stamp();
    ConvArgs c = a.c;
    epilogue_tile<H, MI, NJ, WM, WN>(c, acc, smem, m0, n0, wm, wn, tid, lane);
    stamp();
}


// ---------------------------------------------------------------------------------------------------- fuse, exact fp32
// The same factorisation with fp32 operands on the exact fp32 matrix cores (v_mfma_f32_32x32x2_f32: products exact, fp32 accumulate)
// for the parity modes (DirEngine(dtype=float32)): no rounding of Wgt, G or the weights, so the only difference from the reference's
// bone_proj + conv3x3 is the association of the sum -- fp32 rounding noise, like any other summation order.  One workgroup per
// (sample, 128-pixel strip, 128 output channels); patch rows and G tile rows are 80 floats at a 336-byte pitch (21 x 16 B: conflict-free
// ds_read_b128); MFMA step t consumes e = t from lanes 0-31 and e = 40 + t from lanes 32-63 (any bijection of the reduction index is
// legal as long as both operands use it), so every lane reads ITS 40 values as ten 16-byte loads per tap.
constexpr int F32_PITCH = 336, F32_MAX_ROWS = 208, F32_BM = 128;

__global__ __launch_bounds__(512, 1) void bone_fuse_f32_kernel(FuseArgs a) {
    constexpr int MI = 1, NJ = 2, WM = 4, WN = 2, NT = 512, BM = F32_BM, BN = 128;
    constexpr int P_BYTES = F32_MAX_ROWS * F32_PITCH, G_BYTES = BN * F32_PITCH;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = P_BYTES + 2 * G_BYTES > STAGE_BYTES ? P_BYTES + 2 * G_BYTES : STAGE_BYTES;
    constexpr int GW = 40 * BN / NT;                                 // G float2 words per thread per tap (10)
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    __shared__ float s_uv[84];
    __shared__ float s_bone[40 * 6];

    const int S = a.S, hw = S * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int tn = blockIdx.x & 1, tm = blockIdx.x >> 1;
    const int m0 = tm * BM, n0 = tn * BN;
    const int b = m0 / hw, y0 = (m0 - b * hw) / S;

    const float2* gsrc = reinterpret_cast<const float2*>(a.g) + ((long long)b * NTAP * 40) * NCOUT + n0;
    float2 greg[GW];
    auto g_load = [&](int tap) {
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            const int w = tid + NT * k, hb = w / BN, n = w - hb * BN;
            greg[k] = gsrc[((long long)tap * 40 + hb) * NCOUT + n];
        }
    };
    auto g_store = [&](int buf) {
        char* gb = smem + P_BYTES + buf * G_BYTES;
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            const int w = tid + NT * k, hb = w / BN, n = w - hb * BN;
            *reinterpret_cast<float2*>(gb + n * F32_PITCH + hb * 8) = greg[k];
        }
    };
    g_load(0);

    if (tid < 84) {
#pragma clang fp contract(off)
        const int hand = tid / 42, r = tid - hand * 42;
        const float v = a.uv[hand][(long long)b * 42 + r];
        s_uv[tid] = (v + 1.f) / 2.f * (float)S;                      // models/dir.py:150
    }
    __syncthreads();
    if (tid < 40) {
        const int hand = tid / 20, bone = tid - hand * 20;
        const float* uv = s_uv + hand * 42;
        const int pa = kParent[bone], ch = kChild[bone];
        float dx, dy;
        dir::bone::bone_dir(uv[2 * pa], uv[2 * pa + 1], uv[2 * ch], uv[2 * ch + 1], dx, dy);
        float* sb = s_bone + 6 * tid;
        sb[0] = uv[2 * pa]; sb[1] = uv[2 * pa + 1]; sb[2] = uv[2 * ch]; sb[3] = uv[2 * ch + 1]; sb[4] = dx; sb[5] = dy;
    }
    __syncthreads();
    for (int i = tid; i < a.npr * 40; i += NT) {
        const int hb = convk::div_magic(i, a.mg_npr, a.sh_npr), prow = i - hb * a.npr;
        const int py = convk::div_magic(prow, a.mg_pw, a.sh_pw), px = prow - py * a.PW;
        const int iy = y0 + py - 1, ix = px - 1;                      // 3x3, pad 1
        float2 word = make_float2(0.f, 0.f);
        if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
            const float* sb = s_bone + 6 * hb;
            float wa, wb;
            if (dir::bone::bone_weights_fast((float)ix + 0.5f, (float)iy + 0.5f, sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], a.distance, wa, wb))
                word = make_float2(wa, wb);                            // torch.where(mask, v, 0), models/dir.py:172
        }
        *reinterpret_cast<float2*>(smem + prow * F32_PITCH + hb * 8) = word;
    }
    g_store(0);
    __syncthreads();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int pr0;
    {
        const int r = wm * 32 + (lane & 31);
        const int y = r / S, x = r - y * S;
        pr0 = y * a.PW + x;
    }
    const int hoff = (lane >> 5) * 160;                               // this lane's 40 of the 80 reduction indices
    const int frag_b = (wn * NJ * 32 + (lane & 31)) * F32_PITCH + hoff;

    for (int tap = 0; tap < NTAP; ++tap) {
        if (tap + 1 < NTAP) g_load(tap + 1);
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* gb = smem + P_BYTES + (tap & 1) * G_BYTES + frag_b;
        const char* pa = smem + (pr0 + ky * a.PW + kx) * F32_PITCH + hoff;
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const float4 fa = *reinterpret_cast<const float4*>(pa + q * 16);
            float4 fb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const float4*>(gb + j * 32 * F32_PITCH + q * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[0][j], 0, 0, 0);
            }
        }
        if (tap + 1 < NTAP) g_store((tap + 1) & 1);
        __syncthreads();
    }
    ConvArgs c = a.c;
    epilogue_tile<float, MI, NJ, WM, WN>(c, acc, smem, m0, n0, wm, wn, tid, lane);
}

// ------------------------------------------------------------------------------------------------- fuse, split precision
// The fp32 factorisation above on the f16 matrix cores: both operands (bone weights in [0, 1] times 2^10; G times the calibrated power of two
// g_scale) are split into f16 hi + lo when they are written to LDS and every product is hi*hi + lo*hi + hi*lo with fp32 accumulation -- the
// arithmetic of DIR_DT_F16X3 (conv_common.h), ~2^-22 per product, i.e. the same distance from the reference's bone_proj + conv3x3 as the
// exact kernel (fp32 summation noise) at a tenth of its matrix-core time (15 v_mfma_f32_32x32x16_f16 instead of 40 v_mfma_f32_32x32x2_f32 per
// tile and tap).  Rows are [80 hi | 80 lo] f16 at the same 336-byte pitch (conflict-free ds_read_b128); reduction index e = 2 hb + end.
// LDS writes: lanes vary (row & 7, hb & 3) fastest -- 32 distinct banks per ds_write_b32 -- instead of one row per lane (8-way conflicts).
constexpr float X3_WSCALE = 1024.f;
__device__ __forceinline__ void split2(float x, float y, float s, unsigned& hi, unsigned& lo) {
    constexpr float FMAX = 65504.f;
    const convk::f32x2_t v = {__builtin_amdgcn_fmed3f(x * s, -FMAX, FMAX), __builtin_amdgcn_fmed3f(y * s, -FMAX, FMAX)};
    const convk::f16x2_t h = __builtin_convertvector(v, convk::f16x2_t);
    const convk::f32x2_t r = v - __builtin_convertvector(h, convk::f32x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, convk::f16x2_t));
}
__global__ __launch_bounds__(512, 1) void bone_fuse_x3_kernel(FuseArgs a) {
    constexpr int MI = 1, NJ = 2, WM = 4, WN = 2, NT = 512, BM = F32_BM, BN = 128;
    constexpr int P_BYTES = F32_MAX_ROWS * F32_PITCH, G_BYTES = BN * F32_PITCH;
    constexpr int STAGE_BYTES = BM * BN * 4;
    constexpr int SMEM = P_BYTES + 2 * G_BYTES > STAGE_BYTES ? P_BYTES + 2 * G_BYTES : STAGE_BYTES;
    constexpr int GW = 40 * BN / NT;                                 // G float2 words per thread per tap (10)
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    __shared__ float s_uv[84];
    __shared__ float s_bone[40 * 6];

    const int S = a.S, hw = S * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int tn = blockIdx.x & 1, tm = blockIdx.x >> 1;
    const int m0 = tm * BM, n0 = tn * BN;
    const int b = m0 / hw, y0 = (m0 - b * hw) / S;

    // G word k of this thread: channel n = 8 * ((tid >> 5) & 15) + (tid & 7), hand-bone hb = 4 k + ((tid >> 3) & 3)
    const int gn = 8 * ((tid >> 5) & 15) + (tid & 7), ghb = (tid >> 3) & 3;
    const float2* gsrc = reinterpret_cast<const float2*>(a.g) + ((long long)b * NTAP * 40) * NCOUT + n0 + gn;
    float2 greg[GW];
    auto g_load = [&](int tap) {
#pragma unroll
        for (int k = 0; k < GW; ++k) greg[k] = gsrc[((long long)tap * 40 + 4 * k + ghb) * NCOUT];
    };
    auto g_store = [&](int buf) {
        char* gb = smem + P_BYTES + buf * G_BYTES + gn * F32_PITCH + ghb * 4;
#pragma unroll
        for (int k = 0; k < GW; ++k) {
            unsigned hi, lo;
            split2(greg[k].x, greg[k].y, a.g_scale, hi, lo);
            *reinterpret_cast<unsigned*>(gb + 16 * k) = hi;
            *reinterpret_cast<unsigned*>(gb + 160 + 16 * k) = lo;
        }
    };
    g_load(0);

    if (tid < 84) {
#pragma clang fp contract(off)
        const int hand = tid / 42, r = tid - hand * 42;
        const float v = a.uv[hand][(long long)b * 42 + r];
        s_uv[tid] = (v + 1.f) / 2.f * (float)S;                      // models/dir.py:150
    }
    __syncthreads();
    if (tid < 40) {
        const int hand = tid / 20, bone = tid - hand * 20;
        const float* uv = s_uv + hand * 42;
        const int pa = kParent[bone], ch = kChild[bone];
        float dx, dy;
        dir::bone::bone_dir(uv[2 * pa], uv[2 * pa + 1], uv[2 * ch], uv[2 * ch + 1], dx, dy);
        float* sb = s_bone + 6 * tid;
        sb[0] = uv[2 * pa]; sb[1] = uv[2 * pa + 1]; sb[2] = uv[2 * ch]; sb[3] = uv[2 * ch + 1]; sb[4] = dx; sb[5] = dy;
    }
    __syncthreads();
    const int pr8 = (a.npr + 7) >> 3;
    for (int i = tid; i < pr8 * 320; i += NT) {                      // i = ((hq * pr8 + rg) * 4 + c) * 8 + r: patch row 8 rg + r, hand-bone 4 hq + c
        const int rest = i >> 5, hq = convk::div_magic(rest, a.mg_pr8, a.sh_pr8), rg = rest - hq * pr8;
        const int prow = 8 * rg + (i & 7), hb = 4 * hq + ((i >> 3) & 3);
        if (prow >= a.npr) continue;
        const int py = convk::div_magic(prow, a.mg_pw, a.sh_pw), px = prow - py * a.PW;
        const int iy = y0 + py - 1, ix = px - 1;                      // 3x3, pad 1
        float wa = 0.f, wb = 0.f;
        if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
            const float* sb = s_bone + 6 * hb;
            float ta, tb;
            if (dir::bone::bone_weights_fast((float)ix + 0.5f, (float)iy + 0.5f, sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], a.distance, ta, tb)) {
                wa = ta; wb = tb;                                      // torch.where(mask, v, 0), models/dir.py:172
            }
        }
        unsigned hi, lo;
        split2(wa, wb, X3_WSCALE, hi, lo);
        char* pp = smem + prow * F32_PITCH + hb * 4;
        *reinterpret_cast<unsigned*>(pp) = hi;
        *reinterpret_cast<unsigned*>(pp + 160) = lo;
    }
    g_store(0);
    __syncthreads();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int pr0;
    {
        const int r = wm * 32 + (lane & 31);
        const int y = r / S, x = r - y * S;
        pr0 = y * a.PW + x;
    }
    const int hoff = (lane >> 5) * 16;                                // k16-step s: this lane's 8 reduction indices at byte 32 s + 16 h
    const int frag_b = (wn * NJ * 32 + (lane & 31)) * F32_PITCH + hoff;

    for (int tap = 0; tap < NTAP; ++tap) {
        if (tap + 1 < NTAP) g_load(tap + 1);
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* gb = smem + P_BYTES + (tap & 1) * G_BYTES + frag_b;
        const char* pa = smem + (pr0 + ky * a.PW + kx) * F32_PITCH + hoff;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const convk::f16x8 ah = *reinterpret_cast<const convk::f16x8*>(pa + q * 32), al = *reinterpret_cast<const convk::f16x8*>(pa + 160 + q * 32);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const convk::f16x8 bh = *reinterpret_cast<const convk::f16x8*>(gb + j * 32 * F32_PITCH + q * 32);
                const convk::f16x8 bl = *reinterpret_cast<const convk::f16x8*>(gb + j * 32 * F32_PITCH + 160 + q * 32);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0][j], 0, 0, 0);
            }
        }
        if (tap + 1 < NTAP) g_store((tap + 1) & 1);
        __syncthreads();
    }
    const float inv = 1.f / (X3_WSCALE * a.g_scale);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] *= inv;
    ConvArgs c = a.c;
    epilogue_tile<float, MI, NJ, WM, WN>(c, acc, smem, m0, n0, wm, wn, tid, lane);
}

}  // namespace
}  // namespace dir

// (sized for the exact-fp32 variant, float2 per (tap, hand-bone, channel); the bf16 variant uses the first half)
extern "C" size_t dir_bone_fusion_scratch_bytes(int B) { return (size_t)(B > 0 ? B : 0) * dir::NTAP * 40 * dir::NCOUT * 8; }

extern "C" int dir_bone_fusion_prepare(const dir_bone_fusion_params* p, const float* emb, void* scratch, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(p && p->w_g && emb && scratch, "dir_bone_fusion_prepare: null pointer");
    DIR_REQUIRE(B >= 0, "dir_bone_fusion_prepare: B=%d", B);
    if (B == 0) return DIR_OK;
    GArgs ga{p->w_g, emb, (unsigned*)scratch, B, stamps_begin("bone_g")};
    DIR_REQUIRE(p->exact_f32 >= 0 && p->exact_f32 <= 2, "dir_bone_fusion_prepare: exact_f32 must be 0 (bf16), 1 (fp32) or 2 (f16 storage)");
    if (p->exact_f32 == 1) DIR_LAUNCH(bone_g_kernel<true>, dim3(NTAP * 40, 2), dim3(256), 0, (hipStream_t)stream, ga);
    else if (p->exact_f32 == 2) DIR_LAUNCH((bone_g_kernel<false, f16s_t>), dim3(NTAP * 40, 2), dim3(256), 0, (hipStream_t)stream, ga);
    else DIR_LAUNCH((bone_g_kernel<false, bf16_t>), dim3(NTAP * 40, 2), dim3(256), 0, (hipStream_t)stream, ga);
    stamps_end("bone_g", ga.stamps, (hipStream_t)stream);
    return dir::check_launch("dir_bone_fusion_prepare");
}

extern "C" int dir_bone_fusion_forward(const dir_bone_fusion_params* p, const float* uv_left, const float* uv_right,
                                       const void* scratch, void* y, int B, int S, float distance, int out_cstride,
                                       int out_coff, int relu, void* stream) {
    using namespace dir;
    DIR_REQUIRE(p && uv_left && uv_right && scratch && y, "dir_bone_fusion_forward: null pointer");
    DIR_REQUIRE(B >= 0, "dir_bone_fusion_forward: B=%d", B);
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(S > 0 && 256 % S == 0 && (S * S) % 256 == 0, "dir_bone_fusion_forward: S=%d (needs 256 %% S == 0 and S*S %% 256 == 0)", S);
    const int ocs = out_cstride ? out_cstride : NCOUT;
    DIR_REQUIRE(ocs % 8 == 0 && out_coff % 8 == 0 && out_coff + NCOUT <= ocs, "dir_bone_fusion_forward: output slice must be 16-byte aligned");
    DIR_REQUIRE(p->exact_f32 >= 0 && p->exact_f32 <= 2, "dir_bone_fusion_forward: exact_f32 must be 0 (bf16), 1 (fp32) or 2 (f16 storage)");
    const bool exact = p->exact_f32 == 1;
    const int strip = exact ? F32_BM : 256;                     // output pixels per workgroup
    const long long M = (long long)B * S * S;
    DIR_REQUIRE(M < (1ll << 31), "dir_bone_fusion_forward: too many pixels");
    FuseArgs fa{};
    fa.c.scale = p->scale; fa.c.shift = p->shift; fa.c.res = nullptr; fa.c.y = y;
    fa.c.M = (int)M; fa.c.Cout = NCOUT; fa.c.out_cs = ocs; fa.c.out_co = out_coff; fa.c.res_cs = 0; fa.c.res_co = 0;
    fa.c.flags = (relu ? 1 : 0) | 4;
    fa.uv[0] = uv_left; fa.uv[1] = uv_right; fa.g = (const unsigned*)scratch; fa.S = S; fa.distance = distance;
    DIR_REQUIRE(strip % S == 0, "dir_bone_fusion_forward: S=%d does not divide the %d-pixel strip", S, strip);
    const int rows = strip / S;
    fa.PW = S + 2; fa.PH = rows + 2; fa.npr = fa.PH * fa.PW;
    DIR_REQUIRE(fa.npr <= (exact ? F32_MAX_ROWS : FUSE_MAX_ROWS), "dir_bone_fusion_forward: halo patch of %d rows does not fit", fa.npr);
    convk::magic_u31((unsigned)fa.npr, &fa.mg_npr, &fa.sh_npr);
    convk::magic_u31((unsigned)fa.PW, &fa.mg_pw, &fa.sh_pw);
    fa.stamps = stamps_begin("bone_fuse");
    fa.g_scale = exact ? p->g_scale : 0.f;
    convk::magic_u31((unsigned)((fa.npr + 7) >> 3), &fa.mg_pr8, &fa.sh_pr8);
    if (exact && fa.g_scale > 0.f) {
        int e;
        DIR_REQUIRE(frexpf(fa.g_scale, &e) == 0.5f, "dir_bone_fusion_forward: g_scale must be a power of two");
        DIR_LAUNCH(bone_fuse_x3_kernel, dim3((unsigned)(M / F32_BM) * 2), dim3(512), 0, (hipStream_t)stream, fa);
    } else if (exact) DIR_LAUNCH(bone_fuse_f32_kernel, dim3((unsigned)(M / F32_BM) * 2), dim3(512), 0, (hipStream_t)stream, fa);
    else if (p->exact_f32 == 2) DIR_LAUNCH(bone_fuse_kernel<f16s_t>, dim3((unsigned)(M / 256) * 2), dim3(512), 0, (hipStream_t)stream, fa);
    else DIR_LAUNCH(bone_fuse_kernel<bf16_t>, dim3((unsigned)(M / 256) * 2), dim3(512), 0, (hipStream_t)stream, fa);
    stamps_end("bone_fuse", fa.stamps, (hipStream_t)stream);
    return dir::check_launch("dir_bone_fusion_forward");
}
