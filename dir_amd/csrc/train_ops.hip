// Building blocks of the backward pass over the joint-token path (reference: train.py:66-70 runs
// torch autograd through transformer/mixSTE.py, SemGCN/p_graph_conv.py and the nn.Conv1d / nn.Linear / nn.LayerNorm / nn.BatchNorm1d
// modules of models/dir.py:19-130).  The token path is ~0.1 GFLOP per image: these are plain, exact-fp32, deterministic kernels
// (fixed summation orders, no atomics) -- correctness and reproducibility first; they are composed by dir_amd/train/*.py.
//
//   dir_gemm_f32              C = op(A) op(B) (+ bias | + C), batched / strided, exact fp32 on v_mfma_f32_16x16x4_f32
//   dir_colsum_f32            bias gradients: column sums of a [R, N] matrix
//   dir_layernorm_forward / _backward    nn.LayerNorm over the last dimension (mixSTE.py:177: eps 1e-6; head: 1e-5)
//   dir_gelu_forward / _backward         exact-erf GELU (mixSTE.py:12,27)
//   dir_attention_forward / _backward    softmax(q k^T * scale) v per (sample, head) (mixSTE.py:76-97)
//   dir_bn_train_forward / _backward     BatchNorm1d / 2d in training mode over [R rows, C channels] (batch statistics, running update)
#include "dir_common.h"
#include "dir_mfma.h"

#include <math.h>
#include <stdlib.h>
#include <initializer_list>
#include <mutex>

namespace {

using dir::f32x4;

// ------------------------------------------------------------------------------------------------------------------ GEMM
struct GemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K, lda, ldb, ldc, ta, tb, accumulate;
    long long sA, sB, sC;
    int kchunk; float* part;        // split-K (dir_gemm_f32_splitk): blockIdx.z = K chunk, raw partial tiles to part [chunk][M][N]
};
constexpr int GT = 64, GK = 16, GLD = GK + 1;      // 64 x 64 tile, K step 16 (the weight-gradient kernels below); LDS rows padded (17 floats)
constexpr int MK = 32, MLD = MK + 1;               // K step of gemm_f32_kernel

// 256 threads = 4 waves; wave w owns rows 16w .. 16w+15 of the tile and all 64 columns (4 accumulators of 16x16).
// A tile [64][32] (row m, k) and B tile stored transposed [64][32] (column n, k): both MFMA operands read along k.  The operands of step
// k + 1 are requested (unconditional loads from clamped addresses, masked when stored: 16 loads in flight per thread instead of 16
// serialised conditional ones) before the 32 MFMAs of step k; the k order of the MFMAs is that of the first version (bit-identical sums).
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
    __shared__ float s_a[GT * MLD], s_b[GT * MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const bool split = a.part != nullptr;
    const float* A = a.A + (split ? 0 : blockIdx.z * a.sA);
    const float* B = a.B + (split ? 0 : blockIdx.z * a.sB);
    float* C = a.C + (split ? 0 : blockIdx.z * a.sC);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const int kbeg = split ? blockIdx.z * a.kchunk : 0, kend = split ? min(a.K, kbeg + a.kchunk) : a.K;
    // stage: 64 x 32 elements of each operand, 8 per thread; the faster-varying thread index follows the contiguous dimension
    float ra[8], rb[8];
    unsigned oka = 0, okb = 0;
    auto gload = [&](int k0) {
        oka = okb = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 63; k = e >> 6; } else { k = e & 31; r = e >> 5; }          // A[m][k] (ta = 0) or A[k][m] (ta = 1)
            const int m = m0 + r, kk = k0 + k;
            oka |= (m < a.M && kk < kend ? 1u : 0u) << i;
            const int mc = min(m, a.M - 1), kc = min(kk, kend - 1);
            ra[i] = a.ta ? A[(long long)kc * a.lda + mc] : A[(long long)mc * a.lda + kc];
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }        // B[k][n] (tb = 0) or B[n][k] (tb = 1)
            const int n = n0 + c, kb = k0 + k2;
            okb |= (n < a.N && kb < kend ? 1u : 0u) << i;
            const int nc = min(n, a.N - 1), kbc = min(kb, kend - 1);
            rb[i] = a.tb ? B[(long long)nc * a.ldb + kbc] : B[(long long)kbc * a.ldb + nc];
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 63; k = e >> 6; } else { k = e & 31; r = e >> 5; }
            s_a[r * MLD + k] = (oka >> i) & 1u ? ra[i] : 0.f;
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }
            s_b[c * MLD + k2] = (okb >> i) & 1u ? rb[i] : 0.f;
        }
    };
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += MK) {
        lstore();
        __syncthreads();
        gload(min(k0 + MK, kend - 1));                   // always issued (past the end: clamped re-reads that are never stored)
#pragma unroll
        for (int ks = 0; ks < MK / 4; ++ks) {
            const float av = s_a[(16 * wave + li) * MLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bv = s_b[(16 * j + li) * MLD + 4 * ks + lk];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D layout: lane holds column n0 + 16 j + li, rows m0 + 16 wave + 4 lk + r
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + li;
        if (n >= a.N) continue;
        const float bz = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * lk + r;
            if (m >= a.M) continue;
            if (split) { a.part[((long long)blockIdx.z * a.M + m) * a.N + n] = acc[j][r]; continue; }
            float* p = C + (long long)m * a.ldc + n;
            *p = a.accumulate ? *p + (acc[j][r] + bz) : acc[j][r] + bz;
        }
    }
}

// gemm_f32_kernel on 32 x 32 tiles (a wave owns ONE 16 x 16 block) for the products whose 64 x 64 grid leaves most of the GPU empty -- the
// joint-token path: 317 launches per training step on 2 .. 126 workgroups.  There a K step is a serial chain inside each wave (8 x (5 LDS
// reads -> 4 MFMAs): ~1.3 us, measured; requesting more K at once does not shorten it), so the time goes down with the work per WAVE: a
// quarter of the MFMAs and a third of the LDS reads per step, four times the workgroups.  Same k order per element: the same bits.
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(GemmArgs a) {
    constexpr int ST = 32;
    __shared__ float s_a[ST * MLD], s_b[ST * MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
    const bool split = a.part != nullptr;                  // dir_gemm_f32_splitk: blockIdx.z = K chunk, raw partial tiles to part [chunk][M][N]
    const float* A = a.A + (split ? 0 : blockIdx.z * a.sA);
    const float* B = a.B + (split ? 0 : blockIdx.z * a.sB);
    float* C = a.C + (split ? 0 : blockIdx.z * a.sC);
    const int kbeg = split ? blockIdx.z * a.kchunk : 0, kend = split ? min(a.K, kbeg + a.kchunk) : a.K;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4, wr = 16 * (wave >> 1), wc = 16 * (wave & 1);
    float ra[4], rb[4];
    unsigned oka = 0, okb = 0;
    auto gload = [&](int k0) {
        oka = okb = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 31; k = e >> 5; } else { k = e & 31; r = e >> 5; }
            const int m = m0 + r, kk = k0 + k;
            oka |= (m < a.M && kk < kend ? 1u : 0u) << i;
            const int mc = min(m, a.M - 1), kc = min(kk, kend - 1);
            ra[i] = a.ta ? A[(long long)kc * a.lda + mc] : A[(long long)mc * a.lda + kc];
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 31; k2 = e >> 5; }
            const int n = n0 + c, kb = k0 + k2;
            okb |= (n < a.N && kb < kend ? 1u : 0u) << i;
            const int nc = min(n, a.N - 1), kbc = min(kb, kend - 1);
            rb[i] = a.tb ? B[(long long)nc * a.ldb + kbc] : B[(long long)kbc * a.ldb + nc];
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 31; k = e >> 5; } else { k = e & 31; r = e >> 5; }
            s_a[r * MLD + k] = (oka >> i) & 1u ? ra[i] : 0.f;
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 31; k2 = e >> 5; }
            s_b[c * MLD + k2] = (okb >> i) & 1u ? rb[i] : 0.f;
        }
    };
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += MK) {
        lstore();
        __syncthreads();
        gload(min(k0 + MK, kend - 1));
        float av[MK / 4], bv[MK / 4];
#pragma unroll
        for (int ks = 0; ks < MK / 4; ++ks) { av[ks] = s_a[(wr + li) * MLD + 4 * ks + lk]; bv[ks] = s_b[(wc + li) * MLD + 4 * ks + lk]; }
#pragma unroll
        for (int ks = 0; ks < MK / 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wc + li;
    if (n < a.N) {
        const float bz = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr + 4 * lk + r;
            if (m >= a.M) continue;
            if (split) { a.part[((long long)blockIdx.z * a.M + m) * a.N + n] = acc[r]; continue; }
            float* p = C + (long long)m * a.ldc + n;
            *p = a.accumulate ? *p + (acc[r] + bz) : acc[r] + bz;
        }
    }
}

// The same tile walk over GROUPS of displaced operands (dir_gemm_f32_grouped): group g = (gy, gx) of an ny x nx grid displaces A / B / C by
// gy * *_y + gx * *_x elements (a convolution tap as a pointer shift: csrc/bonefuse_bwd.hip).  reduce = 0: blockIdx.z = batch * groups + g,
// every group its own product (hundreds of co-resident workgroups instead of one launch per group with a wave per SIMD); reduce = 1: one
// workgroup sums the groups' products in group order in its accumulators (one write of C instead of a read-modify-write per group).
struct GemmGroups { int ny, nx, reduce; long long a_y, a_x, b_y, b_x, c_y, c_x; };
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(GemmArgs a, GemmGroups gr) {
    __shared__ float s_a[GT * MLD], s_b[GT * MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int ng = gr.ny * gr.nx;
    const int bz = gr.reduce ? blockIdx.z : blockIdx.z / ng, g_own = gr.reduce ? 0 : blockIdx.z - bz * ng;
    const float* A0 = a.A + bz * a.sA;
    const float* B0 = a.B + bz * a.sB;
    float* C = a.C + bz * a.sC;
    if (!gr.reduce) { const int gy = g_own / gr.nx, gx = g_own - gy * gr.nx; C += gy * gr.c_y + gx * gr.c_x; }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const int spg = (a.K + MK - 1) / MK, steps = gr.reduce ? spg * ng : spg;
    float ra[8], rb[8];
    unsigned oka = 0, okb = 0;
    auto gload = [&](int t) {
        const int gi = gr.reduce ? t / spg : 0, k0 = (t - gi * spg) * MK, g = gr.reduce ? gi : g_own;
        const int gy = g / gr.nx, gx = g - gy * gr.nx;
        const float* A = A0 + gy * gr.a_y + gx * gr.a_x;
        const float* B = B0 + gy * gr.b_y + gx * gr.b_x;
        oka = okb = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 63; k = e >> 6; } else { k = e & 31; r = e >> 5; }
            const int m = m0 + r, kk = k0 + k;
            oka |= (m < a.M && kk < a.K ? 1u : 0u) << i;
            const int mc = min(m, a.M - 1), kc = min(kk, a.K - 1);
            ra[i] = a.ta ? A[(long long)kc * a.lda + mc] : A[(long long)mc * a.lda + kc];
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }
            const int n = n0 + c, kb = k0 + k2;
            okb |= (n < a.N && kb < a.K ? 1u : 0u) << i;
            const int nc = min(n, a.N - 1), kbc = min(kb, a.K - 1);
            rb[i] = a.tb ? B[(long long)nc * a.ldb + kbc] : B[(long long)kbc * a.ldb + nc];
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int r, k;
            if (a.ta) { r = e & 63; k = e >> 6; } else { k = e & 31; r = e >> 5; }
            s_a[r * MLD + k] = (oka >> i) & 1u ? ra[i] : 0.f;
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }
            s_b[c * MLD + k2] = (okb >> i) & 1u ? rb[i] : 0.f;
        }
    };
    gload(0);
    for (int t = 0; t < steps; ++t) {
        lstore();
        __syncthreads();
        gload(min(t + 1, steps - 1));                    // the last step's operands again past the end (never stored)
#pragma unroll
        for (int ks = 0; ks < MK / 4; ++ks) {
            const float av = s_a[(16 * wave + li) * MLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bv = s_b[(16 * j + li) * MLD + 4 * ks + lk];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + li;
        if (n >= a.N) continue;
        const float bz_ = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * lk + r;
            if (m >= a.M) continue;
            float* p = C + (long long)m * a.ldc + n;
            *p = a.accumulate ? *p + (acc[j][r] + bz_) : acc[j][r] + bz_;
        }
    }
}

// Grouped walk for SHORT outputs: a tile of 16 * MB rows x 64 columns, the four waves split the columns (16 each) and every wave holds all MB
// row blocks -- M = 80 (the bone fusion's 80 bone ends) fills an 80-row tile exactly; on the 64 x 64 tiles above it pays for 128 rows.
template <int MB>
__global__ __launch_bounds__(256) void gemm_f32_grouped_short_kernel(GemmArgs a, GemmGroups gr) {
    constexpr int TM = 16 * MB, NA = (TM * MK + 255) / 256;
    __shared__ float s_a[TM * MLD], s_b[GT * MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * GT;
    const int ng = gr.ny * gr.nx;
    const int bz = gr.reduce ? blockIdx.z : blockIdx.z / ng, g_own = gr.reduce ? 0 : blockIdx.z - bz * ng;
    const float* A0 = a.A + bz * a.sA;
    const float* B0 = a.B + bz * a.sB;
    float* C = a.C + bz * a.sC;
    if (!gr.reduce) { const int gy = g_own / gr.nx, gx = g_own - gy * gr.nx; C += gy * gr.c_y + gx * gr.c_x; }
    f32x4 acc[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const int spg = (a.K + MK - 1) / MK, steps = gr.reduce ? spg * ng : spg;
    float ra[NA], rb[8];
    unsigned oka = 0, okb = 0;
    auto a_rk = [&](int e, int& r, int& k) { if (a.ta) { k = e / TM; r = e - k * TM; } else { k = e & 31; r = e >> 5; } };
    auto gload = [&](int t) {
        const int gi = gr.reduce ? t / spg : 0, k0 = (t - gi * spg) * MK, g = gr.reduce ? gi : g_own;
        const int gy = g / gr.nx, gx = g - gy * gr.nx;
        const float* A = A0 + gy * gr.a_y + gx * gr.a_x;
        const float* B = B0 + gy * gr.b_y + gx * gr.b_x;
        oka = okb = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = min(tid + 256 * i, TM * MK - 1);
            int r, k;
            a_rk(e, r, k);
            const int m = m0 + r, kk = k0 + k;
            oka |= (m < a.M && kk < a.K ? 1u : 0u) << i;
            const int mc = min(m, a.M - 1), kc = min(kk, a.K - 1);
            ra[i] = a.ta ? A[(long long)kc * a.lda + mc] : A[(long long)mc * a.lda + kc];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }
            const int n = n0 + c, kb = k0 + k2;
            okb |= (n < a.N && kb < a.K ? 1u : 0u) << i;
            const int nc = min(n, a.N - 1), kbc = min(kb, a.K - 1);
            rb[i] = a.tb ? B[(long long)nc * a.ldb + kbc] : B[(long long)kbc * a.ldb + nc];
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            if (e < TM * MK) {
                int r, k;
                a_rk(e, r, k);
                s_a[r * MLD + k] = (oka >> i) & 1u ? ra[i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i;
            int c, k2;
            if (a.tb) { k2 = e & 31; c = e >> 5; } else { c = e & 63; k2 = e >> 6; }
            s_b[c * MLD + k2] = (okb >> i) & 1u ? rb[i] : 0.f;
        }
    };
    gload(0);
    for (int t = 0; t < steps; ++t) {
        lstore();
        __syncthreads();
        gload(min(t + 1, steps - 1));
#pragma unroll
        for (int ks = 0; ks < MK / 4; ++ks) {
            const float bv = s_b[(16 * wave + li) * MLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float av = s_a[(16 * j + li) * MLD + 4 * ks + lk];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D layout: lane holds column n0 + 16 wave + li, rows m0 + 16 j + 4 lk + r
    const int n = n0 + 16 * wave + li;
    if (n < a.N) {
        const float bz_ = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * j + 4 * lk + r;
                if (m >= a.M) continue;
                float* p = C + (long long)m * a.ldc + n;
                *p = a.accumulate ? *p + (acc[j][r] + bz_) : acc[j][r] + bz_;
            }
    }
}

// split-K second pass: C[m][n] = (accumulate ? C : 0) + (sum over the chunks IN ORDER of part[chunk][m][n]) + bias[n]   (deterministic)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* part, const float* bias, float* C, int M, int N, int ldc, int chunks, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    float s = 0.f;
    s = dir::sum_in_order(part + i, (long long)M * N, chunks, s);
    s += bias ? bias[n] : 0.f;
    float* p = C + (long long)m * ldc + n;
    *p = accumulate ? *p + s : s;
}

// column sums of X [R, N] (row stride ld): one thread per column, rows in order
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, float* out, int R, int N, int ld, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float acc = dir::sum_in_order(x + n, ld, R, 0.f);
    out[n] = accumulate ? out[n] + acc : acc;
}

// A few hundred to a few thousand rows (the joint-token path's bias gradients: R = B * 21 .. 42): ONE launch -- 16 column quads x 64 row lanes,
// every lane its rows in order with four loads in flight, the lanes added in lane order -- instead of chunk partials + a second launch.
__global__ __launch_bounds__(1024) void colsum_mid_kernel(const float* x, float* out, int R, int N, int ld, int accumulate) {
    __shared__ float4 s[64][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c = min(blockIdx.x * 64 + cq * 4, N - 4);
    auto at = [&](int r) { return *reinterpret_cast<const float4*>(x + (long long)r * ld + c); };
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = rl;
    for (; r + 192 < R; r += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = at(r + 64 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < R; r += 64) { const float4 v = at(r); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    s[rl][cq] = a;
    __syncthreads();
    if (rl != 0 || blockIdx.x * 64 + cq * 4 >= N) return;
    float4 t = s[0][cq];
    for (int l = 1; l < 64; ++l) { const float4 q = s[l][cq]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
    float4* o = reinterpret_cast<float4*>(out + c);
    if (accumulate) { const float4 p = *o; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
    *o = t;
}

// ------------------------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row, C <= 256 (4 values per lane).  Statistics as torch: mean, biased variance (two passes in registers).
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                                                           int R, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + (long long)row * C;
    float v[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; v[i] = c < C ? xr[c] : 0.f; s += v[i]; }
    const float mu = dir::wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; const float d = c < C ? v[i] - mu : 0.f; q += d * d; }
    const float rs = 1.f / sqrtf(dir::wave_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; if (c < C) y[(long long)row * C + c] = (v[i] - mu) * rs * w[c] + b[c]; }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// g x = rstd (g^ - mean(g^) - x^ mean(g^ x^)), g^ = g y * w
__global__ __launch_bounds__(256) void layernorm_bwd_x_kernel(const float* gy, const float* x, const float* w, const float* mean, const float* rstd,
                                                             float* gx, int R, int C, int accumulate) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float mu = mean[row], rs = rstd[row];
    float gh[4], xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        gh[i] = c < C ? gy[(long long)row * C + c] * w[c] : 0.f;
        xh[i] = c < C ? (x[(long long)row * C + c] - mu) * rs : 0.f;
        s1 += gh[i]; s2 += gh[i] * xh[i];
    }
    s1 = dir::wave_sum(s1) / C; s2 = dir::wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < C) { float* p = gx + (long long)row * C + c; const float v = rs * (gh[i] - s1 - xh[i] * s2); *p = accumulate ? *p + v : v; }
    }
}
// g w[c] = sum_rows g y x^ ; g b[c] = sum_rows g y : a workgroup = 16 columns x 16 row lanes (lane l takes rows l, l + 16, ...), the lanes'
// partial sums added in lane order (deterministic)
__global__ __launch_bounds__(256) void layernorm_bwd_wb_kernel(const float* gy, const float* x, const float* mean, const float* rstd, float* gw, float* gb,
                                                              int R, int C, int accumulate) {
    __shared__ float s_w[16][17], s_b[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float aw = 0.f, ab = 0.f;
    if (c < C) {
        // four rows' loads in flight before their (in-order) accumulation: the loop was one dependent HBM round trip per row -- 84 of them for the
        // 1344 token rows, 26 us for 1.4 MB.  Same sums in the same order.
        int r = rl;
        for (; r + 48 < R; r += 64) {
            float g[4], xv[4], mu[4], rs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (long long)(r + 16 * u) * C + c;
                g[u] = gy[o]; xv[u] = x[o]; mu[u] = mean[r + 16 * u]; rs[u] = rstd[r + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { aw = fmaf(g[u], (xv[u] - mu[u]) * rs[u], aw); ab += g[u]; }
        }
        for (; r < R; r += 16) {
            const float g = gy[(long long)r * C + c];
            aw = fmaf(g, (x[(long long)r * C + c] - mean[r]) * rstd[r], aw);
            ab += g;
        }
    }
    s_w[rl][cl] = aw; s_b[rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        float tw = 0.f, tb = 0.f;
        for (int l = 0; l < 16; ++l) { tw += s_w[l][cl]; tb += s_b[l][cl]; }
        gw[c] = accumulate ? gw[c] + tw : tw;
        gb[c] = accumulate ? gb[c] + tb : tb;
    }
}

// ------------------------------------------------------------------------------------------------------------------ GELU (exact erf)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* x, float* y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* gy, const float* x, float* gx, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
        gx[i] = gy[i] * (cdf + v * pdf);
    }
}

// ------------------------------------------------------------------------------------------------------------------ attention
// qkv [B, T, 3, H, D] (nn.Linear output of mixSTE.py:77), T <= 64, D = 32.  One workgroup per (sample, head).
constexpr int AT = 64, AD = 32;
struct AttnArgs { const float* qkv; float* probs; float* out; const float* gout; float* gqkv; int B, T, H; float scale; };

__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs a) {
    __shared__ float s_q[AT * (AD + 1)], s_k[AT * (AD + 1)], s_v[AT * (AD + 1)], s_p[AT * (AT + 1)];
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H, tid = threadIdx.x, T = a.T, C = a.H * AD;
    for (int e = tid; e < T * AD; e += 256) {
        const int t = e / AD, d = e - t * AD;
        const float* p = a.qkv + ((long long)(b * T + t) * 3) * C + h * AD + d;
        s_q[t * (AD + 1) + d] = p[0]; s_k[t * (AD + 1) + d] = p[C]; s_v[t * (AD + 1) + d] = p[2 * C];
    }
    __syncthreads();
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e - i * T;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < AD; ++d) acc = fmaf(s_q[i * (AD + 1) + d], s_k[j * (AD + 1) + d], acc);
        s_p[i * (AT + 1) + j] = acc * a.scale;
    }
    __syncthreads();
    if (tid < T) {            // softmax of row tid (max-subtracted, as torch)
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) mx = fmaxf(mx, s_p[tid * (AT + 1) + j]);
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float ev = expf(s_p[tid * (AT + 1) + j] - mx); s_p[tid * (AT + 1) + j] = ev; sum += ev; }
        for (int j = 0; j < T; ++j) s_p[tid * (AT + 1) + j] /= sum;
    }
    __syncthreads();
    if (a.probs)
        for (int e = tid; e < T * T; e += 256) a.probs[((long long)(b * a.H + h) * T + e / T) * T + e % T] = s_p[(e / T) * (AT + 1) + e % T];
    for (int e = tid; e < T * AD; e += 256) {
        const int i = e / AD, d = e - i * AD;
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(s_p[i * (AT + 1) + j], s_v[j * (AD + 1) + d], acc);
        a.out[(long long)(b * T + i) * C + h * AD + d] = acc;            // (attn @ v).transpose(1, 2).reshape(B, N, C)
    }
}

// g v = P^T g o ; g P = g o v^T ; g S = P (g P - rowsum(g P P)) ; g q = g S k scale ; g k = g S^T q scale
__global__ __launch_bounds__(256) void attention_bwd_kernel(AttnArgs a) {
    __shared__ float s_q[AT * (AD + 1)], s_k[AT * (AD + 1)], s_v[AT * (AD + 1)], s_go[AT * (AD + 1)], s_p[AT * (AT + 1)], s_gs[AT * (AT + 1)];
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H, tid = threadIdx.x, T = a.T, C = a.H * AD;
    for (int e = tid; e < T * AD; e += 256) {
        const int t = e / AD, d = e - t * AD;
        const float* p = a.qkv + ((long long)(b * T + t) * 3) * C + h * AD + d;
        s_q[t * (AD + 1) + d] = p[0]; s_k[t * (AD + 1) + d] = p[C]; s_v[t * (AD + 1) + d] = p[2 * C];
        s_go[t * (AD + 1) + d] = a.gout[(long long)(b * T + t) * C + h * AD + d];
    }
    for (int e = tid; e < T * T; e += 256) s_p[(e / T) * (AT + 1) + e % T] = a.probs[((long long)(b * a.H + h) * T + e / T) * T + e % T];
    __syncthreads();
    for (int e = tid; e < T * T; e += 256) {          // g P
        const int i = e / T, j = e - i * T;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < AD; ++d) acc = fmaf(s_go[i * (AD + 1) + d], s_v[j * (AD + 1) + d], acc);
        s_gs[i * (AT + 1) + j] = acc;
    }
    __syncthreads();
    if (tid < T) {            // softmax backward, row tid; the scale of S = q k^T * scale folded in
        float dot = 0.f;
        for (int j = 0; j < T; ++j) dot = fmaf(s_gs[tid * (AT + 1) + j], s_p[tid * (AT + 1) + j], dot);
        for (int j = 0; j < T; ++j) s_gs[tid * (AT + 1) + j] = s_p[tid * (AT + 1) + j] * (s_gs[tid * (AT + 1) + j] - dot) * a.scale;
    }
    __syncthreads();
    for (int e = tid; e < T * AD; e += 256) {
        const int i = e / AD, d = e - i * AD;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int j = 0; j < T; ++j) {
            gq = fmaf(s_gs[i * (AT + 1) + j], s_k[j * (AD + 1) + d], gq);
            gk = fmaf(s_gs[j * (AT + 1) + i], s_q[j * (AD + 1) + d], gk);
            gv = fmaf(s_p[j * (AT + 1) + i], s_go[j * (AD + 1) + d], gv);
        }
        float* p = a.gqkv + ((long long)(b * T + i) * 3) * C + h * AD + d;
        p[0] = gq; p[C] = gk; p[2 * C] = gv;
    }
}

// ------------------------------------------------------------------------------------------------------------------ BatchNorm (training)
// x [R, C] (channels last: rows = samples x positions).  One thread per channel, rows in order (R is small on the token path).
// the BatchNorm output as every kernel here forms it (the ReLU mask of the backward pass re-computes exactly this value from x)
__device__ __forceinline__ float bn_value(float x, float mu, float rs, float g, float be) { return fmaf((x - mu) * rs, g, be); }
__global__ __launch_bounds__(256) void bn_train_fwd_kernel(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd,
                                                          float* running_mean, float* running_var, int R, int C, int ld, float eps, float momentum, int relu, const float* res) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += x[(long long)r * ld + c];
    const float mu = s / R;
    float q = 0.f;
    for (int r = 0; r < R; ++r) { const float d = x[(long long)r * ld + c] - mu; q = fmaf(d, d, q); }
    const float var = q / R, rs = 1.f / sqrtf(var + eps);
    const float g = w ? w[c] : 1.f, be = b ? b[c] : 0.f;
    for (int r = 0; r < R; ++r) {
        float v = bn_value(x[(long long)r * ld + c], mu, rs, g, be);
        if (res) v += res[(long long)r * ld + c];
        y[(long long)r * ld + c] = relu ? fmaxf(v, 0.f) : v;
    }
    save_mean[c] = mu; save_rstd[c] = rs;
    if (running_mean) {       // torch: running = (1 - momentum) running + momentum stat, with the UNBIASED variance
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (R > 1 ? q / (R - 1) : var);
    }
}
// g x = w rstd (g y - mean(g y) - x^ mean(g y x^)) ; g w = sum g y x^ ; g b = sum g y
__global__ __launch_bounds__(256) void bn_train_bwd_kernel(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                                          float* gx, float* gw, float* gb, int R, int C, int ld, int relu) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mu = save_mean[c], rs = save_rstd[c], g = w ? w[c] : 1.f, be = b ? b[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < R; ++r) {
        float gv = gy[(long long)r * ld + c];
        if (relu && !(bn_value(x[(long long)r * ld + c], mu, rs, g, be) > 0.f)) gv = 0.f;
        s1 += gv; s2 = fmaf(gv, (x[(long long)r * ld + c] - mu) * rs, s2);
    }
    if (gw) gw[c] = s2;
    if (gb) gb[c] = s1;
    const float m1 = s1 / R, m2 = s2 / R;
    if (gx)
        for (int r = 0; r < R; ++r) {
            float gv = gy[(long long)r * ld + c];
            if (relu && !(bn_value(x[(long long)r * ld + c], mu, rs, g, be) > 0.f)) gv = 0.f;
            gx[(long long)r * ld + c] = g * rs * (gv - m1 - (x[(long long)r * ld + c] - mu) * rs * m2);
        }
}

// Large R (BatchNorm2d over feature maps: R = B*H*W up to 10^6 rows): the column reductions are cut into row chunks -- a workgroup =
// 64 channels x 4 row lanes over one chunk, lanes combined in a fixed order -- and a per-channel finalise adds the chunk partials in
// chunk order: deterministic, no atomics.  Same formulas as the one-thread-per-channel kernels above (two-pass variance).
constexpr int BN_CHUNK_ROWS = 256, BN_SMALL_R = 512;
// mode 0: p1 = sum x | mode 1: p1 = sum (x - mu)^2 | mode 2: p1 = sum gy, p2 = sum gy (x - mu) rs
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* x, const float* gy, const float* mu, const float* rs, float* p1, float* p2,
                                                        int R, int C, int ld, int mode, const float* w = nullptr, const float* be = nullptr, int relu = 0) {
    __shared__ float s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, ch = blockIdx.y;
    const int r0 = ch * BN_CHUNK_ROWS, r1 = min(R, r0 + BN_CHUNK_ROWS);
    float a = 0.f, b = 0.f;
    if (c < C) {
        const float m = mode ? mu[c] : 0.f, k = mode == 2 ? rs[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float v = x[(long long)r * ld + c];
            if (mode == 0) a += v;
            else if (mode == 1) { const float d = v - m; a = fmaf(d, d, a); }
            else {
                float g = gy[(long long)r * ld + c];
                if (relu && !(bn_value(v, m, k, w ? w[c] : 1.f, be ? be[c] : 0.f) > 0.f)) g = 0.f;
                a += g; b = fmaf(g, (v - m) * k, b);
            }
        }
    }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        p1[(long long)ch * C + c] = ((s1[0][cl] + s1[1][cl]) + s1[2][cl]) + s1[3][cl];
        if (mode == 2) p2[(long long)ch * C + c] = ((s2[0][cl] + s2[1][cl]) + s2[2][cl]) + s2[3][cl];
    }
}
// the same partial sums with 16-byte loads: 256 threads = 16 channel quads (64 channels) x 16 row lanes; lanes combined in order
__global__ __launch_bounds__(256) void bn_partial4_kernel(const float* x, const float* gy, const float* mu, const float* rs, float* p1, float* p2,
                                                         int R, int C, int ld, int mode) {
    __shared__ float4 s1[16][16], s2[16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 64 + cq * 4, ch = blockIdx.y;
    const int r0 = ch * BN_CHUNK_ROWS, r1 = min(R, r0 + BN_CHUNK_ROWS);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c < C) {
        const float4 m = mode ? *reinterpret_cast<const float4*>(mu + c) : a;
        const float4 k = mode == 2 ? *reinterpret_cast<const float4*>(rs + c) : a;
        auto step = [&](const float4 v, const float4 g) {
            if (mode == 0) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            else if (mode == 1) {
                const float dx = v.x - m.x, dy = v.y - m.y, dz = v.z - m.z, dw = v.w - m.w;
                a.x = fmaf(dx, dx, a.x); a.y = fmaf(dy, dy, a.y); a.z = fmaf(dz, dz, a.z); a.w = fmaf(dw, dw, a.w);
            } else {
                a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
                b.x = fmaf(g.x, (v.x - m.x) * k.x, b.x); b.y = fmaf(g.y, (v.y - m.y) * k.y, b.y);
                b.z = fmaf(g.z, (v.z - m.z) * k.z, b.z); b.w = fmaf(g.w, (v.w - m.w) * k.w, b.w);
            }
        };
        // four rows' loads in flight, accumulated in row order (the same sums): a chunk was 16 dependent round trips per thread
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const float4*>(x + (long long)(r + 16 * u) * ld + c);
                g[u] = mode == 2 ? *reinterpret_cast<const float4*>(gy + (long long)(r + 16 * u) * ld + c) : v[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) step(v[u], g[u]);
        }
        for (; r < r1; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long long)r * ld + c);
            step(v, mode == 2 ? *reinterpret_cast<const float4*>(gy + (long long)r * ld + c) : v);
        }
    }
    s1[rl][cq] = a; s2[rl][cq] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        float4 t = s1[0][cq], u = s2[0][cq];
        for (int l = 1; l < 16; ++l) {
            const float4 q = s1[l][cq], w = s2[l][cq];
            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
            u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
        }
        *reinterpret_cast<float4*>(p1 + (long long)ch * C + c) = t;
        if (mode == 2) *reinterpret_cast<float4*>(p2 + (long long)ch * C + c) = u;
    }
}
// Loads and stores that are coherent across the GPU's eight L2s WITHOUT a fence (agent-scope monotonic: `sc1` on gfx950): what workgroups of one
// launch hand each other (the one-launch BatchNorm below).  CO = false: plain accesses.
template <bool CO> __device__ __forceinline__ float ld_co(const float* p) {
    if constexpr (CO) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool CO> __device__ __forceinline__ void st_co(float* p, float v) {
    if constexpr (CO) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool CO> __device__ __forceinline__ float4 ld_co4(const float* p) {
    if constexpr (CO) return make_float4(ld_co<true>(p), ld_co<true>(p + 1), ld_co<true>(p + 2), ld_co<true>(p + 3));
    else return *reinterpret_cast<const float4*>(p);
}
template <bool CO> __device__ __forceinline__ void st_co4(float* p, const float4 v) {
    if constexpr (CO) { st_co<true>(p, v.x); st_co<true>(p + 1, v.y); st_co<true>(p + 2, v.z); st_co<true>(p + 3, v.w); }
    else *reinterpret_cast<float4*>(p) = v;
}
// dir::sum_in_order (the same order, the same bits) over coherent loads
template <bool CO> __device__ __forceinline__ float sum_in_order_co(const float* p, long long stride, int n, float s) {
    if constexpr (!CO) return dir::sum_in_order(p, stride, n, s);
    int c = 0;
    for (; c + 8 <= n; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld_co<true>(p + (long long)(c + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < n; ++c) s += ld_co<true>(p + (long long)c * stride);
    return s;
}
// sums of the chunk partials: a workgroup = 16 channels x 16 lanes (lane l adds chunks l, l + 16, ... in order; lanes combined in lane order)
template <bool CO = false>
__device__ __forceinline__ float chunk_sum16(const float* p, int chunks, int C, int c, float (&s)[16][17]) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float a = 0.f;
    if (c < C && rl < chunks) a = sum_in_order_co<CO>(p + (long long)rl * C + c, 16ll * C, (chunks - rl + 15) / 16, a);
    s[rl][cl] = a;
    __syncthreads();
    float t = 0.f;
    for (int l = 0; l < 16; ++l) t += s[l][cl];
    return t;
}
__global__ __launch_bounds__(256) void bn_colsum_chunks_kernel(const float* p, float* out, int chunks, int C, float scale) {
    __shared__ float s[16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float t = chunk_sum16(p, chunks, C, c, s);
    if ((threadIdx.x >> 4) == 0 && c < C) out[c] = t * scale;
}
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* p, const float* mean, float* save_mean, float* save_rstd, float* running_mean,
                                                               float* running_var, int chunks, int R, int C, float eps, float momentum) {
    __shared__ float s[16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float q = chunk_sum16(p, chunks, C, c, s);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    const float mu = mean[c], var = q / R;
    save_mean[c] = mu; save_rstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (R > 1 ? q / (R - 1) : var);
    }
}
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* x, const float* w, const float* b, const float* mu, const float* rs, float* y,
                                                          long long n, int C, int ld, int relu, const float* res) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long o = (i / C) * ld + c;
    float v = bn_value(x[o], mu[c], rs[c], w ? w[c] : 1.f, b ? b[c] : 0.f);
    if (res) v += res[o];
    y[o] = relu ? fmaxf(v, 0.f) : v;
}
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* gy, const float* x, const float* w, const float* b, const float* mu, const float* rs,
                                                          const float* s1, const float* s2, float* gx, long long n, int R, int C, int ld, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long o = (i / C) * ld + c;
    const float m1 = s1[c] / R, m2 = s2[c] / R, g = w ? w[c] : 1.f;
    float gv = gy[o];
    if (relu && !(bn_value(x[o], mu[c], rs[c], g, b ? b[c] : 0.f) > 0.f)) gv = 0.f;
    gx[o] = g * rs[c] * (gv - m1 - (x[o] - mu[c]) * rs[c] * m2);
}

// frozen statistics (dir_bn_frozen_*): save_mean = running_mean, save_rstd = 1 / sqrt(running_var + eps)
__global__ __launch_bounds__(256) void bn_frozen_stats_kernel(const float* rm, const float* rv, float* mu, float* rs, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) { mu[c] = rm[c]; rs[c] = 1.f / sqrtf(rv[c] + eps); }
}
__global__ __launch_bounds__(256) void zero_f32_kernel(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ---- 16-byte versions for C % 4 == 0 (every BatchNorm2d of the path).  Forward statistics in ONE pass over HBM: a workgroup (64 channels x
// one 256-row chunk) forms the chunk's column sums, then the squared deviations from the CHUNK mean on a second read that hits L2 (a chunk
// is 64 KB); the finalise kernel combines the chunks exactly: var = sum_k [M2_k + n_k (mean_k - mean)^2] / R, in chunk order.
template <bool CO = false>
__device__ __forceinline__ void bn_stats4_tile(const float* x, float* p1, float* p2, int R, int C, int ld, int cg, int ch, float4 (&s1)[16][16]) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c = cg * 64 + cq * 4;
    const int r0 = ch * BN_CHUNK_ROWS, r1 = min(R, r0 + BN_CHUNK_ROWS);
    const bool on = c < C;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {                                              // (four rows' loads in flight, summed in row order: see bn_partial4_kernel)
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (long long)(r + 16 * u) * ld + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; r < r1; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long long)r * ld + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    s1[rl][cq] = a;
    __syncthreads();
    float4 t = s1[0][cq];
    for (int l = 1; l < 16; ++l) { const float4 q = s1[l][cq]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
    __syncthreads();
    const float inv = 1.f / (r1 - r0);
    const float4 m = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        auto sq = [&](const float4 v) {
            const float dx = v.x - m.x, dy = v.y - m.y, dz = v.z - m.z, dw = v.w - m.w;
            b.x = fmaf(dx, dx, b.x); b.y = fmaf(dy, dy, b.y); b.z = fmaf(dz, dz, b.z); b.w = fmaf(dw, dw, b.w);
        };
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (long long)(r + 16 * u) * ld + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) sq(v[u]);
        }
        for (; r < r1; r += 16) sq(*reinterpret_cast<const float4*>(x + (long long)r * ld + c));
    }
    s1[rl][cq] = b;
    __syncthreads();
    if (rl == 0 && on) {
        float4 u = s1[0][cq];
        for (int l = 1; l < 16; ++l) { const float4 q = s1[l][cq]; u.x += q.x; u.y += q.y; u.z += q.z; u.w += q.w; }
        st_co4<CO>(p1 + (long long)ch * C + c, t);
        st_co4<CO>(p2 + (long long)ch * C + c, u);
    }
}
__global__ __launch_bounds__(256) void bn_stats4_kernel(const float* x, float* p1, float* p2, int R, int C, int ld) {
    __shared__ float4 s1[16][16];
    bn_stats4_tile(x, p1, p2, R, C, ld, blockIdx.x, blockIdx.y, s1);
}
// the statistics of 16 channels (c16 = first channel / 16) from the chunk partials
template <bool CO = false>
__device__ __forceinline__ void bn_stats_combine16(const float* p1, const float* p2, float* save_mean, float* save_rstd, float* running_mean,
                                                   float* running_var, int chunks, int R, int C, float eps, float momentum, int c16, float (&s)[16][17],
                                                   int crows = BN_CHUNK_ROWS) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = c16 * 16 + cl;
    const float mu = chunk_sum16<CO>(p1, chunks, C, c, s) / R;
    __syncthreads();
    float a = 0.f;
    if (c < C)
        for (int k = rl; k < chunks; k += 16) {
            const int nk = min(crows, R - k * crows);
            const float d = ld_co<CO>(p1 + (long long)k * C + c) / nk - mu;
            a += fmaf((float)nk * d, d, ld_co<CO>(p2 + (long long)k * C + c));
        }
    s[rl][cl] = a;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    float q = 0.f;
    for (int l = 0; l < 16; ++l) q += s[l][cl];
    const float var = q / R;
    st_co<CO>(save_mean + c, mu); st_co<CO>(save_rstd + c, 1.f / sqrtf(var + eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (R > 1 ? q / (R - 1) : var);
    }
}
__global__ __launch_bounds__(256) void bn_stats_combine_kernel(const float* p1, const float* p2, float* save_mean, float* save_rstd, float* running_mean,
                                                              float* running_var, int chunks, int R, int C, float eps, float momentum) {
    __shared__ float s[16][17];
    bn_stats_combine16(p1, p2, save_mean, save_rstd, running_mean, running_var, chunks, R, C, eps, momentum, blockIdx.x, s);
}
// round 5 (dir_bn_train_stats): the statistics only, plus the affine form the CONSUMING convolution applies where it reads the map --
// pre_scale = gamma rstd, pre_shift = beta - mean gamma rstd (dir_conv2d_forward's pre-activation, dir_split_f16_forward's, dir_conv2d_wgrad_f16x3_pre's):
// the normalised map is never written.  Same statistics kernels, same bits of mean / rstd / running statistics as dir_bn_train_forward.
__global__ __launch_bounds__(256) void bn_stats_combine_pre_kernel(const float* p1, const float* p2, const float* w, const float* b, float* save_mean, float* save_rstd,
                                                                  float* pre_scale, float* pre_shift, float* running_mean, float* running_var, int chunks, int R, int C,
                                                                  float eps, float momentum, int crows) {
    __shared__ float s[16][17];
    bn_stats_combine16(p1, p2, save_mean, save_rstd, running_mean, running_var, chunks, R, C, eps, momentum, blockIdx.x, s, crows);
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    if (rl != 0 || c >= C || !pre_scale) return;         // (the thread that wrote save_mean[c] / save_rstd[c] above)
    const float g = w ? w[c] : 1.f, be = b ? b[c] : 0.f, k = g * save_rstd[c];
    pre_scale[c] = k;
    pre_shift[c] = fmaf(-save_mean[c], k, be);
}
// ---- SyncBN (round 5; SURVEY.md 8e "optionally offer SyncBN": the reference trains 64 images on ONE GPU, config.py:13-15 -- 8 x 32 changes the
// BatchNorm batch unless the statistics are pooled).  The local part of the statistics: this rank's mean and M2 = sum (x - mean_local)^2 per
// channel from the same chunk partials as bn_stats_combine_kernel, combined in the same (chunk) order.
__global__ __launch_bounds__(256) void bn_stats_local_kernel(const float* p1, const float* p2, float* mean_out, float* m2_out, int chunks, int R, int C) {
    __shared__ float s[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    const float mu = chunk_sum16(p1, chunks, C, c, s) / R;
    __syncthreads();
    float a = 0.f;
    if (c < C)
        for (int k = rl; k < chunks; k += 16) {
            const int nk = min(BN_CHUNK_ROWS, R - k * BN_CHUNK_ROWS);
            const float d = p1[(long long)k * C + c] / nk - mu;
            a += fmaf((float)nk * d, d, p2[(long long)k * C + c]);
        }
    s[rl][cl] = a;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    float q = 0.f;
    for (int l = 0; l < 16; ++l) q += s[l][cl];
    mean_out[c] = mu; m2_out[c] = q;
}
// The ranks' parts [W][2 C + 4] = (mean_r [C] | M2_r [C] | rows_r, 0, 0, 0), gathered in rank order, pooled exactly (Chan): n = sum n_r,
// mean = sum n_r mean_r / n, M2 = sum [M2_r + n_r (mean_r - mean)^2]; every rank runs this on the same gathered bytes -> the same statistics
// everywhere.  Writes the pooled mean and BIASED variance (what dir_bn_frozen_forward normalises with) and updates the running statistics with
// the unbiased one over the pooled count, like nn.SyncBatchNorm.
__global__ __launch_bounds__(256) void bn_sync_combine_kernel(const float* parts, int W, int C, float* mean_out, float* var_out, float* running_mean,
                                                             float* running_var, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int stride = 2 * C + 4;
    float n = 0.f, sm = 0.f;
    for (int r = 0; r < W; ++r) { const float nr = parts[(long long)r * stride + 2 * C]; n += nr; sm = fmaf(nr, parts[(long long)r * stride + c], sm); }
    const float mu = sm / n;
    float q = 0.f;
    for (int r = 0; r < W; ++r) {
        const float nr = parts[(long long)r * stride + 2 * C], d = parts[(long long)r * stride + c] - mu;
        q += fmaf(nr * d, d, parts[(long long)r * stride + C + c]);
    }
    mean_out[c] = mu; var_out[c] = q / n;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? q / (n - 1.f) : q / n);
    }
}
// backward: the two column sums of this rank (t1 = sum g, t2 = sum g xhat, g = gy under the ReLU mask) side by side in sums [2 C]
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* p1, const float* p2, float* sums, int chunks, int C) {
    __shared__ float s[16][17];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const float a = chunk_sum16(p1, chunks, C, c, s);
    __syncthreads();
    const float b = chunk_sum16(p2, chunks, C, c, s);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    sums[c] = a; sums[C + c] = b;
}
// one thread = 4 channels of 4 rows
__global__ __launch_bounds__(256) void bn_apply_fwd4_kernel(const float* x, const float* w, const float* b, const float* mu, const float* rs, float* y,
                                                           int R, int C, int ld, int relu, const float* res) {
    const int cq = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(i % cq) * 4;
    const long long r0 = (i / cq) * 4;
    if (r0 >= R) return;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m = *reinterpret_cast<const float4*>(mu + c), k = *reinterpret_cast<const float4*>(rs + c);
    const float4 g = w ? *reinterpret_cast<const float4*>(w + c) : one, be = b ? *reinterpret_cast<const float4*>(b + c) : zero;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long r = r0 + e;
        if (r >= R) break;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        float4 o = make_float4(bn_value(v.x, m.x, k.x, g.x, be.x), bn_value(v.y, m.y, k.y, g.y, be.y), bn_value(v.z, m.z, k.z, g.z, be.z),
                               bn_value(v.w, m.w, k.w, g.w, be.w));
        if (res) { const float4 q = *reinterpret_cast<const float4*>(res + r * ld + c); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        *reinterpret_cast<float4*>(y + r * ld + c) = o;
    }
}
// backward partial sums (the vec4 form of bn_partial_kernel's mode 2) with the ReLU mask re-computed from x
template <bool CO = false>
__device__ __forceinline__ void bn_bwd_partial4_tile(const float* x, const float* gy, const float* w, const float* b, const float* mu, const float* rs,
                                                     float* p1, float* p2, int R, int C, int ld, int relu, int cg, int ch, float4 (&s1)[16][16],
                                                     float4 (&s2)[16][16]) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4, c = cg * 64 + cq * 4;
    const int r0 = ch * BN_CHUNK_ROWS, r1 = min(R, r0 + BN_CHUNK_ROWS);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
    if (c < C) {
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 m = *reinterpret_cast<const float4*>(mu + c), k = *reinterpret_cast<const float4*>(rs + c);
        const float4 g = w ? *reinterpret_cast<const float4*>(w + c) : one, be = b ? *reinterpret_cast<const float4*>(b + c) : zero;
        auto step = [&](const float4 v, float4 q) {
            if (relu) {
                if (!(bn_value(v.x, m.x, k.x, g.x, be.x) > 0.f)) q.x = 0.f;
                if (!(bn_value(v.y, m.y, k.y, g.y, be.y) > 0.f)) q.y = 0.f;
                if (!(bn_value(v.z, m.z, k.z, g.z, be.z) > 0.f)) q.z = 0.f;
                if (!(bn_value(v.w, m.w, k.w, g.w, be.w) > 0.f)) q.w = 0.f;
            }
            a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            bb.x = fmaf(q.x, (v.x - m.x) * k.x, bb.x); bb.y = fmaf(q.y, (v.y - m.y) * k.y, bb.y);
            bb.z = fmaf(q.z, (v.z - m.z) * k.z, bb.z); bb.w = fmaf(q.w, (v.w - m.w) * k.w, bb.w);
        };
        int r = r0 + rl;                                   // (four rows' loads in flight, accumulated in row order: see bn_partial4_kernel)
        for (; r + 48 < r1; r += 64) {
            float4 v[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const float4*>(x + (long long)(r + 16 * u) * ld + c);
                q[u] = *reinterpret_cast<const float4*>(gy + (long long)(r + 16 * u) * ld + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) step(v[u], q[u]);
        }
        for (; r < r1; r += 16)
            step(*reinterpret_cast<const float4*>(x + (long long)r * ld + c), *reinterpret_cast<const float4*>(gy + (long long)r * ld + c));
    }
    s1[rl][cq] = a; s2[rl][cq] = bb;
    __syncthreads();
    if (rl == 0 && c < C) {
        float4 t = s1[0][cq], u = s2[0][cq];
        for (int l = 1; l < 16; ++l) {
            const float4 q = s1[l][cq], z = s2[l][cq];
            t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
            u.x += z.x; u.y += z.y; u.z += z.z; u.w += z.w;
        }
        st_co4<CO>(p1 + (long long)ch * C + c, t);
        st_co4<CO>(p2 + (long long)ch * C + c, u);
    }
}
__global__ __launch_bounds__(256) void bn_bwd_partial4_kernel(const float* x, const float* gy, const float* w, const float* b, const float* mu, const float* rs,
                                                             float* p1, float* p2, int R, int C, int ld, int relu) {
    __shared__ float4 s1[16][16], s2[16][16];
    bn_bwd_partial4_tile(x, gy, w, b, mu, rs, p1, p2, R, C, ld, relu, blockIdx.x, blockIdx.y, s1, s2);
}
// both column sums of the backward pass in one launch: t1 = g b, t2 = g w (also kept for the apply kernel)
template <bool CO = false>
__device__ __forceinline__ void bn_bwd_combine16(const float* p1, const float* p2, float* t1, float* t2, float* gb, float* gw, int chunks, int C, int c16,
                                                 float (&s)[16][17]) {
    const int c = c16 * 16 + (threadIdx.x & 15);
    const float a = chunk_sum16<CO>(p1, chunks, C, c, s);
    __syncthreads();
    const float b = chunk_sum16<CO>(p2, chunks, C, c, s);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    st_co<CO>(t1 + c, a); st_co<CO>(t2 + c, b);
    if (gb) gb[c] = a;
    if (gw) gw[c] = b;
}
__global__ __launch_bounds__(256) void bn_bwd_combine_kernel(const float* p1, const float* p2, float* t1, float* t2, float* gb, float* gw, int chunks, int C) {
    __shared__ float s[16][17];
    bn_bwd_combine16(p1, p2, t1, t2, gb, gw, chunks, C, blockIdx.x, s);
}
__global__ __launch_bounds__(256) void bn_apply_bwd4_kernel(const float* gy, const float* x, const float* w, const float* b, const float* mu, const float* rs,
                                                           const float* s1, const float* s2, float* gx, int R, int C, int ld, int relu, float n_pool = 0.f) {
    const int cq = C >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(i % cq) * 4;
    const long long r0 = (i / cq) * 4;
    if (r0 >= R) return;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m = *reinterpret_cast<const float4*>(mu + c), k = *reinterpret_cast<const float4*>(rs + c);
    const float4 g = w ? *reinterpret_cast<const float4*>(w + c) : one, be = b ? *reinterpret_cast<const float4*>(b + c) : zero;
    const float4 a1 = *reinterpret_cast<const float4*>(s1 + c), a2 = *reinterpret_cast<const float4*>(s2 + c);
    const float nn = n_pool > 0.f ? n_pool : (float)R;          // SyncBN: the sums were pooled over all ranks' rows
    const float4 m1 = make_float4(a1.x / nn, a1.y / nn, a1.z / nn, a1.w / nn), m2 = make_float4(a2.x / nn, a2.y / nn, a2.z / nn, a2.w / nn);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long r = r0 + e;
        if (r >= R) break;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        float4 q = *reinterpret_cast<const float4*>(gy + r * ld + c);
        if (relu) {
            if (!(bn_value(v.x, m.x, k.x, g.x, be.x) > 0.f)) q.x = 0.f;
            if (!(bn_value(v.y, m.y, k.y, g.y, be.y) > 0.f)) q.y = 0.f;
            if (!(bn_value(v.z, m.z, k.z, g.z, be.z) > 0.f)) q.z = 0.f;
            if (!(bn_value(v.w, m.w, k.w, g.w, be.w) > 0.f)) q.w = 0.f;
        }
        float4 o;
        o.x = g.x * k.x * (q.x - m1.x - (v.x - m.x) * k.x * m2.x);
        o.y = g.y * k.y * (q.y - m1.y - (v.y - m.y) * k.y * m2.y);
        o.z = g.z * k.z * (q.z - m1.z - (v.z - m.z) * k.z * m2.z);
        o.w = g.w * k.w * (q.w - m1.w - (v.w - m.w) * k.w * m2.w);
        *reinterpret_cast<float4*>(gx + r * ld + c) = o;
    }
}

// ---- nearest-neighbour upsampling of an NHWC fp32 map by a power-of-two factor, ADDED to dst (the fuse layers of an HRNet module in training
// form: y_i += up(f_ij(x_j)), dir_amd/train/hrnet.py), and its backward (the sum over each f x f block, rows in order).  C % 4 == 0.
__global__ __launch_bounds__(256) void upsample_nearest_add_kernel(const float* src, float* dst, int h, int w, int C, int f, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int cq = C >> 2, W = w * f;
    const int c = (int)(i % cq) * 4;
    const long long p = i / cq;                              // output pixel (b, Y, X)
    const int X = (int)(p % W);
    const long long q = p / W;                               // b * H + Y
    const int H = h * f, Y = (int)(q % H);
    const long long b = q / H;
    const float4 v = *reinterpret_cast<const float4*>(src + ((b * h + Y / f) * w + X / f) * C + c);
    float4* o = reinterpret_cast<float4*>(dst + p * C + c);
    float4 t = *o;
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    *o = t;
}
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float* gy, float* gx, int h, int w, int C, int f, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int cq = C >> 2;
    const int c = (int)(i % cq) * 4;
    const long long p = i / cq;                              // input pixel (b, y, x)
    const int x = (int)(p % w);
    const long long q = p / w;
    const int y = (int)(q % h);
    const long long b = q / h;
    const int W = w * f, H = h * f;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const float4 v = *reinterpret_cast<const float4*>(gy + ((b * H + y * f + dy) * W + x * f + dx) * C + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    *reinterpret_cast<float4*>(gx + p * C + c) = a;
}

// ---- BatchNorm over more than BN_SMALL_R rows in ONE launch (round 5; VERDICT r4 item 3: "BatchNorm out of its six launches").  The three
// launches above (chunk partials -> combine -> apply) are dependent and small: at 32 images per GPU a launch boundary (~5 us) costs as much as
// the median BatchNorm kernel runs, 660 of the step's 2 440 launches.  Here a PERSISTENT grid (every workgroup resident: grid <= what the GPU
// holds) walks the same (64 channels x 256 rows) tiles; a channel group's statistics are combined by the workgroup whose tile arrives LAST
// at the group's counter (in chunk order whatever the arrival order: the same bits as bn_stats_combine_kernel), the others wait on the
// group's flag -- NOT on a grid-wide barrier: a 64-channel group is ready as soon as its own chunks are -- and then normalise the tiles they read.
// words: [arrive | flag | depart][BN_ONE_GROUPS] + an error word, zero before the launch and zero again after it (the last tile to leave a
// group clears it), owned by the library per stream (bn_one_words).  Visibility across the 8 XCDs' L2s: everything workgroups hand each other
// (chunk partials, the group's statistics, the words) moves through agent-scope monotonic loads / stores (`sc1`: ld_co / st_co) ordered by
// s_waitcnt vmcnt(0) -- NOT through release / acquire fences: a fence writes back / invalidates the XCD's whole L2, and one per tile made the
// first version of this kernel a third SLOWER than the three launches (0.041 against 0.031 s per step).
// MEASURED (MI355X, 32 images, profiles/r05_bn_one_launch_ab.txt): the same bits as the three launches, 220 launches instead of 660 per step -- and
// 0.0325 s per step against 0.0315 s.  At this batch a launch boundary inside a busy stream costs ~1 us (the next dispatch is staged while the
// previous kernel drains), less than what the one launch loses: every workgroup of a 64-channel group idles while ONE workgroup combines the
// group's chunks (the combine kernel spreads that over C / 16 workgroups), and the grid is 3/4 of what fits.  So it is OFF by default
// (DIR_BN_ONE_LAUNCH=1 / dir_bn_one_launch_enable(1) turns it on); what BatchNorm costs the step is its 6 passes over the feature maps, which only
// the producing / consuming convolutions can absorb.
constexpr int BN_ONE_GROUPS = 256, BN_ONE_WORDS = 3 * BN_ONE_GROUPS + 4;
struct BnOne {
    const float *x, *gy, *w, *b, *res;
    float *y, *gx, *save_mean, *save_rstd, *running_mean, *running_var, *gw, *gb, *p1, *p2, *t1, *t2;
    int* words;
    int R, C, ld, relu, chunks, cgroups;
    float eps, momentum;
};
__device__ __forceinline__ void bn_one_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int bn_one_arrive(int* counter) {
    __shared__ int old;
    bn_one_stores_done();              // this thread's coherent stores have reached memory
    __syncthreads();                   // ... every thread's
    if (threadIdx.x == 0) old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return old;
}
__device__ __forceinline__ void bn_one_wait(int* flag, int* err) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 400000000ll) {                 // 4 s at 100 MHz: a workgroup of this launch never ran (see dir_bn_one_launch_status)
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();                   // the statistics are then read with ld_co (from memory, not from this XCD's L2)
}
__device__ __forceinline__ void bn_one_leave(int* depart, int* flag, int chunks) {
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(depart, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == chunks - 1) {
        __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every tile of the group has seen the flag: clean for the next launch
        __hip_atomic_store(depart, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(256) void bn_one_fwd_kernel(const BnOne a) {
    __shared__ float4 s1[16][16];
    __shared__ float s[16][17];
    const int ntile = a.cgroups * a.chunks, G = gridDim.x;
    int *arrive = a.words, *flag = a.words + BN_ONE_GROUPS, *depart = a.words + 2 * BN_ONE_GROUPS, *err = a.words + 3 * BN_ONE_GROUPS;
    for (int t = blockIdx.x; t < ntile; t += G) {
        const int cg = t % a.cgroups, ch = t / a.cgroups;
        __syncthreads();
        bn_stats4_tile<true>(a.x, a.p1, a.p2, a.R, a.C, a.ld, cg, ch, s1);
        if (bn_one_arrive(arrive + cg) != a.chunks - 1) continue;
        for (int j = 0; j < 4; ++j) {                                  // the last tile of the group: every chunk partial of these 64 channels is in memory
            __syncthreads();
            bn_stats_combine16<true>(a.p1, a.p2, a.save_mean, a.save_rstd, a.running_mean, a.running_var, a.chunks, a.R, a.C, a.eps, a.momentum, cg * 4 + j, s);
        }
        bn_one_stores_done();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(arrive + cg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag + cg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = blockIdx.x; t < ntile; t += G) {
        const int cg = t % a.cgroups, ch = t / a.cgroups, c = cg * 64 + cq * 4;
        bn_one_wait(flag + cg, err);
        if (c < a.C) {
            const float4 m = ld_co4<true>(a.save_mean + c), k = ld_co4<true>(a.save_rstd + c);
            const float4 g = a.w ? *reinterpret_cast<const float4*>(a.w + c) : one, be = a.b ? *reinterpret_cast<const float4*>(a.b + c) : zero;
            const int r1 = min(a.R, (ch + 1) * BN_CHUNK_ROWS);
            auto put = [&](int r, const float4 v, const float4 q) {
                float4 o = make_float4(bn_value(v.x, m.x, k.x, g.x, be.x), bn_value(v.y, m.y, k.y, g.y, be.y), bn_value(v.z, m.z, k.z, g.z, be.z),
                                       bn_value(v.w, m.w, k.w, g.w, be.w));
                if (a.res) { o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
                if (a.relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                *reinterpret_cast<float4*>(a.y + (long long)r * a.ld + c) = o;
            };
            int r = ch * BN_CHUNK_ROWS + rl;
            for (; r + 48 < r1; r += 64) {
                float4 v[4], q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = *reinterpret_cast<const float4*>(a.x + (long long)(r + 16 * u) * a.ld + c);
                    q[u] = a.res ? *reinterpret_cast<const float4*>(a.res + (long long)(r + 16 * u) * a.ld + c) : zero;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) put(r + 16 * u, v[u], q[u]);
            }
            for (; r < r1; r += 16)
                put(r, *reinterpret_cast<const float4*>(a.x + (long long)r * a.ld + c),
                    a.res ? *reinterpret_cast<const float4*>(a.res + (long long)r * a.ld + c) : zero);
        }
        bn_one_leave(depart + cg, flag + cg, a.chunks);
    }
}
__global__ __launch_bounds__(256) void bn_one_bwd_kernel(const BnOne a) {
    __shared__ float4 s1[16][16], s2[16][16];
    __shared__ float s[16][17];
    const int ntile = a.cgroups * a.chunks, G = gridDim.x;
    int *arrive = a.words, *flag = a.words + BN_ONE_GROUPS, *depart = a.words + 2 * BN_ONE_GROUPS, *err = a.words + 3 * BN_ONE_GROUPS;
    for (int t = blockIdx.x; t < ntile; t += G) {
        const int cg = t % a.cgroups, ch = t / a.cgroups;
        __syncthreads();
        bn_bwd_partial4_tile<true>(a.x, a.gy, a.w, a.b, a.save_mean, a.save_rstd, a.p1, a.p2, a.R, a.C, a.ld, a.relu, cg, ch, s1, s2);
        if (bn_one_arrive(arrive + cg) != a.chunks - 1) continue;
        for (int j = 0; j < 4; ++j) {
            __syncthreads();
            bn_bwd_combine16<true>(a.p1, a.p2, a.t1, a.t2, a.gb, a.gw, a.chunks, a.C, cg * 4 + j, s);
        }
        bn_one_stores_done();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(arrive + cg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.gx) __hip_atomic_store(flag + cg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!a.gx) return;
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float nn = (float)a.R;
    for (int t = blockIdx.x; t < ntile; t += G) {
        const int cg = t % a.cgroups, ch = t / a.cgroups, c = cg * 64 + cq * 4;
        bn_one_wait(flag + cg, err);
        if (c < a.C) {
            const float4 m = *reinterpret_cast<const float4*>(a.save_mean + c), k = *reinterpret_cast<const float4*>(a.save_rstd + c);
            const float4 g = a.w ? *reinterpret_cast<const float4*>(a.w + c) : one, be = a.b ? *reinterpret_cast<const float4*>(a.b + c) : zero;
            const float4 a1 = ld_co4<true>(a.t1 + c), a2 = ld_co4<true>(a.t2 + c);
            const float4 m1 = make_float4(a1.x / nn, a1.y / nn, a1.z / nn, a1.w / nn), m2 = make_float4(a2.x / nn, a2.y / nn, a2.z / nn, a2.w / nn);
            const int r1 = min(a.R, (ch + 1) * BN_CHUNK_ROWS);
            auto put = [&](int r, const float4 v, float4 q) {             // the expressions of bn_apply_bwd4_kernel
                if (a.relu) {
                    if (!(bn_value(v.x, m.x, k.x, g.x, be.x) > 0.f)) q.x = 0.f;
                    if (!(bn_value(v.y, m.y, k.y, g.y, be.y) > 0.f)) q.y = 0.f;
                    if (!(bn_value(v.z, m.z, k.z, g.z, be.z) > 0.f)) q.z = 0.f;
                    if (!(bn_value(v.w, m.w, k.w, g.w, be.w) > 0.f)) q.w = 0.f;
                }
                float4 o;
                o.x = g.x * k.x * (q.x - m1.x - (v.x - m.x) * k.x * m2.x);
                o.y = g.y * k.y * (q.y - m1.y - (v.y - m.y) * k.y * m2.y);
                o.z = g.z * k.z * (q.z - m1.z - (v.z - m.z) * k.z * m2.z);
                o.w = g.w * k.w * (q.w - m1.w - (v.w - m.w) * k.w * m2.w);
                *reinterpret_cast<float4*>(a.gx + (long long)r * a.ld + c) = o;
            };
            int r = ch * BN_CHUNK_ROWS + rl;
            for (; r + 48 < r1; r += 64) {
                float4 v[4], q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = *reinterpret_cast<const float4*>(a.x + (long long)(r + 16 * u) * a.ld + c);
                    q[u] = *reinterpret_cast<const float4*>(a.gy + (long long)(r + 16 * u) * a.ld + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) put(r + 16 * u, v[u], q[u]);
            }
            for (; r < r1; r += 16)
                put(r, *reinterpret_cast<const float4*>(a.x + (long long)r * a.ld + c), *reinterpret_cast<const float4*>(a.gy + (long long)r * a.ld + c));
        }
        bn_one_leave(depart + cg, flag + cg, a.chunks);
    }
}

// ---- BatchNorm over at most BN_SMALL_R rows as a COOPERATIVE kernel (round 4): the kernels above walk a channel's rows with one thread, three
// dependent passes of R loads each -- at 16 images (336 token rows) that made the training step SLOWER than at 32 (0.033 against 0.030 s,
// profiles/r04_k_train_step_batch_sweep.txt).  A workgroup = 16 channels (4 quads) x 64 row lanes over all rows, lanes combined in lane order
// (deterministic); the same formulas (bn_value, two-pass variance, the apply expressions), another summation order.  C % 4 == 0, 16-byte aligned.
__device__ __forceinline__ float4 bn_mid_reduce(float4 v, float4 (*s)[4], int rl, int cq) {
    __syncthreads();                                     // (the previous use of s is over)
    s[rl][cq] = v;
    __syncthreads();
    float4 t = s[0][cq];
    for (int l = 1; l < 64; ++l) { const float4 q = s[l][cq]; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
    return t;                                            // every thread of the quad holds the same sums
}
__global__ __launch_bounds__(256) void bn_mid_fwd_kernel(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd,
                                                        float* running_mean, float* running_var, int R, int C, int ld, float eps, float momentum, int relu,
                                                        const float* res) {
    __shared__ float4 s[64][4];
    const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2, c = min(blockIdx.x * 16 + cq * 4, C - 4);
    const bool on = blockIdx.x * 16 + cq * 4 < C;
    auto at = [&](const float* p, int r) { return *reinterpret_cast<const float4*>(p + (long long)r * ld + c); };
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = rl;
    for (; r + 192 < R; r += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = at(x, r + 64 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < R; r += 64) { const float4 v = at(x, r); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    const float4 t = bn_mid_reduce(a, s, rl, cq);
    const float4 mu = make_float4(t.x / R, t.y / R, t.z / R, t.w / R);
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    auto sq = [&](const float4 v) {
        const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
        q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    };
    r = rl;
    for (; r + 192 < R; r += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = at(x, r + 64 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) sq(v[u]);
    }
    for (; r < R; r += 64) sq(at(x, r));
    const float4 m2 = bn_mid_reduce(q, s, rl, cq);
    const float4 var = make_float4(m2.x / R, m2.y / R, m2.z / R, m2.w / R);
    const float4 rs = make_float4(1.f / sqrtf(var.x + eps), 1.f / sqrtf(var.y + eps), 1.f / sqrtf(var.z + eps), 1.f / sqrtf(var.w + eps));
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = w ? *reinterpret_cast<const float4*>(w + c) : one, be = b ? *reinterpret_cast<const float4*>(b + c) : zero;
    if (rl == 0 && on) {
        *reinterpret_cast<float4*>(save_mean + c) = mu;
        *reinterpret_cast<float4*>(save_rstd + c) = rs;
        if (running_mean) {                              // torch: running = (1 - momentum) running + momentum stat, with the UNBIASED variance
            const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, m2s[4] = {m2.x, m2.y, m2.z, m2.w}, vs[4] = {var.x, var.y, var.z, var.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                running_mean[c + e] = (1.f - momentum) * running_mean[c + e] + momentum * mus[e];
                running_var[c + e] = (1.f - momentum) * running_var[c + e] + momentum * (R > 1 ? m2s[e] / (R - 1) : vs[e]);
            }
        }
    }
    if (!on) return;
    for (r = rl; r < R; r += 64) {
        const float4 v = at(x, r);
        float4 o = make_float4(bn_value(v.x, mu.x, rs.x, g.x, be.x), bn_value(v.y, mu.y, rs.y, g.y, be.y), bn_value(v.z, mu.z, rs.z, g.z, be.z),
                               bn_value(v.w, mu.w, rs.w, g.w, be.w));
        if (res) { const float4 e = at(res, r); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + (long long)r * ld + c) = o;
    }
}
__global__ __launch_bounds__(256) void bn_mid_bwd_kernel(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                                        float* gx, float* gw, float* gb, int R, int C, int ld, int relu) {
    __shared__ float4 s[64][4];
    const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2, c = min(blockIdx.x * 16 + cq * 4, C - 4);
    const bool on = blockIdx.x * 16 + cq * 4 < C;
    auto at = [&](const float* p, int r) { return *reinterpret_cast<const float4*>(p + (long long)r * ld + c); };
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m = *reinterpret_cast<const float4*>(save_mean + c), k = *reinterpret_cast<const float4*>(save_rstd + c);
    const float4 g = w ? *reinterpret_cast<const float4*>(w + c) : one, be = b ? *reinterpret_cast<const float4*>(b + c) : zero;
    auto masked = [&](const float4 v, float4 q) {
        if (relu) {
            if (!(bn_value(v.x, m.x, k.x, g.x, be.x) > 0.f)) q.x = 0.f;
            if (!(bn_value(v.y, m.y, k.y, g.y, be.y) > 0.f)) q.y = 0.f;
            if (!(bn_value(v.z, m.z, k.z, g.z, be.z) > 0.f)) q.z = 0.f;
            if (!(bn_value(v.w, m.w, k.w, g.w, be.w) > 0.f)) q.w = 0.f;
        }
        return q;
    };
    float4 a1 = zero, a2 = zero;
    auto acc = [&](const float4 v, const float4 q0) {
        const float4 q = masked(v, q0);
        a1.x += q.x; a1.y += q.y; a1.z += q.z; a1.w += q.w;
        a2.x = fmaf(q.x, (v.x - m.x) * k.x, a2.x); a2.y = fmaf(q.y, (v.y - m.y) * k.y, a2.y);
        a2.z = fmaf(q.z, (v.z - m.z) * k.z, a2.z); a2.w = fmaf(q.w, (v.w - m.w) * k.w, a2.w);
    };
    int r = rl;
    for (; r + 192 < R; r += 256) {
        float4 v[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = at(x, r + 64 * u); q[u] = at(gy, r + 64 * u); }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(v[u], q[u]);
    }
    for (; r < R; r += 64) acc(at(x, r), at(gy, r));
    const float4 s1 = bn_mid_reduce(a1, s, rl, cq), s2 = bn_mid_reduce(a2, s, rl, cq);
    if (rl == 0 && on) {
        if (gw) *reinterpret_cast<float4*>(gw + c) = s2;
        if (gb) *reinterpret_cast<float4*>(gb + c) = s1;
    }
    if (!gx || !on) return;
    const float4 m1 = make_float4(s1.x / R, s1.y / R, s1.z / R, s1.w / R), m2 = make_float4(s2.x / R, s2.y / R, s2.z / R, s2.w / R);
    for (r = rl; r < R; r += 64) {
        const float4 v = at(x, r), q = masked(v, at(gy, r));
        float4 o;
        o.x = g.x * k.x * (q.x - m1.x - (v.x - m.x) * k.x * m2.x);
        o.y = g.y * k.y * (q.y - m1.y - (v.y - m.y) * k.y * m2.y);
        o.z = g.z * k.z * (q.z - m1.z - (v.z - m.z) * k.z * m2.z);
        o.w = g.w * k.w * (q.w - m1.w - (v.w - m.w) * k.w * m2.w);
        *reinterpret_cast<float4*>(gx + (long long)r * ld + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------ ReLU
__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* x, float* y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* gy, const float* y, float* gx, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) gx[i] = y[i] > 0.f ? gy[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------ P-GCN adjacency
// A_1 = row-softmax of the skeleton mask filled with e_1 (SemGCN/p_graph_conv.py:43-50); the 40 edges in row-major nonzero order.
__constant__ int kAdjOff[22] = {0, 5, 7, 9, 11, 12, 14, 16, 18, 19, 21, 23, 25, 26, 28, 30, 32, 33, 35, 37, 39, 40};
__constant__ int kAdjIdx[40] = {1, 5, 9, 13, 17, 0, 2, 1, 3, 2, 4, 3, 0, 6, 5, 7, 6, 8, 7, 0,
                                10, 9, 11, 10, 12, 11, 0, 14, 13, 15, 14, 16, 15, 0, 18, 17, 19, 18, 20, 19};
__global__ __launch_bounds__(64) void pgcn_adj_fwd_kernel(const float* e1, float* A) {      // A [21][21], one thread per row
    const int j = threadIdx.x;
    if (j >= 21) return;
    for (int k = 0; k < 21; ++k) A[j * 21 + k] = 0.f;
    const int o0 = kAdjOff[j], deg = kAdjOff[j + 1] - o0;
    float mx = -INFINITY, sum = 0.f;
    for (int t = 0; t < deg; ++t) mx = fmaxf(mx, e1[o0 + t]);
    for (int t = 0; t < deg; ++t) sum += expf(e1[o0 + t] - mx);
    for (int t = 0; t < deg; ++t) A[j * 21 + kAdjIdx[o0 + t]] = expf(e1[o0 + t] - mx) / sum;
}
// g A_1[j][k] = sum_b <g z[b][j], h1[b][k]> on the 40 edges (one wave per edge, samples in order), then the softmax chain rule per
// row: g e_1[edge] = A (g A - sum_row g A A).  grid = 1 workgroup of 40 waves would exceed 1024 threads: 40 workgroups + a tail kernel.
__global__ __launch_bounds__(64) void pgcn_adj_bwd_edge_kernel(const float* gz, const float* h1, float* gA_edge, int B) {
    const int e = blockIdx.x, lane = threadIdx.x;
    int j = 0;
    while (kAdjOff[j + 1] <= e) ++j;
    const int k = kAdjIdx[e];
    float acc = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {                           // eight samples' loads in flight, accumulated in sample order (the same sum)
        float g0[8], g1[8], h0[8], h1v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* g = gz + ((long long)(b + u) * 21 + j) * 128;
            const float* h = h1 + ((long long)(b + u) * 21 + k) * 128;
            g0[u] = g[lane]; g1[u] = g[lane + 64]; h0[u] = h[lane]; h1v[u] = h[lane + 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc = fmaf(g0[u], h0[u], acc); acc = fmaf(g1[u], h1v[u], acc); }
    }
    for (; b < B; ++b) {
        const float* g = gz + ((long long)b * 21 + j) * 128;
        const float* h = h1 + ((long long)b * 21 + k) * 128;
        acc = fmaf(g[lane], h[lane], acc);
        acc = fmaf(g[lane + 64], h[lane + 64], acc);
    }
    acc = dir::wave_sum(acc);
    if (lane == 0) gA_edge[e] = acc;
}
__global__ __launch_bounds__(64) void pgcn_adj_bwd_softmax_kernel(const float* e1, const float* gA_edge, float* ge1) {
    const int j = threadIdx.x;
    if (j >= 21) return;
    const int o0 = kAdjOff[j], deg = kAdjOff[j + 1] - o0;
    float mx = -INFINITY, sum = 0.f, p[5], dot = 0.f;
    for (int t = 0; t < deg; ++t) mx = fmaxf(mx, e1[o0 + t]);
    for (int t = 0; t < deg; ++t) { p[t] = expf(e1[o0 + t] - mx); sum += p[t]; }
    for (int t = 0; t < deg; ++t) { p[t] /= sum; dot = fmaf(gA_edge[o0 + t], p[t], dot); }
    for (int t = 0; t < deg; ++t) ge1[o0 + t] = p[t] * (gA_edge[o0 + t] - dot);
}

// ------------------------------------------------------------------------------------------------------------------ grid sample
// F.grid_sample(feat, uv[:, None], bilinear, zeros padding, align_corners False) at 21 joints (models/dir.py:198): feat NHWC fp32
// [B,S,S,C]; rows [B*21, C] (token-major).  Backward w.r.t. feat only (uv comes in detached, models/dir.py:447-453).
struct GridArgs2 { const float* feat; const float* uv; float* rows; const float* grows[2]; const float* uvs[2]; float* gfeat; int B, S, C, hands; };

__device__ __forceinline__ void bilinear_taps(float u, float v, int S, int (&ix)[4], int (&iy)[4], float (&w)[4]) {
#pragma clang fp contract(off)
    const float fx = ((u + 1.f) * S - 1.f) / 2.f, fy = ((v + 1.f) * S - 1.f) / 2.f;
    const float x0 = floorf(fx), y0 = floorf(fy), x1 = x0 + 1.f, y1 = y0 + 1.f;
    const float wx[4] = {x1 - fx, fx - x0, x1 - fx, fx - x0}, wy[4] = {y1 - fy, y1 - fy, fy - y0, fy - y0};
    const float xs[4] = {x0, x1, x0, x1}, ys[4] = {y0, y0, y1, y1};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool ok = xs[t] >= 0.f && xs[t] <= (float)(S - 1) && ys[t] >= 0.f && ys[t] <= (float)(S - 1);
        ix[t] = ok ? (int)xs[t] : 0; iy[t] = ok ? (int)ys[t] : 0;
        w[t] = ok ? wx[t] * wy[t] : 0.f;
    }
}
__global__ __launch_bounds__(256) void grid_rows_fwd_kernel(GridArgs2 a) {          // grid (B*21), thread = channel
    const int bj = blockIdx.x, b = bj / 21;
    int ix[4], iy[4]; float w[4];
    bilinear_taps(a.uv[2 * bj], a.uv[2 * bj + 1], a.S, ix, iy, w);
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc += a.feat[(((long long)b * a.S + iy[t]) * a.S + ix[t]) * a.C + c] * w[t];
        a.rows[(long long)bj * a.C + c] = acc;
    }
}
// g feat[b][pixel][c] += w g rows[b*21 + j][c]: one thread per (sample, channel) walks the hands' 21 x 4 taps in order (deterministic;
// two joints, or the two hands' samplers, may hit the same pixel).  g feat must be zero on entry.
__global__ __launch_bounds__(256) void grid_rows_bwd_kernel(GridArgs2 a) {          // grid (B, ceil(C / 256))
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= a.C) return;
    for (int h = 0; h < a.hands; ++h)
        for (int j = 0; j < 21; ++j) {
            const int bj = b * 21 + j;
            int ix[4], iy[4]; float w[4];
            bilinear_taps(a.uvs[h][2 * bj], a.uvs[h][2 * bj + 1], a.S, ix, iy, w);
            const float g = a.grows[h][(long long)bj * a.C + c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (w[t] != 0.f) { float* p = a.gfeat + (((long long)b * a.S + iy[t]) * a.S + ix[t]) * a.C + c; *p = fmaf(w[t], g, *p); }
        }
}

// dst += alpha src (gradient accumulation of shared modules: global_pos_emb / proj_feat_emb run once per hand, models/dir.py:106-107,118-119)
__global__ __launch_bounds__(256) void axpy_kernel(float* dst, const float* src, long long n, float alpha) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = fmaf(alpha, src[i], dst[i]);
}
// token-path inputs of a stage (models/dir.py:97-98,106-107): pos = xyz / 0.15; gpos = xyz / 0.15 -+ offset / 2 (left -, right +)
__global__ __launch_bounds__(256) void stage_positions_kernel(const float* xyz_l, const float* xyz_r, const float* offset, float* pos_l, float* pos_r,
                                                              float* gpos_l, float* gpos_r, int n) {      // n = B * 63
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / 63, c = i % 3;
    const float o = offset[3 * b + c] / 2.f;
    const float l = xyz_l[i] / 0.15f, r = xyz_r[i] / 0.15f;
    pos_l[i] = l; pos_r[i] = r; gpos_l[i] = l - o; gpos_r[i] = r + o;
}

// ------------------------------------------------------------------------------------------------------------------ conv wgrad
// d loss / d W of nn.Conv2d (NHWC activations, weights [Cout][kh][kw][Cin]):  gW[co][tap][ci] = sum_m gy[m][co] * x[pixel(m, tap)][ci]
// -- per tap a GEMM  gy^T [Cout x M] . X_tap [M x Cin]  whose reduction runs over the M = B*Ho*Wo output pixels.  Grid (ci tile x co
// tile, tap, pixel chunk): a workgroup reduces one contiguous chunk of pixels into a 64 x 64 tile (exact fp32 products on
// v_mfma_f32_16x16x4_f32, the tile layout of gemm_f32_kernel); with more than one chunk the partial tiles go to a workspace and
// wgrad_reduce_kernel adds them in chunk order (deterministic, no atomics).
struct WgradArgs {
    const float* x; const float* gy; float* out;        // out: gw, or the workspace [chunks][Cout][taps][Cin]
    int B, H, W, Cin, in_cs, in_co, Cout, gy_cs, gy_co, kh, kw, stride, pad, Ho, Wo, M, chunk, tiles_ci;
};
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    __shared__ float s_a[GT * GLD], s_b[GT * GLD];
    __shared__ long long s_row[GK];                      // element offset of the input pixel of each of the 16 reduction rows, -1 = padding
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci, tap = blockIdx.y, ch = blockIdx.z;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int co0 = tco * GT, ci0 = tci * GT;
    const int m_beg = ch * a.chunk, m_end = min(a.M, m_beg + a.chunk);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    for (int k0 = m_beg; k0 < m_end; k0 += GK) {
        if (tid < GK) {
            const int m = k0 + tid;
            long long off = -1;
            if (m < m_end) {
                const int b = m / (a.Ho * a.Wo), r = m - b * a.Ho * a.Wo, oy = r / a.Wo, ox = r - oy * a.Wo;
                const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) off = (((long long)b * a.H + iy) * a.W + ix) * a.in_cs + a.in_co;
            }
            s_row[tid] = off;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, c = e & 63, k = e >> 6;           // 64 channels (contiguous in memory) x 16 pixels
            const int m = k0 + k;
            float v = 0.f, w = 0.f;
            if (m < m_end && co0 + c < a.Cout) v = a.gy[(long long)m * a.gy_cs + a.gy_co + co0 + c];
            const long long ro = s_row[k];
            if (ro >= 0 && ci0 + c < a.Cin) w = a.x[ro + ci0 + c];
            s_a[c * GLD + k] = v;
            s_b[c * GLD + k] = w;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const float av = s_a[(16 * wave + li) * GLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_b[(16 * j + li) * GLD + 4 * ks + lk], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int taps = a.kh * a.kw;
    float* out = a.out + (long long)ch * a.Cout * taps * a.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + 16 * j + li;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * wave + 4 * lk + r;
            if (co < a.Cout) out[((long long)co * taps + tap) * a.Cin + ci] = acc[j][r];
        }
    }
}

// The same for a convolution with very few input channels (the 7x7 / 2 stem on the 3-channel image): taps and channels are flattened into ONE
// column index j = tap * Cin + c (147 columns = 3 tiles) instead of a 64-wide channel tile per tap that is 95 % padding.  Grid (column tile x
// co tile, 1, pixel chunk); output element [co][j] is the weight layout [Cout][kh][kw][Cin] itself.
__global__ __launch_bounds__(256) void conv_wgrad_flatk_kernel(WgradArgs a) {
    __shared__ float s_a[GT * GLD], s_b[GT * GLD];
    __shared__ int s_pix[GK * 3];                        // per reduction row: image base index b * H, iy0, ix0 (iy0 = INT_MIN/2: no pixel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KC = a.kh * a.kw * a.Cin;
    const int tj = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci, ch = blockIdx.z;
    const int co0 = tco * GT, j0 = tj * GT;
    const int m_beg = ch * a.chunk, m_end = min(a.M, m_beg + a.chunk);
    // this thread's column (fixed over the loop): tap and channel
    const int jc = j0 + (tid & 63);
    const int tap = jc / a.Cin, cc = jc - tap * a.Cin, ky = tap / a.kw, kx = tap - ky * a.kw;
    const bool col_ok = jc < KC;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    for (int k0 = m_beg; k0 < m_end; k0 += GK) {
        if (tid < GK) {
            const int m = k0 + tid;
            int bh = 0, iy0 = -(1 << 29), ix0 = 0;
            if (m < m_end) {
                const int b = m / (a.Ho * a.Wo), r = m - b * a.Ho * a.Wo, oy = r / a.Wo, ox = r - oy * a.Wo;
                bh = b * a.H; iy0 = oy * a.stride - a.pad; ix0 = ox * a.stride - a.pad;
            }
            s_pix[3 * tid] = bh; s_pix[3 * tid + 1] = iy0; s_pix[3 * tid + 2] = ix0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, c = e & 63, k = e >> 6;           // column c (gy channel / flattened (tap, channel)) x 16 pixels
            const int m = k0 + k;
            float v = 0.f, w = 0.f;
            if (m < m_end && co0 + c < a.Cout) v = a.gy[(long long)m * a.gy_cs + a.gy_co + co0 + c];
            const int iy = s_pix[3 * k + 1] + ky, ix = s_pix[3 * k + 2] + kx;
            if (col_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) w = a.x[((long long)(s_pix[3 * k] + iy) * a.W + ix) * a.in_cs + a.in_co + cc];
            s_a[c * GLD + k] = v;
            s_b[c * GLD + k] = w;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const float av = s_a[(16 * wave + li) * GLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_b[(16 * j + li) * GLD + 4 * ks + lk], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = a.out + (long long)ch * a.Cout * KC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int jj = j0 + 16 * j + li;
        if (jj >= KC) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * wave + 4 * lk + r;
            if (co < a.Cout) out[(long long)co * KC + jj] = acc[j][r];
        }
    }
}
// The same reduction for 16-byte-aligned channel counts (every convolution of the path except the 3-channel stem and the 1- / 3-output
// heads): 32 pixels per step, both operands fetched with float4 loads into registers ONE STEP AHEAD of the MFMAs that consume the
// previous step (software pipeline: global latency hidden behind 32 MFMAs per wave), pixel -> address arithmetic per thread.
constexpr int WK = 32, WLD = WK + 1;
__global__ __launch_bounds__(256) void conv_wgrad4_kernel(WgradArgs a) {
    __shared__ float s_a[GT * WLD], s_b[GT * WLD];       // [channel][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci, tap = blockIdx.y, ch = blockIdx.z;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int co0 = tco * GT, ci0 = tci * GT;
    const int m_beg = ch * a.chunk, m_end = min(a.M, m_beg + a.chunk);
    const int pl = tid >> 4, cq = (tid & 15) * 4;         // loader role: pixels pl, pl + 16 of the step; channels cq .. cq + 3 of the tile
    const bool co_ok = co0 + cq < a.Cout, ci_ok = ci0 + cq < a.Cin;      // (channel counts are multiples of 4: a quad is all in or all out)
    const int hw = a.Ho * a.Wo;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = k0 + pl + 16 * h;
            ra[h] = make_float4(0.f, 0.f, 0.f, 0.f); rb[h] = ra[h];
            if (m < m_end) {
                if (co_ok) ra[h] = *reinterpret_cast<const float4*>(a.gy + (long long)m * a.gy_cs + a.gy_co + co0 + cq);
                const int b = m / hw, r = m - b * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
                const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                if (ci_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    rb[h] = *reinterpret_cast<const float4*>(a.x + (((long long)b * a.H + iy) * a.W + ix) * a.in_cs + a.in_co + ci0 + cq);
            }
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    gload(m_beg);
    for (int k0 = m_beg; k0 < m_end; k0 += WK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int px = pl + 16 * h;
            s_a[(cq + 0) * WLD + px] = ra[h].x; s_a[(cq + 1) * WLD + px] = ra[h].y; s_a[(cq + 2) * WLD + px] = ra[h].z; s_a[(cq + 3) * WLD + px] = ra[h].w;
            s_b[(cq + 0) * WLD + px] = rb[h].x; s_b[(cq + 1) * WLD + px] = rb[h].y; s_b[(cq + 2) * WLD + px] = rb[h].z; s_b[(cq + 3) * WLD + px] = rb[h].w;
        }
        __syncthreads();
        gload(k0 + WK);                                   // next step's operands (all zeros past the chunk's end)
#pragma unroll
        for (int ks = 0; ks < WK / 4; ++ks) {
            const float av = s_a[(16 * wave + li) * WLD + 4 * ks + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, s_b[(16 * j + li) * WLD + 4 * ks + lk], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int taps = a.kh * a.kw;
    float* out = a.out + (long long)ch * a.Cout * taps * a.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + 16 * j + li;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * wave + 4 * lk + r;
            if (co < a.Cout) out[((long long)co * taps + tap) * a.Cin + ci] = acc[j][r];
        }
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* gw, long long n, int chunks, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? gw[i] : 0.f;
    s = dir::sum_in_order(part + i, n, chunks, s);
    gw[i] = s;
}

static void launch_bn_partial(dim3 grid, hipStream_t s, const float* x, const float* gy, const float* mu, const float* rs, float* p1, float* p2, int R, int C, int ld,
                              int mode, const float* w = nullptr, const float* be = nullptr, int relu = 0) {
    if (relu) {                                            // (backward sums through a ReLU mask: the scalar kernel carries it)
        DIR_LAUNCH(bn_partial_kernel, grid, dim3(256), 0, s, x, gy, mu, rs, p1, p2, R, C, ld, mode, w, be, relu);
        return;
    }
    const bool vec = C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && (!gy || ((uintptr_t)gy & 15) == 0) && ((uintptr_t)p1 & 15) == 0 &&
                     (!p2 || ((uintptr_t)p2 & 15) == 0) && (!mu || ((uintptr_t)mu & 15) == 0) && (!rs || ((uintptr_t)rs & 15) == 0);
    if (vec) DIR_LAUNCH(bn_partial4_kernel, grid, dim3(256), 0, s, x, gy, mu, rs, p1, p2, R, C, ld, mode);
    else DIR_LAUNCH(bn_partial_kernel, grid, dim3(256), 0, s, x, gy, mu, rs, p1, p2, R, C, ld, mode, (const float*)nullptr, (const float*)nullptr, 0);
}

}  // namespace

extern "C" int dir_gemm_f32(const dir_gemm_desc* d, const float* A, const float* B, const float* bias, float* C, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && A && B && C, "dir_gemm_f32: null pointer");
    DIR_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0 && d->lda > 0 && d->ldb > 0 && d->ldc >= d->N, "dir_gemm_f32: bad shape");
    GemmArgs a{A, B, bias, C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->trans_a, d->trans_b, d->accumulate, d->stride_a, d->stride_b, d->stride_c, 0, nullptr};
    const dim3 grid((d->N + GT - 1) / GT, (d->M + GT - 1) / GT, d->batch);
    static const bool small = []() { const char* e = getenv("DIR_GEMM_SMALL"); return !(e && e[0] == '0'); }();
    if (small && (long long)grid.x * grid.y * grid.z <= 128)           // the 64 x 64 grid fills at most half the CUs: 32 x 32 tiles (same bits)
        DIR_LAUNCH(gemm_f32_small_kernel, dim3((d->N + 31) / 32, (d->M + 31) / 32, d->batch), dim3(256), 0, (hipStream_t)stream, a);
    else
        DIR_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dir_gemm_f32");
}

extern "C" int dir_gemm_f32_grouped(const dir_gemm_desc* d, const dir_gemm_groups* g, const float* A, const float* B, const float* bias, float* C, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && g && A && B && C, "dir_gemm_f32_grouped: null pointer");
    DIR_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0 && d->lda > 0 && d->ldb > 0 && d->ldc >= d->N, "dir_gemm_f32_grouped: bad shape");
    DIR_REQUIRE(g->ny > 0 && g->nx > 0 && (long long)g->ny * g->nx * d->batch <= 65535, "dir_gemm_f32_grouped: groups %d x %d, batch %d", g->ny, g->nx, d->batch);
    GemmArgs a{A, B, bias, C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->trans_a, d->trans_b, d->accumulate, d->stride_a, d->stride_b, d->stride_c, 0, nullptr};
    GemmGroups gr{g->ny, g->nx, g->reduce ? 1 : 0, g->a_y, g->a_x, g->b_y, g->b_x, g->c_y, g->c_x};
    const int z = g->reduce ? d->batch : d->batch * g->ny * g->nx;
    if (d->M > 64 && d->M <= 80)          // one 80-row tile instead of two 64-row ones (other summation-independent layout: the same k order per element)
        DIR_LAUNCH(gemm_f32_grouped_short_kernel<5>, dim3((d->N + GT - 1) / GT, 1, z), dim3(256), 0, (hipStream_t)stream, a, gr);
    else
        DIR_LAUNCH(gemm_f32_grouped_kernel, dim3((d->N + GT - 1) / GT, (d->M + GT - 1) / GT, z), dim3(256), 0, (hipStream_t)stream, a, gr);
    return check_launch("dir_gemm_f32_grouped");
}

// Tall reductions with a small output (the weight gradients of the token path's Linear layers: gW [<= 384 x <= 256] = gy^T x over K = B * 21 .. 42
// rows): a 64 x 64 tile per workgroup walks K in steps of 16 with two barriers and a memory round trip each -- 84 dependent steps on four
// workgroups.  Split the reduction into chunks of 64 (one workgroup each: hundreds of workgroups, 4 steps each), partial tiles to a workspace,
// summed in chunk order by a second launch.
constexpr int GEMM_SPLIT_CHUNK = 64;
extern "C" long long dir_gemm_f32_splitk_workspace_bytes(const dir_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return -1;
    return (long long)((d->K + GEMM_SPLIT_CHUNK - 1) / GEMM_SPLIT_CHUNK) * d->M * d->N * 4;
}
extern "C" int dir_gemm_f32_splitk(const dir_gemm_desc* d, const float* A, const float* B, const float* bias, float* C, float* workspace,
                                   long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && A && B && C && workspace, "dir_gemm_f32_splitk: null pointer");
    DIR_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch == 1 && d->lda > 0 && d->ldb > 0 && d->ldc >= d->N, "dir_gemm_f32_splitk: bad shape (batch must be 1)");
    DIR_REQUIRE(workspace_bytes >= dir_gemm_f32_splitk_workspace_bytes(d), "dir_gemm_f32_splitk: workspace too small (dir_gemm_f32_splitk_workspace_bytes)");
    const int chunks = (d->K + GEMM_SPLIT_CHUNK - 1) / GEMM_SPLIT_CHUNK;
    GemmArgs a{A, B, nullptr, C, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, d->trans_a, d->trans_b, 0, 0, 0, 0, GEMM_SPLIT_CHUNK, workspace};
    hipStream_t s = (hipStream_t)stream;
    static const bool small = []() { const char* e = getenv("DIR_GEMM_SMALL"); return !(e && e[0] == '0'); }();
    if (small && (long long)((d->N + GT - 1) / GT) * ((d->M + GT - 1) / GT) * chunks <= 128)
        DIR_LAUNCH(gemm_f32_small_kernel, dim3((d->N + 31) / 32, (d->M + 31) / 32, chunks), dim3(256), 0, s, a);
    else
        DIR_LAUNCH(gemm_f32_kernel, dim3((d->N + GT - 1) / GT, (d->M + GT - 1) / GT, chunks), dim3(256), 0, s, a);
    DIR_LAUNCH(gemm_splitk_reduce_kernel, dim3((unsigned)(((long long)d->M * d->N + 255) / 256)), dim3(256), 0, s, (const float*)workspace, bias, C, d->M, d->N,
               d->ldc, chunks, d->accumulate);
    return check_launch("dir_gemm_f32_splitk");
}

extern "C" long long dir_colsum_workspace_bytes(int R, int N) {
    if (R <= 0 || N <= 0) return -1;
    return R <= BN_SMALL_R ? 0 : (long long)((R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS) * N * 4;
}
extern "C" int dir_colsum_f32(const float* x, float* out, int R, int N, int ld, int accumulate, float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && out && R > 0 && N > 0 && ld >= N, "dir_colsum_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    static const bool mid = []() { const char* e = getenv("DIR_COLSUM_MID"); return !(e && e[0] == '0'); }();
    if (mid && R >= 128 && R <= 4096 && N >= 4 && N % 4 == 0 && ld % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0) {
        DIR_LAUNCH(colsum_mid_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, x, out, R, N, ld, accumulate);
        return check_launch("dir_colsum_f32");
    }
    if (R <= BN_SMALL_R) {
        DIR_LAUNCH(colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x, out, R, N, ld, accumulate);
        return check_launch("dir_colsum_f32");
    }
    // tall matrices (bias gradients of the convolutions: R = B*Ho*Wo): 256-row chunk partials, added in chunk order
    DIR_REQUIRE(workspace && workspace_bytes >= dir_colsum_workspace_bytes(R, N), "dir_colsum_f32: workspace too small (dir_colsum_workspace_bytes)");
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    launch_bn_partial(dim3((N + 63) / 64, chunks), s, x, nullptr, nullptr, nullptr, workspace, nullptr, R, N, ld, 0);
    if (accumulate) DIR_LAUNCH(wgrad_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const float*)workspace, out, (long long)N, chunks, accumulate);
    else DIR_LAUNCH(bn_colsum_chunks_kernel, dim3((N + 15) / 16), dim3(256), 0, s, (const float*)workspace, out, chunks, N, 1.f);
    return check_launch("dir_colsum_f32");
}

extern "C" int dir_layernorm_forward(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int R, int C, float eps, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && w && b && y && mean && rstd && R > 0 && C > 0 && C <= 256, "dir_layernorm_forward: bad arguments (C <= 256)");
    DIR_LAUNCH(layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, y, mean, rstd, R, C, eps);
    return check_launch("dir_layernorm_forward");
}

extern "C" int dir_layernorm_backward(const float* gy, const float* x, const float* w, const float* mean, const float* rstd, float* gx, float* gw, float* gb,
                                      int R, int C, int accumulate_x, int accumulate_wb, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && w && mean && rstd && R > 0 && C > 0 && C <= 256, "dir_layernorm_backward: bad arguments (C <= 256)");
    if (gx) DIR_LAUNCH(layernorm_bwd_x_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, gy, x, w, mean, rstd, gx, R, C, accumulate_x);
    if (gw && gb) DIR_LAUNCH(layernorm_bwd_wb_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, gy, x, mean, rstd, gw, gb, R, C, accumulate_wb);
    return check_launch("dir_layernorm_backward");
}

extern "C" int dir_gelu_forward(const float* x, float* y, long long n, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && y && n > 0, "dir_gelu_forward: bad arguments");
    DIR_LAUNCH(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("dir_gelu_forward");
}
extern "C" int dir_gelu_backward(const float* gy, const float* x, float* gx, long long n, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && gx && n > 0, "dir_gelu_backward: bad arguments");
    DIR_LAUNCH(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, x, gx, n);
    return check_launch("dir_gelu_backward");
}

extern "C" int dir_attention_forward(const float* qkv, float* probs, float* out, int B, int T, int H, float scale, void* stream) {
    using namespace dir;
    DIR_REQUIRE(qkv && out && B > 0 && T > 0 && T <= AT && H > 0, "dir_attention_forward: bad arguments (T <= 64, head dim 32)");
    AttnArgs a{qkv, probs, out, nullptr, nullptr, B, T, H, scale};
    DIR_LAUNCH(attention_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dir_attention_forward");
}
extern "C" int dir_attention_backward(const float* qkv, const float* probs, const float* gout, float* gqkv, int B, int T, int H, float scale, void* stream) {
    using namespace dir;
    DIR_REQUIRE(qkv && probs && gout && gqkv && B > 0 && T > 0 && T <= AT && H > 0, "dir_attention_backward: bad arguments");
    AttnArgs a{qkv, const_cast<float*>(probs), nullptr, gout, gqkv, B, T, H, scale};
    DIR_LAUNCH(attention_bwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dir_attention_backward");
}

extern "C" long long dir_bn_train_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return -1;
    if (R <= BN_SMALL_R) return 0;
    const long long chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    return (2 * chunks + 2) * C * 4;
}
static bool bn_vec4(int C, int ld, std::initializer_list<const void*> ps) {
    if (C % 4 || ld % 4) return false;
    for (const void* q : ps) if (q && ((uintptr_t)q & 15)) return false;
    return true;
}
// ---- one-launch BatchNorm: the library-owned sync words (one block per stream that ever ran a BatchNorm, from a pool allocated and zeroed at the
// first call -- never inside a stream capture: a stream first seen while capturing takes a block of the pool, and with the pool absent or spent
// the three-launch path runs) and the persistent grid's size
constexpr int BN_ONE_POOL = 16, BN_ONE_DEVICES = 16;
struct BnOnePool { int* base = nullptr; int used = 0; hipStream_t streams[BN_ONE_POOL]; };
static std::mutex bn_one_mu;
static int bn_one_switch = -1;           // -1: not read yet (DIR_BN_ONE_LAUNCH, default OFF: measured slower, below) | 0 | 1; dir_bn_one_launch_enable sets it
static bool bn_one_enabled() {
    std::lock_guard<std::mutex> lock(bn_one_mu);
    if (bn_one_switch < 0) { const char* e = getenv("DIR_BN_ONE_LAUNCH"); bn_one_switch = e && e[0] == '1'; }
    return bn_one_switch != 0;
}
static BnOnePool bn_one_pools[BN_ONE_DEVICES];
static int* bn_one_words(hipStream_t s) {
    if (!bn_one_enabled()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BN_ONE_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(bn_one_mu);
    BnOnePool& p = bn_one_pools[dev];
    if (!p.base) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
        int* q = nullptr;
        if (hipMalloc(&q, sizeof(int) * BN_ONE_WORDS * BN_ONE_POOL) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(q, 0, sizeof(int) * BN_ONE_WORDS * BN_ONE_POOL) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(q); return nullptr; }
        p.base = q;
    }
    for (int i = 0; i < p.used; ++i) if (p.streams[i] == s) return p.base + (long long)i * BN_ONE_WORDS;
    if (p.used == BN_ONE_POOL) return nullptr;
    p.streams[p.used] = s;
    return p.base + (long long)(p.used++) * BN_ONE_WORDS;
}
template <typename K>
static int bn_one_grid(K kernel, int ntile) {
    static const int slots = [kernel]() {
        int dev = 0, cus = 256, occ = 4;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ < 1) occ = 1;
        return cus * (occ > 1 ? occ * 3 / 4 : 1);          // every workgroup of the launch must be resident at once (they wait on each other): below what fits
    }();
    return ntile < slots ? ntile : slots;
}
// 0, or 1 if a workgroup of a one-launch BatchNorm on the current device ever gave up waiting (that launch's outputs were wrong); the sync words
// of the device are cleared.  Synchronises the device: nothing in the training step calls this -- tests and the eager (calibration) steps do.
extern "C" int dir_bn_one_launch_enable(int on) {
    const bool was = bn_one_enabled();
    std::lock_guard<std::mutex> lock(bn_one_mu);
    bn_one_switch = on != 0;
    return was;
}
extern "C" int dir_bn_one_launch_status(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BN_ONE_DEVICES) return 0;
    std::lock_guard<std::mutex> lock(bn_one_mu);
    BnOnePool& p = bn_one_pools[dev];
    if (!p.base) return 0;
    static int host[BN_ONE_WORDS * BN_ONE_POOL];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, p.base, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int bad = 0;
    for (int i = 0; i < BN_ONE_POOL; ++i) bad |= host[i * BN_ONE_WORDS + 3 * BN_ONE_GROUPS] != 0;
    if (bad && hipMemset(p.base, 0, sizeof(host)) != hipSuccess) return -1;
    return bad;
}
extern "C" int dir_bn_train_forward(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd, float* running_mean,
                                    float* running_var, int R, int C, int ld, float eps, float momentum, int relu, const float* residual,
                                    float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && y && save_mean && save_rstd && R > 0 && C > 0 && ld >= C && ((running_mean == nullptr) == (running_var == nullptr)),
                "dir_bn_train_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (R <= BN_SMALL_R && R >= 16 && C >= 4 && bn_vec4(C, ld, {x, y, w, b, save_mean, save_rstd, residual})) {
        DIR_LAUNCH(bn_mid_fwd_kernel, dim3((C + 15) / 16), dim3(256), 0, s, x, w, b, y, save_mean, save_rstd, running_mean, running_var, R, C, ld, eps, momentum, relu, residual);
        return check_launch("dir_bn_train_forward");
    }
    if (R <= BN_SMALL_R) {
        DIR_LAUNCH(bn_train_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, s, x, w, b, y, save_mean, save_rstd, running_mean, running_var, R, C, ld, eps, momentum, relu, residual);
        return check_launch("dir_bn_train_forward");
    }
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_train_workspace_bytes(R, C), "dir_bn_train_forward: workspace too small (dir_bn_train_workspace_bytes)");
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* part = workspace;
    const dim3 pg((C + 63) / 64, chunks), cg((C + 15) / 16);
    if (bn_vec4(C, ld, {x, y, w, b, save_mean, save_rstd, workspace, residual})) {
        float* p2 = part + (long long)chunks * C;
        int* words = (int)pg.x <= BN_ONE_GROUPS ? bn_one_words(s) : nullptr;
        if (words) {
            BnOne a{};
            a.x = x; a.w = w; a.b = b; a.res = residual; a.y = y; a.save_mean = save_mean; a.save_rstd = save_rstd; a.running_mean = running_mean;
            a.running_var = running_var; a.p1 = part; a.p2 = p2; a.words = words; a.R = R; a.C = C; a.ld = ld; a.relu = relu; a.chunks = chunks;
            a.cgroups = (int)pg.x; a.eps = eps; a.momentum = momentum;
            DIR_LAUNCH(bn_one_fwd_kernel, dim3(bn_one_grid(bn_one_fwd_kernel, a.cgroups * chunks)), dim3(256), 0, s, a);
            return check_launch("dir_bn_train_forward");
        }
        DIR_LAUNCH(bn_stats4_kernel, pg, dim3(256), 0, s, x, part, p2, R, C, ld);
        DIR_LAUNCH(bn_stats_combine_kernel, cg, dim3(256), 0, s, (const float*)part, (const float*)p2, save_mean, save_rstd, running_mean, running_var, chunks, R, C, eps, momentum);
        const long long nt = (long long)((R + 3) / 4) * (C / 4);
        DIR_LAUNCH(bn_apply_fwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, x, w, b, (const float*)save_mean, (const float*)save_rstd, y, R, C, ld, relu, residual);
        return check_launch("dir_bn_train_forward");
    }
    launch_bn_partial(pg, s, x, nullptr, nullptr, nullptr, part, nullptr, R, C, ld, 0);
    DIR_LAUNCH(bn_colsum_chunks_kernel, cg, dim3(256), 0, s, (const float*)part, save_mean, chunks, C, 1.f / R);
    launch_bn_partial(pg, s, x, nullptr, save_mean, nullptr, part, nullptr, R, C, ld, 1);
    DIR_LAUNCH(bn_stats_finalize_kernel, cg, dim3(256), 0, s, (const float*)part, (const float*)save_mean, save_mean, save_rstd, running_mean, running_var, chunks, R, C, eps, momentum);
    const long long n = (long long)R * C;
    DIR_LAUNCH(bn_apply_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, w, b, (const float*)save_mean, (const float*)save_rstd, y, n, C, ld, relu, residual);
    return check_launch("dir_bn_train_forward");
}
extern "C" int dir_bn_train_stats(const float* x, const float* w, const float* b, float* save_mean, float* save_rstd, float* pre_scale, float* pre_shift,
                                  float* running_mean, float* running_var, int R, int C, int ld, float eps, float momentum, float* workspace,
                                  long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && save_mean && save_rstd && pre_scale && pre_shift && R > BN_SMALL_R && C > 0 && ld >= C && ((running_mean == nullptr) == (running_var == nullptr)),
                "dir_bn_train_stats: bad arguments (maps of more than %d rows: smaller ones go through dir_bn_train_forward)", BN_SMALL_R);
    DIR_REQUIRE(bn_vec4(C, ld, {x, w, b, save_mean, save_rstd, workspace}), "dir_bn_train_stats: C and ld must be multiples of 4, pointers 16-byte aligned");
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_train_workspace_bytes(R, C), "dir_bn_train_stats: workspace too small (dir_bn_train_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* part = workspace; float* p2 = part + (long long)chunks * C;
    const dim3 pg((C + 63) / 64, chunks), cg((C + 15) / 16);
    DIR_LAUNCH(bn_stats4_kernel, pg, dim3(256), 0, s, x, part, p2, R, C, ld);
    DIR_LAUNCH(bn_stats_combine_pre_kernel, cg, dim3(256), 0, s, (const float*)part, (const float*)p2, w, b, save_mean, save_rstd, pre_scale, pre_shift, running_mean,
               running_var, chunks, R, C, eps, momentum, BN_CHUNK_ROWS);
    return check_launch("dir_bn_train_stats");
}
// Chunk partials -> partials of GROUPS of `per` consecutive chunks (same meaning: column sum | sum of squared deviations from the group's own mean),
// one workgroup per (16 channels, group): q1[g][c] = sum_k p1[k][c], q2[g][c] = sum_k [p2[k][c] + n_k (p1[k][c] / n_k - mean_g)^2] (exact pooling of
// variances, chunks in order).  A 64-row convolution tile over 131 072 pixels leaves 2 048 chunks per channel, which the 16 lanes per channel of the
// combine kernel walk in 14 us; groups of 32 first, in parallel, then 64 groups: 2 x 3 us.
__global__ __launch_bounds__(256) void bn_partials_coarsen_kernel(const float* p1, const float* p2, float* q1, float* q2, int chunks, int per, int crows, int R, int C) {
    __shared__ float s[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl, g = blockIdx.y;
    const int k0 = g * per, k1 = min(chunks, k0 + per);
    const int rows_g = min(R, k1 * crows) - k0 * crows;
    float a = 0.f;
    if (c < C)
        for (int k = k0 + rl; k < k1; k += 16) a += p1[(long long)k * C + c];
    s[rl][cl] = a;
    __syncthreads();
    float t = 0.f;
    for (int l = 0; l < 16; ++l) t += s[l][cl];
    const float mu = t / rows_g;
    __syncthreads();
    a = 0.f;
    if (c < C)
        for (int k = k0 + rl; k < k1; k += 16) {
            const int nk = min(crows, R - k * crows);
            const float d = p1[(long long)k * C + c] / nk - mu;
            a += fmaf((float)nk * d, d, p2[(long long)k * C + c]);
        }
    s[rl][cl] = a;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    float q = 0.f;
    for (int l = 0; l < 16; ++l) q += s[l][cl];
    q1[(long long)g * C + c] = t; q2[(long long)g * C + c] = q;
}
// the statistics from chunk partials a convolution's epilogue formed (dir_conv2d_forward_stats): p1 / p2 [ceil(R / chunk_rows)][C]; cap_rows: rows
// of C floats p1 and p2 each have room for -- with more than 256 chunks and room for the groups behind the chunks, the partials are pooled in two levels
extern "C" int dir_bn_train_stats_from_partials(float* p1, float* p2, int chunk_rows, int cap_rows, const float* w, const float* b, float* save_mean, float* save_rstd,
                                                float* pre_scale, float* pre_shift, float* running_mean, float* running_var, int R, int C, float eps, float momentum,
                                                void* stream) {
    using namespace dir;
    DIR_REQUIRE(p1 && p2 && chunk_rows > 0 && save_mean && save_rstd && R > 0 && C > 0 && ((pre_scale == nullptr) == (pre_shift == nullptr)) &&
                    ((running_mean == nullptr) == (running_var == nullptr)), "dir_bn_train_stats_from_partials: bad arguments");
    int chunks = (R + chunk_rows - 1) / chunk_rows;
    DIR_REQUIRE(cap_rows >= chunks, "dir_bn_train_stats_from_partials: cap_rows is smaller than the number of chunks");
    hipStream_t s = (hipStream_t)stream;
    const float* c1 = p1; const float* c2 = p2;
    constexpr int PER = 32;
    const int groups = (chunks + PER - 1) / PER;
    if (chunks > 256 && cap_rows >= chunks + groups) {
        float* q1 = p1 + (long long)chunks * C; float* q2 = p2 + (long long)chunks * C;
        DIR_LAUNCH(bn_partials_coarsen_kernel, dim3((C + 15) / 16, groups), dim3(256), 0, s, c1, c2, q1, q2, chunks, PER, chunk_rows, R, C);
        c1 = q1; c2 = q2; chunks = groups; chunk_rows *= PER;
    }
    DIR_LAUNCH(bn_stats_combine_pre_kernel, dim3((C + 15) / 16), dim3(256), 0, s, c1, c2, w, b, save_mean, save_rstd, pre_scale, pre_shift, running_mean,
               running_var, chunks, R, C, eps, momentum, chunk_rows);
    return check_launch("dir_bn_train_stats_from_partials");
}
// the normalisation alone, from statistics already formed (dir_bn_train_stats / _from_partials): y = act(BatchNorm(x) + residual) like dir_bn_train_forward
extern "C" int dir_bn_train_apply(const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* y, int R, int C, int ld,
                                  int relu, const float* residual, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && y && save_mean && save_rstd && R > 0 && C > 0 && ld >= C, "dir_bn_train_apply: bad arguments");
    DIR_REQUIRE(bn_vec4(C, ld, {x, y, w, b, save_mean, save_rstd, residual}), "dir_bn_train_apply: C and ld must be multiples of 4, pointers 16-byte aligned");
    const long long nt = (long long)((R + 3) / 4) * (C / 4);
    DIR_LAUNCH(bn_apply_fwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, save_mean, save_rstd, y, R, C, ld, relu, residual);
    return check_launch("dir_bn_train_apply");
}
// both in one call: statistics from the producing convolution's chunk partials, then y = act(BatchNorm(x) + residual)
extern "C" int dir_bn_train_forward_from_partials(const float* x, float* p1, float* p2, int chunk_rows, int cap_rows, const float* w, const float* b, float* y,
                                                  float* save_mean, float* save_rstd, float* running_mean, float* running_var, int R, int C, int ld, float eps,
                                                  float momentum, int relu, const float* residual, void* stream) {
    const int rc = dir_bn_train_stats_from_partials(p1, p2, chunk_rows, cap_rows, w, b, save_mean, save_rstd, nullptr, nullptr, running_mean, running_var, R, C, eps,
                                                    momentum, stream);
    if (rc != 0) return rc;
    return dir_bn_train_apply(x, w, b, save_mean, save_rstd, y, R, C, ld, relu, residual, stream);
}
extern "C" int dir_bn_train_backward(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* gx,
                                     float* gw, float* gb, int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && save_mean && save_rstd && R > 0 && C > 0 && ld >= C, "dir_bn_train_backward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (R <= BN_SMALL_R && R >= 16 && C >= 4 && bn_vec4(C, ld, {x, gy, gx, w, b, save_mean, save_rstd, gw, gb})) {
        DIR_LAUNCH(bn_mid_bwd_kernel, dim3((C + 15) / 16), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, gx, gw, gb, R, C, ld, relu);
        return check_launch("dir_bn_train_backward");
    }
    if (R <= BN_SMALL_R) {
        DIR_LAUNCH(bn_train_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, gx, gw, gb, R, C, ld, relu);
        return check_launch("dir_bn_train_backward");
    }
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_train_workspace_bytes(R, C), "dir_bn_train_backward: workspace too small (dir_bn_train_workspace_bytes)");
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* p1 = workspace; float* p2 = p1 + (long long)chunks * C; float* t1 = p2 + (long long)chunks * C; float* t2 = t1 + C;
    const dim3 pg((C + 63) / 64, chunks), cg((C + 15) / 16);
    const bool vec = bn_vec4(C, ld, {x, gy, gx, w, b, save_mean, save_rstd, workspace});
    int* words = vec && (int)pg.x <= BN_ONE_GROUPS ? bn_one_words(s) : nullptr;
    if (words) {
        BnOne a{};
        a.x = x; a.gy = gy; a.w = w; a.b = b; a.gx = gx; a.save_mean = const_cast<float*>(save_mean); a.save_rstd = const_cast<float*>(save_rstd);
        a.gw = gw; a.gb = gb; a.p1 = p1; a.p2 = p2; a.t1 = t1; a.t2 = t2; a.words = words; a.R = R; a.C = C; a.ld = ld; a.relu = relu; a.chunks = chunks;
        a.cgroups = (int)pg.x;
        DIR_LAUNCH(bn_one_bwd_kernel, dim3(bn_one_grid(bn_one_bwd_kernel, a.cgroups * chunks)), dim3(256), 0, s, a);
        return check_launch("dir_bn_train_backward");
    }
    if (vec) DIR_LAUNCH(bn_bwd_partial4_kernel, pg, dim3(256), 0, s, x, gy, w, b, save_mean, save_rstd, p1, p2, R, C, ld, relu);
    else launch_bn_partial(pg, s, x, gy, save_mean, save_rstd, p1, p2, R, C, ld, 2, w, b, relu);
    DIR_LAUNCH(bn_bwd_combine_kernel, cg, dim3(256), 0, s, (const float*)p1, (const float*)p2, t1, t2, gb, gw, chunks, C);
    if (gx) {
        if (vec) {
            const long long nt = (long long)((R + 3) / 4) * (C / 4);
            DIR_LAUNCH(bn_apply_bwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, (const float*)t1, (const float*)t2, gx, R, C, ld, relu);
        } else {
            const long long n = (long long)R * C;
            DIR_LAUNCH(bn_apply_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, (const float*)t1, (const float*)t2, gx, n, R, C, ld, relu);
        }
    }
    return check_launch("dir_bn_train_backward");
}

extern "C" int dir_bn_train_backward_from_partials(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                                   const float* p1, const float* p2, int chunks, float* gx, float* gw, float* gb, int R, int C, int ld, int relu,
                                                   float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && save_mean && save_rstd && p1 && p2 && chunks > 0 && R > 0 && C > 0 && ld >= C && workspace && workspace_bytes >= 2ll * C * 4,
                "dir_bn_train_backward_from_partials: bad arguments (workspace: 2 C floats)");
    DIR_REQUIRE(bn_vec4(C, ld, {x, gy, gx, w, b, save_mean, save_rstd, workspace}), "dir_bn_train_backward_from_partials: C and ld must be multiples of 4, pointers 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* t1 = workspace; float* t2 = t1 + C;
    DIR_LAUNCH(bn_bwd_combine_kernel, dim3((C + 15) / 16), dim3(256), 0, s, p1, p2, t1, t2, gb, gw, chunks, C);
    if (gx) {
        const long long nt = (long long)((R + 3) / 4) * (C / 4);
        DIR_LAUNCH(bn_apply_bwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, (const float*)t1, (const float*)t2, gx, R, C, ld, relu);
    }
    return check_launch("dir_bn_train_backward_from_partials");
}

// ---- SyncBN building blocks (dir_amd/train/ops.py: sync_bn_fwd / sync_bn_bwd put the collectives between them)
extern "C" long long dir_bn_sync_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return -1;
    const long long chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    return 2 * chunks * C * 4;
}
extern "C" int dir_bn_sync_local_stats(const float* x, float* part, int R, int C, int ld, float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && part && R > 0 && C > 0 && ld >= C, "dir_bn_sync_local_stats: bad arguments");
    DIR_REQUIRE(bn_vec4(C, ld, {x, part, workspace}), "dir_bn_sync_local_stats: C and ld must be multiples of 4, pointers 16-byte aligned");
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_sync_workspace_bytes(R, C), "dir_bn_sync_local_stats: workspace too small (dir_bn_sync_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* p1 = workspace; float* p2 = p1 + (long long)chunks * C;
    DIR_LAUNCH(bn_stats4_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, s, x, p1, p2, R, C, ld);
    DIR_LAUNCH(bn_stats_local_kernel, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)p1, (const float*)p2, part, part + C, chunks, R, C);
    return check_launch("dir_bn_sync_local_stats");
}
extern "C" int dir_bn_sync_combine(const float* parts, int world, int C, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                   void* stream) {
    using namespace dir;
    DIR_REQUIRE(parts && mean && var && world > 0 && C > 0 && ((running_mean == nullptr) == (running_var == nullptr)), "dir_bn_sync_combine: bad arguments");
    DIR_LAUNCH(bn_sync_combine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, parts, world, C, mean, var, running_mean, running_var, momentum);
    return check_launch("dir_bn_sync_combine");
}
extern "C" int dir_bn_sync_backward_sums(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                         float* sums, int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && save_mean && save_rstd && sums && R > 0 && C > 0 && ld >= C, "dir_bn_sync_backward_sums: bad arguments");
    DIR_REQUIRE(bn_vec4(C, ld, {x, gy, w, b, save_mean, save_rstd, sums, workspace}), "dir_bn_sync_backward_sums: C and ld must be multiples of 4, pointers 16-byte aligned");
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_sync_workspace_bytes(R, C), "dir_bn_sync_backward_sums: workspace too small (dir_bn_sync_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* p1 = workspace; float* p2 = p1 + (long long)chunks * C;
    DIR_LAUNCH(bn_bwd_partial4_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, s, x, gy, w, b, save_mean, save_rstd, p1, p2, R, C, ld, relu);
    DIR_LAUNCH(bn_bwd_sums_kernel, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)p1, (const float*)p2, sums, chunks, C);
    return check_launch("dir_bn_sync_backward_sums");
}
extern "C" int dir_bn_sync_backward_apply(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                          const float* sums_pooled, float* gx, int R, float rows_pooled, int C, int ld, int relu, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && save_mean && save_rstd && sums_pooled && gx && R > 0 && C > 0 && ld >= C && rows_pooled >= (float)R, "dir_bn_sync_backward_apply: bad arguments");
    DIR_REQUIRE(bn_vec4(C, ld, {x, gy, gx, w, b, save_mean, save_rstd, sums_pooled}), "dir_bn_sync_backward_apply: C and ld must be multiples of 4, pointers 16-byte aligned");
    const long long nt = (long long)((R + 3) / 4) * (C / 4);
    DIR_LAUNCH(bn_apply_bwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, x, w, b, save_mean, save_rstd, sums_pooled,
               sums_pooled + C, gx, R, C, ld, relu, rows_pooled);
    return check_launch("dir_bn_sync_backward_apply");
}

// BatchNorm with FROZEN statistics inside a training pass (nn.BatchNorm*.eval() under model.train(): the fine-tuning form, and the form in which
// the reference's whole-step gradient is reproducible to 4e-5 -- tests/golden G20e): y = (x - running_mean) / sqrt(running_var + eps) * w + b, the
// running statistics untouched.  Backward: g w = sum gy (x - mean) rstd, g b = sum gy (the same deterministic chunked column sums as the
// training-mode backward), g x = gy w rstd (no batch-statistics terms).  save_mean / save_rstd are written by the forward for the backward.
extern "C" long long dir_bn_frozen_workspace_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return -1;
    const long long chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    return (2 * chunks + 2) * C * 4;
}
extern "C" int dir_bn_frozen_forward(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd, const float* running_mean,
                                     const float* running_var, int R, int C, int ld, float eps, int relu, const float* residual, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && y && save_mean && save_rstd && running_mean && running_var && R > 0 && C > 0 && ld >= C, "dir_bn_frozen_forward: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DIR_LAUNCH(bn_frozen_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var, save_mean, save_rstd, C, eps);
    if (bn_vec4(C, ld, {x, y, w, b, save_mean, save_rstd, residual})) {
        const long long nt = (long long)((R + 3) / 4) * (C / 4);
        DIR_LAUNCH(bn_apply_fwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, x, w, b, (const float*)save_mean, (const float*)save_rstd, y, R, C, ld, relu, residual);
    } else {
        const long long n = (long long)R * C;
        DIR_LAUNCH(bn_apply_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, w, b, (const float*)save_mean, (const float*)save_rstd, y, n, C, ld, relu, residual);
    }
    return check_launch("dir_bn_frozen_forward");
}
extern "C" int dir_bn_frozen_backward(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* gx,
                                      float* gw, float* gb, int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && x && save_mean && save_rstd && R > 0 && C > 0 && ld >= C, "dir_bn_frozen_backward: bad arguments");
    DIR_REQUIRE(workspace && workspace_bytes >= dir_bn_frozen_workspace_bytes(R, C), "dir_bn_frozen_backward: workspace too small (dir_bn_frozen_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (R + BN_CHUNK_ROWS - 1) / BN_CHUNK_ROWS;
    float* p1 = workspace; float* p2 = p1 + (long long)chunks * C; float* t1 = p2 + (long long)chunks * C; float* t2 = t1 + C;
    const dim3 pg((C + 63) / 64, chunks), cg((C + 15) / 16);
    const bool vec = bn_vec4(C, ld, {x, gy, gx, w, b, save_mean, save_rstd, workspace});
    if (vec) DIR_LAUNCH(bn_bwd_partial4_kernel, pg, dim3(256), 0, s, x, gy, w, b, save_mean, save_rstd, p1, p2, R, C, ld, relu);
    else launch_bn_partial(pg, s, x, gy, save_mean, save_rstd, p1, p2, R, C, ld, 2, w, b, relu);
    DIR_LAUNCH(bn_bwd_combine_kernel, cg, dim3(256), 0, s, (const float*)p1, (const float*)p2, t1, t2, gb, gw, chunks, C);
    if (gx) {
        DIR_LAUNCH(zero_f32_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, s, t1, 2 * C);          // no batch-statistics terms: the apply kernels' column means are zero
        if (vec) {
            const long long nt = (long long)((R + 3) / 4) * (C / 4);
            DIR_LAUNCH(bn_apply_bwd4_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, (const float*)t1, (const float*)t2, gx, R, C, ld, relu);
        } else {
            const long long n = (long long)R * C;
            DIR_LAUNCH(bn_apply_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gy, x, w, b, save_mean, save_rstd, (const float*)t1, (const float*)t2, gx, n, R, C, ld, relu);
        }
    }
    return check_launch("dir_bn_frozen_backward");
}

extern "C" int dir_relu_forward(const float* x, float* y, long long n, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && y && n > 0, "dir_relu_forward: bad arguments");
    DIR_LAUNCH(relu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("dir_relu_forward");
}
extern "C" int dir_relu_backward(const float* gy, const float* y, float* gx, long long n, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && y && gx && n > 0, "dir_relu_backward: bad arguments");
    DIR_LAUNCH(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, y, gx, n);
    return check_launch("dir_relu_backward");
}
extern "C" int dir_upsample_nearest_add_f32(const float* src, float* dst, int B, int h, int w, int C, int factor, void* stream) {
    using namespace dir;
    DIR_REQUIRE(src && dst && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && factor >= 1 && (factor & (factor - 1)) == 0 &&
                    !((uintptr_t)src & 15) && !((uintptr_t)dst & 15), "dir_upsample_nearest_add_f32: bad arguments (C %% 4 == 0, power-of-two factor, 16-byte aligned)");
    const long long n4 = (long long)B * h * factor * w * factor * (C / 4);
    DIR_LAUNCH(upsample_nearest_add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, h, w, C, factor, n4);
    return check_launch("dir_upsample_nearest_add_f32");
}
extern "C" int dir_upsample_nearest_backward_f32(const float* gy, float* gx, int B, int h, int w, int C, int factor, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && gx && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && factor >= 1 && (factor & (factor - 1)) == 0 &&
                    !((uintptr_t)gy & 15) && !((uintptr_t)gx & 15), "dir_upsample_nearest_backward_f32: bad arguments (C %% 4 == 0, power-of-two factor, 16-byte aligned)");
    const long long n4 = (long long)B * h * w * (C / 4);
    DIR_LAUNCH(upsample_nearest_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, h, w, C, factor, n4);
    return check_launch("dir_upsample_nearest_backward_f32");
}
extern "C" int dir_pgcn_adjacency_forward(const float* e1, float* A, void* stream) {
    using namespace dir;
    DIR_REQUIRE(e1 && A, "dir_pgcn_adjacency_forward: null pointer");
    DIR_LAUNCH(pgcn_adj_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e1, A);
    return check_launch("dir_pgcn_adjacency_forward");
}
extern "C" int dir_pgcn_adjacency_backward(const float* e1, const float* gz, const float* h1, float* scratch40, float* g_e1, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(e1 && gz && h1 && scratch40 && g_e1 && B > 0, "dir_pgcn_adjacency_backward: bad arguments");
    DIR_LAUNCH(pgcn_adj_bwd_edge_kernel, dim3(40), dim3(64), 0, (hipStream_t)stream, gz, h1, scratch40, B);
    DIR_LAUNCH(pgcn_adj_bwd_softmax_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, e1, scratch40, g_e1);
    return check_launch("dir_pgcn_adjacency_backward");
}

extern "C" int dir_grid_rows_forward(const float* feat_nhwc, const float* uv, float* rows, int B, int S, int C, void* stream) {
    using namespace dir;
    DIR_REQUIRE(feat_nhwc && uv && rows && B > 0 && S > 0 && C > 0, "dir_grid_rows_forward: bad arguments");
    GridArgs2 a{};
    a.feat = feat_nhwc; a.uv = uv; a.rows = rows; a.B = B; a.S = S; a.C = C;
    DIR_LAUNCH(grid_rows_fwd_kernel, dim3(B * 21), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dir_grid_rows_forward");
}
extern "C" int dir_grid_rows_backward(const float* const* g_rows_h, const float* const* uv_h, int hands, float* g_feat_nhwc, int B, int S, int C,
                                      int zero_first, void* stream) {
    using namespace dir;
    DIR_REQUIRE(g_rows_h && uv_h && g_feat_nhwc && (hands == 1 || hands == 2) && B > 0 && S > 0 && C > 0, "dir_grid_rows_backward: bad arguments");
    GridArgs2 a{};
    for (int h = 0; h < hands; ++h) { DIR_REQUIRE(g_rows_h[h] && uv_h[h], "dir_grid_rows_backward: null hand"); a.grows[h] = g_rows_h[h]; a.uvs[h] = uv_h[h]; }
    a.gfeat = g_feat_nhwc; a.B = B; a.S = S; a.C = C; a.hands = hands;
    hipStream_t s = (hipStream_t)stream;
    if (zero_first && hipMemsetAsync(g_feat_nhwc, 0, (size_t)B * S * S * C * sizeof(float), s) != hipSuccess) { set_error("dir_grid_rows_backward: memset failed"); return DIR_E_LAUNCH; }
    DIR_LAUNCH(grid_rows_bwd_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, s, a);
    return check_launch("dir_grid_rows_backward");
}

// dst_t += alpha * src_t for up to AXPY_MULTI tensors in ONE launch (the table travels in the kernel arguments): moving the 556 parameter
// gradients of a training step into the flat all-reduce bucket was 556 launches of a few microseconds each
namespace {
constexpr int AXPY_MULTI = 40, AXPY_CHUNK = 8192;      // tensors per launch; elements per workgroup (a tensor gets ceil(n / chunk) of them)
struct AxpyTable { float* dst[AXPY_MULTI]; const float* src[AXPY_MULTI]; long long n[AXPY_MULTI]; int first[AXPY_MULTI + 1]; };
__global__ __launch_bounds__(256) void axpy_multi_kernel(AxpyTable t, float alpha, int count) {
    int k = 0;
    while (k + 1 < count && (int)blockIdx.x >= t.first[k + 1]) ++k;          // <= 40 uniform (scalar) comparisons
    float* d = t.dst[k];
    const float* s = t.src[k];
    const long long n = t.n[k], i0 = (long long)(blockIdx.x - t.first[k]) * AXPY_CHUNK, i1 = min(n, i0 + AXPY_CHUNK);
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        for (long long i = i0 + threadIdx.x * 4; i + 3 < i1; i += 1024) {
            float4 a = *reinterpret_cast<float4*>(d + i);
            const float4 b = *reinterpret_cast<const float4*>(s + i);
            a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w;
            *reinterpret_cast<float4*>(d + i) = a;
        }
        for (long long i = i0 + ((i1 - i0) & ~3ll) + threadIdx.x; i < i1; i += 256) d[i] += alpha * s[i];
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) d[i] += alpha * s[i];
    }
}
}  // namespace
extern "C" int dir_axpy_multi_f32(float* const* dst_host, const float* const* src_host, const long long* n_host, int count, float alpha, void* stream) {
    using namespace dir;
    DIR_REQUIRE(count >= 0 && (count == 0 || (dst_host && src_host && n_host)), "dir_axpy_multi_f32: bad arguments");
    for (int base = 0; base < count; base += AXPY_MULTI) {
        AxpyTable t;
        const int m = count - base < AXPY_MULTI ? count - base : AXPY_MULTI;
        long long blocks = 0;
        for (int k = 0; k < m; ++k) {
            DIR_REQUIRE(dst_host[base + k] && src_host[base + k] && n_host[base + k] >= 0, "dir_axpy_multi_f32: null tensor %d", base + k);
            t.dst[k] = dst_host[base + k]; t.src[k] = src_host[base + k]; t.n[k] = n_host[base + k];
            t.first[k] = (int)blocks;
            blocks += (n_host[base + k] + AXPY_CHUNK - 1) / AXPY_CHUNK;
            DIR_REQUIRE(blocks < (1ll << 30), "dir_axpy_multi_f32: too many elements");
        }
        t.first[m] = (int)blocks;
        if (blocks > 0) DIR_LAUNCH(axpy_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, alpha, m);
    }
    return check_launch("dir_axpy_multi_f32");
}

extern "C" int dir_axpy_f32(float* dst, const float* src, long long n, float alpha, void* stream) {
    using namespace dir;
    DIR_REQUIRE(dst && src && n > 0, "dir_axpy_f32: bad arguments");
    DIR_LAUNCH(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, alpha);
    return check_launch("dir_axpy_f32");
}
extern "C" int dir_stage_positions(const float* xyz_left, const float* xyz_right, const float* offset, float* pos_left, float* pos_right,
                                   float* gpos_left, float* gpos_right, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(xyz_left && xyz_right && offset && pos_left && pos_right && gpos_left && gpos_right && B > 0, "dir_stage_positions: bad arguments");
    DIR_LAUNCH(stage_positions_kernel, dim3((B * 63 + 255) / 256), dim3(256), 0, (hipStream_t)stream, xyz_left, xyz_right, offset, pos_left, pos_right,
               gpos_left, gpos_right, B * 63);
    return check_launch("dir_stage_positions");
}

static bool wgrad_flatk(const dir_conv_desc* d) { return d->Cin < 16 && d->kh * d->kw > 1; }      // few input channels: flatten (tap, channel)
static int wgrad_chunks(const dir_conv_desc* d, long long M) {
    const long long per = wgrad_flatk(d) ? (long long)((d->kh * d->kw * d->Cin + GT - 1) / GT) * ((d->Cout + GT - 1) / GT)
                                         : (long long)((d->Cin + GT - 1) / GT) * ((d->Cout + GT - 1) / GT) * d->kh * d->kw;
    long long c = (1024 + per - 1) / per;                 // >= ~1024 workgroups in flight
    const long long cmax = (M + 255) / 256;               // but at least 256 pixels (16 steps) per chunk
    if (c > cmax) c = cmax;
    return (int)(c < 1 ? 1 : c);
}
extern "C" long long dir_conv2d_wgrad_workspace_bytes(const dir_conv_desc* d) {
    if (!d) return -1;
    const int Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    const int Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    const long long M = (long long)d->B * Ho * Wo;
    const int c = wgrad_chunks(d, M);
    return c > 1 ? (long long)c * d->Cout * d->kh * d->kw * d->Cin * 4 : 0;
}
extern "C" int dir_conv2d_wgrad_f32(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                                    long long workspace_bytes, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && x && gy && gw, "dir_conv2d_wgrad_f32: null pointer");
    DIR_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0,
                "dir_conv2d_wgrad_f32: bad geometry");
    WgradArgs a;
    a.x = x; a.gy = gy; a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.in_cs = d->in_cstride ? d->in_cstride : d->Cin; a.in_co = d->in_coff;
    a.Cout = d->Cout; a.gy_cs = d->out_cstride ? d->out_cstride : d->Cout; a.gy_co = d->out_coff;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho > 0 ? d->Ho : (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    a.Wo = d->Wo > 0 ? d->Wo : (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DIR_REQUIRE(a.Ho > 0 && a.Wo > 0, "dir_conv2d_wgrad_f32: empty output");
    const long long M = (long long)d->B * a.Ho * a.Wo;
    DIR_REQUIRE(M < (1ll << 31), "dir_conv2d_wgrad_f32: too many output pixels");
    a.M = (int)M;
    const int chunks = wgrad_chunks(d, M);
    const long long n = (long long)d->Cout * d->kh * d->kw * d->Cin;
    DIR_REQUIRE(chunks == 1 || (workspace && workspace_bytes >= (long long)chunks * n * 4), "dir_conv2d_wgrad_f32: workspace too small (dir_conv2d_wgrad_workspace_bytes)");
    a.chunk = (int)(((M + chunks - 1) / chunks + GK - 1) / GK * GK);
    a.tiles_ci = (d->Cin + GT - 1) / GT;
    const bool direct = chunks == 1 && !accumulate;
    DIR_REQUIRE(direct || workspace, "dir_conv2d_wgrad_f32: accumulate needs a workspace of at least the weight size");
    DIR_REQUIRE(direct || workspace_bytes >= (long long)chunks * n * 4, "dir_conv2d_wgrad_f32: workspace too small");
    a.out = direct ? gw : workspace;
    hipStream_t s = (hipStream_t)stream;
    const bool vec4 = d->Cin % 4 == 0 && d->Cout % 4 == 0 && a.in_cs % 4 == 0 && a.in_co % 4 == 0 && a.gy_cs % 4 == 0 && a.gy_co % 4 == 0 &&
                      ((uintptr_t)x & 15) == 0 && ((uintptr_t)gy & 15) == 0;
    const dim3 wgrid(a.tiles_ci * ((d->Cout + GT - 1) / GT), d->kh * d->kw, chunks);
    if (wgrad_flatk(d)) {
        a.tiles_ci = (d->kh * d->kw * d->Cin + GT - 1) / GT;
        DIR_LAUNCH(conv_wgrad_flatk_kernel, dim3(a.tiles_ci * ((d->Cout + GT - 1) / GT), 1, chunks), dim3(256), 0, s, a);
    } else if (vec4) {
        a.chunk = (a.chunk + WK - 1) / WK * WK;           // whole 32-pixel steps per chunk
        DIR_LAUNCH(conv_wgrad4_kernel, wgrid, dim3(256), 0, s, a);
    } else DIR_LAUNCH(conv_wgrad_kernel, wgrid, dim3(256), 0, s, a);
    if (!direct) DIR_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace, gw, n, chunks, accumulate);
    return check_launch("dir_conv2d_wgrad_f32");
}
