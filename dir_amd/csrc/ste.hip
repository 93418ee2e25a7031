// a6: mixSTE spatial transformer encoder over the 42 joint tokens of both hands, one workgroup per sample.
// Replaces transformer/mixSTE.py:194-205 (STE.forward), :129-131 (Block), :76-97 (Attention), :39-44 (Mlp):
//   x += pos_embed ; for blocks 1..depth-1 (block 0 is never executed by the reference):
//       x += proj(softmax(q k^T * 32^-0.5) v) ; x += fc2(gelu_erf(fc1(LN(x)))) ; x = spatial_norm(x)
//   y = Linear(LN_1e-5(x)).
// All activations ([48][128] residual stream, [48][384] qkv / [48][256] MLP hidden, 4 x [48][44] attention
// probabilities) stay in LDS (~154 KB of the CU's 160 KB); the six Linear layers per block, q k^T and P v run on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32 = exact fp32 fmaf chains, so the 1e-4 mm MANO budget downstream is untouched) with
// the weight fragment prefetched k-major from L2; LayerNorm and softmax reduce with wavefront shuffles (one wave
// per token / per attention row).  ~36 MFLOP per sample.
#include "dir_common.h"
#include "dir_mfma.h"

namespace {

constexpr int NT = 42, NTP = 48, D = 128, HEADS = 4, HD = 32, NTHREADS = 512, NWAVES = 8;
// LDS row strides: +2 floats makes the MFMA A-operand reads (lane -> row l&15, k l>>4) bank-conflict free
constexpr int LDX = D + 2, LDQ = 384 + 2, LDH = 256 + 2;
constexpr int LDP = 45;               // attention rows: 42 keys + 2 zero columns (K of the P.V MFMA is padded to 44);
                                      // odd stride: thread-per-row softmax and the MFMA operand reads are conflict free
constexpr int KP = 44;

using dir::f32x4;

struct SteArgs {
    dir_ste_params p;
    float* x_inout; const float* x_in; float* y; int nblocks;
    long long* stamps;        // DIR_STAMPS=ste: s_memtime at every phase boundary of workgroup 0 (tuning aid, else NULL)
};

// sum over the 16 lanes of a DPP row (all lanes receive it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror.
// (wave shuffles go through ds_bpermute, ~10x the latency of a DPP modifier)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// LayerNorm over the channel dim (two-pass variance like ATen): 16 lanes per token (channels li, li+16, ...), four
// tokens per wave, 32 tokens per round.  s_out may alias s_in (every element is read and written by the same lane).
constexpr int LDB = 136, LDHB = 264;  // bf16 activation rows (elements): [48][128 + 8], [48][256 + 8] -- 16-byte aligned rows whose
                                      // 16-lane ds_read_b128 (row l & 15) tile all 64 banks
__device__ __forceinline__ unsigned short f2bf_rne(float f) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{f, 0.f}, b2_t)) & 0xffffu);
}
// bf16 mode only: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the GELU output that
// follows) -- ocml's erff cost 3 us per block in fc1's epilogue (DIR_STAMPS=ste).  The fp32 mode keeps erff.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x), t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
template <bool BF16OUT = false>
__device__ __forceinline__ void layernorm_tokens(const float* s_in, float* s_out, const float* w, const float* b,
                                                 float eps, int wave, int lane) {
    const int li = lane & 15;
    float wv[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { wv[e] = w[li + 16 * e]; bv[e] = b[li + 16 * e]; }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int t = (rr * NWAVES + wave) * 4 + (lane >> 4);
        const bool live = t < NT;
        const float* src = s_in + (live ? t : 0) * LDX + li;
        float v[8], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = src[16 * e]; sum += v[e]; }
        const float mean = row16_sum(sum) * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; sq = fmaf(v[e], v[e], sq); }
        const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / D) + eps);
        if (live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float o = v[e] * rstd * wv[e] + bv[e];
                if constexpr (BF16OUT) reinterpret_cast<unsigned short*>(s_out)[t * LDB + li + 16 * e] = f2bf_rne(o);   // the Linear's operand
                else s_out[t * LDX + li + 16 * e] = o;
            }
        }
    }
}

// out[t][n] = f( sum_k s_in[t][k] * Wt[k][n] + bias[n] ), t < 42, on the fp32 matrix cores.
// v_mfma_f32_16x16x4_f32: A lane (i = l&15, k = l>>4), B lane (k = l>>4, j = l&15), D lane col = l&15, row = 4*(l>>4)+r;
// the result is a k-ordered fp32 fmaf chain (exact fp32).  A wave owns a 16-column tile for all three 16-row token
// tiles; its whole K x 16 weight fragment is prefetched into registers first (one L2 latency per tile).
template <int K, typename F>
__device__ __forceinline__ void linear_mfma(const float* s_in, int ldi, const float* __restrict__ Wt,
                                            const float* __restrict__ bias, int N, int wave, int lane, F store) {
    const int li = lane & 15, lk = lane >> 4;
    for (int nt = wave; nt < N / 16; nt += NWAVES) {
        const int n = nt * 16 + li;
        float bv[K / 4];
#pragma unroll
        for (int kk = 0; kk < K / 4; ++kk) bv[kk] = Wt[(long long)(4 * kk + lk) * N + n];
        f32x4 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ap = s_in + li * ldi + lk;
#pragma unroll
        for (int kk = 0; kk < K / 4; ++kk) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m * 16 * ldi + 4 * kk], bv[kk], acc[m], 0, 0, 0);
        }
        const float bb = bias[n];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = m * 16 + lk * 4 + r;
                if (t < NT) store(t, n, acc[m][r] + bb);
            }
    }
}

// The same Linear on the bf16 matrix cores (bf16 throughput mode = the reference's autocast semantics: nn.Linear in bf16,
// LayerNorm / softmax / residual stream in fp32).  W is the PyTorch-native [N][K] bf16 matrix: a lane's B operand
// (column n, 8 consecutive k) is one 16-byte load; the A operand is converted from the fp32 LDS activations on the fly.
// v_mfma_f32_16x16x32_bf16: A lane (row l&15, k 8*(l>>4)..+7), B lane (col l&15, same k), D as the fp32 variant.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
template <int K, typename F>
__device__ __forceinline__ void linear_mfma_bf16(const float* s_in, int ldi, const unsigned short* __restrict__ W,
                                                 const float* __restrict__ bias, int N, int wave, int lane, F store) {
    const int li = lane & 15, lk = lane >> 4;
    for (int nt = wave; nt < N / 16; nt += NWAVES) {
        const int n = nt * 16 + li;
        bf16x8_t bv[K / 32];
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) bv[kk] = *reinterpret_cast<const bf16x8_t*>(W + (long long)n * K + kk * 32 + lk * 8);
        f32x4 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float2* ap = reinterpret_cast<const float2*>(s_in + (m * 16 + li) * ldi + kk * 32 + lk * 8);   // 8-byte aligned rows
                bf16x8_t av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 t = ap[e];
                    av[2 * e] = (__bf16)t.x;
                    av[2 * e + 1] = (__bf16)t.y;
                }
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[kk], acc[m], 0, 0, 0);
            }
        }
        const float bb = bias[n];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = m * 16 + lk * 4 + r;
                if (t < NT) store(t, n, acc[m][r] + bb);
            }
    }
}

// bf16 mode, activations already rounded to bf16 in LDS (by the producer: LayerNorm / attention output / GELU): every A fragment
// is ONE 16-byte LDS read instead of four 8-byte fp32 reads + conversion, and it is shared by all of the wave's output tiles
// (tile i = wave + 8 i).  Measured with DIR_STAMPS=ste: the fp32-operand version spent 1.1 us per 16-column tile re-reading the
// whole [48][K] fp32 activation matrix -- LDS bandwidth, not the weights' L2 latency, bounded the Linears.  Same values reach
// the matrix cores (one round-to-nearest-even of the same fp32 numbers), so results are bit-identical.
template <int K, int NTW>
struct WFrag { bf16x8_t bv[NTW][K / 32]; float bb[NTW]; };
// all global loads of the wave's weight tiles + bias; issued one phase before gemm_bf16_act needs them (measured with
// DIR_STAMPS=ste: loaded at the point of use, the first-touch latency of a Linear's weights -- ~2 us, the slowest wave's more --
// was waited for by the whole workgroup at the next barrier)
template <int K, int NTW>
__device__ __forceinline__ void load_wfrag(const float* Wf, const float* __restrict__ bias, int N, int wave, int lane, WFrag<K, NTW>& w) {
    const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(Wf);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = min(wave + NWAVES * i, N / 16 - 1) * 16 + li;
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) w.bv[i][kk] = *reinterpret_cast<const bf16x8_t*>(W + (long long)n * K + kk * 32 + lk * 8);
        w.bb[i] = bias[n];
    }
}
template <int K, int NTW, typename F>
__device__ __forceinline__ void gemm_bf16_act(const unsigned short* s_a, int lda, const WFrag<K, NTW>& w, int N, int wave, int lane, F store) {
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[NTW][3];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < K / 32; ++kk)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(s_a + (m * 16 + li) * lda + kk * 32 + lk * 8);
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w.bv[i][kk], acc[i][m], 0, 0, 0);
        }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + NWAVES * i;
        if (nt < N / 16) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = m * 16 + lk * 4 + r;
                    if (t < NT) store(t, nt * 16 + li, acc[i][m][r] + w.bb[i]);
                }
        }
    }
}

// WBF16: Linear weights are bf16 [out][in] (dir_ste_params.weight_dtype == DIR_DT_BF16), else fp32 k-major [in][out]
template <int K, bool WBF16, typename F>
__device__ __forceinline__ void linear(const float* s_in, int ldi, const float* W, const float* bias, int N, int wave, int lane, F store) {
    if constexpr (WBF16) linear_mfma_bf16<K>(s_in, ldi, reinterpret_cast<const unsigned short*>(W), bias, N, wave, lane, store);
    else linear_mfma<K>(s_in, ldi, W, bias, N, wave, lane, store);
}

template <bool WBF16>
__global__ __launch_bounds__(NTHREADS) void ste_kernel(SteArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[2 * NTP * LDX + NTP * LDQ + HEADS * NTP * LDP];   // 158,592 B
    float* s_x = sm;                      // [48][130] residual stream (rows 42..47: zero padding of the MFMA row tile)
    float* s_n = s_x + NTP * LDX;         // [48][130] LayerNorm output / attention output
    float* s_big = s_n + NTP * LDX;       // [48][386] qkv, later [48][258] MLP hidden
    float* s_p = s_big + NTP * LDQ;       // [4][48][44] attention scores / probabilities (padding: zero columns 42,43)
    // bf16 mode: the Linear operands live as bf16 -- s_n's space holds [48][136] bf16 (LayerNorm / attention output), the MLP hidden
    // [48][264] bf16 sits at the start of s_big (free once the attention has consumed q, k, v)
    unsigned short* s_nb = reinterpret_cast<unsigned short*>(s_n);
    unsigned short* s_hb = reinterpret_cast<unsigned short*>(s_big);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && b == 0 && tid == 0) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();

    for (int i = tid; i < (NTP - NT) * LDX; i += NTHREADS) { s_x[NT * LDX + i] = 0.f; s_n[NT * LDX + i] = 0.f; }
    for (int i = tid; i < (NTP - NT) * LDQ; i += NTHREADS) s_big[NT * LDQ + i] = 0.f;
    for (int i = tid; i < HEADS * NTP * LDP; i += NTHREADS) s_p[i] = 0.f;   // padding rows/columns must stay finite
    const float* xin = a.x_in + (long long)b * NT * D;
    for (int i = tid; i < NT * D; i += NTHREADS) {
        const float v = xin[i] + a.p.pos_embed[i];                      // x += spatial_pos_embed (mixSTE.py:196)
        s_x[(i >> 7) * LDX + (i & 127)] = v;
        if (a.x_inout) a.x_inout[(long long)b * NT * D + i] = v;        // the reference mutates its input in place
    }
    __syncthreads(); stamp();

    // bf16 mode: each Linear's weight fragments are requested one phase ahead (qkv's before LayerNorm 1 / during the previous
    // block's fc2, proj's before the attention, fc1's before LayerNorm 2, fc2's before fc1's GELU epilogue)
    WFrag<D, WBF16 ? 3 : 1> wq;
    WFrag<D, 1> wp, wh;
    WFrag<D, WBF16 ? 2 : 1> w1;
    WFrag<256, 1> w2;
    if constexpr (WBF16) {
        if (a.nblocks > 0) load_wfrag<D, 3>(a.p.blocks[0].qkv_wt, a.p.blocks[0].qkv_b, 384, wave, lane, wq);
    }
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const dir_ste_block& P = a.p.blocks[blk];
        // ---- attention branch
        layernorm_tokens<WBF16>(s_x, s_n, P.ln1_w, P.ln1_b, 1e-6f, wave, lane);
        __syncthreads(); stamp();
        if constexpr (WBF16) {
            load_wfrag<D, 1>(P.proj_wt, P.proj_b, D, wave, lane, wp);
            gemm_bf16_act<D, 3>(s_nb, LDB, wq, 384, wave, lane, [&](int t, int n, float v) { s_big[t * LDQ + n] = v; });
        }
        else linear<D, WBF16>(s_n, LDX, P.qkv_wt, P.qkv_b, 384, wave, lane, [&](int t, int n, float v) { s_big[t * LDQ + n] = v; });
        __syncthreads(); stamp();
        // scores S = q k^T * 32^-0.5 per head on the matrix cores.  qkv column layout is (3, heads, 32) (mixSTE.py:78):
        // q = [0,128), k = [128,256), v = [256,384).  36 tiles of 16x16 (4 heads x 3 x 3), K = 32.
        const float scale = 0.17677669529663687f;                       // 32 ** -0.5
        {
            const int li = lane & 15, lk = lane >> 4;
            // a wave's tiles (wave, wave + 8, ...: five or four of them) advance together through K: five independent accumulator
            // chains hide the MFMA latency a single dependent chain of 8 exposes; per tile the k order is unchanged (exact fp32)
            constexpr int TPW = (HEADS * 9 + NWAVES - 1) / NWAVES;
            const float* qa[TPW]; const float* kb[TPW];
            f32x4 acc[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = min(wave + NWAVES * i, HEADS * 9 - 1);
                const int h = tile / 9, mt = (tile - h * 9) / 3, nt = tile - h * 9 - mt * 3;
                qa[i] = s_big + (mt * 16 + li) * LDQ + h * HD + lk;
                kb[i] = s_big + (nt * 16 + li) * LDQ + 128 + h * HD + lk;
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < HD / 4; ++kk)
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i][4 * kk], kb[i][4 * kk], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = wave + NWAVES * i;
                if (tile < HEADS * 9) {
                    const int h = tile / 9, mt = (tile - h * 9) / 3, nt = tile - h * 9 - mt * 3;
                    const int col = nt * 16 + li;
                    if (col < KP) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            s_p[(h * NTP + mt * 16 + lk * 4 + r) * LDP + col] = col < NT ? acc[i][r] * scale : 0.f;
                    }
                }
            }
        }
        __syncthreads(); stamp();
        // softmax: four lanes per row (elements q, q + 4, ...; max / sum through two DPP quad permutes), 128 rows per pass --
        // every wave works (one thread per row kept 5 of 8 waves idle for ~3 us per block)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 128 + (tid >> 2), q = tid & 3;
            const bool live = row < HEADS * NT;
            float* pr = s_p + (((live ? row : 0) / NT) * NTP + (live ? row : 0) % NT) * LDP;
            float v[11], mx = -INFINITY, sum = 0.f;
#pragma unroll
            for (int e = 0; e < 11; ++e) {
                const int j = q + 4 * e;
                v[e] = j < NT ? pr[j] : -INFINITY;
                mx = fmaxf(mx, v[e]);
            }
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xF, 0xF, true)));
            mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xF, 0xF, true)));
#pragma unroll
            for (int e = 0; e < 11; ++e) { v[e] = q + 4 * e < NT ? expf(v[e] - mx) : 0.f; sum += v[e]; }
            sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0xB1, 0xF, 0xF, true));
            sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x4E, 0xF, 0xF, true));
            if (live) {
#pragma unroll
                for (int e = 0; e < 11; ++e)
                    if (q + 4 * e < NT) pr[q + 4 * e] = v[e] / sum;
            }
        }
        __syncthreads(); stamp();
        // o = P v, heads concatenated (mixSTE.py:94): 24 tiles (4 heads x 3 row tiles x 2 column tiles), K = 44
        // (probability columns 42,43 and v rows 42..47 are zero)
        {
            const int li = lane & 15, lk = lane >> 4;
            constexpr int TPW = HEADS * 6 / NWAVES;                     // 3 tiles per wave, advanced together (independent chains)
            const float* pa[TPW]; const float* vb[TPW];
            f32x4 acc[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = wave + NWAVES * i, h = tile / 6, mt = (tile - h * 6) >> 1, nt = tile & 1;
                pa[i] = s_p + (h * NTP + mt * 16 + li) * LDP + lk;
                vb[i] = s_big + lk * LDQ + 256 + h * HD + nt * 16 + li;
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < KP / 4; ++kk)
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[i][4 * kk], vb[i][4 * kk * LDQ], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = wave + NWAVES * i, h = tile / 6, mt = (tile - h * 6) >> 1, nt = tile & 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = mt * 16 + lk * 4 + r;
                    if (t < NT) {
                        if constexpr (WBF16) s_nb[t * LDB + h * HD + nt * 16 + li] = f2bf_rne(acc[i][r]);
                        else s_n[t * LDX + h * HD + nt * 16 + li] = acc[i][r];
                    }
                }
            }
        }
        __syncthreads(); stamp();
        auto add_x = [&](int t, int n, float v) { s_x[t * LDX + n] += v; };
        if constexpr (WBF16) {
            load_wfrag<D, 2>(P.fc1_wt, P.fc1_b, 256, wave, lane, w1);
            gemm_bf16_act<D, 1>(s_nb, LDB, wp, D, wave, lane, add_x);
        }
        else linear<D, WBF16>(s_n, LDX, P.proj_wt, P.proj_b, D, wave, lane, add_x);
        __syncthreads(); stamp();
        // ---- MLP branch
        layernorm_tokens<WBF16>(s_x, s_n, P.ln2_w, P.ln2_b, 1e-6f, wave, lane);
        __syncthreads(); stamp();
        if constexpr (WBF16) {
            load_wfrag<256, 1>(P.fc2_wt, P.fc2_b, D, wave, lane, w2);
            gemm_bf16_act<D, 2>(s_nb, LDB, w1, 256, wave, lane, [&](int t, int n, float v) {
                s_hb[t * LDHB + n] = f2bf_rne(0.5f * v * (1.f + erf_as(v * 0.70710678118654752f)));       // GELU, rounded once for fc2
            });
            __syncthreads(); stamp();
            if (blk + 1 < a.nblocks) load_wfrag<D, 3>(a.p.blocks[blk + 1].qkv_wt, a.p.blocks[blk + 1].qkv_b, 384, wave, lane, wq);
            else load_wfrag<D, 1>(a.p.head_wt, a.p.head_b, 64, wave, lane, wh);
            gemm_bf16_act<256, 1>(s_hb, LDHB, w2, D, wave, lane, add_x);
        } else {
            linear<D, WBF16>(s_n, LDX, P.fc1_wt, P.fc1_b, 256, wave, lane, [&](int t, int n, float v) {
                s_big[t * LDH + n] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));          // exact GELU
            });
            __syncthreads(); stamp();
            linear<256, WBF16>(s_big, LDH, P.fc2_wt, P.fc2_b, D, wave, lane, add_x);
        }
        __syncthreads(); stamp();
        // ---- spatial_norm after every block (mixSTE.py:200)
        layernorm_tokens(s_x, s_x, a.p.snorm_w, a.p.snorm_b, 1e-6f, wave, lane);      // in place
        __syncthreads(); stamp();
    }
    // ---- head: LayerNorm(eps 1e-5) + Linear 128 -> 64 (mixSTE.py:187-190)
    layernorm_tokens<WBF16>(s_x, s_n, a.p.head_ln_w, a.p.head_ln_b, 1e-5f, wave, lane);
    __syncthreads(); stamp();
    float* y = a.y + (long long)b * NT * 64;
    if constexpr (WBF16) {
        if (a.nblocks == 0) load_wfrag<D, 1>(a.p.head_wt, a.p.head_b, 64, wave, lane, wh);
        gemm_bf16_act<D, 1>(s_nb, LDB, wh, 64, wave, lane, [&](int t, int n, float v) { y[t * 64 + n] = v; });
    }
    else linear<D, WBF16>(s_n, LDX, a.p.head_wt, a.p.head_b, 64, wave, lane, [&](int t, int n, float v) { y[t * 64 + n] = v; });
    stamp();
}

}  // namespace

extern "C" int dir_ste_forward(const dir_ste_params* p, const float* x, float* x_pos_out, float* y, int B,
                               void* stream) {
    DIR_REQUIRE(p && x && y, "dir_ste_forward: null pointer");
    DIR_REQUIRE(B > 0 && p->num_blocks >= 0 && p->num_blocks <= 3, "dir_ste_forward: bad B / num_blocks");
    DIR_REQUIRE(p->pos_embed && p->snorm_w && p->snorm_b && p->head_ln_w && p->head_ln_b && p->head_wt && p->head_b,
                "dir_ste_forward: null parameter");
    for (int i = 0; i < p->num_blocks; ++i) {
        const dir_ste_block& b = p->blocks[i];
        DIR_REQUIRE(b.ln1_w && b.ln1_b && b.qkv_wt && b.qkv_b && b.proj_wt && b.proj_b && b.ln2_w && b.ln2_b &&
                        b.fc1_wt && b.fc1_b && b.fc2_wt && b.fc2_b, "dir_ste_forward: null block parameter");
    }
    SteArgs a;
    a.p = *p; a.x_in = x; a.x_inout = x_pos_out; a.y = y; a.nblocks = p->num_blocks;
    a.stamps = dir::stamps_begin("ste");
    DIR_REQUIRE(p->weight_dtype == DIR_DT_F32 || p->weight_dtype == DIR_DT_BF16, "dir_ste_forward: weight_dtype must be f32 or bf16");
    if (p->weight_dtype == DIR_DT_BF16) DIR_LAUNCH(ste_kernel<true>, dim3(B), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    else DIR_LAUNCH(ste_kernel<false>, dim3(B), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    dir::stamps_end("ste", a.stamps, (hipStream_t)stream);
    return dir::check_launch("dir_ste_forward");
}
