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


// ------------------------------------------------------------------------------------------------- round 4: the bf16-weights kernel on 12 waves
// ste_kernel<true> (8 waves, 11 barrier-separated phases per block) measured 58.8 us alone / 68 us inside a forward for B = 64 with every phase at
// 1.6 - 4 us although none of them holds more than ~0.3 us of bf16 matrix-core work (DIR_STAMPS=ste): 256 VGPRs with 38 spills (weight fragments
// of four Linears prefetched in registers), three LDS round trips through the [4][48][45] probability buffer, and 552 exact-fp32 MFMAs per sample
// and block for q k^T and P v (1.9 us of matrix-core time per SIMD).  ste12_kernel is the same network laid out for 12 waves (768 threads, three
// per SIMD, 168 VGPRs) and 7 phases per block:
//   * attention without the probability buffer: wave u owns (head u / 3, 16-query tile u % 3) -- twelve units, twelve waves.  It forms
//     S^T = K Q^T (keys x queries), so a lane holds, for ONE query (lane & 15), twelve keys' scores (three key tiles x four accumulator
//     registers): the row softmax is twelve in-lane values and two cross-lane steps (lanes +16, +32 hold the same query).  The probabilities are
//     then already the A operand of P V -- MFMA k-step 0 takes its eight k slots from (key tile 0, r 0..3 | key tile 1, r 0..3), k-step 1 from
//     (key tile 2 | zeros), and V's rows are gathered in that order -- so P never leaves the registers;
//   * q k^T and P v in SPLIT PRECISION on the bf16 matrix cores: a = ah + al (bf16 halves made in registers from the fp32 LDS rows), a b = ah bh +
//     ah bl + al bh with fp32 accumulation: 21 v_mfma_f32_16x16x32_bf16 per unit instead of 48 fp32 ones, ~2^-16 per product -- far inside this
//     mode's bf16 Linears (the fp32-weights kernel keeps exact fp32);
//   * spatial_norm of block i and LayerNorm 1 of block i + 1 (or the head's LayerNorm) in one phase; every LayerNorm's parameters in LDS from the
//     start;
//   * each wave requests its next Linear's weight tiles one phase early into one of three register buffers, AFTER the phase's own loads (vmcnt is
//     in order); phases meet at an LDS-only barrier (wg_sync); the phases are straight-line code (all 48 rows computed, padding rows included: a
//     predicate is a branch, and at a join hipcc's wait-count pass waits for the weight prefetch).
// Measured (tools/bench_tokens.py, B = 64, alone): 58.8 -> 47.0 us (16 waves with 127 spills 75.7; 12 waves 58.4; LayerNorm parameters in LDS 53.5;
// straight-line phases 50.8; split-precision attention 47.0); per block: LN1 2.4, qkv 1.2, attention 3.8 (5.3 in exact fp32), proj 0.8, LN2 2.3,
// fc1 2.0, fc2 1.1 us.  What is left is per-phase latency (LDS round trip + cross-lane reduction + barrier ~0.8 us each), not arithmetic.
constexpr int NTH12 = 768, NW12 = 12;      // 12 waves = three per SIMD (<= 168 VGPRs): one attention unit each, two qkv tiles each

// Weight tiles of the NEXT Linear live in one of two register buffers whose uses never overlap: W8 (8 fragments: qkv's two tiles per wave, or fc2's
// one K = 256 tile) and W4 (4 fragments: proj, fc1 or the head) -- 51 VGPRs; one WFrag per Linear kept all five alive across the block loop (spills).
struct W8 { bf16x8_t v[8]; float bb[2]; };
struct W4 { bf16x8_t v[4]; float bb[2]; };
template <int K, int NTW, typename WB>
__device__ __forceinline__ void load_w12(const float* Wf, const float* __restrict__ bias, int N, int wave, int lane, WB& w) {
    static_assert(NTW * (K / 32) <= (int)(sizeof(w.v) / sizeof(w.v[0])), "weight buffer too small");
    const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(Wf);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = min(wave + NW12 * i, N / 16 - 1) * 16 + li;
#pragma unroll
        for (int kk = 0; kk < K / 32; ++kk) w.v[i * (K / 32) + kk] = *reinterpret_cast<const bf16x8_t*>(W + (long long)n * K + kk * 32 + lk * 8);
        w.bb[i] = bias[n];
    }
}
// out tile(s) wave + 16 i of  act[48][K] (bf16, LDS) x W^T : the A fragments are shared by the wave's tiles; a wave without a tile skips the phase
template <int K, int NTW, typename WB, typename F>
__device__ __forceinline__ void gemm12(const unsigned short* s_a, int lda, const WB& w, int N, int wave, int lane, F store) {
    if (wave >= N / 16) return;
    const int li = lane & 15, lk = lane >> 4;
    f32x4 acc[NTW][3];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments one k-step ahead of their MFMAs, and no further (the scheduler otherwise hoists all K / 32 x 3 reads: 96 VGPRs at K = 256, spills)
    bf16x8_t av[2][3];
#pragma unroll
    for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const bf16x8_t*>(s_a + (m * 16 + li) * lda + lk * 8);
#pragma unroll
    for (int kk = 0; kk < K / 32; ++kk) {
        if (kk + 1 < K / 32) {
#pragma unroll
            for (int m = 0; m < 3; ++m) av[(kk + 1) & 1][m] = *reinterpret_cast<const bf16x8_t*>(s_a + (m * 16 + li) * lda + (kk + 1) * 32 + lk * 8);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int i = 0; i < NTW; ++i)
                if (i == 0 || wave + NW12 * i < N / 16)
                    acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[kk & 1][m], w.v[i * (K / 32) + kk], acc[i][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wave + NW12 * i;
        if (nt < N / 16) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) store(m * 16 + lk * 4 + r, nt * 16 + li, acc[i][m][r] + w.bb[i]);      // rows 42..47: padding rows (the caller's store decides)
        }
    }
}
// token t = 4 wave + (lane >> 4) (one round for 42 tokens on 12 waves), 16 lanes per token.  PRE: first x <- LayerNorm_1e-6(x) * pw + pb written
// back to the residual stream (spatial_norm, mixSTE.py:200); then LayerNorm_eps(x) * w + b -> bf16 operand rows (the next Linear's input).
template <bool PRE, typename PF>
__device__ __forceinline__ void ln12(float* s_x, const float* pw, const float* pb, const float* w, const float* b, float eps, unsigned short* s_out,
                                     int wave, int lane, PF prefetch) {
    const int li = lane & 15, t = wave * 4 + (lane >> 4);
    // this phase's own parameters first, THEN the next Linear's weight tiles: vmcnt is in order, so a wait for the parameters must not sit behind
    // the weight stream (measured with the order reversed: 7 us per LayerNorm phase instead of ~1)
    float gw[8], gb[8], gpw[PRE ? 8 : 1], gpb[PRE ? 8 : 1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gw[e] = w[li + 16 * e]; gb[e] = b[li + 16 * e];
        if constexpr (PRE) { gpw[e] = pw[li + 16 * e]; gpb[e] = pb[li + 16 * e]; }
    }
    prefetch();
    float* row = s_x + t * LDX + li;           // all 48 rows, padding rows included: straight-line code (a predicate here is a branch, and a branch
    //                                            makes hipcc's wait-count pass wait for the weight prefetch at the join)
    float v[8], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = row[16 * e]; sum += v[e]; }
    if constexpr (PRE) {
        const float mean = row16_sum(sum) * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; sq = fmaf(v[e], v[e], sq); }
        const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / D) + 1e-6f);
        sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = v[e] * rstd * gpw[e] + gpb[e];
            row[16 * e] = v[e];
            sum += v[e];
        }
    }
    const float mean = row16_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] -= mean; sq = fmaf(v[e], v[e], sq); }
    const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / D) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_out[t * LDB + li + 16 * e] = f2bf_rne(v[e] * rstd * gw[e] + gb[e]);
}

// Phase boundary of ste12_kernel: the phases hand over through LDS only, so the barrier waits for this wave's LDS traffic (lgkmcnt) and NOT for its
// outstanding global loads -- __syncthreads() drains vmcnt too, which made every phase that requests the next Linear's weight tiles last as long
// as that fetch (measured: 3.2 - 5.7 us for the LayerNorm / attention phases against 0.8 - 2 us for the others).
__device__ __forceinline__ void wg_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(NTH12) void ste12_kernel(SteArgs a) {
    __shared__ __attribute__((aligned(16))) float sm16[NTP * LDX + NTP * LDB / 2 + NTP * LDQ];      // 24,960 + 13,056 + 74,112 B
    float* s_x = sm16;                                                // [48][130] residual stream (fp32)
    unsigned short* s_nb = reinterpret_cast<unsigned short*>(s_x + NTP * LDX);      // [48][136] bf16: LayerNorm / attention output (Linear operand)
    // every LayerNorm's weight / bias (3 blocks x (ln1, ln2) + spatial_norm + the head's) in LDS from the start: read from global at the point of
    // use, each LayerNorm phase began with an exposed L2 round trip (3.2 - 3.8 us per phase measured, ~1 us of it arithmetic)
    __shared__ float s_ln[16][D];
    float* s_big = s_x + NTP * LDX + NTP * LDB / 2;                   // [48][386] q | k | v (fp32); later [48][264] bf16 MLP hidden at its start
    unsigned short* s_hb = reinterpret_cast<unsigned short*>(s_big);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && b == 0 && tid == 0 && nstamp < dir::MAX_STAMPS) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    // padding rows 42..47: zero once (they are MFMA rows / V rows that must stay finite; no store ever targets them)
    for (int i = tid; i < (NTP - NT) * LDX; i += NTH12) s_x[NT * LDX + i] = 0.f;
    for (int i = tid; i < (NTP - NT) * LDB; i += NTH12) s_nb[NT * LDB + i] = 0;
    for (int i = tid; i < (NTP - NT) * LDQ; i += NTH12) s_big[NT * LDQ + i] = 0.f;
    for (int i = tid; i < 16 * D; i += NTH12) {
        const int v = i >> 7, c = i & (D - 1), blk = v >> 2, q = v & 3;
        const float* src = v < 12 ? (blk < a.nblocks ? (q == 0 ? a.p.blocks[blk].ln1_w : q == 1 ? a.p.blocks[blk].ln1_b : q == 2 ? a.p.blocks[blk].ln2_w : a.p.blocks[blk].ln2_b) : nullptr)
                                  : (v == 12 ? a.p.snorm_w : v == 13 ? a.p.snorm_b : v == 14 ? a.p.head_ln_w : a.p.head_ln_b);
        s_ln[v][c] = src ? src[c] : 0.f;
    }
    const float* xin = a.x_in + (long long)b * NT * D;
    for (int i = tid; i < NT * D; i += NTH12) {
        const float v = xin[i] + a.p.pos_embed[i];                      // x += spatial_pos_embed (mixSTE.py:196)
        s_x[(i >> 7) * LDX + (i & 127)] = v;
        if (a.x_inout) a.x_inout[(long long)b * NT * D + i] = v;
    }
    W8 w8;                    // qkv (two of its 24 column tiles per wave)  |  fc2 (8 tiles, K = 256)
    W8 w8b;                   // fc1 (16 tiles: two on waves 0..3)
    W4 w4;                    // proj (8 tiles)  |  head (4 tiles)
    wg_sync(); stamp();

    const int li = lane & 15, lk = lane >> 4;
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const dir_ste_block& P = a.p.blocks[blk];
        // ---- 1. (spatial_norm of the previous block +) LayerNorm 1 -> bf16 operand; qkv's weight tiles are requested first
        auto pf_qkv = [&]() { load_w12<D, 2>(P.qkv_wt, P.qkv_b, 384, wave, lane, w8); };
        if (blk == 0) ln12<false>(s_x, nullptr, nullptr, s_ln[4 * blk], s_ln[4 * blk + 1], 1e-6f, s_nb, wave, lane, pf_qkv);
        else ln12<true>(s_x, s_ln[12], s_ln[13], s_ln[4 * blk], s_ln[4 * blk + 1], 1e-6f, s_nb, wave, lane, pf_qkv);
        wg_sync(); stamp();
        // ---- 2. qkv -> s_big (fp32)
        gemm12<D, 2>(s_nb, LDB, w8, 384, wave, lane, [&](int t, int n, float v) { s_big[t * LDQ + n] = v; });
        wg_sync(); stamp();
        // ---- 3. attention, one (head, 16-query tile) per wave, probabilities in registers; proj's weight tile is requested first
        load_w12<D, 1>(P.proj_wt, P.proj_b, D, wave, lane, w4);
        load_w12<D, 2>(P.fc1_wt, P.fc1_b, 256, wave, lane, w8b);
        {
            // Split precision on the bf16 matrix cores (round 4; the exact-fp32 form cost 552 v_mfma_f32_16x16x4_f32 per sample and block: 1.9 us of
            // matrix-core time per SIMD, 5.3 us per phase): every product of two fp32 numbers a = ah + al, b = bh + bl (bf16 halves) is
            // ah bh + ah bl + al bh with fp32 accumulation -- relative error ~2^-16 per product, far inside this mode's bf16 Linears.
            const int h = wave / 3, qt = wave - 3 * h;
            // eight consecutive fp32 values of an LDS row -> bf16 hi | lo fragments (hi = rne(v), lo = rne(v - hi))
            auto split8 = [&](const float* p, bf16x8_t& hi, bf16x8_t& lo) {
                const float2* p2 = reinterpret_cast<const float2*>(p);         // (rows are 8-byte, not 16-byte aligned: pitch 386 floats)
                const float2 a0 = p2[0], a1 = p2[1], a2 = p2[2], a3 = p2[3];
                const float v[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned short hh = f2bf_rne(v[e]);
                    hi[e] = __builtin_bit_cast(__bf16, hh);
                    lo[e] = __builtin_bit_cast(__bf16, f2bf_rne(v[e] - __uint_as_float((unsigned)hh << 16)));
                }
            };
            bf16x8_t qh, ql;
            split8(s_big + (qt * 16 + li) * LDQ + h * HD + 8 * lk, qh, ql);          // B operand: Q[query 16 qt + li][d = 8 lk .. + 7]
            f32x4 sc[3];
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                bf16x8_t kh, kl;
                split8(s_big + (kt * 16 + li) * LDQ + 128 + h * HD + 8 * lk, kh, kl);  // A operand: K[key 16 kt + li][d = 8 lk .. + 7]
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh, sc[kt], 0, 0, 0);
            }
            // sc[kt][r] = <k[key 16 kt + 4 lk + r], q[query 16 qt + li]>: softmax over the keys of this lane's query
            const float scale = 0.17677669529663687f;                                // 32 ** -0.5
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = kt * 16 + 4 * lk + r < NT;
                    sc[kt][r] = on ? sc[kt][r] * scale : -INFINITY;
                    mx = fmaxf(mx, sc[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[kt][r] = kt * 16 + 4 * lk + r < NT ? expf(sc[kt][r] - mx) : 0.f;
                    sum += sc[kt][r];
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            // o = P v.  The probabilities are already P V's A operand: lane (query li, group lk) holds keys 16 kt + 4 lk + r; MFMA k-step 0 takes its
            // eight k slots from (kt 0, r 0..3 | kt 1, r 0..3), k-step 1 from (kt 2, r 0..3 | four zeros) -- and V's rows are gathered in exactly that slot order.
            const float rs = 1.f / sum;
            bf16x8_t ph[2], pl[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p0 = sc[e >> 2][e & 3] * rs, p1 = e < 4 ? sc[2][e] * rs : 0.f;
                const unsigned short h0 = f2bf_rne(p0), h1 = f2bf_rne(p1);
                ph[0][e] = __builtin_bit_cast(__bf16, h0); ph[1][e] = __builtin_bit_cast(__bf16, h1);
                pl[0][e] = __builtin_bit_cast(__bf16, f2bf_rne(p0 - __uint_as_float((unsigned)h0 << 16)));
                pl[1][e] = __builtin_bit_cast(__bf16, f2bf_rne(p1 - __uint_as_float((unsigned)h1 << 16)));
            }
            f32x4 o[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* vc = s_big + 256 + h * HD + nt * 16 + li + (4 * lk) * LDQ;    // column of V, rows = keys 4 lk + ...
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // B operand: V[key of slot 8 lk + e][column]: k-step 0 keys 4 lk + e (e < 4) | 16 + 4 lk + e - 4; k-step 1 keys 32 + 4 lk + e (e < 4) | nothing
                    bf16x8_t vh, vl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int key0 = ks == 0 ? (e < 4 ? e : 16 + e - 4) : 32 + (e & 3);           // + 4 lk (in vc)
                        const float v = (ks == 1 && e >= 4) ? 0.f : vc[key0 * LDQ];
                        const unsigned short hh = f2bf_rne(v);
                        vh[e] = __builtin_bit_cast(__bf16, hh);
                        vl[e] = __builtin_bit_cast(__bf16, f2bf_rne(v - __uint_as_float((unsigned)hh << 16)));
                    }
                    o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[ks], vh, o[nt], 0, 0, 0);
                    o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ks], vl, o[nt], 0, 0, 0);
                    o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ks], vh, o[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_nb[(qt * 16 + 4 * lk + r) * LDB + h * HD + nt * 16 + li] = f2bf_rne(o[nt][r]);
        }
        wg_sync(); stamp();
        // ---- 4. proj + residual
        auto add_x = [&](int t, int n, float v) { s_x[t * LDX + n] += v; };
        gemm12<D, 1>(s_nb, LDB, w4, D, wave, lane, add_x);
        wg_sync(); stamp();
        // ---- 5. LayerNorm 2; fc1's and fc2's weight tiles are requested first (both buffers are free)
        ln12<false>(s_x, nullptr, nullptr, s_ln[4 * blk + 2], s_ln[4 * blk + 3], 1e-6f, s_nb, wave, lane, [&]() { load_w12<256, 1>(P.fc2_wt, P.fc2_b, D, wave, lane, w8); });
        wg_sync(); stamp();
        // ---- 6. fc1 + GELU -> bf16 hidden (q | k | v are dead)
        gemm12<D, 2>(s_nb, LDB, w8b, 256, wave, lane, [&](int t, int n, float v) {
            s_hb[t * LDHB + n] = f2bf_rne(0.5f * v * (1.f + erf_as(v * 0.70710678118654752f)));
        });
        wg_sync(); stamp();
        // ---- 7. fc2 + residual
        gemm12<256, 1>(s_hb, LDHB, w8, D, wave, lane, add_x);
        wg_sync(); stamp();
    }
    // ---- (spatial_norm of the last block +) the head's LayerNorm (eps 1e-5), then Linear 128 -> 64 (mixSTE.py:187-190)
    auto pf_head = [&]() { load_w12<D, 1>(a.p.head_wt, a.p.head_b, 64, wave, lane, w4); };
    if (a.nblocks > 0) ln12<true>(s_x, s_ln[12], s_ln[13], s_ln[14], s_ln[15], 1e-5f, s_nb, wave, lane, pf_head);
    else ln12<false>(s_x, nullptr, nullptr, s_ln[14], s_ln[15], 1e-5f, s_nb, wave, lane, pf_head);
    wg_sync(); stamp();
    float* y = a.y + (long long)b * NT * 64;
    gemm12<D, 1>(s_nb, LDB, w4, 64, wave, lane, [&](int t, int n, float v) { if (t < NT) y[t * 64 + n] = v; });
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
    static const int v1 = getenv("DIR_STE_V1") ? atoi(getenv("DIR_STE_V1")) : 0;          // A/B aid: 1 = the 8-wave kernel of rounds 1-3 for bf16 weights too
    if (p->weight_dtype == DIR_DT_BF16 && !v1) DIR_LAUNCH(ste12_kernel, dim3(B), dim3(NTH12), 0, (hipStream_t)stream, a);
    else if (p->weight_dtype == DIR_DT_BF16) DIR_LAUNCH(ste_kernel<true>, dim3(B), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    else DIR_LAUNCH(ste_kernel<false>, dim3(B), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    dir::stamps_end("ste", a.stamps, (hipStream_t)stream);
    return dir::check_launch("dir_ste_forward");
}
