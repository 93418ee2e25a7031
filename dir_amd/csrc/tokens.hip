// Joint-token kernels of one DIR refinement stage (fp32 throughout: these feed MANO, whose 1e-4 mm parity
// budget leaves no room for reduced precision; the work is ~0.1 GFLOP/image, launch/HBM bound, not MFMA bound).
//
//   dir_grid_tokens_forward  a4+a12: F.grid_sample at 21 joint uv + Conv1d-BN-ReLU-Conv1d, pos_emb, global_pos_emb
//                            (models/dir.py:94-101,106-107,197-200)
//   dir_pgcn_stack_forward   a5: ResSimplePGCN = 4 x [PGraphConv -> BN1d -> ReLU]
//                            (SemGCN/p_graph_conv.py:39-59, SemGCN/p_gcn.py:20-27,71-73)
//   dir_regress_forward      a7+a12: RegressorOffset Linears + proj_feat_emb (models/dir.py:118-119,339-351)
//
// P-GCN decomposition: the per-node weights W[2][21][128][128] (2.75 MB/layer) dominate the bytes, activations are
// 10x smaller.  One workgroup owns (hand, node j, 64-column slice): it streams its 2 x 128x64 weight slice ONCE
// (coalesced along the output column) and applies it to that node's row of every sample in the batch held in LDS.
// The neighbour mix  out[j] = h0[j] + sum_k softmax(e_1)[j,k] h1[k] + b  (<= 5 neighbours on the hand skeleton),
// BatchNorm and ReLU of layer l are applied in the LDS-staging prologue of layer l+1, so a layer is ONE launch.
#include "dir_common.h"
#include "dir_mfma.h"

namespace {

using dir::f32x4;

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
struct f16s_t { unsigned short u; };      // f16 STORAGE of the feature map (DIR_DT_F16, round 5)
template <> __device__ __forceinline__ float ld<f16s_t>(const f16s_t* p) { return (float)__builtin_bit_cast(_Float16, p->u); }

constexpr int NJ = 21;

// hand skeleton adjacency in CSR form; entries are in the row-major nonzero order that indexes e_1
// (SemGCN/utils.py:66-71 edges, SemGCN/p_graph_conv.py:26-30 mask order)
__constant__ int kNbrOff[22] = {0, 5, 7, 9, 11, 12, 14, 16, 18, 19, 21, 23, 25, 26, 28, 30, 32, 33, 35, 37, 39, 40};
__constant__ int kNbrIdx[40] = {1, 5, 9, 13, 17, 0, 2, 1, 3, 2, 4, 3, 0, 6, 5, 7, 6, 8, 7, 0,
                                10, 9, 11, 10, 12, 11, 0, 14, 13, 15, 14, 16, 15, 0, 18, 17, 19, 18, 20, 19};

// --------------------------------------------------------------------------------------------- grid tokens
struct GridArgs {
    const void* feat; int S; int C; int fcs; int fco;
    const float* uv[2]; const float* xyz[2]; const float* offset;
    dir_token_mlp img2joint[2]; dir_token_mlp pos_emb[2]; dir_token_mlp gpos;
    float* x0; float* g;       // [2][B][21][128]
    int B;
    long long* stamps;         // DIR_STAMPS=grid_tokens (tuning aid, else NULL)
};

constexpr int GT_ROWS = 32;            // 21 tokens padded to two 16-row MFMA tiles
constexpr int GT_THR = 512;             // 8 waves: one 16-column output tile per wave (8 tiles of the 128-wide MLP layers)
static_assert(GT_THR / 64 == 128 / 16, "grid_tokens: one output tile per wave");
constexpr int LDS_S = 256 + 2, LDS_H = 128 + 2;

// second Conv1d of a token MLP on the matrix cores: acc[m] += hid[32 x 128] * w2t[128 x 128]; wave w owns output column tile w
__device__ __forceinline__ void mlp_out_mfma(const float* s_hid, const dir_token_mlp& m, int wave, int lane, f32x4 (&acc)[2]) {
    dir::mfma_tile_f32<128, 2>(s_hid, LDS_H, m.w2t, 128, wave * 16, lane, acc);
}

// first Conv1d (K = 3) + folded BN + ReLU of the positional MLPs: plain VALU, writes hid[21][128]
__device__ __forceinline__ void mlp_in3(const float* s_in, const dir_token_mlp& m, float* s_hid, int tid) {
    const int o = tid & 127;
    const float w0 = m.w1t[o], w1 = m.w1t[128 + o], w2 = m.w1t[256 + o], s1 = m.s1[o], b1 = m.b1[o];
    for (int j = tid >> 7; j < NJ; j += GT_THR / 128) {
        const float h = fmaf(w2, s_in[j * 3 + 2], fmaf(w1, s_in[j * 3 + 1], w0 * s_in[j * 3]));
        s_hid[j * LDS_H + o] = fmaxf(fmaf(h, s1, b1), 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(GT_THR) void grid_tokens_kernel(GridArgs a) {
    __shared__ float s_samp[GT_ROWS * LDS_S];
    __shared__ float s_hid[GT_ROWS * LDS_H];
    __shared__ float s_p[NJ * 3], s_q[NJ * 3];
    __shared__ float s_w[NJ * 4];
    __shared__ int s_i[NJ * 4];
    const int b = blockIdx.x, hand = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.S, C = a.C;
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && b == 0 && hand == 0 && tid == 0) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    if (tid < NJ) {
        // F.grid_sample, bilinear / zeros / align_corners=False: ix = ((u + 1) * W - 1) / 2
#pragma clang fp contract(off)
        const float u = a.uv[hand][((long long)b * NJ + tid) * 2], v = a.uv[hand][((long long)b * NJ + tid) * 2 + 1];
        const float ix = ((u + 1.f) * (float)S - 1.f) / 2.f, iy = ((v + 1.f) * (float)S - 1.f) / 2.f;
        const float x0 = floorf(ix), y0 = floorf(iy), x1 = x0 + 1.f, y1 = y0 + 1.f;
        const float wx[2] = {x1 - ix, ix - x0}, wy[2] = {y1 - iy, iy - y0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {            // order nw, ne, sw, se
            const float xx = (t & 1) ? x1 : x0, yy = (t & 2) ? y1 : y0;
            const bool ok = xx >= 0.f && xx <= (float)(S - 1) && yy >= 0.f && yy <= (float)(S - 1);
            s_w[tid * 4 + t] = ok ? wx[t & 1] * wy[t >> 1] : 0.f;
            s_i[tid * 4 + t] = ok ? ((int)yy * S + (int)xx) : -1;
        }
    }
    if (tid >= 64 && tid < 64 + NJ * 3) {
#pragma clang fp contract(off)
        const int i = tid - 64, c = i % 3;
        const float p = a.xyz[hand][(long long)b * NJ * 3 + i] / 0.15f;          // models/dir.py:97
        const float off = a.offset[(long long)b * 3 + c] / 2.f;
        s_p[i] = p;
        s_q[i] = hand == 0 ? p - off : p + off;                                   // models/dir.py:106-107
    }
    // padding rows of the MFMA row tiles stay zero for the whole kernel
    for (int i = tid; i < (GT_ROWS - NJ) * LDS_S; i += GT_THR) s_samp[NJ * LDS_S + i] = 0.f;
    for (int i = tid; i < (GT_ROWS - NJ) * LDS_H; i += GT_THR) s_hid[NJ * LDS_H + i] = 0.f;
    __syncthreads(); stamp();
    const T* fb = (const T*)a.feat + (long long)b * S * S * a.fcs + a.fco;
    // bilinear gather: out-of-range taps carry weight 0 and a clamped (valid) pixel, so the 4 loads of an element -- and, unrolled,
    // of several elements -- are unconditional and in flight together
    if (C == 256) {                                                      // thread = (channel, joint half): 11 | 10 joints x 4 taps
        const int c = tid & 255, jb = (tid >> 8) ? 11 : 0, je = (tid >> 8) ? NJ : 11;
#pragma unroll
        for (int j0 = 0; j0 < 12; j0 += 6) {
            float tv[6][4];
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const int j = min(jb + j0 + jj, je - 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) tv[jj][t] = ld<T>(fb + (long long)max(s_i[j * 4 + t], 0) * a.fcs + c);
            }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const int j = jb + j0 + jj;
                if (j < je) {
                    float acc = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (s_i[j * 4 + t] >= 0) acc += tv[jj][t] * s_w[j * 4 + t];   // same skip rule / order as before
                    s_samp[j * LDS_S + c] = acc;
                }
            }
        }
    } else {
        for (int i = tid; i < NJ * C; i += GT_THR) {
            const int j = i / C, c = i - j * C;
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pix = s_i[j * 4 + t];
                if (pix >= 0) acc += ld<T>(fb + (long long)pix * a.fcs + c) * s_w[j * 4 + t];
            }
            s_samp[j * LDS_S + c] = acc;
        }
    }
    __syncthreads(); stamp();
    const int li = lane & 15, lk = lane >> 4;
    // ---- img2joint: Conv1d 256 -> 128 (+BN+ReLU) on the matrix cores -> hid
    {
        const dir_token_mlp& m = a.img2joint[hand];
        {
            f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const int n0 = wave * 16, n = n0 + li;
            dir::mfma_tile_f32<256, 2>(s_samp, LDS_S, m.w1t, 128, n0, lane, h);
            const float s1 = m.s1[n], b1 = m.b1[n];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = mt * 16 + lk * 4 + r;
                    if (j < NJ) s_hid[j * LDS_H + n] = fmaxf(fmaf(h[mt][r], s1, b1), 0.f);
                }
        }
    }
    __syncthreads(); stamp();
    f32x4 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mlp_out_mfma(s_hid, a.img2joint[hand], wave, lane, acc);
    __syncthreads(); stamp();
    // ---- pos_emb: x0 = pos + img (models/dir.py:100) accumulates into the same tiles
    mlp_in3(s_p, a.pos_emb[hand], s_hid, tid);
    __syncthreads(); stamp();
    mlp_out_mfma(s_hid, a.pos_emb[hand], wave, lane, acc);
    const int n = wave * 16 + li;
    float* x0 = a.x0 + ((long long)hand * a.B + b) * NJ * 128;
    {
        const float b2 = a.img2joint[hand].b2[n] + a.pos_emb[hand].b2[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = mt * 16 + lk * 4 + r;
                if (j < NJ) x0[j * 128 + n] = acc[mt][r] + b2;
            }
    }
    __syncthreads(); stamp();
    // ---- global_pos_emb
    mlp_in3(s_q, a.gpos, s_hid, tid);
    __syncthreads(); stamp();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mlp_out_mfma(s_hid, a.gpos, wave, lane, acc);
    float* gp = a.g + ((long long)hand * a.B + b) * NJ * 128;
    {
        const float b2 = a.gpos.b2[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = mt * 16 + lk * 4 + r;
                if (j < NJ) gp[j * 128 + n] = acc[mt][r] + b2;
            }
    }
    stamp();
}

// --------------------------------------------------------------------------------------------- P-GCN
struct PgcnHand {
    const float* W; const float* x_in;          // layer weights [2][21][128][128] (fp32 [k][o], or bf16 [o][k] when w_bf16); x_in [B][21][128] (layer 0 only)
    int w_bf16;
    const float* h_prev;                        // [B][21][256] = (h0 | h1) of the previous layer (layers >= 1)
    const float* e1_prev; const float* bias_prev; const float* bns_prev; const float* bnb_prev; int relu_prev;
    float* h_out;                               // [B][21][256]
};

__device__ __forceinline__ void edge_softmax_row(const float* e1, int j, float (&w)[5], int (&idx)[5], int& deg) {
    // row j of softmax(A_1) where A_1 = -9e15 off the skeleton edges (SemGCN/p_graph_conv.py:43-50)
    const int o0 = kNbrOff[j];
    deg = kNbrOff[j + 1] - o0;
    float mx = -INFINITY;
    for (int t = 0; t < deg; ++t) mx = fmaxf(mx, e1[o0 + t]);
    float sum = 0.f;
    for (int t = 0; t < deg; ++t) { w[t] = expf(e1[o0 + t] - mx); sum += w[t]; idx[t] = kNbrIdx[o0 + t]; }
    for (int t = 0; t < deg; ++t) w[t] /= sum;
}

constexpr int PG_BC = 16;   // samples per chunk (one 16-row MFMA tile)
constexpr int PG_LD = 128 + 2;
constexpr int PG_T = 512;   // 8 waves: wave w owns output columns 32 (w & 3) .. +31 of W0 (w < 4) or W1 (w >= 4)
constexpr int PG_MAXSPLIT = 8;

struct PgcnArgs {
    PgcnHand h[2];
    int B, nchunk;       // batch, 16-sample chunks
    int npairs, npp;     // (node, hand) pairs = 21 * hands; workgroups per pair (each takes chunks c, c + npp, ...)
    long long* stamps;   // DIR_STAMPS=pgcn (tuning aid, else NULL)
};

// One workgroup = one (node j, hand) pair and every npp-th 16-sample chunk of the batch.
//   * The node's W0[j] | W1[j] (bf16: 64 KB, fp32: 128 KB) are fetched ONCE per workgroup, straight into MFMA B-operand registers
//     (wave w: 32 of the 256 output columns, every k), and stay there for all of the workgroup's chunks.  The npp workgroups of
//     a pair sit on the same XCD (blockIdx -> XCD is round robin), so their concurrent fetches of the same 64 KB merge in that
//     XCD's L2: each weight byte leaves HBM once per launch (the previous grid -- (node, 64-column slice, hand, chunk) -- pulled
//     it through up to eight L2s: 4x the algorithmic traffic in the PMC counters).
//   * Layers >= 1 finish the previous layer in the prologue: per (sample, 4 channels) one thread gathers the node's own h0 row
//     and its <= 5 neighbours' h1 rows with 16-byte loads (all six independent, from clamped addresses), applies the softmax-ed
//     edge weights (row j of A_1, from LDS), bias, BatchNorm, ReLU and writes the MFMA A tile to LDS.  The next chunk's
//     gathers are issued before the current chunk's MFMAs.
//   * Arithmetic per output element is that of the previous kernel (same fmaf chain, same k order): results are bit-identical.
template <bool WBF16>
__global__ __launch_bounds__(PG_T) void pgcn_node_kernel(PgcnArgs args) {
    __shared__ float s_x[2][PG_BC * PG_LD];
    __shared__ float s_adj[8];                 // row j of softmax(A_1) of the previous layer, padded to 5 (+ pad)
    __shared__ int s_nidx[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pi = slot / args.npp, c0 = slot - pi * args.npp;
    // XCD -> (hand, contiguous range of nodes): the joints are numbered finger by finger, so most of a node's graph neighbours are
    // produced and consumed under the same L2 (round-robin pairs re-fetched every h1 row through ~3 L2s: 1.8x the algorithmic bytes)
    const int nh = args.npairs / NJ, groups = 8 / nh;       // nh = 1: 8 node groups; nh = 2: XCDs 0-3 left hand, 4-7 right hand
    const int hand = xcd / groups, g = xcd - hand * groups;
    const int j = (NJ * g) / groups + pi;
    if (j >= (NJ * (g + 1)) / groups) return;
    const PgcnHand& a = args.h[hand];
    int nstamp = 0;
    auto stamp = [&]() { if (args.stamps && blockIdx.x == 0 && tid == 0 && nstamp < dir::MAX_STAMPS) args.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();

    // ---- this wave's weight fragments -> registers (in flight while the first chunk is gathered and mixed)
    const int half = wave >> 2, ncol0 = (wave & 3) * 32, li = lane & 15, lk = lane >> 4;
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    bf16x8_t bw[WBF16 ? 2 : 1][WBF16 ? 4 : 1];
    float bv[WBF16 ? 1 : 2][WBF16 ? 1 : 32];
    if constexpr (WBF16) {          // W^T as bf16 [2][21][o][k]: one 16-byte load per lane and k-step
        const unsigned short* w = reinterpret_cast<const unsigned short*>(a.W) + (((long long)half * NJ + j) * 128 + ncol0 + li) * 128 + lk * 8;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bw[t][kk] = *reinterpret_cast<const bf16x8_t*>(w + t * 16 * 128 + kk * 32);
    } else {                        // fp32 [2][21][k][o] (the reference layout)
        const float* w = a.W + ((long long)half * NJ + j) * 128 * 128 + ncol0 + li;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) bv[t][kk] = w[(4 * kk + lk) * 128 + t * 16];
    }

    // ---- adjacency row (LDS) and the per-channel epilogue constants of the previous layer
    if (tid == 0) {
        float wgt[5]; int nidx[5]; int deg = 0;
        if (a.h_prev) edge_softmax_row(a.e1_prev, j, wgt, nidx, deg);
#pragma unroll
        for (int t = 0; t < 5; ++t) { s_adj[t] = t < deg ? wgt[t] : 0.f; s_nidx[t] = t < deg ? nidx[t] : j; }   // padded: every gather is unconditional
    }
    const int bb = tid >> 5, cq = (tid & 31) * 4;          // mix role: sample bb of the chunk, channels cq .. cq+3
    float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), ps = make_float4(1.f, 1.f, 1.f, 1.f), pn = pb;
    if (a.h_prev) {
        pb = *reinterpret_cast<const float4*>(a.bias_prev + cq);
        ps = *reinterpret_cast<const float4*>(a.bns_prev + cq);
        pn = *reinterpret_cast<const float4*>(a.bnb_prev + cq);
    }
    __syncthreads();
    float wgt[5]; int nidx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { wgt[t] = s_adj[t]; nidx[t] = s_nidx[t]; }

    float4 self, nbv[5];
    auto gather = [&](int chunk) {
        const int b0 = chunk * PG_BC, nb = min(PG_BC, args.B - b0);
        const long long b = b0 + min(bb, nb - 1);
        if (!a.h_prev) self = *reinterpret_cast<const float4*>(a.x_in + (b * NJ + j) * 128 + cq);
        else {
            const float* hb = a.h_prev + b * NJ * 256;
            self = *reinterpret_cast<const float4*>(hb + j * 256 + cq);
#pragma unroll
            for (int t = 0; t < 5; ++t) nbv[t] = *reinterpret_cast<const float4*>(hb + nidx[t] * 256 + 128 + cq);
        }
    };
    auto mix1 = [&](float sv, float n0, float n1, float n2, float n3, float n4, float b_, float s_, float h_) -> float {
        float acc = 0.f;
        acc = fmaf(wgt[0], n0, acc); acc = fmaf(wgt[1], n1, acc); acc = fmaf(wgt[2], n2, acc);
        acc = fmaf(wgt[3], n3, acc); acc = fmaf(wgt[4], n4, acc);           // padded terms add +0 (same sum as the deg-term chain)
        float v = sv + acc + b_;                                            // output_0 + output_1 + bias
        v = fmaf(v, s_, h_);                                                // BN1d (eval)
        return a.relu_prev ? fmaxf(v, 0.f) : v;
    };
    auto mix_store = [&](int chunk, int buf) {
        const int nb = min(PG_BC, args.B - chunk * PG_BC);
        float4 v = self;
        if (a.h_prev) {
            v.x = mix1(self.x, nbv[0].x, nbv[1].x, nbv[2].x, nbv[3].x, nbv[4].x, pb.x, ps.x, pn.x);
            v.y = mix1(self.y, nbv[0].y, nbv[1].y, nbv[2].y, nbv[3].y, nbv[4].y, pb.y, ps.y, pn.y);
            v.z = mix1(self.z, nbv[0].z, nbv[1].z, nbv[2].z, nbv[3].z, nbv[4].z, pb.z, ps.z, pn.z);
            v.w = mix1(self.w, nbv[0].w, nbv[1].w, nbv[2].w, nbv[3].w, nbv[4].w, pb.w, ps.w, pn.w);
        }
        if (bb >= nb) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = s_x[buf] + bb * PG_LD + cq;                               // row pitch 130 floats: 8-byte aligned
        *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    };

    int buf = 0;
    if (c0 < args.nchunk) gather(c0);
    stamp();
    for (int chunk = c0; chunk < args.nchunk; chunk += args.npp, buf ^= 1) {
        mix_store(chunk, buf);
        __syncthreads();                       // the A tile of this chunk is complete (and the other buffer is free again)
        if (chunk + args.npp < args.nchunk) gather(chunk + args.npp);
        const float* sx = s_x[buf];
        f32x4 acc[2];
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (WBF16) {
            // bf16 throughput mode (torch.autocast semantics for the matmuls of SemGCN/p_graph_conv.py:47-48): the A operand is
            // converted from the fp32 LDS rows; fp32 accumulation
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float2* ap = reinterpret_cast<const float2*>(sx + li * PG_LD + kk * 32 + lk * 8);
                bf16x8_t av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 t = ap[e];
                    av[2 * e] = (__bf16)t.x;
                    av[2 * e + 1] = (__bf16)t.y;
                }
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw[0][kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw[1][kk], acc[1], 0, 0, 0);
            }
        } else {
            const float* ap = sx + li * PG_LD + lk;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float av = ap[4 * kk];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[0][kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[1][kk], acc[1], 0, 0, 0);
            }
        }
        const int b0 = chunk * PG_BC, nb = min(PG_BC, args.B - b0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk * 4 + r;
            if (row < nb) {
                float* hb = a.h_out + ((long long)(b0 + row) * NJ + j) * 256 + half * 128 + ncol0 + li;
                hb[0] = acc[0][r];
                hb[16] = acc[1][r];
            }
        }
    }
    stamp();
}

struct MixHand {
    const float* h; const float* e1; const float* bias; const float* bns; const float* bnb;
    const float* add; float* out; int relu;
};
struct MixArgs { MixHand h[2]; long long out_bstride; int B; };
// finishes the last layer: out[b][j][:] = relu(bn(h0 + A1 h1 + bias)) (+ add).  One thread per (sample, node, 4 channels): six
// independent 16-byte gathers, one 16-byte store; grid (ceil(B * 21 * 32 / 256), hands)
__global__ __launch_bounds__(256) void pgcn_mix_kernel(MixArgs args) {
    const MixHand& a = args.h[blockIdx.y];
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int cq = (g & 31) * 4, bj = g >> 5;
    if (bj >= args.B * NJ) return;
    const int b = bj / NJ, j = bj - b * NJ;
    float wgt[5]; int nidx[5]; int deg;
    edge_softmax_row(a.e1, j, wgt, nidx, deg);
    const float* hb = a.h + (long long)b * NJ * 256;
    float4 nv[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) nv[t] = *reinterpret_cast<const float4*>(hb + (t < deg ? nidx[t] : j) * 256 + 128 + cq);   // unconditional
    const float4 self = *reinterpret_cast<const float4*>(hb + j * 256 + cq);
    const float4 bi = *reinterpret_cast<const float4*>(a.bias + cq), sc = *reinterpret_cast<const float4*>(a.bns + cq),
                 sh = *reinterpret_cast<const float4*>(a.bnb + cq);
    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.add) ad = *reinterpret_cast<const float4*>(a.add + ((long long)b * NJ + j) * 128 + cq);
    auto one = [&](float sv, float n0, float n1, float n2, float n3, float n4, float b_, float s_, float h_, float ad_) -> float {
        float acc = 0.f;
        const float n[5] = {n0, n1, n2, n3, n4};
#pragma unroll
        for (int t = 0; t < 5; ++t) acc = t < deg ? fmaf(wgt[t], n[t], acc) : acc;
        float v = sv + acc + b_;
        v = fmaf(v, s_, h_);
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.add) v += ad_;
        return v;
    };
    float4 o;
    o.x = one(self.x, nv[0].x, nv[1].x, nv[2].x, nv[3].x, nv[4].x, bi.x, sc.x, sh.x, ad.x);
    o.y = one(self.y, nv[0].y, nv[1].y, nv[2].y, nv[3].y, nv[4].y, bi.y, sc.y, sh.y, ad.y);
    o.z = one(self.z, nv[0].z, nv[1].z, nv[2].z, nv[3].z, nv[4].z, bi.z, sc.z, sh.z, ad.z);
    o.w = one(self.w, nv[0].w, nv[1].w, nv[2].w, nv[3].w, nv[4].w, bi.w, sc.w, sh.w, ad.w);
    *reinterpret_cast<float4*>(a.out + (long long)b * args.out_bstride + j * 128 + cq) = o;
}

// --------------------------------------------------------------------------------------------- P-GCN, the whole stack in ONE launch
// the 4-layer stack of both hands cost five launches of ~7 us each (35.6 us at B = 64), every one of them far below
// a launch's worth of work (1.4 MB of weights per hand and layer).  Here the layers are separated by per-node FLAGS instead of launches:
//   * one PERSISTENT workgroup per (hand, node j, batch split s): it runs all layers of its node for its share of the 16-sample chunks;
//     layer l's W0[j] | W1[j] go straight to MFMA B-operand registers and are requested BEFORE the workgroup waits for its neighbours,
//     so the weight stream (the algorithmic bytes of this operator) is never on the critical path after layer 0;
//   * layer l of node j needs layer l-1 of j itself and of its <= 5 skeleton neighbours, nothing else: each workgroup publishes one flag
//     per layer (flags[l][hand][j][s] = launch epoch + 1) after its h rows are out, and polls the <= 6 flags it depends on -- a dataflow
//     over the hand skeleton, not a grid barrier (the thumb tip never waits for the little finger);
//   * h rows and flags cross XCDs (one non-coherent L2 each): they are written / read as relaxed AGENT-scope atomics (sc1 accesses that go
//     to the device-coherent level) ordered by hand -- vmcnt(0), workgroup barrier, then one lane stores the flag (the split-K idiom of
//     conv.hip); weights, parameters, the stack input and its output are ordinary accesses;
//   * flags are never reset: every launch raises each flag it owns by one, and a workgroup reads its own flags' values at entry (the
//     launch epoch) before it publishes anything -- the sync words only have to start at zero once (HIP-graph replays carry no
//     per-launch argument);
//   * forward progress: all 42 x S workgroups of a launch must become resident while the others spin.  Kernels of other kinds always finish
//     and free their CUs; what must never happen is the chip filling up with PARTLY resident launches of this kernel from several streams.
//     Occupancy is pinned (bf16 weights: 4 waves per SIMD = 2 workgroups per CU = 512 on the chip; fp32: 3 waves per SIMD = 1 per CU = 256)
//     and the default S (bf16 2 -> 84 workgroups, fp32 1 -> 42) keeps SIX concurrent launches inside that; a caller that knows it runs
//     fewer streams may pass more splits.  A poll that sees nothing for ~1 s gives up, raises the error word
//     (sync[PGF_ERR_WORD]) and lets the kernel end -- wrong tokens, never a hung queue.
// Arithmetic per output element is that of pgcn_node_kernel + pgcn_mix_kernel (same fmaf chains, same k order): bit-identical results
// (tests/test_gpu_tokens.py::test_pgcn_fused_equals_layered).
constexpr int PGF_MAXL = 4, PGF_MAXS = 8;
constexpr int PGF_ERR_WORD = PGF_MAXL * 2 * NJ * PGF_MAXS;           // sync words: flags [l][hand][node][split] | error word
struct PgcnFusedHand {
    const float* W[PGF_MAXL]; const float* e1[PGF_MAXL]; const float* bias[PGF_MAXL]; const float* bns[PGF_MAXL]; const float* bnb[PGF_MAXL];
    int relu[PGF_MAXL];
    const float* x_in; const float* add; float* out; float* hbuf[2];
};
struct PgcnFusedArgs {
    PgcnFusedHand h[2];
    int B, nchunk, S, L;
    long long out_bstride;
    unsigned* sync;
    long long* stamps;
};

// 16-byte device-coherent accesses (what __hip_atomic_load/store(relaxed, agent) lower to, four words at a time: a scalar sc1 access is one
// fabric transaction each -- measured 36 us for the stack with dword accesses against 31.6 us for the five launches).  The loads are issued
// and waited for inside ONE asm block: the compiler does not track an asm load's completion.
typedef float __attribute__((ext_vector_type(4))) pgf4;
__device__ __forceinline__ void st_dev4(float* p, pgf4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ld_dev4x6(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, const float* p5,
                                          pgf4& v0, pgf4& v1, pgf4& v2, pgf4& v3, pgf4& v4, pgf4& v5) {
    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %7, off sc1\n\tglobal_load_dwordx4 %2, %8, off sc1\n\t"
                 "global_load_dwordx4 %3, %9, off sc1\n\tglobal_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %11, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5)
                 : "memory");
}
__device__ __forceinline__ float4 f4(pgf4 v) { return make_float4(v.x, v.y, v.z, v.w); }

template <bool WBF16>
__device__ __forceinline__ void pgcn_fused_body(const PgcnFusedArgs& args) {
    __shared__ float s_x[2][PG_BC * PG_LD];
    __shared__ float s_adj[8];
    __shared__ int s_nidx[8];
    __shared__ unsigned s_epoch[PGF_MAXL];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = args.S, sp = blockIdx.x % S, pj = blockIdx.x / S, hand = pj / NJ, j = pj - hand * NJ;
    const PgcnFusedHand& a = args.h[hand];
    int nstamp = 0;
    auto stamp = [&]() { if (args.stamps && blockIdx.x == 0 && tid == 0 && nstamp < dir::MAX_STAMPS) args.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    auto flag = [&](int l, int node) -> unsigned* { return args.sync + ((l * 2 + hand) * NJ + node) * PGF_MAXS + sp; };
    if (tid < args.L) s_epoch[tid] = __hip_atomic_load(flag(tid, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // before anything is published

    const int half = wave >> 2, ncol0 = (wave & 3) * 32, li = lane & 15, lk = lane >> 4;
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    bf16x8_t bw[WBF16 ? 2 : 1][WBF16 ? 4 : 1];
    float bv[WBF16 ? 1 : 2][WBF16 ? 1 : 32];
    const int bb = tid >> 5, cq = (tid & 31) * 4;          // mix role: sample bb of the chunk, channels cq .. cq+3

    // waits until node j's and its neighbours' rows of layer `l` are out (lanes 0..5 of wave 0 poll one flag each), then the workgroup meets
    auto wait_layer = [&](int l) {
        if (tid < 6) {
            const int o0 = kNbrOff[j], deg = kNbrOff[j + 1] - o0;
            const int node = tid == 0 ? j : (tid - 1 < deg ? kNbrIdx[o0 + tid - 1] : j);
            const unsigned want = s_epoch[l] + 1u;
            const unsigned* f = flag(l, node);
            int spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) { __hip_atomic_store(args.sync + PGF_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
    };
    // row j of softmax(A_1) of layer `l` (padded to 5 terms: every gather is unconditional) -> LDS
    auto adjacency = [&](int l) {
        if (tid == 0) {
            float wgt[5]; int nidx[5]; int deg = 0;
            edge_softmax_row(a.e1[l], j, wgt, nidx, deg);
#pragma unroll
            for (int t = 0; t < 5; ++t) { s_adj[t] = t < deg ? wgt[t] : 0.f; s_nidx[t] = t < deg ? nidx[t] : j; }
        }
    };

    for (int l = 0; l < args.L; ++l) {
        // ---- this layer's weight fragments -> registers, requested before the wait below
        if constexpr (WBF16) {
            const unsigned short* w = reinterpret_cast<const unsigned short*>(a.W[l]) + (((long long)half * NJ + j) * 128 + ncol0 + li) * 128 + lk * 8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bw[t][kk] = *reinterpret_cast<const bf16x8_t*>(w + t * 16 * 128 + kk * 32);
        } else {
            const float* w = a.W[l] + ((long long)half * NJ + j) * 128 * 128 + ncol0 + li;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) bv[t][kk] = w[(4 * kk + lk) * 128 + t * 16];
        }
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), ps = make_float4(1.f, 1.f, 1.f, 1.f), pn = pb;
        int relu_prev = 0;
        if (l) {
            adjacency(l - 1);
            pb = *reinterpret_cast<const float4*>(a.bias[l - 1] + cq);
            ps = *reinterpret_cast<const float4*>(a.bns[l - 1] + cq);
            pn = *reinterpret_cast<const float4*>(a.bnb[l - 1] + cq);
            relu_prev = a.relu[l - 1];
            wait_layer(l - 1);                    // (its barrier also publishes s_adj / s_nidx; layer 0: s_epoch is ordered by the first chunk's barrier)
        }
        const float* h_prev = l ? a.hbuf[(l - 1) & 1] : nullptr;
        float* h_out = a.hbuf[l & 1];
        float wgt[5]; int nidx[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) { wgt[t] = l ? s_adj[t] : 0.f; nidx[t] = l ? s_nidx[t] : j; }

        float4 self, nbv[5];
        auto gather = [&](int chunk) {
            const int b0 = chunk * PG_BC, nb = min(PG_BC, args.B - b0);
            const long long b = b0 + min(bb, nb - 1);
            if (!h_prev) self = *reinterpret_cast<const float4*>(a.x_in + (b * NJ + j) * 128 + cq);
            else {
                const float* hb = h_prev + b * NJ * 256;
                pgf4 v0, v1, v2, v3, v4, v5;
                ld_dev4x6(hb + j * 256 + cq, hb + nidx[0] * 256 + 128 + cq, hb + nidx[1] * 256 + 128 + cq, hb + nidx[2] * 256 + 128 + cq,
                          hb + nidx[3] * 256 + 128 + cq, hb + nidx[4] * 256 + 128 + cq, v0, v1, v2, v3, v4, v5);
                self = f4(v0); nbv[0] = f4(v1); nbv[1] = f4(v2); nbv[2] = f4(v3); nbv[3] = f4(v4); nbv[4] = f4(v5);
            }
        };
        auto mix1 = [&](float sv, float n0, float n1, float n2, float n3, float n4, float b_, float s_, float h_) -> float {
            float acc = 0.f;
            acc = fmaf(wgt[0], n0, acc); acc = fmaf(wgt[1], n1, acc); acc = fmaf(wgt[2], n2, acc);
            acc = fmaf(wgt[3], n3, acc); acc = fmaf(wgt[4], n4, acc);
            float v = sv + acc + b_;
            v = fmaf(v, s_, h_);
            return relu_prev ? fmaxf(v, 0.f) : v;
        };
        auto mix_store = [&](int chunk, int buf) {
            const int nb = min(PG_BC, args.B - chunk * PG_BC);
            float4 v = self;
            if (h_prev) {
                v.x = mix1(self.x, nbv[0].x, nbv[1].x, nbv[2].x, nbv[3].x, nbv[4].x, pb.x, ps.x, pn.x);
                v.y = mix1(self.y, nbv[0].y, nbv[1].y, nbv[2].y, nbv[3].y, nbv[4].y, pb.y, ps.y, pn.y);
                v.z = mix1(self.z, nbv[0].z, nbv[1].z, nbv[2].z, nbv[3].z, nbv[4].z, pb.z, ps.z, pn.z);
                v.w = mix1(self.w, nbv[0].w, nbv[1].w, nbv[2].w, nbv[3].w, nbv[4].w, pb.w, ps.w, pn.w);
            }
            if (bb >= nb) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float* d = s_x[buf] + bb * PG_LD + cq;
            *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
        };

        int buf = 0;
        if (sp < args.nchunk) gather(sp);
        for (int chunk = sp; chunk < args.nchunk; chunk += S, buf ^= 1) {
            mix_store(chunk, buf);
            __syncthreads();
            const float* sx = s_x[buf];
            f32x4 acc[2];
            acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (WBF16) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float2* ap = reinterpret_cast<const float2*>(sx + li * PG_LD + kk * 32 + lk * 8);
                    bf16x8_t av;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float2 t = ap[e];
                        av[2 * e] = (__bf16)t.x;
                        av[2 * e + 1] = (__bf16)t.y;
                    }
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[0][kk], av, acc[0], 0, 0, 0);       // operands swapped: D = (x W)^T, the same
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[1][kk], av, acc[1], 0, 0, 0);       // products in the same k order
                }
            } else {
                const float* ap = sx + li * PG_LD + lk;
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    const float av = ap[4 * kk];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0][kk], av, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1][kk], av, acc[1], 0, 0, 0);
                }
            }
            // transposed accumulators: lane (sample li, output columns 4 lk .. +3 of each 16-column tile) -> one 16-byte write-through store per tile
            const int b0 = chunk * PG_BC, nb = min(PG_BC, args.B - b0);
            if (li < nb) {
                float* hb = h_out + ((long long)(b0 + li) * NJ + j) * 256 + half * 128 + ncol0 + 4 * lk;
                st_dev4(hb, pgf4{acc[0][0], acc[0][1], acc[0][2], acc[0][3]});
                st_dev4(hb + 16, pgf4{acc[1][0], acc[1][1], acc[1][2], acc[1][3]});
            }
            if (chunk + S < args.nchunk) gather(chunk + S);          // (waited for inside the asm block: after the stores, not before the MFMAs)
        }
        // ---- publish: every wave's rows are out (vmcnt(0)), the workgroup meets, one lane raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag(l, j), s_epoch[l] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp();
    }

    // ---- the last layer's mix (pgcn_mix_kernel's arithmetic): out[b][j][:] = relu(bn(h0 + A1 h1 + bias)) (+ add)
    const int lz = args.L - 1;
    adjacency(lz);
    wait_layer(lz);
    {
        const float4 bi = *reinterpret_cast<const float4*>(a.bias[lz] + cq), sc = *reinterpret_cast<const float4*>(a.bns[lz] + cq),
                     sh = *reinterpret_cast<const float4*>(a.bnb[lz] + cq);
        const int o0 = kNbrOff[j], deg = kNbrOff[j + 1] - o0;
        float wgt[5]; int nidx[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) { wgt[t] = s_adj[t]; nidx[t] = s_nidx[t]; }
        const int relu = a.relu[lz];
        const float* hz = a.hbuf[lz & 1];
        for (int chunk = sp; chunk < args.nchunk; chunk += S) {
            const int b0 = chunk * PG_BC, nb = min(PG_BC, args.B - b0);
            if (bb >= nb) continue;
            const long long b = b0 + bb;
            const float* hb = hz + b * NJ * 256;
            float4 nv[5];
            pgf4 v0, v1, v2, v3, v4, v5;
            ld_dev4x6(hb + j * 256 + cq, hb + nidx[0] * 256 + 128 + cq, hb + nidx[1] * 256 + 128 + cq, hb + nidx[2] * 256 + 128 + cq,
                      hb + nidx[3] * 256 + 128 + cq, hb + nidx[4] * 256 + 128 + cq, v0, v1, v2, v3, v4, v5);
            const float4 self = f4(v0);
            nv[0] = f4(v1); nv[1] = f4(v2); nv[2] = f4(v3); nv[3] = f4(v4); nv[4] = f4(v5);
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.add) ad = *reinterpret_cast<const float4*>(a.add + (b * NJ + j) * 128 + cq);
            auto one = [&](float sv, float n0, float n1, float n2, float n3, float n4, float b_, float s_, float h_, float ad_) -> float {
                float acc = 0.f;
                const float n[5] = {n0, n1, n2, n3, n4};
#pragma unroll
                for (int t = 0; t < 5; ++t) acc = t < deg ? fmaf(wgt[t], n[t], acc) : acc;
                float v = sv + acc + b_;
                v = fmaf(v, s_, h_);
                if (relu) v = fmaxf(v, 0.f);
                if (a.add) v += ad_;
                return v;
            };
            float4 o;
            o.x = one(self.x, nv[0].x, nv[1].x, nv[2].x, nv[3].x, nv[4].x, bi.x, sc.x, sh.x, ad.x);
            o.y = one(self.y, nv[0].y, nv[1].y, nv[2].y, nv[3].y, nv[4].y, bi.y, sc.y, sh.y, ad.y);
            o.z = one(self.z, nv[0].z, nv[1].z, nv[2].z, nv[3].z, nv[4].z, bi.z, sc.z, sh.z, ad.z);
            o.w = one(self.w, nv[0].w, nv[1].w, nv[2].w, nv[3].w, nv[4].w, bi.w, sc.w, sh.w, ad.w);
            *reinterpret_cast<float4*>(a.out + b * args.out_bstride + j * 128 + cq) = o;
        }
    }
    stamp();
}
// occupancy is part of the forward-progress argument above, so it is pinned (second argument: waves per SIMD): bf16 weights 128 VGPRs, fp32 168, no spills
__global__ __launch_bounds__(PG_T, 4) void pgcn_fused_bf16_kernel(PgcnFusedArgs args) { pgcn_fused_body<true>(args); }
__global__ __launch_bounds__(PG_T, 3) void pgcn_fused_f32_kernel(PgcnFusedArgs args) { pgcn_fused_body<false>(args); }

// --------------------------------------------------------------------------------------------- regressor
struct RegArgs {
    dir_regress_params p;
    const float* tok; const float* prev_para[2]; const float* prev_off;
    float* para[2]; float* off; float* emb;
    long long* stamps;         // DIR_STAMPS=regress (tuning aid, else NULL)
};

// one 512-thread workgroup per sample
__global__ __launch_bounds__(512) void regress_kernel(RegArgs a) {
    __shared__ float s_in[2][1408];        // per hand: 21*64 token features | previous mano_para (models/dir.py:344-345)
    __shared__ float s_hid[42 * 64];
    __shared__ float s_part[4][128];
    __shared__ float s_off[3], s_red[8][3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && b == 0 && tid == 0) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    for (int i = tid; i < 42 * 64; i += 512) {
        const int hand = i / 1344;
        s_in[hand][i - hand * 1344] = a.tok[(long long)b * 42 * 64 + i];
    }
    if (tid < 128) s_in[tid >> 6][1344 + (tid & 63)] = a.prev_para[tid >> 6][(long long)b * 64 + (tid & 63)];
    if (tid >= 128 && tid < 131) s_off[tid - 128] = a.prev_off[(long long)b * 3 + tid - 128];
    __syncthreads(); stamp();
    {   // Linear(1408 -> 64) x 2 (models/dir.py:350-351): thread = (output o of 128, K quarter), weights k-major
        const int o = tid & 127, ks = tid >> 7;
        const float* in = s_in[o >> 6] + ks * 352;
        const float* w = a.p.mano_wt + (long long)(ks * 352) * 128 + o;
        float acc = 0.f;
        for (int k0 = 0; k0 < 352; k0 += 88) {                      // 88 loads in flight per thread: 4 L2 round trips (was 22)
            float wv[88];
#pragma unroll
            for (int u = 0; u < 88; ++u) wv[u] = w[(k0 + u) * 128];
#pragma unroll
            for (int u = 0; u < 88; ++u) acc = fmaf(in[k0 + u], wv[u], acc);
        }
        s_part[ks][o] = acc;
    }
    {   // Linear(2691 -> 3) (models/dir.py:347-348) over cat(tokL, tokR, prev_offset)
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
#pragma unroll
        for (int it = 0; it < 6; ++it) {                            // 2688 = 5.25 x 512: unrolled, all weight loads independent
            const int k = tid + 512 * it;
            if (k < 2688) {
                const float x = k < 1344 ? s_in[0][k] : s_in[1][k - 1344];
                p0 = fmaf(x, a.p.off_w[k], p0); p1 = fmaf(x, a.p.off_w[2691 + k], p1); p2 = fmaf(x, a.p.off_w[2 * 2691 + k], p2);
            }
        }
        p0 = dir::wave_sum(p0); p1 = dir::wave_sum(p1); p2 = dir::wave_sum(p2);
        if (lane == 0) { s_red[wave][0] = p0; s_red[wave][1] = p1; s_red[wave][2] = p2; }
    }
    __syncthreads(); stamp();
    if (tid < 128) {
        const int s = tid >> 6, oo = tid & 63;
        a.para[s][(long long)b * 64 + oo] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid] + a.p.mano_b[s][oo];
    } else if (tid < 131) {
        const int oo = tid - 128;
        float acc = 0.f;
        for (int w8 = 0; w8 < 8; ++w8) acc += s_red[w8][oo];
        for (int q = 0; q < 3; ++q) acc = fmaf(s_off[q], a.p.off_w[oo * 2691 + 2688 + q], acc);
        a.off[(long long)b * 3 + oo] = acc + a.p.off_b[oo];
    }
    // proj_feat_emb: Conv1d(64,64,1) -> BN -> ReLU -> Conv1d(64,64,1) on every token (models/dir.py:51-56,118-119)
    const dir_token_mlp& m = a.p.emb;
    const int o = tid & 63, g = tid >> 6;            // 8 token groups: tokens g, g+8, ...
    float h[6];
    float wv[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wv[k] = m.w1t[k * 64 + o];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int j = g + 8 * t;
        float acc = 0.f;
        if (j < 42) {
            const float* x = (j < 21) ? s_in[0] + j * 64 : s_in[1] + (j - 21) * 64;
#pragma unroll
            for (int k = 0; k < 64; ++k) acc = fmaf(wv[k], x[k], acc);
            s_hid[j * 64 + o] = fmaxf(fmaf(acc, m.s1[o], m.b1[o]), 0.f);
        }
    }
    __syncthreads(); stamp();
#pragma unroll
    for (int k = 0; k < 64; ++k) wv[k] = m.w2t[k * 64 + o];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int j = g + 8 * t;
        if (j < 42) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 64; ++k) acc = fmaf(wv[k], s_hid[j * 64 + k], acc);
            h[t] = acc;
            a.emb[((long long)b * 42 + j) * 64 + o] = h[t] + m.b2[o];
        }
    }
    stamp();
}

bool mlp_ok(const dir_token_mlp& m) { return m.w1t && m.s1 && m.b1 && m.w2t && m.b2; }

}  // namespace

extern "C" int dir_grid_tokens_forward(const void* feat, int feat_dtype, int S, int C, int feat_cstride, int feat_coff,
                                       const float* uv_left,
                                       const float* uv_right, const float* xyz_left, const float* xyz_right,
                                       const float* offset, const dir_token_mlp* img2joint_lr,
                                       const dir_token_mlp* pos_emb_lr, const dir_token_mlp* global_pos_emb,
                                       float* x0, float* gpos, int B, void* stream) {
    DIR_REQUIRE(feat && uv_left && uv_right && xyz_left && xyz_right && offset && img2joint_lr && pos_emb_lr &&
                    global_pos_emb && x0 && gpos, "dir_grid_tokens_forward: null pointer");
    DIR_REQUIRE(B > 0 && S > 0 && C == 256, "dir_grid_tokens_forward: need B>0, S>0, C==256");
    DIR_REQUIRE(mlp_ok(img2joint_lr[0]) && mlp_ok(img2joint_lr[1]) && mlp_ok(pos_emb_lr[0]) && mlp_ok(pos_emb_lr[1]) &&
                    mlp_ok(*global_pos_emb), "dir_grid_tokens_forward: incomplete MLP parameters");
    GridArgs a;
    a.feat = feat; a.S = S; a.C = C; a.fcs = feat_cstride ? feat_cstride : C; a.fco = feat_coff; a.uv[0] = uv_left; a.uv[1] = uv_right; a.xyz[0] = xyz_left; a.xyz[1] = xyz_right;
    a.offset = offset; a.img2joint[0] = img2joint_lr[0]; a.img2joint[1] = img2joint_lr[1];
    a.pos_emb[0] = pos_emb_lr[0]; a.pos_emb[1] = pos_emb_lr[1]; a.gpos = *global_pos_emb; a.x0 = x0; a.g = gpos; a.B = B;
    hipStream_t s = (hipStream_t)stream;
    a.stamps = dir::stamps_begin("grid_tokens");
    if (feat_dtype == DIR_DT_F32) DIR_LAUNCH((grid_tokens_kernel<float>), dim3(B, 2), dim3(GT_THR), 0, s, a);
    else if (feat_dtype == DIR_DT_BF16) DIR_LAUNCH((grid_tokens_kernel<bf16_t>), dim3(B, 2), dim3(GT_THR), 0, s, a);
    else if (feat_dtype == DIR_DT_F16) DIR_LAUNCH((grid_tokens_kernel<f16s_t>), dim3(B, 2), dim3(GT_THR), 0, s, a);
    else DIR_REQUIRE(false, "dir_grid_tokens_forward: bad dtype");
    dir::stamps_end("grid_tokens", a.stamps, s);
    return dir::check_launch("dir_grid_tokens_forward");
}

static int pgcn_run(const dir_pgcn_layer* const* layers, int nh, int num_layers, const float* const* x,
                    const float* const* add, float* const* out, long long out_bstride, float* const* scratch, int B,
                    hipStream_t s) {
    DIR_REQUIRE(num_layers >= 1 && B > 0 && out_bstride >= NJ * 128, "dir_pgcn_stack_forward: bad arguments");
    const int nchunk = (B + PG_BC - 1) / PG_BC;
    bool wbf16 = false;
    for (int l = 0; l < num_layers; ++l) {
        PgcnArgs a;
        a.B = B; a.nchunk = nchunk;
        a.npairs = NJ * nh; a.npp = nchunk < PG_MAXSPLIT ? nchunk : PG_MAXSPLIT;
        for (int h = 0; h < 2; ++h) {
            const int hh = h < nh ? h : 0;
            const dir_pgcn_layer& L = layers[hh][l];
            DIR_REQUIRE(L.W && L.e1 && L.bias && L.bn_scale && L.bn_shift, "dir_pgcn_stack_forward: null layer parameter");
            DIR_REQUIRE(L.w_dtype == DIR_DT_F32 || L.w_dtype == DIR_DT_BF16, "dir_pgcn_stack_forward: w_dtype must be f32 or bf16");
            float* hbuf[2] = {scratch[hh], scratch[hh] + (long long)B * NJ * 256};
            PgcnHand& g = a.h[h];
            g.W = L.W; g.w_bf16 = L.w_dtype == DIR_DT_BF16; g.x_in = x[hh]; g.h_prev = l ? hbuf[(l - 1) & 1] : nullptr;
            g.e1_prev = l ? layers[hh][l - 1].e1 : nullptr; g.bias_prev = l ? layers[hh][l - 1].bias : nullptr;
            g.bns_prev = l ? layers[hh][l - 1].bn_scale : nullptr; g.bnb_prev = l ? layers[hh][l - 1].bn_shift : nullptr;
            g.relu_prev = l ? layers[hh][l - 1].relu : 0;
            g.h_out = hbuf[l & 1];
            if (h == 0) wbf16 = g.w_bf16 != 0;
            DIR_REQUIRE((g.w_bf16 != 0) == wbf16, "dir_pgcn_stack_forward: both hands must use the same weight dtype");
        }
        a.stamps = dir::stamps_begin("pgcn");
        const int per_xcd = (NJ + 8 / nh - 1) / (8 / nh);         // nodes per XCD group (kernel: XCD -> hand, node range)
        const dim3 grid(8 * per_xcd * a.npp);
        if (wbf16) DIR_LAUNCH(pgcn_node_kernel<true>, grid, dim3(PG_T), 0, s, a);
        else DIR_LAUNCH(pgcn_node_kernel<false>, grid, dim3(PG_T), 0, s, a);
        dir::stamps_end("pgcn", a.stamps, s);
    }
    MixArgs m;
    m.out_bstride = out_bstride; m.B = B;
    for (int h = 0; h < 2; ++h) {
        const int hh = h < nh ? h : 0;
        const dir_pgcn_layer& L = layers[hh][num_layers - 1];
        float* hb = scratch[hh] + (((num_layers - 1) & 1) ? (long long)B * NJ * 256 : 0);
        m.h[h] = MixHand{hb, L.e1, L.bias, L.bn_scale, L.bn_shift, add ? add[hh] : nullptr, out[hh], L.relu};
    }
    DIR_LAUNCH(pgcn_mix_kernel, dim3((B * NJ * 32 + 255) / 256, nh), dim3(256), 0, s, m);
    return dir::check_launch("dir_pgcn_stack_forward");
}

extern "C" int dir_pgcn_stack_forward(const dir_pgcn_layer* layers, int num_layers, const float* x, const float* add,
                                      float* out, long long out_bstride, float* scratch, int B, void* stream) {
    DIR_REQUIRE(layers && x && out && scratch, "dir_pgcn_stack_forward: null pointer");
    const dir_pgcn_layer* lp[1] = {layers};
    const float* xp[1] = {x};
    const float* ap[1] = {add};
    float* op[1] = {out};
    float* sp[1] = {scratch};
    return pgcn_run(lp, 1, num_layers, xp, add ? ap : nullptr, op, out_bstride, sp, B, (hipStream_t)stream);
}

extern "C" int dir_pgcn_stack_forward_pair(const dir_pgcn_layer* layers_left, const dir_pgcn_layer* layers_right,
                                           int num_layers, const float* x_lr, const float* add_lr, float* tokens,
                                           float* scratch, int B, void* stream) {
    DIR_REQUIRE(layers_left && layers_right && x_lr && tokens && scratch, "dir_pgcn_stack_forward_pair: null pointer");
    DIR_REQUIRE(B > 0, "dir_pgcn_stack_forward_pair: bad B");
    const long long hs = (long long)B * NJ * 128;
    const dir_pgcn_layer* lp[2] = {layers_left, layers_right};
    const float* xp[2] = {x_lr, x_lr + hs};
    const float* ap[2] = {add_lr, add_lr ? add_lr + hs : nullptr};
    float* op[2] = {tokens, tokens + NJ * 128};
    float* sp[2] = {scratch, scratch + 2 * (long long)B * NJ * 256};
    return pgcn_run(lp, 2, num_layers, xp, add_lr ? ap : nullptr, op, 42 * 128, sp, B, (hipStream_t)stream);
}

extern "C" long long dir_pgcn_fused_sync_bytes(void) { return (long long)(PGF_ERR_WORD + 4) * sizeof(unsigned); }

extern "C" int dir_pgcn_stack_forward_fused(const dir_pgcn_layer* layers_left, const dir_pgcn_layer* layers_right, int num_layers,
                                            const float* x_lr, const float* add_lr, float* tokens, float* scratch, void* sync_ws,
                                            int splits, int B, void* stream) {
    DIR_REQUIRE(layers_left && layers_right && x_lr && tokens && scratch && sync_ws, "dir_pgcn_stack_forward_fused: null pointer");
    DIR_REQUIRE(B > 0 && num_layers >= 1 && num_layers <= PGF_MAXL && splits >= 0 && splits <= PGF_MAXS, "dir_pgcn_stack_forward_fused: bad arguments");
    const long long hs = (long long)B * NJ * 128;
    const dir_pgcn_layer* lp[2] = {layers_left, layers_right};
    PgcnFusedArgs a;
    a.B = B; a.nchunk = (B + PG_BC - 1) / PG_BC; a.L = num_layers; a.out_bstride = 42 * 128; a.sync = (unsigned*)sync_ws;
    bool wbf16 = false;
    for (int h = 0; h < 2; ++h) {
        PgcnFusedHand& g = a.h[h];
        for (int l = 0; l < PGF_MAXL; ++l) {
            const dir_pgcn_layer& L = lp[h][l < num_layers ? l : 0];
            DIR_REQUIRE(L.W && L.e1 && L.bias && L.bn_scale && L.bn_shift, "dir_pgcn_stack_forward_fused: null layer parameter");
            DIR_REQUIRE(L.w_dtype == DIR_DT_F32 || L.w_dtype == DIR_DT_BF16, "dir_pgcn_stack_forward_fused: w_dtype must be f32 or bf16");
            if (h == 0 && l == 0) wbf16 = L.w_dtype == DIR_DT_BF16;
            DIR_REQUIRE((L.w_dtype == DIR_DT_BF16) == wbf16, "dir_pgcn_stack_forward_fused: every layer of both hands must use the same weight dtype");
            g.W[l] = L.W; g.e1[l] = L.e1; g.bias[l] = L.bias; g.bns[l] = L.bn_scale; g.bnb[l] = L.bn_shift; g.relu[l] = L.relu;
        }
        g.x_in = x_lr + h * hs; g.add = add_lr ? add_lr + h * hs : nullptr; g.out = tokens + h * NJ * 128;
        g.hbuf[0] = scratch + (long long)h * 2 * B * NJ * 256; g.hbuf[1] = g.hbuf[0] + (long long)B * NJ * 256;
    }
    // batch splits per node: bounded so that six such launches on six streams stay co-resident (see the kernel's header comment)
    int S = splits ? splits : (wbf16 ? 2 : 1);
    if (const char* e = getenv("DIR_PGCN_SPLITS")) { const int v = atoi(e); if (v >= 1 && v <= PGF_MAXS) S = v; }
    if (S > a.nchunk) S = a.nchunk;
    a.S = S;
    hipStream_t s = (hipStream_t)stream;
    a.stamps = dir::stamps_begin("pgcn_fused");
    if (wbf16) DIR_LAUNCH(pgcn_fused_bf16_kernel, dim3(2 * NJ * S), dim3(PG_T), 0, s, a);
    else DIR_LAUNCH(pgcn_fused_f32_kernel, dim3(2 * NJ * S), dim3(PG_T), 0, s, a);
    dir::stamps_end("pgcn_fused", a.stamps, s);
    return dir::check_launch("dir_pgcn_stack_forward_fused");
}

extern "C" int dir_regress_forward(const dir_regress_params* p, const float* tok, const float* prev_para_left,
                                   const float* prev_para_right, const float* prev_offset, float* para_left,
                                   float* para_right, float* offset, float* emb, int B, void* stream) {
    DIR_REQUIRE(p && tok && prev_para_left && prev_para_right && prev_offset && para_left && para_right && offset && emb,
                "dir_regress_forward: null pointer");
    DIR_REQUIRE(B > 0 && p->mano_wt && p->mano_b[0] && p->mano_b[1] && p->off_w && p->off_b && mlp_ok(p->emb),
                "dir_regress_forward: bad arguments");
    RegArgs a;
    a.p = *p; a.tok = tok; a.prev_para[0] = prev_para_left; a.prev_para[1] = prev_para_right; a.prev_off = prev_offset;
    a.para[0] = para_left; a.para[1] = para_right; a.off = offset; a.emb = emb;
    a.stamps = dir::stamps_begin("regress");
    DIR_LAUNCH(regress_kernel, dim3(B), dim3(512), 0, (hipStream_t)stream, a);
    dir::stamps_end("regress", a.stamps, (hipStream_t)stream);
    return dir::check_launch("dir_regress_forward");
}
