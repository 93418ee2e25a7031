// SURVEY 8a row a13 / 8f rank 2 (forward half): the training objective of the reference, models/dir.py:542-594, as device kernels.
//   * stage_loss_kernel: the 13 terms of one refinement stage (models/dir.py:571-592) -- SmoothL1 with the 0.01 knee
//     (models/loss.py:68-85) on joint / mesh uv and on the /0.15-normalised joint / mesh xyz, EdgeLengthLoss (:41-60) and
//     NormalVectorLoss (:11-33) over the 1538 MANO triangles, SmoothL1 on the inter-hand offset.  One workgroup per (sample,
//     hand): both normalised meshes staged once in LDS (18.7 KB), triangles gathered from LDS, per-sample partial sums written to a
//     scratch table and reduced over the batch in a fixed order by a second tiny launch (deterministic; no float atomics).
//   * dense_pixel_kernel / lovasz_kernel / dense_final_kernel: models/dir.py:562-569 -- F.interpolate (nearest labels, bilinear
//     dense map), class-weighted cross entropy, SmoothL1 on the dense map, and lovasz_softmax (models/lovasz_loss.py:155-202) on the
//     raw logits exactly as the reference calls it.  The descending sort of the per-class errors (models/lovasz_loss.py:107-111: torch.sort) is a
//     hand-written stable LSD radix sort, one segment per class (seg_radix_sort_desc below; rounds 1-5 called rocPRIM here -- the last vendor-library
//     primitive of libdir_hip.so); the Jaccard scan runs one workgroup per class with a carried prefix.
// Elementwise arithmetic is fp32 in the reference's operation order (contraction off); sums accumulate in fp64.
// HBM-bound and tiny: 2 x (778 x 5 + ...) floats per sample and stage.
#include <cstring>

#include "dir_common.h"

namespace dir {
namespace {

constexpr int NV = 778, NJ = 21, LT = 256, NTERM = 13;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum of v over the workgroup (every thread gets it); s_red: NW doubles
template <int NW> __device__ __forceinline__ double block_sum_d(double v, double* s_red, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[w];
    return t;
}

// models/loss.py:74-81 for one element
__device__ __forceinline__ float smooth_l1_term(float x, float y) {
#pragma clang fp contract(off)
    const float z = x - y, az = fabsf(z);
    return az < 0.01f ? 0.5f * (z * z) : 0.01f * (az - 0.005f);
}

// utils/utils.py:47-63 (projection_batch_xy): scale * xyz[..., :2] + trans2d
__device__ __forceinline__ float project_xy(float scale, float trans, float x) {
#pragma clang fp contract(off)
    return scale * x + trans;
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) {
#pragma clang fp contract(off)
    return V3{a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ float sumsq(V3 a) {
#pragma clang fp contract(off)
    return a.x * a.x + a.y * a.y + a.z * a.z;
}
__device__ __forceinline__ V3 normalize(V3 a) {           // F.normalize(p=2, eps=1e-12)
    const float n = fmaxf(sqrtf(sumsq(a)), 1e-12f);
    return V3{a.x / n, a.y / n, a.z / n};
}
__device__ __forceinline__ float dot(V3 a, V3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
#pragma clang fp contract(off)
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct StageArgs {
    dir_loss_pred p;
    dir_loss_target g;
    double* scratch;      // [B][NTERM]
    int B;
};

// slots: 0,1 joint uv L/R; 2,3 mesh uv; 4,5 joint xyz; 6,7 mesh xyz; 8,9 edge; 10,11 normal; 12 offset
__global__ __launch_bounds__(LT) void stage_loss_kernel(StageArgs a) {
    __shared__ float s_pm[NV * 3], s_gm[NV * 3];
    __shared__ double s_red[LT / 64];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const float* cen = a.g.center[h] + (size_t)b * 3;
    const float c0 = cen[0], c1 = cen[1], c2 = cen[2];
    double* out = a.scratch + (size_t)b * NTERM;
    // ---- mesh xyz: pred / 0.15 vs (gt - center) / 0.15 (models/dir.py:556-560,576-577), kept in LDS for the triangle terms
    double acc = 0;
    {
        const float* pm = a.p.mesh_xyz[h] + (size_t)b * NV * 3;
        const float* gm = a.g.mesh_3d[h] + (size_t)b * NV * 3;
        for (int i = tid; i < NV * 3; i += LT) {
            const int c = i % 3;
            const float x = pm[i] / 0.15f, y = (gm[i] - (c == 0 ? c0 : c == 1 ? c1 : c2)) / 0.15f;
            s_pm[i] = x; s_gm[i] = y;
            acc += (double)smooth_l1_term(x, y);
        }
    }
    double t = block_sum_d<LT / 64>(acc, s_red, tid);        // (also the barrier that publishes s_pm / s_gm)
    if (tid == 0) out[6 + h] = t / (NV * 3);
    // ---- joint xyz
    acc = 0;
    if (tid < NJ * 3) {
        const int c = tid % 3;
        const float x = a.p.joint_xyz[h][(size_t)b * NJ * 3 + tid] / 0.15f;
        const float y = (a.g.joint_3d[h][(size_t)b * NJ * 3 + tid] - (c == 0 ? c0 : c == 1 ? c1 : c2)) / 0.15f;
        acc = (double)smooth_l1_term(x, y);
    }
    t = block_sum_d<LT / 64>(acc, s_red, tid);
    if (tid == 0) out[4 + h] = t / (NJ * 3);
    // ---- joint uv / mesh uv against the first two columns of the [.., c2] targets
    acc = 0;
    if (tid < NJ * 2) {
        const int j = tid >> 1, c = tid & 1;
        acc = (double)smooth_l1_term(a.p.joint_uv[h][(size_t)b * NJ * 2 + tid], a.g.joint_2d[h][((size_t)b * NJ + j) * a.g.c2 + c]);
    }
    t = block_sum_d<LT / 64>(acc, s_red, tid);
    if (tid == 0) out[0 + h] = t / (NJ * 2);
    acc = 0;
    for (int i = tid; i < NV * 2; i += LT) {
        const int v = i >> 1, c = i & 1;
        float x;
        if (a.p.mesh_uv[h]) x = a.p.mesh_uv[h][(size_t)b * NV * 2 + i];
        else x = project_xy(a.p.proj[h][(size_t)b * 3], a.p.proj[h][(size_t)b * 3 + 1 + c], a.p.mesh_xyz[h][((size_t)b * NV + v) * 3 + c]);
        acc += (double)smooth_l1_term(x, a.g.mesh_2d[h][((size_t)b * NV + v) * a.g.c2 + c]);
    }
    t = block_sum_d<LT / 64>(acc, s_red, tid);
    if (tid == 0) out[2 + h] = t / (NV * 2);
    // ---- triangles: edge lengths (models/loss.py:46-60) and normals (:16-33)
    double e_acc = 0, n_acc = 0;
    const int32_t* face = a.g.faces[h];
    for (int f = tid; f < a.g.n_faces; f += LT) {
        const int i0 = face[f * 3], i1 = face[f * 3 + 1], i2 = face[f * 3 + 2];
        const V3 p0 = ld3(s_pm + i0 * 3), p1 = ld3(s_pm + i1 * 3), p2 = ld3(s_pm + i2 * 3);
        const V3 g0 = ld3(s_gm + i0 * 3), g1 = ld3(s_gm + i1 * 3), g2 = ld3(s_gm + i2 * 3);
        {
            const float d1o = sqrtf(sumsq(sub(p0, p1)) + 1e-12f), d2o = sqrtf(sumsq(sub(p0, p2)) + 1e-12f), d3o = sqrtf(sumsq(sub(p1, p2)) + 1e-12f);
            const float d1g = sqrtf(sumsq(sub(g0, g1)) + 1e-12f), d2g = sqrtf(sumsq(sub(g0, g2)) + 1e-12f), d3g = sqrtf(sumsq(sub(g1, g2)) + 1e-12f);
            e_acc += (double)fabsf(d1o - d1g) + (double)fabsf(d2o - d2g) + (double)fabsf(d3o - d3g);
        }
        {
            const V3 v1o = normalize(sub(p1, p0)), v2o = normalize(sub(p2, p0)), v3o = normalize(sub(p2, p1));
            const V3 v1g = normalize(sub(g1, g0)), v2g = normalize(sub(g2, g0));
            const V3 ng = normalize(cross(v1g, v2g));
            n_acc += (double)fabsf(dot(v1o, ng)) + (double)fabsf(dot(v2o, ng)) + (double)fabsf(dot(v3o, ng));
        }
    }
    t = block_sum_d<LT / 64>(e_acc, s_red, tid);
    if (tid == 0) out[8 + h] = t / (3.0 * a.g.n_faces);
    t = block_sum_d<LT / 64>(n_acc, s_red, tid);
    if (tid == 0) out[10 + h] = t / (3.0 * a.g.n_faces);
    // ---- offset (models/dir.py:561,591-592): (center_right - center_left) / 0.15
    if (h == 0) {
        acc = 0;
        if (tid < 3) {
            const float y = (a.g.center[1][(size_t)b * 3 + tid] - a.g.center[0][(size_t)b * 3 + tid]) / 0.15f;
            acc = (double)smooth_l1_term(a.p.offset[(size_t)b * 3 + tid], y);
        }
        acc = wave_sum_d(acc);
        if (tid == 0) out[12] = acc / 3.0;
    }
}

// batch means: wave k owns term k (lanes stride over the samples, fixed-shape reduction tree): term k = weight_k * sum_b scratch[b][k] / B
__global__ __launch_bounds__(64 * NTERM) void stage_reduce_kernel(const double* __restrict__ scratch, float* __restrict__ out, int B, float coord_weight) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0;
    for (int b = lane; b < B; b += 64) s += scratch[(size_t)b * NTERM + k];
    s = wave_sum_d(s);
    const double w = (k < 8 || k == 12) ? (double)coord_weight : (k < 10 ? 1.0 : 0.1);
    if (lane == 0) out[k] = (float)(s / B * w);
}

// ------------------------------------------------------------------------------------------------ seg / dense / lovasz
struct DenseArgs {
    const float* seg; const float* dense; const float* gt_seg; const float* gt_dense;
    unsigned long long* keys; unsigned char* vals;       // [3][P]: (2 - class) << 32 | bits of the (non-negative) error -- ONE device-wide descending sort orders class 0 | 1 | 2, each by error
    double* partial;                        // [nwg][6]: sum w nll, sum w, dense smooth-l1 sum, fg count of class 0..2
    int B, S, H, W, P, chunks;
    float cw[3];
};

__global__ __launch_bounds__(LT) void dense_pixel_kernel(DenseArgs a) {
#pragma clang fp contract(off)
    __shared__ double s_red[LT / 64];
    const int b = blockIdx.y, tid = threadIdx.x, S = a.S, pix = blockIdx.x * LT + tid;
    const bool ok = pix < S * S;
    const int oy = ok ? pix / S : 0, ox = ok ? pix - oy * S : 0;
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (ok) {
        // F.interpolate(nearest): source index floor(dst * in / out) (models/dir.py:565)
        const int sy = min((int)floorf(oy * ((float)a.H / S)), a.H - 1), sx = min((int)floorf(ox * ((float)a.W / S)), a.W - 1);
        int lab = (int)a.gt_seg[((size_t)b * a.H + sy) * a.W + sx];
        lab = lab < 0 ? 0 : lab > 2 ? 2 : lab;
        const size_t plane = (size_t)S * S;
        const float* sg = a.seg + (size_t)b * 3 * plane + pix;
        const float l0 = sg[0], l1 = sg[plane], l2 = sg[2 * plane];
        // weighted cross entropy (models/dir.py:511,567)
        const double m = fmax((double)l0, fmax((double)l1, (double)l2));
        const double lse = m + log(exp(l0 - m) + exp(l1 - m) + exp(l2 - m));
        const double nll = lse - (double)(lab == 0 ? l0 : lab == 1 ? l1 : l2);
        const double w = a.cw[lab];
        v[0] = w * nll; v[1] = w;
        // lovasz errors |fg - logit| per class (models/lovasz_loss.py:184-193; the reference passes raw logits)
        const size_t p = (size_t)b * plane + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float fg = lab == c ? 1.f : 0.f, lc = c == 0 ? l0 : c == 1 ? l1 : l2;
            a.keys[(size_t)c * a.P + p] = ((unsigned long long)(2 - c) << 32) | (unsigned long long)__float_as_uint(fabsf(fg - lc));
            a.vals[(size_t)c * a.P + p] = lab == c;
            v[3 + c] = lab == c;
        }
        // F.interpolate(bilinear, align_corners=False) of the dense target (models/dir.py:566) + SmoothL1
        const float fy = fmaxf((oy + 0.5f) * ((float)a.H / S) - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * ((float)a.W / S) - 0.5f, 0.f);
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx), y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
        const float wy1 = fy - y0, wy0 = 1.f - wy1, wx1 = fx - x0, wx0 = 1.f - wx1;
        for (int c = 0; c < 3; ++c) {
            const float* gd = a.gt_dense + ((size_t)b * 3 + c) * a.H * a.W;
            const float r0 = wx0 * gd[(size_t)y0 * a.W + x0] + wx1 * gd[(size_t)y0 * a.W + x1];
            const float r1 = wx0 * gd[(size_t)y1 * a.W + x0] + wx1 * gd[(size_t)y1 * a.W + x1];
            const float t = wy0 * r0 + wy1 * r1;
            v[2] += (double)smooth_l1_term(a.dense[((size_t)b * 3 + c) * plane + pix], t);
        }
    }
    double* out = a.partial + ((size_t)b * a.chunks + blockIdx.x) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double t = block_sum_d<LT / 64>(v[k], s_red, tid);
        if (tid == 0) out[k] = t;
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Stable descending radix sort of `nseg` equal segments of P (key, value) pairs, keys ordered by their LOW 32 bits (the error's float bits: errors
// are >= 0, so unsigned order = float order; the class sits in the bits above and is constant inside a segment).  Four LSD passes of 8 bits, two
// launches each, ping-pong between (k0, v0) and (k1, v1): the sorted pairs end up where they started, in (k0, v0).
//   radix_hist_kernel     workgroup = (2048-element tile, segment): the tile's 256-bin digit histogram -> hist[(seg * 256 + digit) * nblk + tile]
//   radix_scatter_kernel  the same workgroup: its 256 start offsets from the table (bins before its digit in the whole segment + the same digit in
//                         earlier tiles), then the tile in chunks of 256 elements IN INDEX ORDER: a wave finds the lanes that share a digit with
//                         eight ballots (rank among them = popcount below the lane), per-wave counts meet in LDS, offsets follow wave order --
//                         so equal keys keep their input order (what rocPRIM's radix sort gave; the goldens G8 / G12 hold ties bit for bit).
// "Descending" = ascending on the complemented digit.  Integer counts only: deterministic.  3 x 65 536 pairs (B = 64, S = 32): 8 launches of
// <= 96 workgroups, 2.4 MB moved per pass.
constexpr int RS_T = 256, RS_E = 8, RS_TILE = RS_T * RS_E;

__global__ __launch_bounds__(RS_T) void radix_hist_kernel(const unsigned long long* __restrict__ keys, int P, int shift, int nblk, unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[256];
    const int tid = threadIdx.x, blk = blockIdx.x, seg = blockIdx.y;
    s_h[tid] = 0;
    __syncthreads();
    const unsigned long long* k = keys + (size_t)seg * P;
#pragma unroll
    for (int e = 0; e < RS_E; ++e) {
        const int i = blk * RS_TILE + e * RS_T + tid;
        if (i < P) atomicAdd(&s_h[255u - (((unsigned)k[i] >> shift) & 255u)], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * 256 + tid) * nblk + blk] = s_h[tid];
}

template <typename V>
__global__ __launch_bounds__(RS_T) void radix_scatter_kernel(const unsigned long long* __restrict__ kin, const V* __restrict__ vin, unsigned long long* __restrict__ kout,
                                                             V* __restrict__ vout, int P, int shift, int nblk, const unsigned* __restrict__ hist) {
    __shared__ unsigned s_run[256];                 // next free output slot of every digit for this tile
    __shared__ unsigned s_cnt[RS_T / 64][256];      // per-wave digit counts of the current chunk
    __shared__ unsigned s_scan[RS_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x, seg = blockIdx.y;
    {   // digit `tid`: elements of the whole segment with this digit, and those in earlier tiles
        const unsigned* hrow = hist + ((size_t)seg * 256 + tid) * nblk;
        unsigned tot = 0, below = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned h = hrow[b];
            below += b < blk ? h : 0u;
            tot += h;
        }
        unsigned x = tot;                           // exclusive scan of tot over the 256 digits
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_scan[wave] = x;
        __syncthreads();
        unsigned woff = 0;
#pragma unroll
        for (int w = 0; w < RS_T / 64; ++w) woff += w < wave ? s_scan[w] : 0u;
        s_run[tid] = woff + x - tot + below;
    }
    const unsigned long long* k = kin + (size_t)seg * P;
    const V* v = vin + (size_t)seg * P;
    unsigned long long* ko = kout + (size_t)seg * P;
    V* vo = vout + (size_t)seg * P;
    for (int e = 0; e < RS_E; ++e) {
        const int i = blk * RS_TILE + e * RS_T + tid;
        const bool ok = i < P;
        const unsigned long long key = ok ? k[i] : 0ull;
        const V val = ok ? v[i] : V(0);
        const unsigned d = 255u - (((unsigned)key >> shift) & 255u);
#pragma unroll
        for (int w = 0; w < RS_T / 64; ++w) s_cnt[w][tid] = 0;
        unsigned long long same = __ballot(ok);     // valid lanes of this wave with my digit
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long b = __ballot((d >> bit) & 1u);
            same &= ((d >> bit) & 1u) ? b : ~b;
        }
        const unsigned rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        __syncthreads();                            // counts zeroed; s_run of the previous chunk (or the set-up) final
        if (ok && rank == 0) s_cnt[wave][d] = (unsigned)__popcll(same);
        __syncthreads();
        if (ok) {
            unsigned off = s_run[d] + rank;
#pragma unroll
            for (int w = 0; w < RS_T / 64; ++w) off += w < wave ? s_cnt[w][d] : 0u;
            ko[off] = key;
            vo[off] = val;
        }
        __syncthreads();                            // every offset read s_run before it moves
        {
            unsigned add = 0;
#pragma unroll
            for (int w = 0; w < RS_T / 64; ++w) add += s_cnt[w][tid];
            s_run[tid] += add;
        }
        __syncthreads();                            // ... and s_cnt before the next chunk zeroes it
    }
}

inline int radix_tiles(size_t P) { return (int)((P + RS_TILE - 1) / RS_TILE); }
inline size_t radix_hist_bytes(size_t P, int nseg) { return (size_t)nseg * 256 * radix_tiles(P) * sizeof(unsigned); }

// sorted pairs end in (k0, v0); (k1, v1) and hist are scratch
template <typename V>
int seg_radix_sort_desc(unsigned long long* k0, V* v0, unsigned long long* k1, V* v1, unsigned* hist, int P, int nseg, hipStream_t s) {
    const int nblk = radix_tiles((size_t)P);
    for (int pass = 0; pass < 4; ++pass) {
        const unsigned long long* ki = pass & 1 ? k1 : k0;
        const V* vi = pass & 1 ? v1 : v0;
        unsigned long long* ko = pass & 1 ? k0 : k1;
        V* vo = pass & 1 ? v0 : v1;
        DIR_LAUNCH(radix_hist_kernel, dim3(nblk, nseg), dim3(RS_T), 0, s, ki, P, 8 * pass, nblk, hist);
        DIR_LAUNCH((radix_scatter_kernel<V>), dim3(nblk, nseg), dim3(RS_T), 0, s, ki, vi, ko, vo, P, 8 * pass, nblk, (const unsigned*)hist);
    }
    return 0;
}

constexpr int LV_T = 1024, LV_E = 8;
// one workgroup per class: loss_c = sum_i err_sorted[i] * (J_i - J_{i-1}), J_i = 1 - (G - cumfg_i) / (G + cumbg_i)
// (models/lovasz_loss.py:19-31,194-197); result[c] = loss, result[3 + c] = G (0 -> class absent, skipped by 'present').
// A thread owns LV_E consecutive sorted elements per pass (local scan), the wave / workgroup scans run on the thread totals and the
// running foreground count and last J are carried from pass to pass.
__global__ __launch_bounds__(LV_T) void lovasz_kernel(const unsigned long long* __restrict__ keys, const unsigned char* __restrict__ vals,
                                                      const double* __restrict__ partial, int nwg, int P, double* __restrict__ result) {
    __shared__ double s_red[LV_T / 64];
    __shared__ int s_wsum[LV_T / 64];
    __shared__ double s_prevJ;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double G = 0;
    for (int i = 0; i < nwg; ++i) G += partial[(size_t)i * 6 + 3 + c];           // same order in every thread: identical value
    const unsigned long long* e = keys + (size_t)c * P;
    const unsigned char* fgv = vals + (size_t)c * P;
    long long carry = 0;                      // foreground count before this pass
    double prevJ = 0, acc = 0;                // J of the last element of the previous pass (J_{-1} := 0: jaccard[0] is kept)
    for (int base = 0; base < P; base += LV_T * LV_E) {
        const int i0 = base + tid * LV_E;
        int fg[LV_E]; float er[LV_E];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            const bool ok = i0 + k < P;
            fg[k] = ok ? (int)fgv[ok ? i0 + k : 0] : 0;
            er[k] = ok ? __uint_as_float((unsigned)e[ok ? i0 + k : 0]) : 0.f;
            mine += fg[k];
        }
        int x = mine;                         // inclusive scan of the thread totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < LV_T / 64; ++w) {
            const int sw = s_wsum[w];
            if (w < wave) woff += sw;
            total += sw;
        }
        long long cum = carry + woff + x - mine;          // foreground count before this thread's first element
        double J[LV_E];
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            cum += fg[k];
            const long long cumbg = (long long)(i0 + k + 1) - cum;
            J[k] = i0 + k < P ? 1.0 - (G - (double)cum) / (G + (double)cumbg) : 0.0;
        }
        // J of the element before this thread's first one: previous lane's last, previous wave's last, or the previous pass's last
        double Jm = __shfl_up(J[LV_E - 1], 1, 64);
        if (lane == 63) s_red[wave] = J[LV_E - 1];
        __syncthreads();
        if (lane == 0) Jm = wave == 0 ? prevJ : s_red[wave - 1];
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            if (i0 + k < P) acc += (double)er[k] * (J[k] - Jm);
            Jm = J[k];
        }
        if (tid == LV_T - 1) s_prevJ = J[LV_E - 1];
        __syncthreads();
        prevJ = s_prevJ;
        carry += total;
    }
    const double t = block_sum_d<LV_T / 64>(acc, s_red, tid);
    if (tid == 0) { result[c] = t; result[3 + c] = G; }
}

// one wave: lanes stride over the per-workgroup partials, fixed-shape reduction tree (deterministic)
__global__ __launch_bounds__(64) void dense_final_kernel(const double* __restrict__ partial, int chunks, int B, int S, const double* __restrict__ lov,
                                                          float dense_weight, float* __restrict__ out) {
    const int lane = threadIdx.x;
    double wn = 0, w = 0, dsum = 0;
    for (int b = lane; b < B; b += 64) {
        double d = 0;
        for (int k = 0; k < chunks; ++k) {
            const double* p = partial + ((size_t)b * chunks + k) * 6;
            wn += p[0]; w += p[1]; d += p[2];
        }
        dsum += d / (3.0 * S * S);              // SmoothL1Loss: per-sample mean, then the batch mean
    }
    wn = wave_sum_d(wn); w = wave_sum_d(w); dsum = wave_sum_d(dsum);
    if (lane != 0) return;
    double ls = 0; int n = 0;
    for (int c = 0; c < 3; ++c)
        if (lov[3 + c] > 0) { ls += lov[c]; ++n; }
    out[0] = (float)(wn / w * 0.1 * dense_weight);
    out[1] = (float)(dsum / B * dense_weight);
    out[2] = (float)((n ? ls / n : 0.0) * 0.1 * dense_weight);
}

// workspace carve-up (all offsets 256-byte aligned)
struct DenseWs { size_t keys_in, keys_out, vals_in, vals_out, partial, lov, temp, temp_bytes, total; };
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
int dense_ws(int B, int S, DenseWs& w) {
    const size_t P = (size_t)B * S * S;
    const int chunks = (S * S + LT - 1) / LT;
    size_t o = 0;
    w.keys_in = o; o = al(o + 3 * P * 8);
    w.keys_out = o; o = al(o + 3 * P * 8);
    w.vals_in = o; o = al(o + 3 * P);
    w.vals_out = o; o = al(o + 3 * P);
    w.partial = o; o = al(o + (size_t)B * chunks * 6 * sizeof(double));
    w.lov = o; o = al(o + 6 * sizeof(double));
    const size_t tb = radix_hist_bytes(P, 3);     // the sort's digit histograms
    w.temp = o; w.temp_bytes = tb; o = al(o + tb);
    w.total = o;
    return 0;
}

}  // namespace
}  // namespace dir


// ================================================================================================ backward (gradients w.r.t. the predictions)
namespace dir {
namespace {

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.f) - (x < 0.f)); }
// d smooth_l1_term / d x (models/loss.py:74-81 under autograd: z inside the knee, 0.01 sign(z) outside)
__device__ __forceinline__ float smooth_l1_dterm(float x, float y) {
#pragma clang fp contract(off)
    const float z = x - y;
    return fabsf(z) < 0.01f ? z : 0.01f * sgn(z);
}

struct StageBwdArgs {
    dir_loss_pred p;
    dir_loss_target g;
    dir_loss_pred_grad o;
    const float* gout;          // [13] upstream gradients (NULL: ones)
    const int32_t* vf_off[2];   // CSR vertex -> (face * 3 + corner) lists: offsets [NV + 1]
    const int32_t* vf_idx[2];   // entries [3 F]
    float coord_weight;
    int B;
};

// One workgroup per (sample, hand).  Phase 1: every triangle's edge-length and normal-vector gradient contributions to its three
// corners go to LDS (9 floats per triangle); phase 2: every vertex sums its corners in CSR order (deterministic, no atomics), adds the
// SmoothL1 term and applies the 1 / 0.15 of the normalisation.
__global__ __launch_bounds__(LT) void stage_loss_bwd_kernel(StageBwdArgs a) {
    extern __shared__ float s_dyn[];
    float* s_pm = s_dyn;                 // [NV * 3]
    float* s_gm = s_pm + NV * 3;         // [NV * 3]
    float* s_fg = s_gm + NV * 3;         // [F * 9]
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, F = a.g.n_faces;
    const float* cen = a.g.center[h] + (size_t)b * 3;
    const float c0 = cen[0], c1 = cen[1], c2 = cen[2];
    auto gk = [&](int k) { return a.gout ? a.gout[k] : 1.f; };
    const float invB = 1.f / (float)a.B;
    {
        const float* pm = a.p.mesh_xyz[h] + (size_t)b * NV * 3;
        const float* gm = a.g.mesh_3d[h] + (size_t)b * NV * 3;
        for (int i = tid; i < NV * 3; i += LT) {
            const int c = i % 3;
            s_pm[i] = pm[i] / 0.15f;
            s_gm[i] = (gm[i] - (c == 0 ? c0 : c == 1 ? c1 : c2)) / 0.15f;
        }
    }
    __syncthreads();
    const float ke = gk(8 + h) * invB / (3.f * F), kn = gk(10 + h) * 0.1f * invB / (3.f * F);
    const int32_t* face = a.g.faces[h];
    for (int f = tid; f < F; f += LT) {
        const int idx[3] = {face[f * 3], face[f * 3 + 1], face[f * 3 + 2]};
        V3 p[3], g[3], acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = ld3(s_pm + idx[c] * 3); g[c] = ld3(s_gm + idx[c] * 3); acc[c] = V3{0.f, 0.f, 0.f}; }
        const V3 n = normalize(cross(normalize(sub(g[1], g[0])), normalize(sub(g[2], g[0]))));
        auto add = [&](int c, V3 v, float sc) { acc[c].x += sc * v.x; acc[c].y += sc * v.y; acc[c].z += sc * v.z; };
        const int ea[3] = {0, 0, 1}, eb[3] = {1, 2, 2};               // edge pairs of EdgeLengthLoss; NormalVectorLoss: (hi, lo) = (eb, ea)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const V3 d = sub(p[ea[e]], p[eb[e]]);
            const float len = sqrtf(sumsq(d) + 1e-12f), leng = sqrtf(sumsq(sub(g[ea[e]], g[eb[e]])) + 1e-12f);
            const float s = sgn(len - leng) / len * ke;
            add(ea[e], d, s); add(eb[e], d, -s);
            const V3 ev = sub(p[eb[e]], p[ea[e]]);                    // hi - lo
            const float ln = fmaxf(sqrtf(sumsq(ev)), 1e-12f);
            const V3 v = V3{ev.x / ln, ev.y / ln, ev.z / ln};
            const float dt = dot(v, n), sc = sgn(dt) / ln * kn;
            const V3 t = V3{n.x - dt * v.x, n.y - dt * v.y, n.z - dt * v.z};
            add(eb[e], t, sc); add(ea[e], t, -sc);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_fg[(f * 3 + c) * 3] = acc[c].x; s_fg[(f * 3 + c) * 3 + 1] = acc[c].y; s_fg[(f * 3 + c) * 3 + 2] = acc[c].z; }
    }
    __syncthreads();
    const float km = gk(6 + h) * a.coord_weight * invB / (NV * 3.f);
    float* gmx = a.o.mesh_xyz[h] + (size_t)b * NV * 3;
    for (int v = tid; v < NV; v += LT) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = a.vf_off[h][v]; e < a.vf_off[h][v + 1]; ++e) {
            const int k = a.vf_idx[h][e] * 3;
            sx += s_fg[k]; sy += s_fg[k + 1]; sz += s_fg[k + 2];
        }
        sx += km * smooth_l1_dterm(s_pm[3 * v], s_gm[3 * v]);
        sy += km * smooth_l1_dterm(s_pm[3 * v + 1], s_gm[3 * v + 1]);
        sz += km * smooth_l1_dterm(s_pm[3 * v + 2], s_gm[3 * v + 2]);
        gmx[3 * v] = sx / 0.15f; gmx[3 * v + 1] = sy / 0.15f; gmx[3 * v + 2] = sz / 0.15f;
    }
    // joints, uv, offset: plain SmoothL1 gradients
    if (tid < NJ * 3) {
        const int c = tid % 3;
        const float x = a.p.joint_xyz[h][(size_t)b * NJ * 3 + tid] / 0.15f;
        const float y = (a.g.joint_3d[h][(size_t)b * NJ * 3 + tid] - (c == 0 ? c0 : c == 1 ? c1 : c2)) / 0.15f;
        a.o.joint_xyz[h][(size_t)b * NJ * 3 + tid] = gk(4 + h) * a.coord_weight * invB / (NJ * 3.f) * smooth_l1_dterm(x, y) / 0.15f;
    }
    if (tid < NJ * 2) {
        const int j = tid >> 1, c = tid & 1;
        a.o.joint_uv[h][(size_t)b * NJ * 2 + tid] = gk(0 + h) * a.coord_weight * invB / (NJ * 2.f) *
            smooth_l1_dterm(a.p.joint_uv[h][(size_t)b * NJ * 2 + tid], a.g.joint_2d[h][((size_t)b * NJ + j) * a.g.c2 + c]);
    }
    for (int i = tid; i < NV * 2; i += LT) {
        const int v = i >> 1, c = i & 1;
        a.o.mesh_uv[h][(size_t)b * NV * 2 + i] = gk(2 + h) * a.coord_weight * invB / (NV * 2.f) *
            smooth_l1_dterm(a.p.mesh_uv[h][(size_t)b * NV * 2 + i], a.g.mesh_2d[h][((size_t)b * NV + v) * a.g.c2 + c]);
    }
    if (h == 0 && tid < 3) {
        const float y = (a.g.center[1][(size_t)b * 3 + tid] - a.g.center[0][(size_t)b * 3 + tid]) / 0.15f;
        a.o.offset[(size_t)b * 3 + tid] = gk(12) * a.coord_weight * invB / 3.f * smooth_l1_dterm(a.p.offset[(size_t)b * 3 + tid], y);
    }
}

// ---- seg / dense / lovasz
struct DenseBwdArgs {
    const float* seg; const float* dense; const float* gt_seg; const float* gt_dense;
    unsigned long long* keys; unsigned* vals;      // [3][P]: composite (class, error) keys; values = pixel index | fg << 31
    double* partial;                                // [nwg][4]: sum of class weights, fg count of class 0..2
    unsigned char* label;                           // [P]
    float* gdense;
    const float* gout;                              // [3] upstream gradients (NULL: ones)
    int B, S, H, W, P, chunks;
    float cw[3], dense_weight;
};

__global__ __launch_bounds__(LT) void dense_bwd_pixel_kernel(DenseBwdArgs a) {
#pragma clang fp contract(off)
    __shared__ double s_red[LT / 64];
    const int b = blockIdx.y, tid = threadIdx.x, S = a.S, pix = blockIdx.x * LT + tid;
    const bool ok = pix < S * S;
    const int oy = ok ? pix / S : 0, ox = ok ? pix - oy * S : 0;
    double v[4] = {0, 0, 0, 0};
    if (ok) {
        const int sy = min((int)floorf(oy * ((float)a.H / S)), a.H - 1), sx = min((int)floorf(ox * ((float)a.W / S)), a.W - 1);
        int lab = (int)a.gt_seg[((size_t)b * a.H + sy) * a.W + sx];
        lab = lab < 0 ? 0 : lab > 2 ? 2 : lab;
        const size_t plane = (size_t)S * S, p = (size_t)b * plane + pix;
        a.label[p] = (unsigned char)lab;
        v[0] = a.cw[lab];
        const float* sg = a.seg + (size_t)b * 3 * plane + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float fg = lab == c ? 1.f : 0.f;
            a.keys[(size_t)c * a.P + p] = ((unsigned long long)(2 - c) << 32) | (unsigned long long)__float_as_uint(fabsf(fg - sg[c * plane]));
            a.vals[(size_t)c * a.P + p] = (unsigned)p | (lab == c ? 0x80000000u : 0u);
            v[1 + c] = lab == c;
        }
        // dense SmoothL1 gradient (target = bilinear F.interpolate of the GT map, as in the forward)
        const float fy = fmaxf((oy + 0.5f) * ((float)a.H / S) - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * ((float)a.W / S) - 0.5f, 0.f);
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx), y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
        const float wy1 = fy - y0, wy0 = 1.f - wy1, wx1 = fx - x0, wx0 = 1.f - wx1;
        const float kd = (a.gout ? a.gout[1] : 1.f) * a.dense_weight / ((float)a.B * 3.f * S * S);
        for (int c = 0; c < 3; ++c) {
            const float* gd = a.gt_dense + ((size_t)b * 3 + c) * a.H * a.W;
            const float r0 = wx0 * gd[(size_t)y0 * a.W + x0] + wx1 * gd[(size_t)y0 * a.W + x1];
            const float r1 = wx0 * gd[(size_t)y1 * a.W + x0] + wx1 * gd[(size_t)y1 * a.W + x1];
            const float t = wy0 * r0 + wy1 * r1;
            const size_t o = ((size_t)b * 3 + c) * plane + pix;
            a.gdense[o] = kd * smooth_l1_dterm(a.dense[o], t);
        }
    }
    double* out = a.partial + ((size_t)b * a.chunks + blockIdx.x) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t = block_sum_d<LT / 64>(v[k], s_red, tid);
        if (tid == 0) out[k] = t;
    }
}

// same scan as lovasz_kernel; writes d loss_c / d logit = sign(logit - fg) (J_i - J_{i-1}) at the element's pixel
__global__ __launch_bounds__(LV_T) void lovasz_bwd_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals,
                                                          const float* __restrict__ seg, const double* __restrict__ partial, int nwg, int P,
                                                          int plane, float* __restrict__ glov, double* __restrict__ result) {
    __shared__ double s_red[LV_T / 64];
    __shared__ int s_wsum[LV_T / 64];
    __shared__ double s_prevJ;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double G = 0, W = 0;
    for (int i = 0; i < nwg; ++i) { G += partial[(size_t)i * 4 + 1 + c]; W += partial[(size_t)i * 4]; }
    if (tid == 0) { result[c] = G; if (c == 0) result[3] = W; }
    const unsigned* vv = vals + (size_t)c * P;
    long long carry = 0;
    double prevJ = 0;
    for (int base = 0; base < P; base += LV_T * LV_E) {
        const int i0 = base + tid * LV_E;
        int fg[LV_E]; unsigned px[LV_E];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            const bool ok = i0 + k < P;
            const unsigned v = ok ? vv[i0 + k] : 0u;
            fg[k] = ok ? (int)(v >> 31) : 0;
            px[k] = v & 0x7fffffffu;
            mine += fg[k];
        }
        int x = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < LV_T / 64; ++w) {
            const int sw = s_wsum[w];
            if (w < wave) woff += sw;
            total += sw;
        }
        long long cum = carry + woff + x - mine;
        double J[LV_E];
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            cum += fg[k];
            const long long cumbg = (long long)(i0 + k + 1) - cum;
            J[k] = i0 + k < P ? 1.0 - (G - (double)cum) / (G + (double)cumbg) : 0.0;
        }
        double Jm = __shfl_up(J[LV_E - 1], 1, 64);
        if (lane == 63) s_red[wave] = J[LV_E - 1];
        __syncthreads();
        if (lane == 0) Jm = wave == 0 ? prevJ : s_red[wave - 1];
#pragma unroll
        for (int k = 0; k < LV_E; ++k) {
            if (i0 + k < P) {
                const unsigned p = px[k];
                const int bb = p / plane, pix = p - bb * plane;
                const float lg = seg[((size_t)bb * 3 + c) * plane + pix];
                glov[(size_t)c * P + p] = sgn(lg - (float)fg[k]) * (float)(J[k] - Jm);
            }
            Jm = J[k];
        }
        if (tid == LV_T - 1) s_prevJ = J[LV_E - 1];
        __syncthreads();
        prevJ = s_prevJ;
        carry += total;
    }
}

// grad_seg[b][c][pix] = g_seg 0.1 dw w[y] (softmax_c - [c == y]) / sum w  +  g_lov 0.1 dw / n_present * glov[c][p]
__global__ __launch_bounds__(LT) void dense_bwd_final_kernel(DenseBwdArgs a, const float* __restrict__ glov, const double* __restrict__ result,
                                                             float* __restrict__ gseg) {
    const int b = blockIdx.y, pix = blockIdx.x * LT + threadIdx.x, S = a.S;
    if (pix >= S * S) return;
    const size_t plane = (size_t)S * S, p = (size_t)b * plane + pix;
    const int lab = a.label[p];
    const float* sg = a.seg + (size_t)b * 3 * plane + pix;
    const double l0 = sg[0], l1 = sg[plane], l2 = sg[2 * plane];
    const double m = fmax(l0, fmax(l1, l2));
    const double e0 = exp(l0 - m), e1 = exp(l1 - m), e2 = exp(l2 - m), es = e0 + e1 + e2;
    int npres = 0;
    for (int c = 0; c < 3; ++c) npres += result[c] > 0;
    const double kce = (double)(a.gout ? a.gout[0] : 1.f) * 0.1 * a.dense_weight * a.cw[lab] / result[3];
    const double klv = npres ? (double)(a.gout ? a.gout[2] : 1.f) * 0.1 * a.dense_weight / npres : 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double sm = (c == 0 ? e0 : c == 1 ? e1 : e2) / es;
        const double gl = result[c] > 0 ? (double)glov[(size_t)c * a.P + p] : 0.0;
        gseg[((size_t)b * 3 + c) * plane + pix] = (float)(kce * (sm - (lab == c ? 1.0 : 0.0)) + klv * gl);
    }
}

struct DenseBwdWs { size_t keys_in, keys_out, vals_in, vals_out, partial, result, label, glov, temp, temp_bytes, total; };
int dense_bwd_ws(int B, int S, DenseBwdWs& w) {
    const size_t P = (size_t)B * S * S;
    const int chunks = (S * S + LT - 1) / LT;
    size_t o = 0;
    w.keys_in = o; o = al(o + 3 * P * 8);
    w.keys_out = o; o = al(o + 3 * P * 8);
    w.vals_in = o; o = al(o + 3 * P * 4);
    w.vals_out = o; o = al(o + 3 * P * 4);
    w.partial = o; o = al(o + (size_t)B * chunks * 4 * sizeof(double));
    w.result = o; o = al(o + 4 * sizeof(double));
    w.label = o; o = al(o + P);
    w.glov = o; o = al(o + 3 * P * 4);
    const size_t tb = radix_hist_bytes(P, 3);
    w.temp = o; w.temp_bytes = tb; o = al(o + tb);
    w.total = o;
    return 0;
}

}  // namespace
}  // namespace dir

extern "C" int dir_stage_losses_backward(const dir_loss_pred* pred_host, const dir_loss_target* gt_host, float coord_weight,
                                         const float* grad_out13, const int32_t* const* vert_face_offsets, const int32_t* const* vert_face_index,
                                         const dir_loss_pred_grad* grads_host, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(pred_host && gt_host && grads_host && vert_face_offsets && vert_face_index, "dir_stage_losses_backward: null pointer");
    DIR_REQUIRE(B > 0 && gt_host->c2 >= 2 && gt_host->n_faces > 0, "dir_stage_losses_backward: bad arguments");
    for (int h = 0; h < 2; ++h)
        DIR_REQUIRE(pred_host->joint_uv[h] && pred_host->mesh_uv[h] && pred_host->joint_xyz[h] && pred_host->mesh_xyz[h] &&
                    gt_host->joint_2d[h] && gt_host->mesh_2d[h] && gt_host->joint_3d[h] && gt_host->mesh_3d[h] && gt_host->center[h] &&
                    gt_host->faces[h] && grads_host->joint_uv[h] && grads_host->mesh_uv[h] && grads_host->joint_xyz[h] &&
                    grads_host->mesh_xyz[h] && vert_face_offsets[h] && vert_face_index[h],
                    "dir_stage_losses_backward: null tensor (pd_mesh_uv must be given: it is a leaf of this gradient)");
    DIR_REQUIRE(pred_host->offset && grads_host->offset, "dir_stage_losses_backward: null offset");
    const size_t lds = ((size_t)2 * NV * 3 + (size_t)gt_host->n_faces * 9) * sizeof(float);
    DIR_REQUIRE(lds <= 150 * 1024, "dir_stage_losses_backward: too many faces for LDS (%d)", gt_host->n_faces);
    StageBwdArgs a;
    a.p = *pred_host; a.g = *gt_host; a.o = *grads_host; a.gout = grad_out13; a.coord_weight = coord_weight; a.B = B;
    for (int h = 0; h < 2; ++h) { a.vf_off[h] = vert_face_offsets[h]; a.vf_idx[h] = vert_face_index[h]; }
    static bool attr_set = false;
    if (!attr_set) {
        DIR_REQUIRE(hipFuncSetAttribute((const void*)stage_loss_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess,
                    "dir_stage_losses_backward: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    DIR_LAUNCH(stage_loss_bwd_kernel, dim3(B, 2), dim3(LT), lds, (hipStream_t)stream, a);
    return check_launch("dir_stage_losses_backward");
}

extern "C" long long dir_dense_losses_backward_workspace_bytes(int B, int S) {
    using namespace dir;
    if (B <= 0 || S <= 0) return -1;
    DenseBwdWs w;
    if (dense_bwd_ws(B, S, w)) return -1;
    return (long long)w.total;
}

extern "C" int dir_dense_losses_backward(const float* seg_logits, const float* dense_pred, const float* gt_seg, const float* gt_dense,
                                         const float* class_weight_host, float dense_weight, const float* grad_out3, void* workspace,
                                         long long workspace_bytes, float* grad_seg, float* grad_dense, int B, int S, int H, int W,
                                         void* stream) {
    using namespace dir;
    DIR_REQUIRE(seg_logits && dense_pred && gt_seg && gt_dense && class_weight_host && workspace && grad_seg && grad_dense,
                "dir_dense_losses_backward: null pointer");
    DIR_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && (long long)B * S * S * 3 < (1ll << 31), "dir_dense_losses_backward: bad shape");
    DenseBwdWs w;
    DIR_REQUIRE(dense_bwd_ws(B, S, w) == 0, "dir_dense_losses_backward: workspace layout failed");
    DIR_REQUIRE(workspace_bytes >= (long long)w.total, "dir_dense_losses_backward: workspace too small (%lld < %zu)", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    DenseBwdArgs a;
    a.seg = seg_logits; a.dense = dense_pred; a.gt_seg = gt_seg; a.gt_dense = gt_dense; a.gdense = grad_dense; a.gout = grad_out3;
    a.keys = (unsigned long long*)(ws + w.keys_in); a.vals = (unsigned*)(ws + w.vals_in); a.partial = (double*)(ws + w.partial);
    a.label = (unsigned char*)(ws + w.label);
    a.B = B; a.S = S; a.H = H; a.W = W; a.P = B * S * S; a.chunks = (S * S + LT - 1) / LT; a.dense_weight = dense_weight;
    for (int c = 0; c < 3; ++c) a.cw[c] = class_weight_host[c];
    DIR_LAUNCH(dense_bwd_pixel_kernel, dim3(a.chunks, B), dim3(LT), 0, s, a);
    // per-class descending stable sort; the sorted pairs come back in (keys_in, vals_in)
    seg_radix_sort_desc<unsigned>(a.keys, a.vals, (unsigned long long*)(ws + w.keys_out), (unsigned*)(ws + w.vals_out), (unsigned*)(ws + w.temp), a.P, 3, s);
    double* result = (double*)(ws + w.result);
    float* glov = (float*)(ws + w.glov);
    DIR_LAUNCH(lovasz_bwd_kernel, dim3(3), dim3(LV_T), 0, s, (const unsigned long long*)a.keys, (const unsigned*)a.vals,
                       seg_logits, (const double*)a.partial, B * a.chunks, a.P, S * S, glov, result);
    DIR_LAUNCH(dense_bwd_final_kernel, dim3(a.chunks, B), dim3(LT), 0, s, a, (const float*)glov, (const double*)result, grad_seg);
    return check_launch("dir_dense_losses_backward");
}

extern "C" int dir_stage_losses_forward(const dir_loss_pred* pred_host, const dir_loss_target* gt_host, float coord_weight,
                                        double* scratch, float* out13, int B, void* stream) {
    using namespace dir;
    DIR_REQUIRE(pred_host && gt_host && scratch && out13, "dir_stage_losses_forward: null pointer");
    DIR_REQUIRE(B > 0 && gt_host->c2 >= 2 && gt_host->n_faces > 0, "dir_stage_losses_forward: bad arguments");
    for (int h = 0; h < 2; ++h)
        DIR_REQUIRE(pred_host->joint_uv[h] && (pred_host->mesh_uv[h] || pred_host->proj[h]) && pred_host->joint_xyz[h] && pred_host->mesh_xyz[h] &&
                    gt_host->joint_2d[h] && gt_host->mesh_2d[h] && gt_host->joint_3d[h] && gt_host->mesh_3d[h] && gt_host->center[h] &&
                    gt_host->faces[h], "dir_stage_losses_forward: null tensor");
    DIR_REQUIRE(pred_host->offset, "dir_stage_losses_forward: null offset");
    StageArgs a;
    a.p = *pred_host; a.g = *gt_host; a.scratch = scratch; a.B = B;
    hipStream_t s = (hipStream_t)stream;
    DIR_LAUNCH(stage_loss_kernel, dim3(B, 2), dim3(LT), 0, s, a);
    DIR_LAUNCH(stage_reduce_kernel, dim3(1), dim3(64 * NTERM), 0, s, scratch, out13, B, coord_weight);
    return check_launch("dir_stage_losses_forward");
}

extern "C" long long dir_dense_losses_workspace_bytes(int B, int S) {
    using namespace dir;
    if (B <= 0 || S <= 0) return -1;
    DenseWs w;
    if (dense_ws(B, S, w)) return -1;
    return (long long)w.total;
}

extern "C" int dir_dense_losses_forward(const float* seg_logits, const float* dense_pred, const float* gt_seg, const float* gt_dense,
                                        const float* class_weight_host, float dense_weight, void* workspace, long long workspace_bytes,
                                        float* out3, int B, int S, int H, int W, void* stream) {
    using namespace dir;
    DIR_REQUIRE(seg_logits && dense_pred && gt_seg && gt_dense && class_weight_host && workspace && out3,
                "dir_dense_losses_forward: null pointer");
    DIR_REQUIRE(B > 0 && S > 0 && H > 0 && W > 0 && (long long)B * S * S * 3 < (1ll << 31), "dir_dense_losses_forward: bad shape");
    DenseWs w;
    DIR_REQUIRE(dense_ws(B, S, w) == 0, "dir_dense_losses_forward: workspace layout failed");
    DIR_REQUIRE(workspace_bytes >= (long long)w.total, "dir_dense_losses_forward: workspace too small (%lld < %zu)", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    DenseArgs a;
    a.seg = seg_logits; a.dense = dense_pred; a.gt_seg = gt_seg; a.gt_dense = gt_dense;
    a.keys = (unsigned long long*)(ws + w.keys_in); a.vals = (unsigned char*)(ws + w.vals_in); a.partial = (double*)(ws + w.partial);
    a.B = B; a.S = S; a.H = H; a.W = W; a.P = B * S * S; a.chunks = (S * S + LT - 1) / LT;
    for (int c = 0; c < 3; ++c) a.cw[c] = class_weight_host[c];
    DIR_LAUNCH(dense_pixel_kernel, dim3(a.chunks, B), dim3(LT), 0, s, a);
    seg_radix_sort_desc<unsigned char>(a.keys, a.vals, (unsigned long long*)(ws + w.keys_out), (unsigned char*)(ws + w.vals_out), (unsigned*)(ws + w.temp), a.P, 3, s);
    double* lov = (double*)(ws + w.lov);
    DIR_LAUNCH(lovasz_kernel, dim3(3), dim3(LV_T), 0, s, (const unsigned long long*)a.keys, (const unsigned char*)a.vals,
                       (const double*)a.partial, B * a.chunks, a.P, lov);
    DIR_LAUNCH(dense_final_kernel, dim3(1), dim3(64), 0, s, (const double*)a.partial, a.chunks, B, S, (const double*)lov, dense_weight, out3);
    return check_launch("dir_dense_losses_forward");
}
