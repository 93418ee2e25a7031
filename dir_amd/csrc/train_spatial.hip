// Backward pass, image half: the spatial operators between the convolutions -- what torch autograd runs for
//   nn.MaxPool2d(3, 2, 1)                       models/backbone/resnet.py:247
//   nn.Upsample(scale_factor=2, 'bilinear')     models/dir.py:392,398 (align_corners False)
//   InitRegressor's attention pooling           models/dir.py:263-270  (sigmoid attention map, attention-weighted mean, plain mean)
//   Joint2BoneFeature.bone_proj                 models/dir.py:132-174  (gradients w.r.t. the re-embedded joint features and the joint uv)
// fp32, NHWC, deterministic (gather formulations with fixed summation orders; no atomics).  Correctness first: these run once per
// training step on maps of at most [B,128,128,64].
#include "dir_common.h"
#include "bone_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------------------------------ max pool 3x3 / 2 / 1
// g x[p] = sum over the (at most four) windows that contain p of g y[window] * [p is the window's argmax]; argmax = the FIRST maximum in
// (ky, kx) scan order (ATen max_pool2d_with_indices: `val > maxval`), recomputed from x.  One thread per input element.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int Ho, int Wo) {
    const long long n = (long long)B * H * W * C, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H), b = (int)(p / H);
    float acc = 0.f;
    for (int oy = max(0, (iy - 1 + 1) / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy)          // windows rows 2 oy - 1 .. 2 oy + 1
        for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
            if (iy < 2 * oy - 1 || iy > 2 * oy + 1 || ix < 2 * ox - 1 || ix > 2 * ox + 1) continue;
            float m = -INFINITY; int ay = -1, ax = -1;
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = 2 * oy - 1 + ky;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = 2 * ox - 1 + kx;
                    if (xx < 0 || xx >= W) continue;
                    const float v = x[(((long long)b * H + yy) * W + xx) * C + c];
                    if (v > m || ay < 0) { m = v; ay = yy; ax = xx; }
                }
            }
            if (ay == iy && ax == ix) acc += gy[(((long long)b * Ho + oy) * Wo + ox) * C + c];
        }
    gx[i] = acc;
}

// the same, four channels per thread (C % 4 == 0): 16-byte loads -- a quarter of the load instructions of the gather (each input element looks
// at up to four windows of nine taps), same per-channel comparisons and sums
__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long n = (long long)B * H * W * C4, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    long long p = i / C4;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H), b = (int)(p / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oy = max(0, iy / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy)
        for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
            if (iy < 2 * oy - 1 || iy > 2 * oy + 1 || ix < 2 * ox - 1 || ix > 2 * ox + 1) continue;
            float m[4]; int arg[4] = {-1, -1, -1, -1};
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = 2 * oy - 1 + ky;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = 2 * ox - 1 + kx;
                    if (xx < 0 || xx >= W) continue;
                    const float4 v4 = *reinterpret_cast<const float4*>(x + (((long long)b * H + yy) * W + xx) * C + c);
                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (arg[e] < 0 || v[e] > m[e]) { m[e] = v[e]; arg[e] = yy * W + xx; }
                }
            }
            const float4 g4 = *reinterpret_cast<const float4*>(gy + (((long long)b * Ho + oy) * Wo + ox) * C + c);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (arg[e] == iy * W + ix) acc[e] += g[e];
        }
    *reinterpret_cast<float4*>(gx + i * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// round 5: one thread = a 2 x 2 block of input pixels (rows 2 by, 2 by + 1; columns 2 bx, 2 bx + 1) x 4 channels.  Only the windows (by, bx),
// (by, bx + 1), (by + 1, bx), (by + 1, bx + 1) contain those pixels, and together they cover a 5 x 5 patch: 25 loads of 16 bytes per four pixels
// instead of 81 (an element of the per-pixel kernel above scans 1 / 2 / 4 windows of nine taps).  Same comparisons (first maximum in (ky, kx)
// order) and the same order of the (at most four) additions per element -- oy outer, ox inner: the same bits.  0.287 -> ~0.1 ms on the stem's map.
__global__ __launch_bounds__(256) void maxpool_bwd4_block_kernel(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long long n = (long long)B * Hb * Wb * C4, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C4) * 4;
    long long p = i / C4;
    const int bx = (int)(p % Wb); p /= Wb;
    const int by = (int)(p % Hb), b = (int)(p / Hb);
    const int y0 = 2 * by - 1, x0 = 2 * bx - 1;                      // patch origin (may be -1)
    float4 v[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int yy = y0 + r, xx = x0 + q;
            v[r][q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? *reinterpret_cast<const float4*>(x + (((long long)b * H + yy) * W + xx) * C + c)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float acc[2][2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][q][e] = 0.f;
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
            const int oy = by + wy, ox = bx + wx;
            if (oy >= Ho || ox >= Wo) continue;
            float m[4]; int arg[4] = {-1, -1, -1, -1};              // arg = 5 r + q inside the patch
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = 2 * wy + ky, yy = y0 + r;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int q = 2 * wx + kx, xx = x0 + q;
                    if (xx < 0 || xx >= W) continue;
                    const float t[4] = {v[r][q].x, v[r][q].y, v[r][q].z, v[r][q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (arg[e] < 0 || t[e] > m[e]) { m[e] = t[e]; arg[e] = 5 * r + q; }
                }
            }
            const float4 g4 = *reinterpret_cast<const float4*>(gy + (((long long)b * Ho + oy) * Wo + ox) * C + c);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (arg[e] == 5 * (r + 1) + (q + 1)) acc[r][q][e] += g[e];          // the block's pixels sit at patch (1..2, 1..2)
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int yy = 2 * by + r, xx = 2 * bx + q;
            if (yy < H && xx < W)
                *reinterpret_cast<float4*>(gx + (((long long)b * H + yy) * W + xx) * C + c) = make_float4(acc[r][q][0], acc[r][q][1], acc[r][q][2], acc[r][q][3]);
        }
}

// ------------------------------------------------------------------------------------------------------------------ bilinear 2x upsample
// transposed interpolation, gather form: input row iy receives from output rows 2 iy - 1 .. 2 iy + 2 with the weights the forward used
// (src = max((dst + 0.5) / 2 - 0.5, 0); rows y0 = floor(src), y1 = min(y0 + 1, H - 1) with 1 - l, l).  gx may be a channel slice.
__device__ __forceinline__ float up_weight(int o, int i, int n) {        // weight of input index i in output index o
    const float s = fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.f);
    const int i0 = (int)s, i1 = min(i0 + 1, n - 1);
    const float l = s - i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* gy, float* gx, int B, int H, int W, int C, int gy_cs, int gy_co) {
    const long long n = (long long)B * H * W * C, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    long long p = i / C;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H), b = (int)(p / H);
    float acc = 0.f;
    for (int oy = max(0, 2 * iy - 1); oy <= min(2 * H - 1, 2 * iy + 2); ++oy) {
        const float wy = up_weight(oy, iy, H);
        if (wy == 0.f) continue;
        for (int ox = max(0, 2 * ix - 1); ox <= min(2 * W - 1, 2 * ix + 2); ++ox) {
            const float wx = up_weight(ox, ix, W);
            if (wx != 0.f) acc = fmaf(wy * wx, gy[(((long long)b * 2 * H + oy) * 2 * W + ox) * gy_cs + gy_co + c], acc);
        }
    }
    gx[i] = acc;
}

// ------------------------------------------------------------------------------------------------------------------ attention pooling
// models/dir.py:263-270:  a = sigmoid(logit) [B,HW];  F[c] = sum_p f[p,c] a[p] / (sum_p a[p] + 1e-8);  mean[c] = sum_p f[p,c] / HW.
// One workgroup per sample; thread = channel (strided), pixels in order.
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* feat, const float* logit, float* attn, float* pooled, float* mean, int HW, int C) {
    extern __shared__ float s_a[];                   // [HW]
    __shared__ float s_sum;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int p = tid; p < HW; p += 256) { const float a = 1.f / (1.f + expf(-logit[(long long)b * HW + p])); s_a[p] = a; if (attn) attn[(long long)b * HW + p] = a; }
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int p = 0; p < HW; ++p) s += s_a[p]; s_sum = s + 1e-8f; }
    __syncthreads();
    const float* f = feat + (long long)b * HW * C;
    for (int c = tid; c < C; c += 256) {
        float num = 0.f, m = 0.f;
        for (int p = 0; p < HW; ++p) { const float v = f[(long long)p * C + c]; num = fmaf(v, s_a[p], num); m += v; }
        if (pooled) pooled[(long long)b * C + c] = num / s_sum;
        if (mean) mean[(long long)b * C + c] = m / HW;
    }
}
// g f[p,c] (+)= gF[c] a[p] / S + gmean[c] / HW;   g a[p] = sum_c gF[c] (f[p,c] - F[c]) / S;   g logit[p] = g a[p] a[p] (1 - a[p])
// One workgroup per (sample, pixel) -- B x HW of them instead of B: the same expressions and the same fixed reduction tree per pixel as the
// one-workgroup-per-sample form it replaces (bit-identical), 0.48 ms -> a few microseconds at B = 32, HW = 64, C = 2048.
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* feat, const float* attn, const float* pooled, const float* g_pooled, const float* g_mean,
                                                           float* g_feat, float* g_logit, int HW, int C, int accumulate) {
    extern __shared__ float s_a[];                   // [HW]
    __shared__ float s_sum, s_red[256];
    const int b = blockIdx.x / HW, p = blockIdx.x - b * HW, tid = threadIdx.x;
    for (int q = tid; q < HW; q += 256) s_a[q] = attn[(long long)b * HW + q];
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int q = 0; q < HW; ++q) s += s_a[q]; s_sum = s + 1e-8f; }
    __syncthreads();
    const float S = s_sum, ap = s_a[p];
    const float* f = feat + ((long long)b * HW + p) * C;
    float* gf = g_feat + ((long long)b * HW + p) * C;
    for (int c = tid; c < C; c += 256) {
        const float gp = g_pooled ? g_pooled[(long long)b * C + c] : 0.f, gm = g_mean ? g_mean[(long long)b * C + c] / HW : 0.f;
        const float v = gp * ap / S + gm;
        gf[c] = accumulate ? gf[c] + v : v;
    }
    if (!g_logit) return;
    float part = 0.f;                                // channel partials per thread, combined in a fixed tree
    if (g_pooled)
        for (int c = tid; c < C; c += 256) part = fmaf(g_pooled[(long long)b * C + c], f[c] - pooled[(long long)b * C + c], part);
    s_red[tid] = part;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) s_red[tid] += s_red[tid + w]; __syncthreads(); }
    if (tid == 0) g_logit[(long long)b * HW + p] = s_red[0] / S * ap * (1.f - ap);
}

// ------------------------------------------------------------------------------------------------------------------ bone_proj backward
// Forward (models/dir.py:148-174), per sample, hand, bone k = (parent a, child b), pixel p = (x + 0.5, y + 0.5):
//   A = (uv_a + 1) / 2 * S, B likewise;  mask = lineseg_dist(p, A, B) < distance (no gradient; recomputed with the forward's bone_weights);
//   d_a = |p - A + 1e-6|, d_b = |p - B + 1e-6| (F.pairwise_distance adds its eps to the difference);
//   w_a = 1 - d_a / (d_a + d_b), w_b = 1 - d_b / (d_a + d_b);   img[p, k, :] = mask (f_a w_a + f_b w_b)
// Backward from g img [B,S,S,hands*20*64] (NHWC: channel = (hand*20 + k)*64 + c):
//   g f_a[c] = sum_p mask w_a g[p,c],  g f_b[c] = sum_p mask w_b g[p,c]            -> scattered to the 21 joints (index_select backward)
//   g w_a = <g[p,:], f_a>, g w_b = <g[p,:], f_b>;  g d_a = (g w_b - g w_a) d_b / (d_a + d_b)^2,  g d_b = (g w_a - g w_b) d_a / (d_a + d_b)^2
//   g A = - g d_a (p - A + eps) / d_a  (summed over p),  g uv_a = g A * S / 2
// One workgroup per (sample, hand, bone): 64 threads = channels; pixels in order.  Then one workgroup per (sample, hand) adds the bones'
// contributions into the joints in bone order (deterministic).
struct BoneBwdArgs {
    const float* uv[2]; const float* emb; const float* g_img; float distance;
    float* g_bone_f; float* g_bone_uv;               // scratch [B][hands][20][2 ends][64] and [B][hands][20][2 ends][2]
    float* g_emb; float* g_uv[2];
    int B, S, hands, img_cs, img_co;
};
__constant__ int kBoneParent[20] = {0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19};
__constant__ int kBoneChild[20] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20};

__global__ __launch_bounds__(64) void bone_bwd_kernel(BoneBwdArgs a) {
#pragma clang fp contract(off)
    const int k = blockIdx.x % 20, hb = blockIdx.x / 20, hand = hb % a.hands, b = hb / a.hands, c = threadIdx.x;
    const int S = a.S, ja = kBoneParent[k], jb = kBoneChild[k];
    const float* uv = a.uv[hand] + (long long)b * 42;
    const float Ax = (uv[2 * ja] + 1.f) / 2.f * S, Ay = (uv[2 * ja + 1] + 1.f) / 2.f * S;
    const float Bx = (uv[2 * jb] + 1.f) / 2.f * S, By = (uv[2 * jb + 1] + 1.f) / 2.f * S;
    const float* emb = a.emb + ((long long)b * 42 + hand * 21) * 64;
    const float fa = emb[ja * 64 + c], fb = emb[jb * 64 + c];
    float gfa = 0.f, gfb = 0.f, gAx = 0.f, gAy = 0.f, gBx = 0.f, gBy = 0.f;
    const float* gbase = a.g_img + (long long)b * S * S * a.img_cs + a.img_co + (hand * 20 + k) * 64 + c;
    auto pixel = [&](int p, float g) {               // one pixel inside the bone's mask: the sums in pixel order
        const int y = p / S, x = p - y * S;
        const float px = x + 0.5f, py = y + 0.5f;
        float wa, wb; bool in;
        dir::bone::bone_weights(px, py, Ax, Ay, Bx, By, a.distance, wa, wb, in);     // the forward's own mask and weights, bit for bit
        const float dax = px - Ax + 1e-6f, day = py - Ay + 1e-6f, dbx = px - Bx + 1e-6f, dby = py - By + 1e-6f;
        const float da = sqrtf(dax * dax + day * day), db = sqrtf(dbx * dbx + dby * dby), sum = da + db;
        gfa += wa * g; gfb += wb * g;
        float gwa = g * fa, gwb = g * fb;            // <g, f> over the 64 channels = one wave
        for (int o = 32; o > 0; o >>= 1) { gwa += __shfl_xor(gwa, o); gwb += __shfl_xor(gwb, o); }
        // w_a = 1 - d_a / (d_a + d_b): d w_a / d d_a = -d_b / sum^2, d w_a / d d_b = d_a / sum^2; w_b symmetric
        const float gda = (gwb - gwa) * db / (sum * sum), gdb = (gwa - gwb) * da / (sum * sum);
        gAx -= gda * dax / da; gAy -= gda * day / da;
        gBx -= gdb * dbx / db; gBy -= gdb * dby / db;
    };
    // The pixels inside the mask (wave-uniform: it does not depend on the channel), compacted IN PIXEL ORDER into LDS by ballots, then their
    // gradients loaded eight at a time: the loop used to find each pixel and wait for its load in turn -- 100-200 dependent round trips, 166 us.
    __shared__ unsigned short s_list[1024];
    const int npix = S * S;
    if (npix <= 1024) {
        int n = 0;
        for (int p0 = 0; p0 < npix; p0 += 64) {
            const int p = p0 + c;
            bool in = false;
            if (p < npix) {
                const int y = p / S, x = p - y * S;
                float wa, wb;
                dir::bone::bone_weights(x + 0.5f, y + 0.5f, Ax, Ay, Bx, By, a.distance, wa, wb, in);
            }
            const unsigned long long m = __ballot(in);
            if (in) s_list[n + __popcll(m & ((1ull << c) - 1ull))] = (unsigned short)p;
            n += __popcll(m);
        }
        __syncthreads();
        int i = 0;
        for (; i + 8 <= n; i += 8) {
            int pp[8]; float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { pp[u] = s_list[i + u]; g[u] = gbase[(long long)pp[u] * a.img_cs]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) pixel(pp[u], g[u]);
        }
        for (; i < n; ++i) { const int p = s_list[i]; pixel(p, gbase[(long long)p * a.img_cs]); }
    } else {
        for (int p = 0; p < npix; ++p) {
            const int y = p / S, x = p - y * S;
            float wa, wb; bool in;
            dir::bone::bone_weights(x + 0.5f, y + 0.5f, Ax, Ay, Bx, By, a.distance, wa, wb, in);
            if (in) pixel(p, gbase[(long long)p * a.img_cs]);
        }
    }
    float* gf = a.g_bone_f + ((((long long)b * a.hands + hand) * 20 + k) * 2) * 64;
    gf[c] = gfa; gf[64 + c] = gfb;
    if (c == 0) {
        float* gu = a.g_bone_uv + ((((long long)b * a.hands + hand) * 20 + k) * 2) * 2;
        gu[0] = gAx * S / 2.f; gu[1] = gAy * S / 2.f; gu[2] = gBx * S / 2.f; gu[3] = gBy * S / 2.f;
    }
}
__global__ __launch_bounds__(64) void bone_bwd_scatter_kernel(BoneBwdArgs a) {
    const int hand = blockIdx.x % a.hands, b = blockIdx.x / a.hands, c = threadIdx.x;
    const float* gf = a.g_bone_f + (((long long)b * a.hands + hand) * 20) * 2 * 64;
    const float* gu = a.g_bone_uv + (((long long)b * a.hands + hand) * 20) * 2 * 2;
    for (int j = 0; j < 21; ++j) {
        float s = 0.f, ux = 0.f, uy = 0.f;
        for (int k = 0; k < 20; ++k) {
            if (kBoneParent[k] == j) { s += gf[(k * 2) * 64 + c]; ux += gu[(k * 2) * 2]; uy += gu[(k * 2) * 2 + 1]; }
            if (kBoneChild[k] == j) { s += gf[(k * 2 + 1) * 64 + c]; ux += gu[(k * 2 + 1) * 2]; uy += gu[(k * 2 + 1) * 2 + 1]; }
        }
        a.g_emb[(((long long)b * 42) + hand * 21 + j) * 64 + c] = s;
        if (c == 0 && a.g_uv[hand]) { a.g_uv[hand][(long long)b * 42 + 2 * j] = ux; a.g_uv[hand][(long long)b * 42 + 2 * j + 1] = uy; }
    }
}

}  // namespace

extern "C" int dir_maxpool3x3s2_backward(const float* x, const float* gy, float* gx, int B, int H, int W, int C, void* stream) {
    using namespace dir;
    DIR_REQUIRE(x && gy && gx && B > 0 && H > 0 && W > 0 && C > 0, "dir_maxpool3x3s2_backward: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long n = (long long)B * H * W * C;
    static const int per_pixel = getenv("DIR_MAXPOOL_BWD_PER_PIXEL") ? atoi(getenv("DIR_MAXPOOL_BWD_PER_PIXEL")) : 0;      // A/B aid: the round-4 kernel (same bits)
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx) & 15) == 0 && !per_pixel) {
        const long long nb = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
        DIR_LAUNCH(maxpool_bwd4_block_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, B, H, W, C, Ho, Wo);
    } else if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx) & 15) == 0)
        DIR_LAUNCH(maxpool_bwd4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, B, H, W, C, Ho, Wo);
    else
        DIR_LAUNCH(maxpool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, B, H, W, C, Ho, Wo);
    return check_launch("dir_maxpool3x3s2_backward");
}
extern "C" int dir_upsample2x_bilinear_backward(const float* gy, float* gx, int B, int H, int W, int C, int gy_cstride, int gy_coff, void* stream) {
    using namespace dir;
    DIR_REQUIRE(gy && gx && B > 0 && H > 0 && W > 0 && C > 0, "dir_upsample2x_bilinear_backward: bad arguments");
    const long long n = (long long)B * H * W * C;
    DIR_LAUNCH(upsample_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, gx, B, H, W, C, gy_cstride ? gy_cstride : C, gy_coff);
    return check_launch("dir_upsample2x_bilinear_backward");
}
extern "C" int dir_attn_pool_forward(const float* feat, const float* logit, float* attn, float* pooled, float* mean, int B, int HW, int C, void* stream) {
    using namespace dir;
    DIR_REQUIRE(feat && logit && B > 0 && HW > 0 && HW <= 8192 && C > 0, "dir_attn_pool_forward: bad arguments (HW <= 8192)");
    DIR_LAUNCH(attn_pool_fwd_kernel, dim3(B), dim3(256), (size_t)HW * 4, (hipStream_t)stream, feat, logit, attn, pooled, mean, HW, C);
    return check_launch("dir_attn_pool_forward");
}
extern "C" int dir_attn_pool_backward(const float* feat, const float* attn, const float* pooled, const float* g_pooled, const float* g_mean, float* g_feat,
                                      float* g_logit, int B, int HW, int C, int accumulate, void* stream) {
    using namespace dir;
    DIR_REQUIRE(feat && attn && g_feat && B > 0 && HW > 0 && HW <= 8192 && C > 0 && (!g_logit || !g_pooled || pooled), "dir_attn_pool_backward: bad arguments");
    DIR_LAUNCH(attn_pool_bwd_kernel, dim3((unsigned)B * (unsigned)HW), dim3(256), (size_t)HW * 4, (hipStream_t)stream, feat, attn, pooled, g_pooled, g_mean, g_feat, g_logit, HW, C, accumulate);
    return check_launch("dir_attn_pool_backward");
}
extern "C" long long dir_bone_proj_backward_scratch_bytes(int B, int hands) { return B > 0 && hands > 0 ? (long long)B * hands * 20 * 2 * (64 + 2) * 4 : -1; }
extern "C" int dir_bone_proj_backward(const float* const* uv_lr, const float* emb, const float* g_img, int img_cstride, int img_coff, float distance,
                                      float* g_emb, float* const* g_uv_lr, float* scratch, int B, int S, int hands, void* stream) {
    using namespace dir;
    DIR_REQUIRE(uv_lr && emb && g_img && g_emb && scratch && B > 0 && S > 0 && (hands == 1 || hands == 2), "dir_bone_proj_backward: bad arguments");
    BoneBwdArgs a{};
    for (int h = 0; h < hands; ++h) { DIR_REQUIRE(uv_lr[h], "dir_bone_proj_backward: null uv"); a.uv[h] = uv_lr[h]; a.g_uv[h] = g_uv_lr ? g_uv_lr[h] : nullptr; }
    a.emb = emb; a.g_img = g_img; a.distance = distance; a.g_emb = g_emb; a.B = B; a.S = S; a.hands = hands;
    a.img_cs = img_cstride ? img_cstride : hands * 20 * 64; a.img_co = img_coff;
    a.g_bone_f = scratch; a.g_bone_uv = scratch + (long long)B * hands * 20 * 2 * 64;
    hipStream_t s = (hipStream_t)stream;
    DIR_LAUNCH(bone_bwd_kernel, dim3(B * hands * 20), dim3(64), 0, s, a);
    DIR_LAUNCH(bone_bwd_scatter_kernel, dim3(B * hands), dim3(64), 0, s, a);
    return check_launch("dir_bone_proj_backward");
}
