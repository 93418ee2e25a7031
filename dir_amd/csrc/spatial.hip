// HBM-bound spatial kernels of the DIR path (NHWC feature maps, fp32 or bf16 storage, fp32 math):
//   dir_stem_prep            NCHW fp32 image -> zero-padded NHWC4 so the 7x7/s2 stem becomes an implicit GEMM
//   dir_maxpool3x3s2         models/backbone/resnet.py:247 (MaxPool2d(3,2,1))
//   dir_upsample2x_bilinear  models/dir.py:392,398,442,459 (nn.Upsample(scale_factor=2,'bilinear'), align_corners=False),
//                            written straight into a channel slice of the concat buffer
//   dir_init_head_forward    models/dir.py:263-270 (1x1 conv -> sigmoid attention, attention-weighted pooling, Linears)
//   dir_bone_proj_forward    models/dir.py:132-174 (bone_proj / lineseg_dists) for both hands, NHWC [B,S,S,2560]
#include "bone_common.h"

namespace {

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
// f16 STORAGE (DIR_DT_F16, round 5; csrc/conv_common.h): the same 2-byte layouts, IEEE binary16, stores clamped to +-65504
struct f16s_t { unsigned short u; };
__device__ __forceinline__ float h2f(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned short f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f)); }
template <> __device__ __forceinline__ float ld<f16s_t>(const f16s_t* p) { return h2f(p->u); }
template <> __device__ __forceinline__ void st<f16s_t>(f16s_t* p, float v) { p->u = f2h(v); }

// 16-byte vectors of T
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = bf2f((bf16_t)(u[e] & 0xffffu)); v[2 * e + 1] = bf2f((bf16_t)(u[e] >> 16)); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (uint32_t)f2bf(v[2 * e]) | ((uint32_t)f2bf(v[2 * e + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

template <> struct Vec<f16s_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const f16s_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = h2f((unsigned short)(u[e] & 0xffffu)); v[2 * e + 1] = h2f((unsigned short)(u[e] >> 16)); }
    }
    static __device__ __forceinline__ void store(f16s_t* p, const float (&v)[8]) {
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (uint32_t)f2h(v[2 * e]) | ((uint32_t)f2h(v[2 * e + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(u[0], u[1], u[2], u[3]);
    }
};

// ------------------------------------------------------------------------------------------ stem prep
template <typename T>
__global__ void stem_prep_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int Hp,
                                 int Wp, int pad) {
    // one thread per padded pixel; writes 4 channels (RGB + 0)
    const long long n = (long long)B * Hp * Wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const int yp = (int)((i / Wp) % Hp);
        const int b = (int)(i / ((long long)Wp * Hp));
        const int x = xp - pad, y = yp - pad;
        float v[3] = {0.f, 0.f, 0.f};
        if (x >= 0 && x < W && y >= 0 && y < H) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = img[((long long)(b * 3 + c) * H + y) * W + x];
        }
        T* o = out + i * 4;
        st<T>(o, v[0]); st<T>(o + 1, v[1]); st<T>(o + 2, v[2]); st<T>(o + 3, 0.f);
    }
}

// Space-to-depth staging of the stem input: out[b][Y][X][16], channel (dy*2 + dx)*4 + c = img[b][c][2Y - 4 + dy][2X - 4 + dx]
// (zero outside the image and for c = 3).  The 7x7 stride-2 pad-3 stem convolution (models/backbone/resnet.py:244) becomes
// a 4x4 stride-1 convolution over 2x2 pixel blocks: output (oy, ox) reads blocks Y = oy..oy+3, X = ox..ox+3, and one
// K-slab of the implicit GEMM is the 128-byte window of 4 blocks x 16 channels of one block row -- K = 4 x 64 = 256
// instead of the 7 x 64 = 448 of the row-window formulation over single pixels (147 of them non-zero either way).
template <typename T>
__global__ void stem_prep_s2d_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int Hs, int Ws) {
    const long long n = (long long)B * Hs * Ws * 4;                     // one thread per (block, dy*2+dx)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        long long p = i >> 2;
        const int X = (int)(p % Ws); p /= Ws;
        const int Y = (int)(p % Hs);
        const int b = (int)(p / Hs);
        const int y = 2 * Y - 4 + (q >> 1), x = 2 * X - 4 + (q & 1);
        float v[3] = {0.f, 0.f, 0.f};
        if (x >= 0 && x < W && y >= 0 && y < H) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = img[((long long)(b * 3 + c) * H + y) * W + x];
        }
        T* o = out + i * 4;
        st<T>(o, v[0]); st<T>(o + 1, v[1]); st<T>(o + 2, v[2]); st<T>(o + 3, 0.f);
    }
}

// ---- SURVEY 8f rank 3 (tensor side of the input pipeline): apps/eval.py:59-61 == dataset/interhand.py:223-225
//      uint8 BGR HWC -> RGB, / 255, (t - mean) / std, in the reference's fp32 operation order (bit-identical to torch on CPU)
__device__ __forceinline__ float norm_px(unsigned char v, float mean, float stdv) {
#pragma clang fp contract(off)
    return ((float)v / 255.f - mean) / stdv;
}
struct NormArgs { float mean[3], stdv[3]; };

__global__ void image_normalize_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int B, int H, int W, NormArgs nm) {
    const long long n = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)H * W), p = i - b * H * W;
        const unsigned char* px = img + i * 3;                                  // B, G, R
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(b * 3 + c) * H * W + p] = norm_px(px[2 - c], nm.mean[c], nm.stdv[c]);
    }
}

// the same, fused with the space-to-depth staging of the stem (no fp32 NCHW image in HBM)
template <typename T>
__global__ void stem_prep_s2d_u8_kernel(const unsigned char* __restrict__ img, T* __restrict__ out, int B, int H, int W, int Hs, int Ws,
                                        NormArgs nm) {
    const long long n = (long long)B * Hs * Ws * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        long long p = i >> 2;
        const int X = (int)(p % Ws); p /= Ws;
        const int Y = (int)(p % Hs);
        const int b = (int)(p / Hs);
        const int y = 2 * Y - 4 + (q >> 1), x = 2 * X - 4 + (q & 1);
        float v[3] = {0.f, 0.f, 0.f};
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const unsigned char* px = img + (((long long)b * H + y) * W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = norm_px(px[2 - c], nm.mean[c], nm.stdv[c]);
        }
        T* o = out + i * 4;
        st<T>(o, v[0]); st<T>(o + 1, v[1]); st<T>(o + 2, v[2]); st<T>(o + 3, 0.f);
    }
}

// ------------------------------------------------------------------------------------------ maxpool
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int VN = Vec<T>::N;
    const int CV = C / VN;
    const long long n = (long long)B * Ho * Wo * CV;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VN;
        long long p = i / CV;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float m[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float v[VN];
                Vec<T>::load(x + ((long long)(b * H + iy) * W + ix) * C + c, v);
#pragma unroll
                for (int e = 0; e < VN; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        Vec<T>::store(y + ((long long)(b * Ho + oy) * Wo + ox) * C + c, m);
    }
}

// ------------------------------------------------------------------------------------------ add (nearest-upsampled) -- HRNet fuse layers
// acc[b, y, x, c] = act(acc[b, y, x, c] + src[b, y / f, x / f, c]): the sum of an HRNet fuse layer, one term at a time (f = 1: a plain add)
template <typename T>
__global__ __launch_bounds__(256) void add_upsampled_kernel(T* __restrict__ acc, const T* __restrict__ src, int B, int H, int W, int C, int f, int relu) {
    constexpr int VN = Vec<T>::N;
    const long long n = (long long)B * H * W * (C / VN);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % (C / VN));
        long long p = i / (C / VN);
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float a[VN], s[VN];
        T* ap = acc + (((long long)b * H + y) * W + x) * C + cv * VN;
        Vec<T>::load(ap, a);
        Vec<T>::load(src + (((long long)b * (H / f) + y / f) * (W / f) + x / f) * C + cv * VN, s);
#pragma unroll
        for (int e = 0; e < VN; ++e) { a[e] += s[e]; if (relu) a[e] = fmaxf(a[e], 0.f); }
        Vec<T>::store(ap, a);
    }
}

// round 5: the whole sum of an HRNet fuse row in ONE pass -- out = act(base + sum_t nearest_upsample(src_t, f_t)), fp32 sum, one rounding.
// The term-by-term form above makes 2 passes over the row's (largest) map per term plus a copy for the first one: 8 passes for the
// full-resolution row of a four-branch module; this is 2.  out may be base (in place).
struct FuseSumArgs { const void* src[4]; int f[4]; int nsrc; };
template <typename T>
__global__ __launch_bounds__(256) void fuse_sum_kernel(T* out, const T* base, FuseSumArgs fs, int B, int H, int W, int C, int relu) {
    constexpr int VN = Vec<T>::N;
    const long long n = (long long)B * H * W * (C / VN);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % (C / VN));
        long long p = i / (C / VN);
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H);
        const int b = (int)(p / H);
        float a[VN], s[4][VN];
        const long long o = (((long long)b * H + y) * W + x) * C + cv * VN;
        Vec<T>::load(base + o, a);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < fs.nsrc) {
                const int f = fs.f[t];
                Vec<T>::load((const T*)fs.src[t] + (((long long)b * (H / f) + y / f) * (W / f) + x / f) * C + cv * VN, s[t]);
            }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < fs.nsrc)
#pragma unroll
                for (int e = 0; e < VN; ++e) a[e] += s[t][e];
        if (relu)
#pragma unroll
            for (int e = 0; e < VN; ++e) a[e] = fmaxf(a[e], 0.f);
        Vec<T>::store(out + o, a);
    }
}

// ------------------------------------------------------------------------------------------ upsample
// One thread = one INPUT pixel x one 16-byte channel vector -> its 2x2 output quad: the quad's four bilinear footprints lie inside the
// pixel's 3x3 neighbourhood, so 9 vector loads (clamped addresses, all independent) serve 4 output vectors instead of 16, and the
// index arithmetic is 32-bit.  Every output is computed with the reference's own expressions (ATen area_pixel_compute_source_index,
// align_corners=False: src = max((dst + 0.5) * 0.5 - 0.5, 0)) -- bit-identical to the one-thread-per-output form it replaces.
template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int ocs, int oco) {
    constexpr int VN = Vec<T>::N;
    const unsigned CV = C / VN, n = (unsigned)B * H * W * CV;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const unsigned cv = i % CV, p = i / CV;
    const int ix = (int)(p % (unsigned)W), q = (int)(p / (unsigned)W), iy = q % H, b = q / H;
    const int c = (int)cv * VN, Wo = 2 * W;
    const T* base = x + (long long)b * H * W * C + c;
    float v[3][3][VN];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            Vec<T>::load(base + ((long long)min(max(iy - 1 + r, 0), H - 1) * W + min(max(ix - 1 + t, 0), W - 1)) * C, v[r][t]);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int oy = 2 * iy + dy;
        const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.f);
        const float ly = sy - (int)sy;                  // 0.75 (rows iy-1, iy) for dy = 0, 0.25 (rows iy, iy+1) for dy = 1; 0 on a clamped border
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ox = 2 * ix + dx;
            const float sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.f);
            const float lx = sx - (int)sx;
            // footprint rows (dy, dy + 1) and columns (dx, dx + 1) of the loaded 3 x 3: on a clamped border the reference's second
            // row / column carries weight 0 (first output row) or is the same clamped pixel (last one) -- the same values either way
            float o[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float top = v[dy][dx][e] * (1.f - lx) + v[dy][dx + 1][e] * lx, bot = v[dy + 1][dx][e] * (1.f - lx) + v[dy + 1][dx + 1][e] * lx;
                o[e] = top * (1.f - ly) + bot * ly;
            }
            Vec<T>::store(y + ((long long)(b * 2 * H + oy) * Wo + ox) * ocs + oco + c, o);
        }
    }
}

// ------------------------------------------------------------------------------------------ init head
struct InitHeadArgs {
    dir_init_head_params p;
    const void* c4; const void* h[2];
    float* para[2]; float* offset;
    int HW, C, Ch, hcs;
    long long* stamps;     // DIR_STAMPS=init_head (tuning aid, else NULL)
};

template <typename T>
__global__ __launch_bounds__(512) void init_head_kernel(InitHeadArgs a) {
    // one 512-thread block per sample.  LDS: attention weights [2][HW], pooled features [3][C] (left, right, mean)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int VN = Vec<T>::N;
    float* s_attn = sm;                 // [2][HW]
    float* s_feat = sm + 2 * a.HW;      // [3][C]
    __shared__ float s_den[2];
    __shared__ float s_part[4][128];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = a.HW, C = a.C, Ch = a.Ch;
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && b == 0 && tid == 0 && nstamp < dir::MAX_STAMPS) a.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    const T* c4 = (const T*)a.c4 + (long long)b * HW * C;
    // 1) attention logits: 1x1 conv Ch -> 1, sigmoid (models/dir.py:231-232); one wave per (hand, pixel) dot product, lanes over
    //    channels (1 KB contiguous per load instruction).  Fast path (Ch = 2 x 64 x VN, 2 HW a multiple of 64): a wave owns 2 HW / 8
    //    consecutive outputs of ONE hand, keeps that hand's weights in registers and issues the loads of sixteen outputs (32 per
    //    lane) together -- one memory round trip per wave instead of eight; DPP wave sums.  Per-lane fmaf order unchanged (bit-identical).
    if (Ch == 2 * 64 * VN && (2 * HW) % 128 == 0) {
        const int per = 2 * HW / 8, o_first = wave * per, s = o_first / HW;
        float wv[2][VN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < VN; ++e) wv[kk][e] = a.p.attn_w[s][kk * 64 * VN + lane * VN + e];
        const float bias = a.p.attn_b[s];
        for (int o0 = o_first; o0 < o_first + per; o0 += 16) {
            float v[16][2][VN];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int px = o0 + u - s * HW;
                const T* hp = (const T*)a.h[s] + ((long long)b * HW + px) * a.hcs + lane * VN;
                Vec<T>::load(hp, v[u][0]);
                Vec<T>::load(hp + 64 * VN, v[u][1]);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc = fmaf(v[u][kk][e], wv[kk][e], acc);
                const float t = dir::wave_sum_dpp(acc);
                if (lane == 0) s_attn[o0 + u] = 1.f / (1.f + expf(-(t + bias)));
            }
        }
    } else {
        for (int o0 = wave * 4; o0 < 2 * HW; o0 += 32) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = lane * VN; k < Ch; k += 64 * VN) {
                float v[4][VN], wv[4][VN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = min(o0 + u, 2 * HW - 1), s = o / HW, px = o - s * HW;
                    Vec<T>::load((const T*)a.h[s] + ((long long)b * HW + px) * a.hcs + k, v[u]);
#pragma unroll
                    for (int e = 0; e < VN; ++e) wv[u][e] = a.p.attn_w[s][k + e];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[u] = fmaf(v[u][e], wv[u][e], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = dir::wave_sum(acc[u]);
                const int o = o0 + u;
                if (lane == 0 && o < 2 * HW) s_attn[o] = 1.f / (1.f + expf(-(t + a.p.attn_b[o / HW])));
            }
        }
    }
    __syncthreads(); stamp();
    if (tid < 2) {
        float d = 0.f;
        for (int px = 0; px < HW; ++px) d += s_attn[tid * HW + px];
        s_den[tid] = d + 1e-8f;                                     // models/dir.py:264
    }
    __syncthreads(); stamp();
    // 2) attention-weighted pooling and the plain spatial mean (models/dir.py:264-268): VN channels per thread
    for (int c = tid * VN; c < C; c += 512 * VN) {
        float fl[VN], fr[VN], fm[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) { fl[e] = 0.f; fr[e] = 0.f; fm[e] = 0.f; }
#pragma unroll 16
        for (int px = 0; px < HW; ++px) {
            float v[VN];
            Vec<T>::load(c4 + (long long)px * C + c, v);
            const float al = s_attn[px], ar = s_attn[HW + px];
#pragma unroll
            for (int e = 0; e < VN; ++e) { fl[e] = fmaf(v[e], al, fl[e]); fr[e] = fmaf(v[e], ar, fr[e]); fm[e] += v[e]; }
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            s_feat[c + e] = fl[e] / s_den[0];
            s_feat[C + c + e] = fr[e] / s_den[1];
            s_feat[2 * C + c + e] = fm[e] / (float)HW;
        }
    }
    __syncthreads(); stamp();
    // 3) Linears (models/dir.py:268-270).  mano_{left,right}: thread = (output o of 128, K quarter), k-major weights,
    //    16 loads in flight; offset (3 outputs): waves 0..2, lane-strided.
    {
        const int o = tid & 127, ks = tid >> 7, kq = C / 4;
        const float* f = s_feat + (o >> 6) * C + ks * kq;
        const float* w = a.p.mano_wt + (long long)(ks * kq) * 128 + o;
        float acc = 0.f;
        int k0 = 0;
        for (; k0 + 64 <= kq; k0 += 64) {                           // 64 loads in flight per thread (was 16: 32 L2 round trips at C = 2048)
            float wv[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) wv[u] = w[(k0 + u) * 128];
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = fmaf(f[k0 + u], wv[u], acc);
        }
        for (; k0 < kq; k0 += 16) {                                 // kq % 16 == 0 (C % 64 == 0)
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w[(k0 + u) * 128];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fmaf(f[k0 + u], wv[u], acc);
        }
        s_part[ks][o] = acc;
    }
    if (wave < 3) {
        const float* w = a.p.off_w + (long long)wave * C;
        const float* f = s_feat + 2 * C;
        float acc = 0.f;
#pragma unroll 8
        for (int k = lane; k < C; k += 64) acc = fmaf(f[k], w[k], acc);
        acc = dir::wave_sum(acc);
        if (lane == 0) a.offset[(long long)b * 3 + wave] = acc + a.p.off_b[wave];
    }
    __syncthreads(); stamp();
    if (tid < 128) {
        const int s = tid >> 6, oo = tid & 63;
        a.para[s][(long long)b * 64 + oo] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid] + a.p.mano_b[s][oo];
    }
    stamp();
}

// ------------------------------------------------------------------------------------------ bone_proj
struct BoneArgs {
    const float* uv[2]; const float* emb; void* out; float* vis; int* bbox;
    int B, S; float distance;
};

using dir::bone::bone_weights;
using dir::bone::kChild;
using dir::bone::kParent;

template <typename T>
__global__ __launch_bounds__(256) void bone_proj_kernel(BoneArgs a) {
    // one block per (sample, image row).  LDS: token features [42][64], per-(x,hand,bone) weights and mask.
    extern __shared__ float sm[];
    const int S = a.S;
    float* s_emb = sm;                       // [42*64]
    float* s_wa = s_emb + 42 * 64;           // [S*40]
    float* s_wb = s_wa + S * 40;             // [S*40]
    float* s_uv = s_wb + S * 40;             // [42*2] pixel units
    unsigned char* s_in = reinterpret_cast<unsigned char*>(s_uv + 84);   // [S*40] capsule mask
    const int b = blockIdx.x / S, y = blockIdx.x - b * S, tid = threadIdx.x;
    for (int i = tid; i < 42 * 64; i += 256) s_emb[i] = a.emb[(long long)b * 42 * 64 + i];
    if (tid < 84) {
#pragma clang fp contract(off)
        const int hand = tid / 42, r = tid - hand * 42;
        const float v = a.uv[hand][(long long)b * 42 + r];
        s_uv[tid] = (v + 1.f) / 2.f * (float)S;                      // models/dir.py:150
    }
    __syncthreads();
    if (a.bbox && y == 0 && tid < 40) {
        // conservative pixel bounding box of the capsule of (hand, bone) = channel group tid of the NHWC output:
        // a pixel centre (x+.5, y+.5) can only be inside if it is within `distance` of the segment's own box.
        const int hand = tid / 20, bone = tid - hand * 20;
        const float* uv = s_uv + hand * 42;
        const float ax = uv[2 * kParent[bone]], ay = uv[2 * kParent[bone] + 1];
        const float bx = uv[2 * kChild[bone]], by = uv[2 * kChild[bone] + 1];
        int* bb = a.bbox + ((long long)b * 40 + tid) * 4;
        const float d = a.distance;
        const float ylo = fminf(ay, by) - d - 0.5f, yhi = fmaxf(ay, by) + d - 0.5f;
        const float xlo = fminf(ax, bx) - d - 0.5f, xhi = fmaxf(ax, bx) + d - 0.5f;
        const bool finite = (ax - ax == 0.f) && (ay - ay == 0.f) && (bx - bx == 0.f) && (by - by == 0.f);
        // NaN / inf joints give a NaN or inf distance for every pixel -> mask all false -> empty box
        if (!finite) { bb[0] = 1; bb[1] = 0; bb[2] = 1; bb[3] = 0; }
        else {
            const float fs = (float)(S - 1);
            bb[0] = (int)fminf(fmaxf(floorf(ylo), 0.f), fs + 1.f); bb[1] = (int)fmaxf(fminf(ceilf(yhi), fs), -1.f);
            bb[2] = (int)fminf(fmaxf(floorf(xlo), 0.f), fs + 1.f); bb[3] = (int)fmaxf(fminf(ceilf(xhi), fs), -1.f);
        }
    }
    const float py = (float)y + 0.5f;                                 // img_gird: (x+0.5, y+0.5), models/dir.py:66-70
    for (int i = tid; i < S * 40; i += 256) {
        const int x = i / 40, hb = i - x * 40, hand = hb / 20, bone = hb - hand * 20;
        const float* uv = s_uv + hand * 42;
        const int pa = kParent[bone], ch = kChild[bone];
        float wa, wb; bool in;
        bone_weights((float)x + 0.5f, py, uv[2 * pa], uv[2 * pa + 1], uv[2 * ch], uv[2 * ch + 1], a.distance, wa, wb, in);
        s_wa[i] = wa;
        s_wb[i] = wb;
        s_in[i] = in ? 1 : 0;
    }
    __syncthreads();
    // NHWC write: [b][y][x][hand*1280 + bone*64 + c], 8 channels per thread-item, fully coalesced
    T* orow = (T*)a.out + ((long long)(b * S + y) * S) * 2560;
    for (int i = tid; i < (a.out ? S * 40 * 8 : 0); i += 256) {
        const int c8 = i & 7, xhb = i >> 3;
        const int x = xhb / 40, hb = xhb - x * 40, hand = hb / 20, bone = hb - hand * 20;
        const float wa = s_wa[xhb], wb = s_wb[xhb];
        const bool masked = s_in[xhb] == 0;                            // torch.where(mask, v, 0), models/dir.py:172
        const float* fa = s_emb + (hand * 21 + kParent[bone]) * 64 + c8 * 8;
        const float* fb = s_emb + (hand * 21 + kChild[bone]) * 64 + c8 * 8;
        T* o = orow + (long long)x * 2560 + hb * 64 + c8 * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)
            v[e] = masked ? 0.f : fa[e] * wa + fb[e] * wb;             // models/dir.py:170-172
        }
        if constexpr (sizeof(T) == 2) Vec<T>::store(o, v);
        else {
            const float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
            Vec<float>::store(reinterpret_cast<float*>(o), lo);
            Vec<float>::store(reinterpret_cast<float*>(o) + 4, hi);
        }
    }
    if (a.vis) {
        // vis_img_feat = left + right, NCHW fp32 [B,1280,S,S] (models/dir.py:128,481): threads run along x
        float* vrow = a.vis + (long long)b * 1280 * S * S + (long long)y * S;
        const int S4 = S / 4;
        for (int i = tid; i < 1280 * S4; i += 256) {
            const int x4 = (i % S4) * 4, ch = i / S4, bone = ch >> 6, c = ch & 63;
            float out4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = 0.f;
#pragma unroll
                for (int hand = 0; hand < 2; ++hand) {
#pragma clang fp contract(off)
                    const int xhb = (x4 + q) * 40 + hand * 20 + bone;
                    const float wa = s_wa[xhb], wb = s_wb[xhb];
                    const float v = (s_in[xhb] == 0) ? 0.f
                                                     : s_emb[(hand * 21 + kParent[bone]) * 64 + c] * wa +
                                                           s_emb[(hand * 21 + kChild[bone]) * 64 + c] * wb;
                    acc = hand == 0 ? v : acc + v;
                }
                out4[q] = acc;
            }
            Vec<float>::store(vrow + (long long)ch * S * S + x4, out4);
        }
    }
}

// proj_feat only (models/dir.py:128,481): vis[b, bone*64 + c, y, x] = left + right, fp32 NCHW.  One workgroup per
// (sample, bone) owns the 64 consecutive channel planes of that bone (a contiguous 64 * S*S * 4-byte region): the
// per-pixel (mask * wa, mask * wb) of both hands are computed once into LDS and every plane is streamed out with
// 16-byte stores.  Same formulas and evaluation order as bone_proj_kernel (bit-identical values).
#ifndef DIR_VIS_NT
#define DIR_VIS_NT 0      // A/B switch: non-temporal stores of proj_feat (never re-read on the GPU): kernel 54 -> 69 us alone, step -0.3 % under overlap (noise): off
#endif
__global__ __launch_bounds__(256) void bone_vis_kernel(BoneArgs a) {
    extern __shared__ float sm[];
    const int S = a.S, hw = S * S;
    float* s_w = sm;                          // [2 hands][2 (wa, wb)][hw]; masked pixels hold NaN-free zeros + flag
    unsigned char* s_in = reinterpret_cast<unsigned char*>(s_w + 4 * hw);   // [2][hw]
    __shared__ float s_f[2][2][64];           // [hand][end][c]
    __shared__ float s_uv[2][4];              // [hand][ax, ay, bx, by] in pixel units
    const int b = blockIdx.x / 20, bone = blockIdx.x - b * 20, tid = threadIdx.x;
    if (tid < 8) {
#pragma clang fp contract(off)
        const int hand = tid >> 2, k = tid & 3, j = (k < 2) ? kParent[bone] : kChild[bone];
        const float v = a.uv[hand][(long long)b * 42 + 2 * j + (k & 1)];
        s_uv[hand][k] = (v + 1.f) / 2.f * (float)S;                   // models/dir.py:150
    }
    {
        const int hand = tid >> 7, end = (tid >> 6) & 1, c = tid & 63;
        s_f[hand][end][c] = a.emb[((long long)b * 42 + hand * 21 + (end ? kChild[bone] : kParent[bone])) * 64 + c];
    }
    __syncthreads();
    for (int i = tid; i < 2 * hw; i += 256) {
        const int hand = i / hw, p = i - hand * hw, y = p / S, x = p - y * S;
        float wa, wb;
        bool in;
        bone_weights((float)x + 0.5f, (float)y + 0.5f, s_uv[hand][0], s_uv[hand][1], s_uv[hand][2], s_uv[hand][3], a.distance, wa, wb, in);
        s_w[(hand * 2) * hw + p] = wa;
        s_w[(hand * 2 + 1) * hw + p] = wb;
        s_in[i] = in ? 1 : 0;
    }
    __syncthreads();
    float* out = a.vis + ((long long)b * 1280 + bone * 64) * hw;
    const int Q = hw >> 2;                                            // pixel quads per plane
    if (Q <= 256 && 256 % Q == 0) {
        // S = 16 / 32: a thread owns ONE pixel quad for every channel it visits: its eight (mask, wa, wb) values live in registers and
        // the channel loop reads only the four broadcast feature values per plane from LDS (was ~10 LDS reads per output element); the
        // arithmetic -- two products and a sum per hand with contraction off, masked pixels exactly 0 -- is that of the loop below
        const int p4 = (tid % Q) * 4, cstep = 256 / Q;
        float wa[2][4], wb[2][4];
        bool in[2][4];
#pragma unroll
        for (int hand = 0; hand < 2; ++hand)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wa[hand][q] = s_w[(hand * 2) * hw + p4 + q];
                wb[hand][q] = s_w[(hand * 2 + 1) * hw + p4 + q];
                in[hand][q] = s_in[hand * hw + p4 + q] != 0;
            }
        for (int c = tid / Q; c < 64; c += cstep) {
            const float f00 = s_f[0][0][c], f01 = s_f[0][1][c], f10 = s_f[1][0][c], f11 = s_f[1][1][c];
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma clang fp contract(off)
                const float v0 = in[0][q] ? f00 * wa[0][q] + f01 * wb[0][q] : 0.f;
                const float v1 = in[1][q] ? f10 * wa[1][q] + f11 * wb[1][q] : 0.f;
                o[q] = v0 + v1;
            }
#if DIR_VIS_NT
            typedef float __attribute__((ext_vector_type(4))) f4_t;
            __builtin_nontemporal_store(f4_t{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4_t*>(out + (long long)c * hw + p4));
#else
            Vec<float>::store(out + (long long)c * hw + p4, o);
#endif
        }
        return;
    }
    for (int i = tid; i < 64 * hw / 4; i += 256) {
        const int c = i / (hw / 4), p4 = (i - c * (hw / 4)) * 4;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int hand = 0; hand < 2; ++hand) {
#pragma clang fp contract(off)
                const int p = p4 + q;
                const float v = s_in[hand * hw + p] == 0 ? 0.f
                                                         : s_f[hand][0][c] * s_w[(hand * 2) * hw + p] + s_f[hand][1][c] * s_w[(hand * 2 + 1) * hw + p];
                acc = hand == 0 ? v : acc + v;
            }
            o[q] = acc;
        }
        Vec<float>::store(out + (long long)c * hw + p4, o);
    }
}

inline int grid_for(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int dir_stem_prep(const float* img_nchw, void* out, int B, int H, int W, int Hp, int Wp, int pad,
                             int dtype, void* stream) {
    DIR_REQUIRE(img_nchw && out && B > 0 && H > 0 && W > 0 && Hp >= H + pad && Wp >= W + pad, "dir_stem_prep: bad args");
    const long long n = (long long)B * Hp * Wp;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((stem_prep_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (float*)out, B, H, W, Hp, Wp, pad);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((stem_prep_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (bf16_t*)out, B, H, W, Hp, Wp, pad);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((stem_prep_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (f16s_t*)out, B, H, W, Hp, Wp, pad);
    else DIR_REQUIRE(false, "dir_stem_prep: bad dtype");
    return dir::check_launch("dir_stem_prep");
}

extern "C" int dir_stem_prep_s2d(const float* img_nchw, void* out, int B, int H, int W, int Hs, int Ws, int dtype, void* stream) {
    DIR_REQUIRE(img_nchw && out && B > 0 && H > 0 && W > 0, "dir_stem_prep_s2d: bad args");
    DIR_REQUIRE(H % 2 == 0 && W % 2 == 0 && Hs >= H / 2 + 3 && Ws >= W / 2 + 3, "dir_stem_prep_s2d: need even H, W and Hs >= H/2+3, Ws >= W/2+3");
    const long long n = (long long)B * Hs * Ws * 4;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((stem_prep_s2d_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (float*)out, B, H, W, Hs, Ws);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((stem_prep_s2d_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (bf16_t*)out, B, H, W, Hs, Ws);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((stem_prep_s2d_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, img_nchw, (f16s_t*)out, B, H, W, Hs, Ws);
    else DIR_REQUIRE(false, "dir_stem_prep_s2d: bad dtype");
    return dir::check_launch("dir_stem_prep_s2d");
}

static int norm_args(const float* mean, const float* stdv, NormArgs* nm, const char* who) {
    DIR_REQUIRE(mean && stdv, "%s: null mean / std (host pointers to 3 floats)", who);
    for (int c = 0; c < 3; ++c) {
        DIR_REQUIRE(stdv[c] != 0.f, "%s: std[%d] is zero", who, c);
        nm->mean[c] = mean[c];
        nm->stdv[c] = stdv[c];
    }
    return DIR_OK;
}

extern "C" int dir_image_normalize_forward(const uint8_t* img_bgr_hwc, float* out_nchw, const float* mean_host, const float* std_host,
                                           int B, int H, int W, void* stream) {
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(img_bgr_hwc && out_nchw && B > 0 && H > 0 && W > 0, "dir_image_normalize_forward: bad args");
    NormArgs nm;
    if (int rc = norm_args(mean_host, std_host, &nm, "dir_image_normalize_forward")) return rc;
    DIR_LAUNCH(image_normalize_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, img_bgr_hwc, out_nchw, B, H,
                       W, nm);
    return dir::check_launch("dir_image_normalize_forward");
}

extern "C" int dir_stem_prep_s2d_u8(const uint8_t* img_bgr_hwc, void* out, const float* mean_host, const float* std_host, int B, int H, int W,
                                    int Hs, int Ws, int dtype, void* stream) {
    DIR_REQUIRE(img_bgr_hwc && out && B > 0 && H > 0 && W > 0, "dir_stem_prep_s2d_u8: bad args");
    DIR_REQUIRE(H % 2 == 0 && W % 2 == 0 && Hs >= H / 2 + 3 && Ws >= W / 2 + 3, "dir_stem_prep_s2d_u8: need even H, W and Hs >= H/2+3, Ws >= W/2+3");
    NormArgs nm;
    if (int rc = norm_args(mean_host, std_host, &nm, "dir_stem_prep_s2d_u8")) return rc;
    const long long n = (long long)B * Hs * Ws * 4;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((stem_prep_s2d_u8_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, img_bgr_hwc, (float*)out, B, H, W, Hs, Ws, nm);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((stem_prep_s2d_u8_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, img_bgr_hwc, (bf16_t*)out, B, H, W, Hs, Ws, nm);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((stem_prep_s2d_u8_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, img_bgr_hwc, (f16s_t*)out, B, H, W, Hs, Ws, nm);
    else DIR_REQUIRE(false, "dir_stem_prep_s2d_u8: bad dtype");
    return dir::check_launch("dir_stem_prep_s2d_u8");
}

extern "C" int dir_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    DIR_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "dir_maxpool3x3s2: bad args");
    DIR_REQUIRE(C % 8 == 0, "dir_maxpool3x3s2: C must be a multiple of 8 (16-byte channel vectors)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long n = (long long)B * Ho * Wo * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((maxpool_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C, Ho, Wo);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((maxpool_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((maxpool_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, (const f16s_t*)x, (f16s_t*)y, B, H, W, C, Ho, Wo);
    else DIR_REQUIRE(false, "dir_maxpool3x3s2: bad dtype");
    return dir::check_launch("dir_maxpool3x3s2");
}

extern "C" int dir_add_upsampled(void* acc, const void* src, int B, int H, int W, int C, int factor, int relu, int dtype, void* stream) {
    DIR_REQUIRE(acc && src && B > 0 && H > 0 && W > 0 && C > 0 && factor >= 1, "dir_add_upsampled: bad args");
    DIR_REQUIRE(C % 8 == 0 && H % factor == 0 && W % factor == 0, "dir_add_upsampled: C must be a multiple of 8, H and W multiples of the factor");
    const long long n = (long long)B * H * W * (C / (dtype == DIR_DT_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((add_upsampled_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (float*)acc, (const float*)src, B, H, W, C, factor, relu);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((add_upsampled_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)acc, (const bf16_t*)src, B, H, W, C, factor, relu);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((add_upsampled_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, (f16s_t*)acc, (const f16s_t*)src, B, H, W, C, factor, relu);
    else DIR_REQUIRE(false, "dir_add_upsampled: bad dtype");
    return dir::check_launch("dir_add_upsampled");
}

extern "C" int dir_fuse_sum(void* out, const void* base, const void* const* srcs, const int* factors, int nsrc, int B, int H, int W, int C, int relu, int dtype,
                            void* stream) {
    DIR_REQUIRE(out && base && B > 0 && H > 0 && W > 0 && C > 0 && nsrc >= 0 && nsrc <= 4 && (nsrc == 0 || (srcs && factors)), "dir_fuse_sum: bad args (at most 4 sources)");
    DIR_REQUIRE(C % 8 == 0, "dir_fuse_sum: C must be a multiple of 8");
    FuseSumArgs fs;
    for (int t = 0; t < 4; ++t) { fs.src[t] = nullptr; fs.f[t] = 1; }
    fs.nsrc = nsrc;
    for (int t = 0; t < nsrc; ++t) {
        DIR_REQUIRE(srcs[t] && factors[t] >= 1 && H % factors[t] == 0 && W % factors[t] == 0, "dir_fuse_sum: H and W must be multiples of every factor");
        fs.src[t] = srcs[t]; fs.f[t] = factors[t];
    }
    const long long n = (long long)B * H * W * (C / (dtype == DIR_DT_F32 ? 4 : 8));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DIR_DT_F32) DIR_LAUNCH((fuse_sum_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (float*)out, (const float*)base, fs, B, H, W, C, relu);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((fuse_sum_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)out, (const bf16_t*)base, fs, B, H, W, C, relu);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((fuse_sum_kernel<f16s_t>), dim3(grid_for(n)), dim3(256), 0, s, (f16s_t*)out, (const f16s_t*)base, fs, B, H, W, C, relu);
    else DIR_REQUIRE(false, "dir_fuse_sum: bad dtype");
    return dir::check_launch("dir_fuse_sum");
}

extern "C" int dir_upsample2x_bilinear(const void* x, void* y, int B, int H, int W, int C, int out_cstride,
                                       int out_coff, int dtype, void* stream) {
    DIR_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "dir_upsample2x_bilinear: bad args");
    const int ocs = out_cstride ? out_cstride : C;
    DIR_REQUIRE(C % 8 == 0 && ocs % 8 == 0 && out_coff % 8 == 0, "dir_upsample2x_bilinear: channel counts/offsets must be multiples of 8");
    const long long n = (long long)B * H * W * (C / (dtype == DIR_DT_F32 ? 4 : 8));        // one thread per input pixel and 16-byte channel vector
    DIR_REQUIRE(n < (1ll << 31), "dir_upsample2x_bilinear: too many elements");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == DIR_DT_F32) DIR_LAUNCH((upsample_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C, ocs, out_coff);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((upsample_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, ocs, out_coff);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((upsample_kernel<f16s_t>), grid, dim3(256), 0, s, (const f16s_t*)x, (f16s_t*)y, B, H, W, C, ocs, out_coff);
    else DIR_REQUIRE(false, "dir_upsample2x_bilinear: bad dtype");
    return dir::check_launch("dir_upsample2x_bilinear");
}

extern "C" int dir_init_head_forward(const dir_init_head_params* p, const void* c4, const void* h_left,
                                     const void* h_right, int h_cstride, float* para_left, float* para_right,
                                     float* offset, int B, int HW, int C, int Ch, int dtype, void* stream) {
    DIR_REQUIRE(p && c4 && h_left && h_right && para_left && para_right && offset, "dir_init_head_forward: null pointer");
    DIR_REQUIRE(B > 0 && HW > 0 && C > 0 && Ch > 0, "dir_init_head_forward: bad shape");
    InitHeadArgs a;
    a.p = *p; a.c4 = c4; a.h[0] = h_left; a.h[1] = h_right; a.para[0] = para_left; a.para[1] = para_right;
    a.offset = offset; a.HW = HW; a.C = C; a.Ch = Ch; a.hcs = h_cstride ? h_cstride : Ch;
    DIR_REQUIRE(a.hcs % 8 == 0, "dir_init_head_forward: h_cstride must be a multiple of 8");
    const size_t lds = (size_t)(2 * HW + 3 * C) * sizeof(float);
    DIR_REQUIRE(lds <= 60000, "dir_init_head_forward: feature too large for LDS");
    DIR_REQUIRE(C % 64 == 0 && Ch % 8 == 0 && (2 * HW) % 4 == 0 && p->mano_wt, "dir_init_head_forward: need C % 64 == 0, Ch % 8 == 0");
    hipStream_t s = (hipStream_t)stream;
    a.stamps = dir::stamps_begin("init_head");
    if (dtype == DIR_DT_F32) DIR_LAUNCH((init_head_kernel<float>), dim3(B), dim3(512), lds, s, a);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((init_head_kernel<bf16_t>), dim3(B), dim3(512), lds, s, a);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((init_head_kernel<f16s_t>), dim3(B), dim3(512), lds, s, a);
    else DIR_REQUIRE(false, "dir_init_head_forward: bad dtype");
    dir::stamps_end("init_head", a.stamps, s);
    return dir::check_launch("dir_init_head_forward");
}

extern "C" int dir_bone_proj_forward(const float* uv_left, const float* uv_right, const float* emb, void* out,
                                     float* vis_nchw, int32_t* group_bbox, int B, int S, float distance, int dtype,
                                     void* stream) {
    DIR_REQUIRE(uv_left && uv_right && emb && (out || vis_nchw), "dir_bone_proj_forward: null pointer (out and vis_nchw may not both be null)");
    DIR_REQUIRE(B > 0 && S > 0 && S <= 64 && S % 4 == 0, "dir_bone_proj_forward: bad shape (S must be a multiple of 4, <= 64)");
    BoneArgs a;
    a.uv[0] = uv_left; a.uv[1] = uv_right; a.emb = emb; a.out = out; a.vis = vis_nchw; a.bbox = group_bbox; a.B = B; a.S = S;
    a.distance = distance;
    const size_t lds = (size_t)(42 * 64 + 2 * S * 40 + 84) * sizeof(float) + (size_t)S * 40;
    hipStream_t s = (hipStream_t)stream;
    if (!out && S <= 32) {                             // proj_feat only: contiguous per-bone channel planes
        DIR_REQUIRE(group_bbox == nullptr, "dir_bone_proj_forward: group_bbox needs the NHWC output");
        const size_t vlds = (size_t)4 * S * S * sizeof(float) + (size_t)2 * S * S;
        DIR_LAUNCH(bone_vis_kernel, dim3(B * 20), dim3(256), vlds, s, a);
        return dir::check_launch("dir_bone_proj_forward");
    }
    if (dtype == DIR_DT_F32) DIR_LAUNCH((bone_proj_kernel<float>), dim3(B * S), dim3(256), lds, s, a);
    else if (dtype == DIR_DT_BF16) DIR_LAUNCH((bone_proj_kernel<bf16_t>), dim3(B * S), dim3(256), lds, s, a);
    else if (dtype == DIR_DT_F16) DIR_LAUNCH((bone_proj_kernel<f16s_t>), dim3(B * S), dim3(256), lds, s, a);
    else DIR_REQUIRE(false, "dir_bone_proj_forward: bad dtype");
    return dir::check_launch("dir_bone_proj_forward");
}
