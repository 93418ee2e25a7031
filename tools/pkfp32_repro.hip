// Standalone reproducer (no torch) for "Packed FP32 beside another kernel" (DESIGN.md): a VICTIM kernel made of v_pk_fma_f32 loops that
// checks itself against the same arithmetic on scalar v_fma_f32, on one stream; an AGGRESSOR on another stream -- one of libdir_hip.so's
// convolution kernels (linked, called through the C ABI).  Round 1 singled out the 64x128 tile of conv_igemm_kernel on the 3-buffer
// LDS-DMA ring (DIR_CONV_VARIANT 19, needs DIR_RING_64x128=1) running the decoder's dual-source 1x1 (M = 16384, N = 256, K = 128 + 1024).
//
//   hipcc --offload-arch=gfx950 -O2 tools/pkfp32_repro.hip -Iinclude -Ldir_amd/lib -ldir_hip -Wl,-rpath,$PWD/dir_amd/lib -o /tmp/pkrepro
//   DIR_RING_64x128=1 /tmp/pkrepro [variant=19] [rounds=200] [victim=0 pk_fma registers | 1 pk_fma memory-fed | 2 ds_read2_b64 | 3 ds_read2_b32 | 4 ds_read_b128 | 5 the library's MANO launch | 6 / 7 / 8 global_load_dword / x2 / x4]
//   (victim 5 against the packed-FP32 investigation build: -l:libdir_hip_pk.so after `DIR_PACKED_FP32=1 python -m dir_amd.build`)
//
// Prints, per configuration, how many victim launches saw a low-half / high-half mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>

#include "dir_hip.h"

// The victim's and the aggressor's entry points can come from DIFFERENT builds of the library (DIR_VICTIM_LIB / DIR_AGGR_LIB = path of a
// libdir_hip*.so, dlopen'ed RTLD_LOCAL): which side has to be compiled with packed FP32 for the effect to appear?
typedef int (*mano_pair_fn)(const dir_mano_tables*, const float* const*, int, const float* const*, int, const float* const*, int, float* const*, float* const*,
                            float* const*, float* const*, int32_t* const*, int, void*);
typedef int (*dual_fn)(const dir_conv_desc*, const void*, const dir_conv_src2*, const void*, const void*, const float*, void*, void*);
static mano_pair_fn p_mano = dir_mano_forward_pair;
static dual_fn p_dual = dir_conv2d_dual_forward;
static void* sym_from(const char* env, const char* name) {
    const char* path = getenv(env);
    if (!path) return nullptr;
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(1); }
    void* f = dlsym(h, name);
    if (!f) { fprintf(stderr, "dlsym %s\n", name); exit(1); }
    printf("%s from %s\n", name, path);
    return f;
}

typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every lane: acc <- acc * a + b, ITER times, once with the packed instruction on a register pair and once with two scalar FMAs
template <bool MEM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ tab, unsigned* bad_lo, unsigned* bad_hi, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f2 acc = {1.0f + 1e-3f * (t & 255), 0.5f + 1e-3f * (t & 127)};
    float s0 = acc.x, s1 = acc.y;
    f2 a = {0.999f, 1.0005f}, b = {1e-3f, -2e-3f};
    for (int i = 0; i < iters; ++i) {
        if (MEM) {
            const f2 v = *reinterpret_cast<const f2*>(tab + 2 * ((t * 7 + i * 64) & 65535));
            a = v;
        }
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1) : "v"(a.y), "v"(b.y));
    }
    if (__float_as_uint(acc.x) != __float_as_uint(s0)) atomicAdd(bad_lo, 1u);
    if (__float_as_uint(acc.y) != __float_as_uint(s1)) atomicAdd(bad_hi, 1u);
}

// Synthetic aggressors (round 4; variant < -1): which ingredient of the library's 64x128 ring tile does the victim need beside it?  One
// 8-wave workgroup per CU, 128 KB of LDS (a 4-wave victim workgroup still fits beside it), per wave and iteration:
//   -2  16 x v_mfma_f32_32x32x16_bf16 on registers      -3  16 ds_read_b128      -4  3 LDS-DMA pieces (buffer_load_dwordx4 ... lds) + vmcnt
//   -5  MFMA + ds_read + DMA together (the K loop of the convolution)            -6  DMA + ds_read      -7  s_sleep only (resident, idle)
//   -8  MFMA + ds_read      -9  MFMA + DMA
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(16))) float af32x16;
typedef int __attribute__((ext_vector_type(4))) ai32x4;
typedef unsigned __attribute__((ext_vector_type(4))) au32x4;
template <bool MFMA, bool LDSR, bool DMA>
__global__ __launch_bounds__(512, 1) void aggr(const char* src, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* base = src + (size_t)(blockIdx.x % 16) * 262144;
    const ai32x4 rs = {(int)(unsigned)(unsigned long long)base, (int)(unsigned)((unsigned long long)base >> 32), 262144, 0x00020000};
    const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < 32 * 1024; i += 512)
        reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u ^ ((i * 2654435761u) >> 9 & 0x03ff03ffu) ^ ((i & 1) ? 0x80000000u : 0) ^ ((i & 2) ? 0x8000u : 0);
    __syncthreads();
    au32x4 fa[2][4], fb[2][4];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 4; ++q) {
            fa[i][q] = *reinterpret_cast<const au32x4*>(lds + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
            fb[i][q] = *reinterpret_cast<const au32x4*>(lds + 65536 + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
        }
    af32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    au32x4 x = fa[0][0];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDSR) {
            const int off = (it & 1) * 16384;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i][q] = *reinterpret_cast<const au32x4*>(lds + (off + (wave * 64 + i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
                    fb[i][q] = *reinterpret_cast<const au32x4*>(lds + 65536 + (off + (i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
                }
            if constexpr (!MFMA) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) x ^= fa[i][q] ^ fb[i][q];
            }
        }
        if constexpr (DMA) {
            for (int u = 0; u < 3; ++u) {
                const unsigned voff = ((unsigned)((it * 3 + u) * 8 + wave) * 1024u + lane * 16u) & 262143u;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lbase + 98304 + wave * 1024 + u * 8192), "s"(rs) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if constexpr (MFMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, fa[i][q]), __builtin_bit_cast(abf16x8, fb[j][q]), acc[i][j], 0, 0, 0);
        }
        if constexpr (!MFMA && !LDSR && !DMA) __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sres = __uint_as_float(x.x ^ x.y ^ x.z ^ x.w);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sres += acc[i][j][e];
    if (sres == 1.2345f) sink[0] = sres;
}
static void launch_synthetic_aggressor(int variant, const char* src, float* sink, hipStream_t s) {
    const int iters = 600;
    switch (variant) {
        case -2: hipLaunchKernelGGL((aggr<true, false, false>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -3: hipLaunchKernelGGL((aggr<false, true, false>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -4: hipLaunchKernelGGL((aggr<false, false, true>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -5: hipLaunchKernelGGL((aggr<true, true, true>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -6: hipLaunchKernelGGL((aggr<false, true, true>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -8: hipLaunchKernelGGL((aggr<true, true, false>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        case -9: hipLaunchKernelGGL((aggr<true, false, true>), dim3(256), dim3(512), 0, s, src, iters, sink); break;
        default: hipLaunchKernelGGL((aggr<false, false, false>), dim3(256), dim3(512), 0, s, src, iters * 8, sink); break;
    }
}

// victim 9 / 10 (round 4): the ONE form the ISA-level bisect of the MANO kernel left standing (tools/pkfp32_patch_build.py: replacing the 43
// `v_pk_fma_f32 ... op_sel:[0,1,0]` of mano_forward_kernel<256,1> by their scalar halves clears all differences; replacing any other class of
// packed instruction does not): the packed FMA whose LOW result takes src1's HIGH dword (src1.hi broadcast to both halves).  9: operands in
// registers; 10: src1 re-read from LDS (ds_read_b64) in every iteration, as the MANO kernel's pose-map weights are.
template <bool LDSFED>
__global__ __launch_bounds__(256) void victim_opsel(unsigned* bad_lo, unsigned* bad_hi, int iters) {
    __shared__ __attribute__((aligned(8))) float sm[256 * 2 * 8];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
    for (int e = 0; e < 8; ++e) { sm[(e * 256 + tid) * 2] = 123.0f + e; sm[(e * 256 + tid) * 2 + 1] = 0.9990f + 1e-4f * ((tid + e) & 15); }
    __syncthreads();
    f2 acc = {1.0f + 1e-3f * (t & 255), 0.5f + 1e-3f * (t & 127)};
    float s0 = acc.x, s1 = acc.y;
    f2 a = {123.0f, 0.9995f}, b = {1e-3f, -2e-3f};                 // a.x is a decoy: a correct op_sel:[0,1,0] never reads it
    for (int i = 0; i < iters; ++i) {
        if (LDSFED) a = *reinterpret_cast<const f2*>(sm + (((i & 7) * 256 + tid) * 2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(a.y), "v"(b.x));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1) : "v"(a.y), "v"(b.y));
    }
    if (__float_as_uint(acc.x) != __float_as_uint(s0)) atomicAdd(bad_lo, 1u);
    if (__float_as_uint(acc.y) != __float_as_uint(s1)) atomicAdd(bad_hi, 1u);
}

// victim 2 / 3 / 4: LDS reads only.  The packed-FP32 build of the MANO kernel differs from the scalar one not just in its FMAs but in how it
// READS LDS (43 ds_read2_b64 + 82 ds_read2_b32 against 11 + 37; 15 ds_read_b128 against 56): maybe what goes wrong beside the LDS-DMA
// aggressor is an operand fetch.  Every thread fills its own LDS words with a known pattern, then re-reads them ITER times with the
// chosen instruction and counts mismatches.
template <int KIND>
__global__ __launch_bounds__(256) void victim_lds(unsigned* bad_lo, unsigned* bad_hi, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned sm[256 * 8];
    const int tid = threadIdx.x;
    for (int e = 0; e < 8; ++e) sm[tid * 8 + e] = 0x1000000u * e + blockIdx.x * 256 + tid;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)sm + tid * 32;
    unsigned lo = 0, hi = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned v[4];
        if (KIND == 2) {          // ds_read2_b64: two 8-byte elements at offsets 0 and 2 (x 8 bytes)
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 r;
            asm volatile("ds_read2_b64 %0, %1 offset1:2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base) : "memory");
            v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
            lo += (v[0] != 0x0000000u + blockIdx.x * 256 + tid) + (v[2] != 0x4000000u + blockIdx.x * 256 + tid);
            hi += (v[1] != 0x1000000u + blockIdx.x * 256 + tid) + (v[3] != 0x5000000u + blockIdx.x * 256 + tid);
        } else if (KIND == 3) {   // ds_read2_b32: elements 0 and 3
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 r;
            asm volatile("ds_read2_b32 %0, %1 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base) : "memory");
            lo += r.x != 0x0000000u + blockIdx.x * 256 + tid;
            hi += r.y != 0x3000000u + blockIdx.x * 256 + tid;
        } else {                  // ds_read_b128
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 r;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base) : "memory");
            lo += (r.x != 0x0000000u + blockIdx.x * 256 + tid) + (r.z != 0x2000000u + blockIdx.x * 256 + tid);
            hi += (r.y != 0x1000000u + blockIdx.x * 256 + tid) + (r.w != 0x3000000u + blockIdx.x * 256 + tid);
        }
    }
    if (lo) atomicAdd(bad_lo, 1u);
    if (hi) atomicAdd(bad_hi, 1u);
}

// victim 6 / 7 / 8: GLOBAL loads only (global_load_dword / dwordx2 / dwordx4) of a table whose every word encodes its own index: the packed
// build of the MANO kernel fetches its blend-shape tables with 34 global_load_dwordx2 where the scalar build has one.  A wrong operand
// FETCH would look exactly like a wrong packed FMA.
template <int W>
__global__ __launch_bounds__(256) void victim_gload(const unsigned* __restrict__ tab, unsigned* bad_lo, unsigned* bad_hi, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned lo = 0, hi = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned idx = ((unsigned)(t * 9 + i * 2336) % (1u << 20)) & ~3u;         // a k-major table walk like posedirs_t[k * 2336 + column]
        if (W == 1) {
            unsigned v;
            asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tab + idx) : "memory");
            lo += v != idx;
        } else if (W == 2) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 v;
            asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tab + idx) : "memory");
            lo += v.x != idx; hi += v.y != idx + 1;
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 v;
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tab + idx) : "memory");
            lo += (v.x != idx) + (v.z != idx + 2); hi += (v.y != idx + 1) + (v.w != idx + 3);
        }
    }
    if (lo) atomicAdd(bad_lo, 1u);
    if (hi) atomicAdd(bad_hi, 1u);
}

// victim 5: the library's own MANO launch (dir_mano_forward_pair, 64 samples x 2 hands) -- the victim of round 1.  Link against the
// packed-FP32 investigation build (DIR_PACKED_FP32=1 python -m dir_amd.build -> lib/libdir_hip_pk.so) to see the effect, against the product
// library (built without packed FP32) for the control.  Random finite tables; the reference output is the same launch run alone.
struct ManoVictim {
    dir_mano_tables tab[2];
    float *pose[2], *verts[2], *joints[2], *ref[2];
    int B;
    static float* dev_random(size_t n, float scale, unsigned seed) {
        float* h = (float*)malloc(n * 4);
        unsigned s = seed;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
        float* d;
        CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
        free(h);
        return d;
    }
    void init(int B_) {
        B = B_;
        for (int h = 0; h < 2; ++h) {
            tab[h].shapedirs_t = dev_random(10 * 2336, 0.004f, 11 + h); tab[h].posedirs_t = dev_random(135 * 2336, 0.0015f, 13 + h);
            tab[h].v_template = dev_random(2334, 0.05f, 15 + h); tab[h].j_template = dev_random(48, 0.05f, 17 + h);
            tab[h].j_shapedirs = dev_random(480, 0.004f, 19 + h); tab[h].hands_mean = dev_random(45, 0.15f, 23 + h);
            tab[h].comps = dev_random(45 * 45, 0.2f, 29 + h);
            float* w = (float*)malloc(778 * 16 * 4);                       // skinning weights: rows sum to 1
            for (int v = 0; v < 778; ++v) for (int k = 0; k < 16; ++k) w[v * 16 + k] = (k == v % 16) ? 0.7f : 0.02f;
            float* dw; CK(hipMalloc(&dw, 778 * 16 * 4)); CK(hipMemcpy(dw, w, 778 * 16 * 4, hipMemcpyHostToDevice)); free(w);
            tab[h].weights = dw; tab[h].side = 1 - h; tab[h].center_idx = 0; tab[h].root_palm = 0;
            pose[h] = dev_random((size_t)B * 64, 0.4f, 31 + h);
            CK(hipMalloc(&verts[h], (size_t)B * 778 * 3 * 4)); CK(hipMalloc(&joints[h], (size_t)B * 21 * 3 * 4));
            ref[h] = (float*)malloc((size_t)B * 778 * 3 * 4);
        }
    }
    int launch(hipStream_t s) {
        const float* p[2] = {pose[0], pose[1]};
        const float* b[2] = {pose[0] + 51, pose[1] + 51};
        return p_mano(tab, p, 64, b, 64, nullptr, 0, verts, joints, nullptr, nullptr, nullptr, B, s);
    }
};

int main(int argc, char** argv) {
    if (void* f = sym_from("DIR_VICTIM_LIB", "dir_mano_forward_pair")) p_mano = (mano_pair_fn)f;
    if (void* f = sym_from("DIR_AGGR_LIB", "dir_conv2d_dual_forward")) p_dual = (dual_fn)f;
    const int variant = argc > 1 ? atoi(argv[1]) : 19;
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    const int vmem = argc > 3 ? atoi(argv[3]) : 0;
    const int B = 64, S = 16, C1 = 128, C2 = 1024, N = 256;
    hipStream_t sa, sv;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sv));
    // aggressor operands: bf16 patterns that are finite (0x3c00.. small values)
    const size_t ny = (size_t)B * S * S * C1, nx = (size_t)B * S * S * C2, nw = (size_t)N * (C1 + C2), no = (size_t)B * S * S * N;
    unsigned short *y, *x, *w, *o;
    float* shift;
    CK(hipMalloc(&y, ny * 2)); CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&o, no * 2)); CK(hipMalloc(&shift, N * 4));
    {
        unsigned short* h = (unsigned short*)malloc(nx * 2);
        // random bf16 values in (-2, 2) with random mantissas: high toggle rate on the matrix cores (argv[4] = 0: the quiet pattern instead)
        unsigned st = 12345u;
        const int quiet = argc > 4 && atoi(argv[4]) == 0;
        for (size_t i = 0; i < nx; ++i) {
            st = st * 1664525u + 1013904223u;
            h[i] = quiet ? (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 0x180) ^ (unsigned short)((i & 1) << 15)
                         : (unsigned short)(((st >> 9) & 0x807f) | (0x3e00 + ((st >> 3) & 0x0180)));
        }
        CK(hipMemcpy(x, h, nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(y, h, ny * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, h, nw * 2, hipMemcpyHostToDevice));
        free(h);
        CK(hipMemset(shift, 0, N * 4));
    }
    float* tab;
    CK(hipMalloc(&tab, 2 * 65536 * 4));
    {
        float* h = (float*)malloc(2 * 65536 * 4);
        for (int i = 0; i < 2 * 65536; ++i) h[i] = 0.999f + 1e-6f * (i % 1000);
        CK(hipMemcpy(tab, h, 2 * 65536 * 4, hipMemcpyHostToDevice));
        free(h);
    }
    unsigned* itab;
    CK(hipMalloc(&itab, (1u << 20) * 4 + 64));
    {
        unsigned* h = (unsigned*)malloc((1u << 20) * 4 + 64);
        for (unsigned i = 0; i < (1u << 20) + 16; ++i) h[i] = i;
        CK(hipMemcpy(itab, h, (1u << 20) * 4 + 64, hipMemcpyHostToDevice));
        free(h);
    }
    unsigned *bad, hbad[2];
    CK(hipMalloc(&bad, 8));
    dir_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.H = S; d.W = S; d.Cin = C1; d.in_cstride = C1; d.Cout = N; d.out_cstride = N; d.kh = d.kw = 1; d.stride = 1;
    d.in_dtype = d.out_dtype = DIR_DT_BF16; d.flags = DIR_CONV_RELU | DIR_CONV_VARIANT(variant);
    dir_conv_src2 d2 = {S, S, C2, C2, 0, 1};
    int bad_launches[2] = {0, 0}, total = 0;
    if (vmem == 5) {
        ManoVictim mv;
        mv.init(B);
        if (mv.launch(sv) != 0) { fprintf(stderr, "victim: %s\n", dir_last_error()); return 1; }
        CK(hipStreamSynchronize(sv));
        for (int h = 0; h < 2; ++h) CK(hipMemcpy(mv.ref[h], mv.verts[h], (size_t)B * 778 * 3 * 4, hipMemcpyDeviceToHost));
        float* got = (float*)malloc((size_t)B * 778 * 3 * 4);
        {   // FNV-1a over the reference launch's vertices: the same for every build of the victim that computes the same bits
            unsigned long long cs = 1469598103934665603ull;
            for (int h = 0; h < 2; ++h)
                for (size_t i = 0; i < (size_t)B * 778 * 3; ++i) { unsigned u; memcpy(&u, &mv.ref[h][i], 4); cs = (cs ^ u) * 1099511628211ull; }
            printf("victim alone: vertices checksum %016llx, v[0] = %.9g %.9g %.9g\n", cs, mv.ref[0][0], mv.ref[0][1], mv.ref[0][2]);
        }
        int wrong_even = 0, wrong_odd = 0;
        for (int r = 0; r < rounds; ++r) {
            for (int k = 0; k < 24; ++k)
                if (variant >= 0 && p_dual(&d, y, &d2, x, w, shift, o, sa) != 0) { fprintf(stderr, "aggressor: %s\n", dir_last_error()); return 1; }
            for (int k = 0; k < 12; ++k) mv.launch(sv);                   // (the last launch's output is the one compared)
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            int bad_this = 0;
            for (int h = 0; h < 2; ++h) {
                CK(hipMemcpy(got, mv.verts[h], (size_t)B * 778 * 3 * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < (size_t)B * 778 * 3; ++i)
                    if (memcmp(&got[i], &mv.ref[h][i], 4) != 0) { bad_this = 1; ((i & 1) ? wrong_odd : wrong_even)++; }
            }
            bad_launches[0] += bad_this; ++total;
        }
        printf("aggressor variant %d, victim dir_mano_forward_pair: %d launches, %d differ from the launch run alone (wrong floats at even / odd flat index: %d / %d)\n",
               variant, total, bad_launches[0], wrong_even, wrong_odd);
        return 0;
    }
    for (int r = 0; r < rounds; ++r) {
        CK(hipMemsetAsync(bad, 0, 8, sv));
        for (int k = 0; k < 24; ++k) {
            if (variant >= 0 && p_dual(&d, y, &d2, x, w, shift, o, sa) != 0) { fprintf(stderr, "aggressor: %s\n", dir_last_error()); return 1; }
            if (variant < -1) launch_synthetic_aggressor(variant, (const char*)x, (float*)o, sa);
        }
        if (vmem == 1) hipLaunchKernelGGL(victim<true>, dim3(1024), dim3(256), 0, sv, tab, bad, bad + 1, 512);
        else if (vmem == 2) hipLaunchKernelGGL(victim_lds<2>, dim3(2048), dim3(256), 0, sv, bad, bad + 1, 2048);
        else if (vmem == 3) hipLaunchKernelGGL(victim_lds<3>, dim3(2048), dim3(256), 0, sv, bad, bad + 1, 2048);
        else if (vmem == 4) hipLaunchKernelGGL(victim_lds<4>, dim3(2048), dim3(256), 0, sv, bad, bad + 1, 2048);
        else if (vmem == 9) hipLaunchKernelGGL(victim_opsel<false>, dim3(1024), dim3(256), 0, sv, bad, bad + 1, 4096);
        else if (vmem == 10) hipLaunchKernelGGL(victim_opsel<true>, dim3(1024), dim3(256), 0, sv, bad, bad + 1, 4096);
        else if (vmem >= 6 && vmem <= 8) {
            for (int k = 0; k < 12; ++k) {
                if (vmem == 6) hipLaunchKernelGGL(victim_gload<1>, dim3(512), dim3(256), 0, sv, itab, bad, bad + 1, 135);
                else if (vmem == 7) hipLaunchKernelGGL(victim_gload<2>, dim3(512), dim3(256), 0, sv, itab, bad, bad + 1, 135);
                else hipLaunchKernelGGL(victim_gload<4>, dim3(512), dim3(256), 0, sv, itab, bad, bad + 1, 135);
            }
        }
        else hipLaunchKernelGGL(victim<false>, dim3(1024), dim3(256), 0, sv, tab, bad, bad + 1, 4096);
        CK(hipMemcpyAsync(hbad, bad, 8, hipMemcpyDeviceToHost, sv));
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        bad_launches[0] += hbad[0] != 0; bad_launches[1] += hbad[1] != 0; ++total;
    }
    printf("aggressor variant %d, victim %s: %d launches, low-half mismatches in %d, high-half mismatches in %d\n", variant,
           vmem == 0 ? "pk_fma registers" : vmem == 1 ? "pk_fma memory-fed" : vmem == 2 ? "ds_read2_b64" : vmem == 3 ? "ds_read2_b32" : vmem == 4 ? "ds_read_b128" : vmem == 6 ? "global_load_dword" : vmem == 7 ? "global_load_dwordx2" : vmem == 9 ? "v_pk_fma_f32 op_sel:[0,1,0] registers" : vmem == 10 ? "v_pk_fma_f32 op_sel:[0,1,0] LDS-fed" : "global_load_dwordx4", total, bad_launches[0], bad_launches[1]);
    return 0;
}
