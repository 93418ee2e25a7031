// Scratch microbenchmark: sustained shader clock and MFMA rate under load on every CU of the chip.
//   8 waves per CU; each iteration = 16 x v_mfma_f32_32x32x16_bf16 per wave (+ optional 16 ds_read_b128, + optional 3 LDS-DMA
//   pieces per wave).  clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.  data: 0 = zeros, 1 = random bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int __attribute__((ext_vector_type(4))) i32x4;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <bool LDSR, bool DMA, int MI, int NJ, int WAVES, int BAR = 0>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const char* src, int iters, unsigned long long* out, float* sink, int rnd) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* base = src + (size_t)(blockIdx.x % 16) * 262144;
    const i32x4 rs = {(int)(unsigned)(unsigned long long)base, (int)(unsigned)((unsigned long long)base >> 32), 262144, 0x00020000};
    const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < 32 * 1024; i += 64 * WAVES) {
        unsigned v = rnd ? (0x3c003c00u ^ ((i * 2654435761u) >> 9 & 0x03ff03ffu) ^ ((i & 1) ? 0x80000000u : 0) ^ ((i & 2) ? 0x8000u : 0)) : 0u;
        reinterpret_cast<unsigned*>(lds)[i] = v;
    }
    __syncthreads();
    u32x4 fa[MI][4], fb[NJ][4];
    for (int i = 0; i < MI; ++i)
        for (int q = 0; q < 4; ++q)
            fa[i][q] = *reinterpret_cast<const u32x4*>(lds + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
    for (int i = 0; i < NJ; ++i)
        for (int q = 0; q < 4; ++q)
            fb[i][q] = *reinterpret_cast<const u32x4*>(lds + 65536 + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
    f32x16 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (BAR >= 1) __syncthreads();
        if constexpr (LDSR) {
            const int off = (it & 1) * 16384;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[i][q] = *reinterpret_cast<const u32x4*>(lds + (off + (wave * 64 + i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    fb[i][q] = *reinterpret_cast<const u32x4*>(lds + 65536 + (off + (i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
            }
        }
        if constexpr (DMA) {
            for (int u = 0; u < 3; ++u) {
                const unsigned voff = ((unsigned)((it * 3 + u) * WAVES + wave) * 1024u + lane * 16u) & 262143u;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lbase + 98304 + wave * 1024 + u * 8192), "s"(rs) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if constexpr (BAR >= 2) __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][q]), __builtin_bit_cast(bf16x8, fb[j][q]), acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    if (tid == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    float s = 0;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 1.2345f) sink[0] = s;
}

template <bool LDSR, bool DMA, int MI = 2, int NJ = 2, int WAVES = 8, int BAR = 0>
void run(const char* src, int iters, int nblk, unsigned long long* dout, float* sink, int rnd) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<LDSR, DMA, MI, NJ, WAVES, BAR>), dim3(nblk), dim3(64 * WAVES), 0, 0, src, iters, dout, sink, rnd);
        (void)hipDeviceSynchronize();
    }
    static unsigned long long h[2048];
    (void)hipMemcpy(h, dout, nblk * 16, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nblk; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    c /= nblk; r /= nblk;
    const double secs = r / 100e6, flop = (double)iters * WAVES * (MI * NJ * 4) * 32768.0 * nblk;
    printf("bar %d wave tile %dx%d x%d waves ldsread %d dma %d data %s blocks %4d: %.0f us, memtime/realtime = %.3f, cycles/iter (memtime) %.0f, %.0f TFLOP/s\n", BAR, MI * 32, NJ * 32, WAVES, (int)LDSR, (int)DMA,
           rnd ? "random" : "zeros ", nblk, secs * 1e6, c / r, c / iters, flop / secs / 1e12);
}

int main() {
    char* src; unsigned long long* dout; float* sink;
    (void)hipMalloc(&src, 16 * 262144); (void)hipMemset(src, 0x3c, 16 * 262144);
    (void)hipMalloc(&dout, 2048 * 8); (void)hipMalloc(&sink, 4);
    for (int rnd : {1}) {
        run<true, true, 2, 2, 8, 0>(src, 20000, 256, dout, sink, rnd);
        run<true, true, 2, 2, 8, 1>(src, 20000, 256, dout, sink, rnd);
        run<true, true, 2, 2, 8, 2>(src, 20000, 256, dout, sink, rnd);
        run<true, true, 4, 2, 8, 1>(src, 10000, 256, dout, sink, rnd);
        run<true, false, 2, 2, 8, 1>(src, 20000, 256, dout, sink, rnd);
    }
    return 0;
}
