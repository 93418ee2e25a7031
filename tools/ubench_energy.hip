// Scratch microbenchmark for the ENERGY side of the roofline: one loop shape per run, on every CU, for a given number of
// seconds, while tools/ubench_energy.py samples rocm-smi.  Prints what the loop achieved; the wrapper divides watts above idle by it.
//   mode 0  spin     : waves resident, s_sleep loop (clocks up, nothing switching)
//   mode 1  mfma     : 16 x v_mfma_f32_32x32x16_bf16 per wave and iteration, operands in registers (8 waves per CU, 64 x 64 wave tiles)
//   mode 2  mfma+lds : + 16 ds_read_b128 per wave and iteration (the fragment traffic of a 64 x 64 wave tile)
//   mode 3  mfma+lds+dma : + 3 LDS-DMA pieces (16 B per lane) per wave and iteration from an L2-resident 4 MB source
//   mode 4  lds      : the 16 ds_read_b128 alone
//   mode 5  dma      : the 3 DMA pieces alone (L2 -> LDS)
//   mode 6  hbm read : 16-byte loads streaming a 1 GiB buffer        mode 7  hbm copy : load + store
//   mode 8  mall read: the same loads over a 96 MB buffer (Infinity-Cache resident after the first pass)      mode 9  l2 read: over 2 MB per XCD-ish (16 MB)
//   arg 3: data 1 = random bf16 (default), 0 = zeros (modes 1-3)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int __attribute__((ext_vector_type(4))) i32x4;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <bool MFMA, bool LDSR, bool DMA>
__global__ __launch_bounds__(512, 1) void k(const char* src, int iters, float* sink, int rnd) {
    __shared__ __attribute__((aligned(16))) char lds[128 * 1024];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* base = src + (size_t)(blockIdx.x % 16) * 262144;
    const i32x4 rs = {(int)(unsigned)(unsigned long long)base, (int)(unsigned)((unsigned long long)base >> 32), 262144, 0x00020000};
    const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
    for (int i = tid; i < 32 * 1024; i += 512) {
        unsigned v = rnd ? (0x3c003c00u ^ ((i * 2654435761u) >> 9 & 0x03ff03ffu) ^ ((i & 1) ? 0x80000000u : 0) ^ ((i & 2) ? 0x8000u : 0)) : 0u;
        reinterpret_cast<unsigned*>(lds)[i] = v;
    }
    __syncthreads();
    u32x4 fa[2][4], fb[2][4];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 4; ++q) {
            fa[i][q] = *reinterpret_cast<const u32x4*>(lds + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
            fb[i][q] = *reinterpret_cast<const u32x4*>(lds + 65536 + ((wave * 64 + i * 32 + (lane & 31)) * 128 + ((lane >> 5) * 4 + q) * 16) % 65536);
        }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    u32x4 x = fa[0][0];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDSR) {
            const int off = (it & 1) * 16384;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i][q] = *reinterpret_cast<const u32x4*>(lds + (off + (wave * 64 + i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
                    fb[i][q] = *reinterpret_cast<const u32x4*>(lds + 65536 + (off + (i * 32 + (lane & 31)) * 128 + (((lane >> 5) * 4 + q) ^ ((lane >> 1) & 7)) * 16) % 65536);
                }
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][q]), __builtin_bit_cast(bf16x8, fb[j][q]), acc[i][j], 0, 0, 0);
        }
        if constexpr (!MFMA && !LDSR && !DMA) __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = __uint_as_float(x.x ^ x.y ^ x.z ^ x.w);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 1.2345f) sink[0] = s;
}

__global__ void k_read(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    const int rnd = argc > 3 ? atoi(argv[3]) : 1;
    char* src; float* sink;
    (void)hipMalloc(&src, 16 * 262144); (void)hipMemset(src, 0x3c, 16 * 262144); (void)hipMalloc(&sink, 4);
    uint4 *a = nullptr, *b = nullptr;
    const size_t nb = 1ull << 30;
    if (mode >= 6) { (void)hipMalloc(&a, nb); (void)hipMalloc(&b, nb); (void)hipMemset(a, 1, nb); (void)hipMemset(b, 2, nb); }
    (void)hipDeviceSynchronize();
    const int iters = mode == 0 ? 20000 : 20000;
    double units = 0;       // FLOP (modes 1-3), LDS bytes (4), DMA bytes (5), HBM bytes (6, 7), launches (0)
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    while (elapsed() < seconds) {
        for (int r = 0; r < 4; ++r) {
            switch (mode) {
                case 0: hipLaunchKernelGGL((k<false, false, false>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 1: hipLaunchKernelGGL((k<true, false, false>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 2: hipLaunchKernelGGL((k<true, true, false>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 3: hipLaunchKernelGGL((k<true, true, true>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 4: hipLaunchKernelGGL((k<false, true, false>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 5: hipLaunchKernelGGL((k<false, false, true>), dim3(256), dim3(512), 0, 0, src, iters, sink, rnd); break;
                case 6: hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, nb / 16, (unsigned*)sink); break;
                case 8: hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, (96ull << 20) / 16, (unsigned*)sink); break;
                case 9: hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, (16ull << 20) / 16, (unsigned*)sink); break;
                default: hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, a, b, nb / 16); break;
            }
            ++launches;
        }
        (void)hipDeviceSynchronize();
    }
    const double dt = elapsed();
    const double per_launch_cu = (double)iters * 8;          // wave-iterations per CU and launch
    const char* unit = "launches";
    if (mode >= 1 && mode <= 3) { units = launches * per_launch_cu * 256 * 16 * 32768.0; unit = "FLOP"; }
    else if (mode == 4) { units = launches * per_launch_cu * 256 * 16 * 1024.0; unit = "LDS bytes"; }
    else if (mode == 5) { units = launches * per_launch_cu * 256 * 3 * 1024.0; unit = "DMA bytes"; }
    else if (mode == 6) { units = (double)launches * nb; unit = "HBM bytes"; }
    else if (mode == 7) { units = (double)launches * nb * 2; unit = "HBM bytes"; }
    else if (mode == 8) { units = (double)launches * (96ull << 20); unit = "HBM bytes"; }
    else if (mode == 9) { units = (double)launches * (16ull << 20); unit = "HBM bytes"; }
    else units = launches;
    printf("{\"mode\": %d, \"seconds\": %.3f, \"units\": %.6e, \"unit\": \"%s\", \"rate\": %.6e, \"data\": %d}\n", mode, dt, units, unit, units / dt, rnd);
    return 0;
}
