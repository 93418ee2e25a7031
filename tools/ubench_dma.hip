// Scratch microbenchmark: per-CU throughput of global -> LDS transfers on gfx950, L2-resident source.
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA), mode 1: buffer_load_dwordx4 -> VGPR -> ds_write_b128, mode 2: VGPR only
// Each workgroup (8 waves) streams REPS x 1 KiB pieces per wave from a 256 KiB window.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int __attribute__((ext_vector_type(4))) i32x4;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const char* src, unsigned bytes, int reps, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const char* base = src + (size_t)(blockIdx.x % 16) * 262144;
    const i32x4 rs = {(int)(unsigned)(unsigned long long)base, (int)(unsigned)((unsigned long long)base >> 32), 262144, 0x00020000};
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 262144, 0x00020000);
    const unsigned lbase = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds + wave * 1024;
    unsigned acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned voff = ((unsigned)((r + u) * WAVES + wave) * 1024u + lane * 16u) & 262143u;
            if constexpr (MODE == 0) {
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" ::"v"(voff), "s"(lbase + (u & 3) * 8192), "s"(rs) : "memory", "m0");
            } else {
                u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, voff, 0, 0));
                if constexpr (MODE == 1) *reinterpret_cast<u32x4*>(lds + wave * 1024 + (u & 3) * 8192 + lane * 16) = v;
                else acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (MODE != 0) acc += lds[tid * 4];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int WAVES>
void run(const char* src, int reps, int nblk, unsigned long long* dout, unsigned* sink) {
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(nblk), dim3(64 * WAVES), 0, 0, src, 0u, reps, dout, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(nblk), dim3(64 * WAVES), 0, 0, src, 0u, reps, dout, sink);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, dout, nblk * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nblk; ++i) s += h[i];
    s /= nblk;
    const double kb = (double)reps * WAVES;
    printf("mode %d waves %d blocks %4d: %8.0f cycles for %5.0f KiB per CU -> %6.1f cycles/KiB, %5.1f B/clk/CU\n", MODE, WAVES, nblk, s, kb, s / kb,
           kb * 1024 / s);
}

int main() {
    char* src; unsigned long long* dout; unsigned* sink;
    hipMalloc(&src, 16 * 262144); hipMemset(src, 1, 16 * 262144);
    hipMalloc(&dout, 1024 * 8); hipMalloc(&sink, 4);
    for (int nblk : {1, 256}) {
        run<0, 8>(src, 256, nblk, dout, sink);
        run<1, 8>(src, 256, nblk, dout, sink);
        run<2, 8>(src, 256, nblk, dout, sink);
        run<0, 4>(src, 256, nblk, dout, sink);
        run<1, 4>(src, 256, nblk, dout, sink);
        run<0, 1>(src, 256, nblk, dout, sink);
        run<1, 1>(src, 256, nblk, dout, sink);
    }
    return 0;
}
