// Scratch: VGPR canary.  Every lane keeps NR known values in registers across an idle period (opaque to the compiler), then verifies
// them: does a kernel running beside it on the same SIMD write into registers it does not own?
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int NR = 200;
__global__ __launch_bounds__(256) void canary_vgpr_kernel(unsigned* report, int spin, int max_rep) {
    unsigned x[NR];
    const unsigned seed = threadIdx.x * 2654435761u ^ (blockIdx.x << 20);
#pragma unroll
    for (int i = 0; i < NR; ++i) { x[i] = seed + 0x9E3779B9u * (unsigned)i; asm volatile("" : "+v"(x[i])); }
    for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(100);
#pragma unroll
    for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(x[i]));
    unsigned bad = 0, first = 0, val = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const unsigned want = seed + 0x9E3779B9u * (unsigned)i;
        if (x[i] != want) { if (!bad) { first = i; val = x[i]; } ++bad; }
    }
    if (bad) {
        const unsigned slot = atomicAdd(report, 1u);
        if (slot < (unsigned)max_rep) { report[1 + 4 * slot] = blockIdx.x * 256 + threadIdx.x; report[2 + 4 * slot] = first; report[3 + 4 * slot] = val; report[4 + 4 * slot] = bad; }
    }
}
extern "C" int canary_vgpr_launch(unsigned* report, int blocks, int spin, int max_rep, void* stream) {
    hipLaunchKernelGGL(canary_vgpr_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, report, spin, max_rep);
    return (int)hipGetLastError();
}
