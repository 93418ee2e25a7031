// Scratch: LDS canary.  Every workgroup fills LDS_WORDS words of LDS with a pattern, idles ~spin iterations, verifies, and reports the
// first corrupted words: does a kernel running beside it on another stream write into LDS it does not own?
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int LDS_WORDS = 8192;      // 32 KB
__global__ __launch_bounds__(256) void canary_kernel(unsigned* report, int spin, int max_rep) {
    __shared__ unsigned s[LDS_WORDS];
    const int tid = threadIdx.x;
    for (int i = tid; i < LDS_WORDS; i += 256) s[i] = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 16);
    __syncthreads();
    for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(100);
    __syncthreads();
    for (int i = tid; i < LDS_WORDS; i += 256) {
        const unsigned v = s[i], want = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 16);
        if (v != want) {
            const unsigned slot = atomicAdd(report, 1u);
            if (slot < (unsigned)max_rep) { report[1 + 3 * slot] = blockIdx.x; report[2 + 3 * slot] = i; report[3 + 3 * slot] = v; }
        }
    }
}
extern "C" int canary_launch(unsigned* report, int blocks, int spin, int max_rep, void* stream) {
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, report, spin, max_rep);
    return (int)hipGetLastError();
}
