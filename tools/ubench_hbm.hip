// Scratch microbenchmark: what a plain streaming kernel reaches on this box, by size (Infinity Cache = 256 MB) and access kind.
//   read  : 16 B/lane loads, summed        write : 16 B/lane stores        copy : load + store
//   hot   : the same buffer read again right after it was written (producer -> consumer through the Infinity Cache)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_read(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_write(uint4* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, 1, 2, 3);
}
__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// 4 independent loads in flight per thread
__global__ void k_read4(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const size_t maxb = 1ull << 30;
    uint4 *a, *b; unsigned* sink;
    (void)hipMalloc(&a, maxb); (void)hipMalloc(&b, maxb); (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, maxb); (void)hipMemset(b, 2, maxb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t mb : {16, 64, 128, 256, 1024}) {
        const size_t n = mb * (1ull << 20) / 16;
        for (int wg : {1024, 4096, 16384}) {
            float t[5] = {0, 0, 0, 0, 0};
            const int reps = 20;
            auto run = [&](int which) {
                for (int r = 0; r < reps + 2; ++r) {
                    if (r == 2) (void)hipEventRecord(e0);
                    if (which == 0) hipLaunchKernelGGL(k_read, dim3(wg), dim3(256), 0, 0, a, n, sink);
                    if (which == 1) hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, a, n);
                    if (which == 2) hipLaunchKernelGGL(k_copy, dim3(wg), dim3(256), 0, 0, a, b, n);
                    if (which == 3) hipLaunchKernelGGL(k_read4, dim3(wg), dim3(256), 0, 0, a, n, sink);
                    if (which == 4) { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, a, n); hipLaunchKernelGGL(k_read4, dim3(wg), dim3(256), 0, 0, a, n, sink); }
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
            };
            for (int w = 0; w < 5; ++w) t[w] = run(w);
            const double gb = mb * 1.048576e-3;
            printf("%5zu MB wg %5d: read %.2f  write %.2f  copy(r+w) %.2f  read4 %.2f  write-then-read pair(2x) %.2f  TB/s\n", mb, wg, gb / t[0], gb / t[1], 2 * gb / t[2],
                   gb / t[3], 2 * gb / t[4]);
        }
    }
    return 0;
}
