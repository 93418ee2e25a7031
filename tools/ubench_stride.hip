// Scratch: does the ORDER in which a workgroup walks its [128 pixels][1 KB] slab of an NHWC activation matter to HBM?  (stream.hip reads it as
// eight K-chunks of 128 px x 128 B -- 128-byte pieces 1 KB apart, the next piece of the same row one chunk later; the alternative is whole 1 KB rows.)
//   mode 0: chunk-major (stream.hip's order), two chunks (32 KB) in flight per workgroup     mode 1: row-major, 32 rows x 1 KB per step, two steps in flight
//   mode 2: like 0 with four chunks in flight      mode 3: plain grid-stride copy-style read of the same bytes
// hipcc --offload-arch=gfx950 -O3 tools/ubench_stride.hip -o /tmp/ustride && /tmp/ustride
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ x, int M, unsigned* sink) {
    const int tid = threadIdx.x;
    unsigned acc = 0;
    const size_t base = (size_t)blockIdx.x * 128 * 64;          // uint4 units: 128 rows x 64 (1 KB)
    if (MODE == 0 || MODE == 2) {
        constexpr int D = MODE == 0 ? 2 : 4;
        uint4 r[D][4];
        auto load = [&](int c, uint4 (&v)[4]) {
            const int cc = c < 8 ? c : 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = x[base + (size_t)((tid >> 3) + 32 * i) * 64 + cc * 8 + (tid & 7)];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, r[d]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc ^= r[c % D][i].x ^ r[c % D][i].w;
            load(c + D, r[c % D]);
            __syncthreads();
        }
    } else if (MODE == 1) {
        uint4 r[2][8];
        auto load = [&](int s, uint4 (&v)[8]) {
            const int ss = s < 4 ? s : 3;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = x[base + (size_t)(ss * 32 + (tid >> 6) * 8 + i) * 64 + (tid & 63)];
        };
        load(0, r[0]); load(1, r[1]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= r[s & 1][i].x ^ r[s & 1][i].w;
            load(s + 2, r[s & 1]);
            __syncthreads();
        }
    } else {
        for (int i = tid; i < 128 * 64; i += 256 * 4) {
            uint4 a = x[base + i], b = x[base + i + 256], c = x[base + i + 512], d = x[base + i + 768];
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const int M = 65536 * 4;                                    // 256 MB of bf16 [M][512]: larger than the Infinity Cache
    uint4* x; unsigned* sink;
    (void)hipMalloc(&x, (size_t)M * 1024); (void)hipMemset(x, 1, (size_t)M * 1024); (void)hipMalloc(&sink, 4);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            const int n = 20;
            for (int i = 0; i < n; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(M / 128), dim3(256), 0, 0, x, M, sink);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(M / 128), dim3(256), 0, 0, x, M, sink);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(M / 128), dim3(256), 0, 0, x, M, sink);
                else hipLaunchKernelGGL(k<3>, dim3(M / 128), dim3(256), 0, 0, x, M, sink);
            }
            (void)hipDeviceSynchronize();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n;
            if (rep) printf("mode %d: %.1f us per pass over %d MB = %.2f TB/s\n", mode, dt * 1e6, M / 1024, (double)M * 1024 / dt / 1e12);
        }
    }
    return 0;
}
