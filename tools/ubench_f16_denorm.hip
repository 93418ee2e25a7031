// Does v_mfma_f32_32x32x16_f16 honour f16 denormal INPUTS, and does the f32 -> f16 conversion produce them?  (Design input for the f16x3
// split-precision convolution: lo = f16(x - f32(f16(x))) is a denormal whenever |x| < 0.125.)
// hipcc --offload-arch=gfx950 -O2 tools/ubench_f16_denorm.hip -o tools/_ubench/f16_denorm && tools/_ubench/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 __attribute__((ext_vector_type(8))) f16x8;
typedef float __attribute__((ext_vector_type(16))) f32x16;
typedef _Float16 __attribute__((ext_vector_type(2))) f16x2;
typedef float __attribute__((ext_vector_type(2))) f32x2;

__global__ void k(const float* xs, float* out, int n) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        const float x = xs[t];
        const f16x2 hi2 = __builtin_convertvector(f32x2{x, x}, f16x2);
        const float hif = (float)hi2[0];
        const f16x2 lo2 = __builtin_convertvector(f32x2{x - hif, x - hif}, f16x2);
        f16x8 a, b, al;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0; al[e] = (_Float16)0; b[e] = (_Float16)0; }
        // row (lane & 31), k = 8*(lane >> 5) + e.  Put the value at k = 0 only, B = 1 at k = 0, column (lane & 31)
        if ((lane >> 5) == 0) { a[0] = hi2[0]; al[0] = lo2[0]; b[0] = (_Float16)1.0f; }
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        f32x16 acc2;
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc2, 0, 0, 0);
        if (lane == 0) {
            out[4 * t + 0] = hif;
            out[4 * t + 1] = (float)lo2[0];
            out[4 * t + 2] = acc[0];
            out[4 * t + 3] = acc2[0];
        }
    }
}
int main() {
    const int n = 8;
    float h[n] = {1.0f, 0.1f, 0.01f, 1e-3f, 1e-4f, 3e-5f, 1e-6f, 100.7f};
    float *dx, *dout;
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, 4 * n * sizeof(float));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n);
    float o[4 * n];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t)
        printf("x=%.9g  hi=%.9g lo=%.9g | mfma(hi)=%.9g mfma(lo)=%.9g | hi+lo-x=%.3g  mfma_sum-x=%.3g (rel %.3g)\n", h[t], o[4 * t], o[4 * t + 1], o[4 * t + 2],
               o[4 * t + 3], (double)o[4 * t] + o[4 * t + 1] - h[t], (double)o[4 * t + 2] + o[4 * t + 3] - h[t], ((double)o[4 * t + 2] + o[4 * t + 3] - h[t]) / h[t]);
    return 0;
}
