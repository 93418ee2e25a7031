// Does MODE.FP16_OVFL (bit 23 of the MODE hardware register) make the fp32 -> f16 conversions of gfx950 saturate at +-65504 instead of
// producing inf?  (The f16-storage kernels want a clamp that costs no VALU instruction.)   hipcc --offload-arch=gfx950 -O3 tools/ubench_f16_ovfl.hip -o /tmp/ovfl/t && /tmp/ovfl/t
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__global__ void k(const float* in, uint32_t* out, int n, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);          // hwreg(HW_REG_MODE, 23, 1) = 1
    const int i = threadIdx.x;
    if (i < n) {
        const f32x2 v = {in[2 * i], in[2 * i + 1]};
        out[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    }
}
int main() {
    const float h[8] = {70000.f, 1e6f, -1e6f, INFINITY, 65504.f, 65519.f, 65520.f, -65536.f};
    float* d; uint32_t* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 4, ovfl);
        uint32_t r[4];
        hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", ovfl);
        for (int i = 0; i < 4; ++i) printf("  %g -> %04x  %g -> %04x", h[2 * i], r[i] & 0xffff, h[2 * i + 1], r[i] >> 16);
        printf("\n");
    }
    return 0;
}
