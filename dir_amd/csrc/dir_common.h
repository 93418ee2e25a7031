// Shared helpers for libdir_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dir_hip.h"

namespace dir {

void set_error(const char* fmt, ...);

// Launch log (measurement aid, dir_launch_log_* in include/dir_hip.h): every kernel launch of the library records its kernel's name
// for the calling thread, so that a per-call HIP-event timing (bench.py's live roofline) can be labelled with the kernel
// symbol rocprofv3 reports for it.  One pointer store per launch.
void note_kernel(const char* name);
#define DIR_LAUNCH(kernel, ...)                   \
    do {                                          \
        ::dir::note_kernel(#kernel);              \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);  \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DIR_E_LAUNCH;
    }
    return DIR_OK;
}

#define DIR_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            dir::set_error(__VA_ARGS__); \
            return DIR_E_INVALID;       \
        }                               \
    } while (0)

// Tuning aid shared by the latency-bound token kernels: with DIR_STAMPS=<kernel name> in the environment the launcher passes a
// device buffer and thread 0 of workgroup 0 writes s_memtime at every phase boundary; the launcher then prints the phase
// durations (ticks of the shader clock counter) to stderr.  NULL (the normal case) costs one uniform branch per stamp.
long long* stamps_begin(const char* kernel);                       // NULL unless DIR_STAMPS names this kernel
void stamps_end(const char* kernel, long long* buf, hipStream_t s);
constexpr int MAX_STAMPS = 64;

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the 64 lanes on DPP row operations (a few cycles each) + four v_readlane: the shuffle tree above goes through
// ds_bpermute (~100 cycles per step, six dependent steps).  Different association order than wave_sum (last-ulp differences).
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f32<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);      // row_half_mirror
    v += dpp_f32<0x140>(v);      // row_mirror: every lane holds its row-of-16 sum
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace dir

namespace dir {
// s + p[0] + p[stride] + ... + p[(n - 1) stride], added IN THAT ORDER (the same bits as the plain loop), with eight loads in flight: the
// chunk-partial reductions of the training step (weight gradients, split-K GEMMs, BatchNorm, column sums) were one dependent memory round trip
// per term -- up to 64 per thread.
__device__ __forceinline__ float sum_in_order(const float* p, long long stride, int n, float s) {
    int c = 0;
    for (; c + 8 <= n; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < n; ++c) s += p[(long long)c * stride];
    return s;
}
}  // namespace dir
