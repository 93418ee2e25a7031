// SURVEY 8f rank 2 (optimiser half): train.py:227 `optim.AdamW(model.parameters(), cfg.lr)` as ONE launch over a flat parameter
// store.  MI355X-first layout: all 92.7 M parameters, their gradients and the two moment buffers live in four contiguous fp32
// buffers (the model's tensors are views into the first), so the step is a single 16-byte-vectorised streaming kernel --
// 4 reads + 3 writes x 4 B per parameter = 2.6 GB per step, HBM-bound -- instead of 963 tensors x 6 elementwise launches, and the
// flat gradient buffer is exactly the bucket a data-parallel all-reduce wants.
// Arithmetic = torch.optim.AdamW's single-tensor update (decoupled weight decay, bias-corrected moments), fp32, in its operation
// order: p *= 1 - lr wd;  m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
#include "dir_common.h"

namespace dir {
namespace {

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long long n;
    float decay, w1, beta2, w2, bc2_sqrt, eps, step_size;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
#pragma clang fp contract(off)
    p = p * a.decay;
    m = m + a.w1 * (g - m);                         // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (a.w2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2): value * t1 * t2
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    const long long n4 = a.n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        // every byte is touched once per step and the four buffers (1.5 GB) exceed every cache: non-temporal loads / stores
        v4 p = __builtin_nontemporal_load(reinterpret_cast<v4*>(a.p) + i);
        const v4 g = __builtin_nontemporal_load(reinterpret_cast<const v4*>(a.g) + i);
        v4 m = __builtin_nontemporal_load(reinterpret_cast<v4*>(a.m) + i);
        v4 v = __builtin_nontemporal_load(reinterpret_cast<v4*>(a.v) + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = p[e], me = m[e], ve = v[e];
            adam1(pe, g[e], me, ve, a);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
        __builtin_nontemporal_store(p, reinterpret_cast<v4*>(a.p) + i);
        __builtin_nontemporal_store(m, reinterpret_cast<v4*>(a.m) + i);
        __builtin_nontemporal_store(v, reinterpret_cast<v4*>(a.v) + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {            // tail (n not a multiple of 4)
        const long long i = (n4 << 2) + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        adam1(p, a.g[i], m, v, a);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}

}  // namespace
}  // namespace dir

extern "C" int dir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                              double beta1, double beta2, double eps, double weight_decay, long long step, void* stream) {
    using namespace dir;
    DIR_REQUIRE(param && grad && exp_avg && exp_avg_sq, "dir_adamw_step: null pointer");
    DIR_REQUIRE(n > 0 && step >= 1, "dir_adamw_step: n and step must be positive");
    DIR_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                "dir_adamw_step: buffers must be 16-byte aligned");
    // scalar prefactors exactly as torch computes them on the host (python doubles), then rounded to the tensors' dtype
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    a.decay = (float)(1.0 - lr * weight_decay);
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.eps = (float)eps;
    a.step_size = (float)(lr / bc1);
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;                 // 16 workgroups per CU, grid-stride beyond
    if (blocks < 1) blocks = 1;
    DIR_LAUNCH(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dir_adamw_step");
}
