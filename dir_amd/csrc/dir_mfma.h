// fp32 matrix-core helper shared by the token kernels (exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain).
#pragma once
#include <hip/hip_runtime.h>

namespace dir {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// acc[m] += A[m*16 .. m*16+15][0..K) * B[0..K)[n0 .. n0+15]   for m < MT
//   A: LDS, row-major, leading dimension lda floats.  lda % 32 == 2 makes the operand reads bank-conflict free
//      (lane -> row l & 15, k-offset l >> 4).
//   B: global, k-major [K][ldb] (column n contiguous): the wave's whole K x 16 fragment is fetched into registers
//      first, so the L2 latency is paid once per tile.
//   D layout: lane holds column n0 + (l & 15), rows m*16 + 4*(l >> 4) + r, r = 0..3.
template <int K, int MT>
__device__ __forceinline__ void mfma_tile_f32(const float* s_a, int lda, const float* __restrict__ Bt, int ldb, int n0,
                                              int lane, f32x4 (&acc)[MT]) {
    const int li = lane & 15, lk = lane >> 4;
    float bv[K / 4];
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk) bv[kk] = Bt[(long long)(4 * kk + lk) * ldb + n0 + li];
    const float* ap = s_a + li * lda + lk;
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[m * 16 * lda + 4 * kk], bv[kk], acc[m], 0, 0, 0);
    }
}

}  // namespace dir
