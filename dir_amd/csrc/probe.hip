// Measurement aid (bench.py `roofline.measured_ceilings`): what THIS board sustains, measured inside the bench run
// instead of quoted from another box.  Two loops, launched back to back by the caller for about a second each:
//   mode 0  bf16 MFMA, operands in registers: 16 x v_mfma_f32_32x32x16_bf16 per wave and iteration on pseudo-random bf16 data (2 x 2 tiles
//           of 32 x 32 per wave: the arithmetic of a 64 x 64 wave tile), 8 waves per workgroup, one workgroup per CU.  Random data matters:
//           on all-zero operands the same loop runs 1.4x faster (the board's power / current limits pull the clock on real data).
//   mode 1  HBM read: 16-byte loads streaming `bytes` of `buf` (make it larger than the 256 MB Infinity Cache).
//   mode 2  HBM copy (round 5): the first half of `buf` copied to the second half with 16-byte loads and stores -- what a
//           streaming kernel does (it reads AND writes); returns bytes read + bytes written.  A read-only loop under-reports the ceiling (5.1 TB/s
//           on these boards, below what bone_vis_kernel achieves).
//           Round 6: the first version (4096 x 256 grid-stride, four loads 16 MB apart, dst exactly 512 MiB behind src)
//           reported 4.1 TB/s where the guide's float4 copy reaches 6.29 and bone_vis_kernel 6.5: power-of-two strides and a power-of-two
//           src -> dst distance put a wave's four streams and its stores on the same channels.  Now: CUs x 2 persistent workgroups, each
//           iteration moves one CONTIGUOUS 16 KB block (4 independent 16-byte loads per lane in flight, 4 KB apart), non-temporal stores
//           (every byte is touched once), dst placed an odd number of 4 KB pages (+ 256 B) behind the middle of the buffer.  Shipped shape after a
//           sweep (tools/probe_sweep.py): 4 loads per lane, 2 workgroups per CU, non-temporal: 6.1 TB/s.
//   mode 3  as mode 0 on the f16 matrix-core instruction (v_mfma_f32_32x32x16_f16) with pseudo-random f16 operands: the f16-storage mode's ceiling.
//   mode 4  (round 6) L2 -> CU "weight stream": EVERY workgroup (CUs x 2, 256 threads) reads the SAME first `bytes` of buf (1 - 2 MB: what a small-map
//           convolution's weights are) over and over, 8 independent 16-byte loads per lane in flight -- the aggregate rate at which the eight L2s can
//           feed all CUs the same stream.  This is the roof of every convolution whose pixel tile is small: a 64-pixel tile does 64 FLOP per weight
//           byte, so at R TB/s of this rate the matrix cores cannot exceed 64 R TFLOP/s on it (balance table).  Returns bytes delivered.
//   mode 5  (round 6) LDS read: every CU's 8 waves issue ds_read_b128 from a 64 KB tile in a loop; returns bytes read.  The other roof of an
//           operand-from-LDS MFMA loop (64 x 64 wave tiles read 1 KB per 32x32x16 MFMA).
// Nothing in the product path calls this.
#include "dir_common.h"

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__global__ __launch_bounds__(512, 1) void probe_mfma_kernel(int iters, float* sink) {
    const unsigned t = blockIdx.x * 512u + threadIdx.x;
    u32x4 fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // bf16 pairs in [1, 2) with pseudo-random mantissas and signs: finite sums, every operand bit toggling
            unsigned h = (t * 8u + i * 4u + q) * 2654435761u;
            u32x4 v, w;
            // (F16: f16 pairs in [1, 2) and [2^-5, 2^-4) with all 10 mantissa bits and the sign pseudo-random)
            const unsigned ba = F16 ? 0x3c003c00u : 0x3f803f80u, bb = F16 ? 0x28002800u : 0x3c003c00u, mk = F16 ? 0x83ff83ffu : 0x807f807fu;
            v.x = ba ^ (h & mk); h = h * 1664525u + 1013904223u;
            v.y = ba ^ (h & mk); h = h * 1664525u + 1013904223u;
            v.z = ba ^ (h & mk); h = h * 1664525u + 1013904223u;
            v.w = ba ^ (h & mk); h = h * 1664525u + 1013904223u;
            w.x = bb ^ (h & mk); h = h * 1664525u + 1013904223u;
            w.y = bb ^ (h & mk); h = h * 1664525u + 1013904223u;
            w.z = bb ^ (h & mk); h = h * 1664525u + 1013904223u;
            w.w = bb ^ (h & mk);
            fa[i][q] = v; fb[i][q] = w;
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][q]), __builtin_bit_cast(f16x8, fb[j][q]), acc[i][j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][q]), __builtin_bit_cast(bf16x8, fb[j][q]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 1.2345f) sink[0] = s;
}

__global__ __launch_bounds__(256) void probe_read_kernel(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// persistent workgroups; block b of the buffer = U x 256 consecutive 16-byte elements, blocks dealt round-robin
template <int U, bool NT>
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t nblocks) {
    for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const size_t base = b * (U * 256) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + base + u * 256) : src[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + base + u * 256); else dst[base + u * 256] = v[u]; }
    }
}
// every workgroup streams the same n16 16-byte elements `reps` times (mode 4)
__global__ __launch_bounds__(256) void probe_l2_stream_kernel(const u32x4* __restrict__ p, size_t n16, int reps, unsigned* sink) {
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + 7 * 256 < n16; i += 8 * 256) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// mode 5: 8 waves per CU read a 64 KB LDS tile with ds_read_b128, `reps` passes of 16 reads per lane
__global__ __launch_bounds__(512, 1) void probe_lds_read_kernel(int reps, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) u32x4 tile[4096];            // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 512) tile[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    unsigned acc = 0;
    const int base = threadIdx.x & 255;
    for (int r = 0; r < reps; ++r) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = tile[base + 256 * u];      // lane-consecutive 16-byte slots: conflict-free
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u].x ^ v[u].z;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
}  // namespace

extern "C" long long dir_probe_launch(int mode, void* buf, long long bytes, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!buf || bytes < 64) { dir::set_error("dir_probe_launch: need a device buffer of at least 64 bytes"); return DIR_E_INVALID; }
    if (mode == 0 || mode == 3) {
        if (iters <= 0) { dir::set_error("dir_probe_launch: iters must be positive"); return DIR_E_INVALID; }
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (mode == 3) hipLaunchKernelGGL(probe_mfma_kernel<true>, dim3(ncu), dim3(512), 0, s, iters, (float*)buf);
        else hipLaunchKernelGGL(probe_mfma_kernel<false>, dim3(ncu), dim3(512), 0, s, iters, (float*)buf);
        if (dir::check_launch("dir_probe_launch") != 0) return DIR_E_LAUNCH;
        return (long long)ncu * 8 * 16 * 32768LL * iters;               // FLOPs of this launch: CUs x waves x MFMAs x 2*32*32*16
    }
    if (mode == 1) {
        const size_t n = (size_t)bytes / 16;
        hipLaunchKernelGGL(probe_read_kernel, dim3(4096), dim3(256), 0, s, (const uint4*)buf, n, (unsigned*)buf);
        if (dir::check_launch("dir_probe_launch") != 0) return DIR_E_LAUNCH;
        const size_t stride = 4096ull * 256, groups = n / (4 * stride);
        return (long long)(groups * 4 * stride * 16);                    // bytes the loop really reads
    }
    if (mode == 2) {
        // iters selects the shape (0 = the default the bench uses): bits 0-3 loads in flight per lane (4 | 8 | 16, 0 = 4), bits 4-7 workgroups per CU
        // (0 = 2), bit 8 plain instead of non-temporal accesses -- tools/probe_sweep.py; measured (profiles/r06_probe_sweep.txt): non-temporal 4 x 2
        // 6.12 TB/s, 5.7 - 6.1 for the other non-temporal shapes, 5.3 - 5.6 with plain accesses; the round-5 kernel 4.1
        if (bytes < (4ll << 20)) { dir::set_error("dir_probe_launch: the copy probe needs at least 4 MiB"); return DIR_E_INVALID; }
        const int U = (iters & 15) ? (iters & 15) : 4, wpc = ((iters >> 4) & 15) ? ((iters >> 4) & 15) : 2;
        const bool nt = !((iters >> 8) & 1);
        const size_t skew = 37 * 4096 + 256;                             // dst starts an odd number of pages (+ 256 B) behind the middle
        const size_t half = (size_t)bytes / 2, blk = (size_t)U * 256 * 16;
        const size_t nblocks = (half - skew) / blk;
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const u32x4* sp = (const u32x4*)buf;
        u32x4* dp = (u32x4*)((char*)buf + half + skew);
        const dim3 grid(ncu * wpc);
#define DIR_CPY(U_) do { if (nt) hipLaunchKernelGGL((probe_copy_kernel<U_, true>), grid, dim3(256), 0, s, sp, dp, nblocks); \
                         else hipLaunchKernelGGL((probe_copy_kernel<U_, false>), grid, dim3(256), 0, s, sp, dp, nblocks); } while (0)
        if (U == 4) DIR_CPY(4); else if (U == 16) DIR_CPY(16); else if (U == 8) DIR_CPY(8);
        else { dir::set_error("dir_probe_launch: copy shape: 4, 8 or 16 loads per lane"); return DIR_E_INVALID; }
#undef DIR_CPY
        if (dir::check_launch("dir_probe_launch") != 0) return DIR_E_LAUNCH;
        return (long long)(nblocks * blk * 2);                           // bytes read + bytes written
    }
    if (mode == 4 || mode == 6) {          // 6: ONE workgroup per CU (no second workgroup whose loads could hit the first one's L1 lines)
        if (iters <= 0 || bytes < 65536) { dir::set_error("dir_probe_launch: the L2 stream probe needs iters > 0 and at least 64 KB"); return DIR_E_INVALID; }
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const size_t n16 = (size_t)bytes / 16, per_pass = (n16 / (8 * 256)) * (8 * 256);
        const int wpc = mode == 4 ? 2 : 1;
        hipLaunchKernelGGL(probe_l2_stream_kernel, dim3(ncu * wpc), dim3(256), 0, s, (const u32x4*)buf, n16, iters, (unsigned*)buf);
        if (dir::check_launch("dir_probe_launch") != 0) return DIR_E_LAUNCH;
        return (long long)((size_t)ncu * wpc * iters * per_pass * 16);
    }
    if (mode == 5) {
        if (iters <= 0) { dir::set_error("dir_probe_launch: iters must be positive"); return DIR_E_INVALID; }
        int dev = 0, ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        hipLaunchKernelGGL(probe_lds_read_kernel, dim3(ncu), dim3(512), 0, s, iters, (unsigned*)buf);
        if (dir::check_launch("dir_probe_launch") != 0) return DIR_E_LAUNCH;
        return (long long)ncu * 512 * 16 * 16 * iters;
    }
    dir::set_error("dir_probe_launch: mode must be 0 (bf16 MFMA), 1 (HBM read), 2 (HBM copy), 3 (f16 MFMA), 4 / 6 (L2 weight stream, 2 / 1 workgroups per CU) or 5 (LDS read)");
    return DIR_E_INVALID;
}
