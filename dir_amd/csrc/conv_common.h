// Pieces shared by the two implicit-GEMM convolution kernels (conv.hip: 4-wave general kernel; conv_pipe.hip: 8-wave
// deep-pipelined kernel for the MFMA-bound layers).  gfx950 only.
#pragma once
#include "dir_common.h"

#include <type_traits>

namespace dir {
namespace convk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;
typedef int __attribute__((ext_vector_type(4))) i32x4;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

// 16-byte-per-lane LDS-DMA: global -> LDS without a VGPR destination.  `lds` must be wave-uniform; lane L lands at
// lds + 16*L.  An out-of-range voff writes zeros.  (Kept in a __device__ function: used directly inside the __global__
// template, the host pass silently drops the kernel stub.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// Same transfer, but issued through inline asm so that hipcc's wait-count pass does NOT see it: the compiler would
// otherwise drain every in-flight DMA (vmcnt(0)) before the next ds_read / barrier, which caps the pipeline at one slab
// in flight.  No VGPR is written, so the register hazard of an untracked load does not exist here; completion is
// tracked by hand with counted s_waitcnt vmcnt(N).  M0 (LDS base of the transfer) is saved and restored inside the
// statement.  `lds_addr` = LDS byte address (wave-uniform).
__device__ __forceinline__ void lds_dma16_untracked(i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}
// Variant that leaves M0 pointing at the transfer's LDS base (no save / restore: one SALU write per piece).  gfx9+ DS
// instructions do not read M0, and nothing else in the kernels that use this touches it; M0 is on the clobber list.
__device__ __forceinline__ void lds_dma16_m0(i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
                 : : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory", "m0");
#pragma clang diagnostic pop
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NOTE (kept as a warning): hiding REGISTER-destination loads from the compiler (inline-asm buffer_load +
// hand-counted vmcnt) to deepen the register pipeline is UNSAFE -- hipcc may split / copy the live range of an asm-loaded VGPR
// before the data has landed (intermittent garbage on full grids; tests/test_gpu_conv.py::
// test_conv_full_size_chunk_consistency is the regression test).  The deep pipelines here use LDS-DMA (no VGPR destination).
constexpr int LDS_STRIDE = 144;   // one K-slab row = 128 data bytes (+16 pad)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 pair -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32 (gfx950) instead of ~6 integer ops per value
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
// ReLU on a packed bf16 pair: sign bit set = negative int16 -> 0 (v_pk_max_i16)
__device__ __forceinline__ uint32_t relu2bf(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, v), i16x2_t{0, 0}));
}

// f16 STORAGE (DIR_DT_F16, round 5): feature maps and weights held as IEEE binary16 -- the bytes, addressing, LDS-DMA path and MFMA rate of
// the bf16 mode with an 11-bit significand instead of 8: the rounding of every stored map is 8x finer (the bf16 mode's 0.036 / 0.050 mm at
// the init stage is the accumulated rounding of the backbone's 53 stored maps).  The range is 65504: stores saturate (MODE.FP16_OVFL,
// half_kernel_init) instead of producing inf.  A distinct tag type so that every kernel template instantiates twice; 2 bytes like bf16_t.
struct f16s_t { unsigned short u; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <typename T> struct is_half { static constexpr bool value = std::is_same<T, bf16_t>::value || std::is_same<T, f16s_t>::value; };
constexpr float F16_MAX = 65504.f;

// the two 16-bit storage kinds behind one interface: H = bf16_t | f16s_t
template <typename H> struct Half;
template <> struct Half<bf16_t> {
    static constexpr int DT = DIR_DT_BF16;
    static __device__ __forceinline__ float to_f32(unsigned short v) { return __uint_as_float(((uint32_t)v) << 16); }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack2bf(lo, hi); }
    static __device__ __forceinline__ uint32_t pack2_relu(float lo, float hi) { return relu2bf(pack2bf(lo, hi)); }
    template <typename A, typename B> static __device__ __forceinline__ f32x16 mfma32(const A a, const B b, const f32x16 c) {      // A, B: any 16-byte type
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    template <typename A, typename B> static __device__ __forceinline__ f32x4_t mfma16(const A a, const B b, const f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Half<f16s_t> {
    static constexpr int DT = DIR_DT_F16;
    static __device__ __forceinline__ float to_f32(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
    // Round to nearest even; SATURATING at +-65504 through the wave's MODE.FP16_OVFL bit, which every f16-storage kernel sets first thing
    // (half_kernel_init below): measured on gfx950 (tools/ubench_f16_ovfl.hip) v_cvt_pk_f16_f32 then turns 70000 / 1e6 / 65520 into 0x7bff and
    // keeps a true inf -- a clamp that costs no VALU instruction (the first version clamped with v_med3_f32 per value: +1.5 % on the step).
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
    }
    // ReLU on the packed bit patterns, as for bf16 (sign bit set = negative int16 -> 0)
    static __device__ __forceinline__ uint32_t pack2_relu(float lo, float hi) { return relu2bf(pack2(lo, hi)); }
    template <typename A, typename B> static __device__ __forceinline__ f32x16 mfma32(const A a, const B b, const f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    template <typename A, typename B> static __device__ __forceinline__ f32x4_t mfma16(const A a, const B b, const f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
// First statement of every kernel that STORES the 16-bit kind T: f16 -> MODE.FP16_OVFL = 1 for this wave (hwreg(HW_REG_MODE, 23, 1)): fp32 -> f16
// conversions saturate instead of overflowing to inf.  Nothing for bf16 / fp32.
template <typename T> __device__ __forceinline__ void half_kernel_init() {
    if constexpr (std::is_same<T, f16s_t>::value) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
}
// a packed pair of 16-bit values -> two floats
template <typename H> __device__ __forceinline__ void unpack2(uint32_t u, float& lo, float& hi) {
    lo = Half<H>::to_f32((unsigned short)(u & 0xffffu));
    hi = Half<H>::to_f32((unsigned short)(u >> 16));
}

// f16x3 "split precision" operand tag (DIR_DT_F16X3): tensors are fp32 in memory, every product a*w is evaluated as
//   hi(a)*hi(w) + lo(a)*hi(w) + hi(a)*lo(w),   hi(x) = f16(x), lo(x) = f16(x - hi(x))           (3 x v_mfma_f32_32x32x16_f16, fp32 accumulate)
// hi + lo carries 22 significant bits (the dropped lo*lo and residual terms are ~2^-22 |a w|), against 24 for the exact fp32 MFMA at
// 1/5.3 of its matrix-core time.  The matrix cores honour f16 denormal inputs (tools/ubench_f16_denorm.hip), so lo needs no scaling for
// |a| >= 2^-14 ~ 6e-5 * 2^11; the WEIGHTS are pre-scaled per output channel by a power of two on the host (max |w| -> [2^12, 2^13)) so
// that their lo parts are normal numbers, and split there: the weight "tensor" of this mode is [Cout][kh][kw][Cin/32][hi 32 | lo 32]
// f16 -- byte for byte the size and addressing of the fp32 tensor it replaces.  Activations are split when a K-slab is staged into LDS.
struct f16x3_t { float v; };
// f16x1 (DIR_DT_F16X1): the same data path with the hi parts only -- ONE f16 MFMA per product, operands rounded to f16 (11 significant
// bits, 8x finer than bf16), fp32 accumulation: what torch.autocast(float16) does to a convolution, on fp32 tensors.  The "fp16 MFMA path"
// of BASELINE config 5; weights come in the f16x3 packing (their lo halves are simply not read into the products).
struct f16x1_t { float v; };

// hi parts only (f16x1): {hi01, hi23, 0, 0}
__device__ __forceinline__ uint4 split_f16x1(const uint4 v, const float s) {
    constexpr float FMAX = 65504.f;
    const f32x2_t a = {__builtin_amdgcn_fmed3f(__uint_as_float(v.x) * s, -FMAX, FMAX), __builtin_amdgcn_fmed3f(__uint_as_float(v.y) * s, -FMAX, FMAX)};
    const f32x2_t b = {__builtin_amdgcn_fmed3f(__uint_as_float(v.z) * s, -FMAX, FMAX), __builtin_amdgcn_fmed3f(__uint_as_float(v.w) * s, -FMAX, FMAX)};
    return make_uint4(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, f16x2_t)), __builtin_bit_cast(uint32_t, __builtin_convertvector(b, f16x2_t)), 0u, 0u);
}

// f16x3p / f16x1p (DIR_DT_F16X3P / F16X1P): the ACTIVATIONS arrive pre-split too -- dir_split_f16_forward wrote them once as
// [pixel][C/32][hi 32 | lo 32] f16 (the bytes and addressing of the fp32 tensor), already multiplied by in_scale and through the
// pre-activation -- so both operands travel global -> LDS by DMA like the bf16 path (no register staging, no conversion per N tile).
struct f16x3p_t { float v; };
struct f16x1p_t { float v; };

template <typename T> struct Tr;
template <> struct Tr<float> { static constexpr int EPC = 4, BK = 32; };    // EPC = elems per 16-B chunk
template <> struct Tr<bf16_t> { static constexpr int EPC = 8, BK = 64; };
template <> struct Tr<f16s_t> { static constexpr int EPC = 8, BK = 64; };
template <> struct Tr<f16x3_t> { static constexpr int EPC = 4, BK = 32; };
template <> struct Tr<f16x1_t> { static constexpr int EPC = 4, BK = 32; };
template <> struct Tr<f16x3p_t> { static constexpr int EPC = 4, BK = 32; };
template <> struct Tr<f16x1p_t> { static constexpr int EPC = 4, BK = 32; };

// four fp32 values, times the power of two s, clamped to the f16 range -> {hi01, hi23, lo01, lo23} as packed f16 pairs (round to
// nearest even both times)
__device__ __forceinline__ uint4 split_f16x3(const uint4 v, const float s) {
    constexpr float FMAX = 65504.f;
    const f32x2_t a = {__builtin_amdgcn_fmed3f(__uint_as_float(v.x) * s, -FMAX, FMAX), __builtin_amdgcn_fmed3f(__uint_as_float(v.y) * s, -FMAX, FMAX)};
    const f32x2_t b = {__builtin_amdgcn_fmed3f(__uint_as_float(v.z) * s, -FMAX, FMAX), __builtin_amdgcn_fmed3f(__uint_as_float(v.w) * s, -FMAX, FMAX)};
    const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);
    const f16x2_t la = __builtin_convertvector(ra, f16x2_t), lb = __builtin_convertvector(rb, f16x2_t);
    return make_uint4(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb), __builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

struct ConvArgs {
    const void* x; const void* w; const float* scale; const float* shift;
    const float* pre_scale; const float* pre_shift; const void* res; void* y;
    int B, H, W, Cin, in_cs, in_co, Cout, out_cs, out_co, res_cs, res_co;
    int kh, kw, stride, pad, Ho, Wo, M, K, nk, tiles_m, tiles_n, flags;
    int variant;                        // DIR_CONV_VARIANT code (0 = heuristic)
    float a_scale;                      // f16x3: power of two applied to the activations before the hi / lo split (dir_conv_desc.in_scale)
    float out_split_scale;              // > 0 (fp32 outputs only): write y as the NEXT convolution's pre-split operand ([pixel][Cout/32][hi 32 | lo 32]
    int out_split_hi_only;              //   f16 of act(y) * out_split_scale; dir_conv_desc.out_split_scale) instead of fp32 -- the same bytes, no pass
    // optional second source, a 1x1 (strided) convolution accumulated into the same output: K-slabs ks >= nk1 read x2;
    // weight rows are [kh*kw*Cin | Cin2] (dir_conv2d_dual_forward)
    const void* x2; unsigned x2_bytes; int H2, W2, in_cs2, in_co2, stride2, nk1;
    unsigned x_bytes, w_bytes;          // buffer sizes for the hardware bounds check
    unsigned mg_hw, sh_hw, mg_w, sh_w;  // magic multipliers: m / (Ho*Wo) and r / Wo for 0 <= m < 2^31 (set_magic)
    long long* stamps;                  // DIR_STAMPS=conv_pipe (tuning aid, else NULL): phase times of the first / last workgroup
    const int* bbox; int bbox_groups;   // optional [B][bbox_groups][4] = ymin,ymax,xmin,xmax of the non-zero support of
                                        // each 64-channel input group; K-slabs that cannot touch a tile are skipped
    // split-K (dir_conv2d_splitk_forward): `splits` workgroups (blockIdx.y) share one output tile, each reducing a contiguous range
    // of K-slabs; raw fp32 partial tiles go to ws_part [split][tile][128*128], the last workgroup to arrive at ws_cnt[tile] sums
    // them in split order (deterministic) and runs the epilogue.  0 / 1 = off.
    int splits; float* ws_part; unsigned* ws_cnt;
    // round 5 (dir_conv2d_forward_stats): the chunk partials of the BatchNorm (training mode) that FOLLOWS this convolution, from the output tile while
    // it is still in registers -- st_p1 [tiles_m][Cout] = per tile row and output channel the sum of the tile's valid rows of y, st_p2 = sum (y - tile
    // mean)^2 (what bn_stats4_kernel forms from the stored map with 256-row chunks; here the chunk is the M tile).  NULL = off.
    float* st_p1 = nullptr; float* st_p2 = nullptr;
    // round 5 (dir_conv2d_forward_masked, fp32 outputs): y = mask > 0 ? (conv + residual) : 0 with mask a tensor of y's shape (row stride out_cs, offset
    // out_co) -- the ReLU backward of the block that PRODUCED this convolution's input, applied where a data-gradient convolution writes that
    // input's gradient (the mask is the block's stored output): the separate pass over (gradient, output) -> masked gradient goes away.  NULL = off.
    const float* mask = nullptr;
    // round 5 (dir_conv2d_forward_ex): this convolution is a DATA-GRADIENT convolution whose output g [M][Cout] is the gradient of a training-mode
    // BatchNorm's (+ ReLU's) output; bs_z [M][Cout] is that BatchNorm's input, bs_mu / bs_rs / bs_w / bs_b its statistics and affine.  The
    // epilogue then also forms the BatchNorm backward's chunk partials over the tile's rows -- bs_p1 [tiles_m][Cout] = sum of g under the ReLU mask
    // (bs_relu: BatchNorm(z) > 0), bs_p2 = sum of the same times xhat = (z - mean) rstd -- what bn_bwd_partial4_kernel forms from the stored maps.
    const float* bs_z = nullptr; const float* bs_mu = nullptr; const float* bs_rs = nullptr; const float* bs_w = nullptr; const float* bs_b = nullptr;
    int bs_relu = 0; float* bs_p1 = nullptr; float* bs_p2 = nullptr;
};
// rows of the M tile the last launch on this thread formed the statistics over (0: the kernel that took the launch does not form them)
extern thread_local int stats_rows_launched;

constexpr int MAX_SLABS = 768;

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void store_out<f16s_t>(f16s_t* p, float v) { p->u = (unsigned short)(Half<f16s_t>::pack2(v, 0.f) & 0xffffu); }
template <typename TO> __device__ __forceinline__ float load_res(const TO* p);
template <> __device__ __forceinline__ float load_res<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_res<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float load_res<f16s_t>(const f16s_t* p) { return Half<f16s_t>::to_f32(p->u); }

template <typename TI>
__device__ __forceinline__ void mma_slab(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc);
template <>
__device__ __forceinline__ void mma_slab<float>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[q].x), __uint_as_float(bf[q].x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[q].y), __uint_as_float(bf[q].y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[q].z), __uint_as_float(bf[q].z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[q].w), __uint_as_float(bf[q].w), acc, 0, 0, 0);
    }
}
template <>
__device__ __forceinline__ void mma_slab<bf16_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[q]),
                                                      __builtin_bit_cast(bf16x8, bf[q]), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slab<f16s_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = Half<f16s_t>::mfma32(af[q], bf[q], acc);
}
template <>
__device__ __forceinline__ void mma_slab<f16x3p_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc);
template <>
__device__ __forceinline__ void mma_slab<f16x1p_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc);
// f16x1: the hi fragments only
template <>
__device__ __forceinline__ void mma_slab<f16x1_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[2 * s]), __builtin_bit_cast(f16x8, bf[2 * s]), acc, 0, 0, 0);
}
// f16x3: fragment q = 2*s + part of a 32-channel slab: k16-step s in {0, 1}, part 0 = hi, 1 = lo (both operands laid out alike)
template <>
__device__ __forceinline__ void mma_slab<f16x3_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 ah = __builtin_bit_cast(f16x8, af[2 * s]), al = __builtin_bit_cast(f16x8, af[2 * s + 1]);
        const f16x8 bh = __builtin_bit_cast(f16x8, bf[2 * s]), bl = __builtin_bit_cast(f16x8, bf[2 * s + 1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    }
}
template <>
__device__ __forceinline__ void mma_slab<f16x3p_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) { mma_slab<f16x3_t>(af, bf, acc); }
template <>
__device__ __forceinline__ void mma_slab<f16x1p_t>(const uint4 (&af)[4], const uint4 (&bf)[4], f32x16& acc) { mma_slab<f16x1_t>(af, bf, acc); }


// Exact unsigned division of n < 2^31 by a launch constant d as one 64-bit multiply and a shift: sh = 31 + ceil(log2 d),
// mg = ceil(2^sh / d) < 2^32 (Granlund-Montgomery round-up; n * mg < 2^63).  The tile setup of every workgroup decomposes
// BM output-pixel indices into (image, row, column); with hardware-less integer division that was ~40 VALU instructions each.
inline void magic_u31(unsigned d, unsigned* mg, unsigned* sh) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    *sh = 31 + l;
    *mg = (unsigned)(((1ull << *sh) + d - 1) / d);
}
inline void set_magic(ConvArgs& a) {
    magic_u31((unsigned)(a.Ho * a.Wo), &a.mg_hw, &a.sh_hw);
    magic_u31((unsigned)a.Wo, &a.mg_w, &a.sh_w);
}
__device__ __forceinline__ int div_magic(int n, unsigned mg, unsigned sh) { return (int)(((unsigned long long)(unsigned)n * mg) >> sh); }

// output pixel m -> byte offset of its first tap (may be negative) and the bit mask of taps inside the image
__device__ __forceinline__ void pixel_setup(const ConvArgs& a, int m, int col_bytes, int es, int& voff, unsigned& msk, int& b, int& oy, int& ox) {
    b = div_magic(m, a.mg_hw, a.sh_hw);
    const int rem = m - b * (a.Ho * a.Wo);
    oy = div_magic(rem, a.mg_w, a.sh_w);
    ox = rem - oy * a.Wo;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    voff = (((b * a.H + iy0) * a.W + ix0) * a.in_cs + a.in_co) * es + col_bytes;
    unsigned colbits = 0;
    for (int kx = 0; kx < a.kw; ++kx) colbits |= (unsigned)(ix0 + kx >= 0 && ix0 + kx < a.W) << kx;
    msk = 0;
    for (int ky = 0; ky < a.kh; ++ky)
        if (iy0 + ky >= 0 && iy0 + ky < a.H) msk |= colbits << (ky * a.kw);
}

// 16-byte vector of output elements (coalesced epilogue)
template <typename TO> struct OutVec;
template <> struct OutVec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4 t, float (&v)[4]) {
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void store_act(float* p, const float (&v)[4], bool relu) {
        *reinterpret_cast<float4*>(p) = relu ? make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f))
                                             : make_float4(v[0], v[1], v[2], v[3]);
    }
};
// four consecutive output channels n .. n+3 of pixel m as the pre-split operand of the next convolution (out_cs == Cout, out_co == 0)
__device__ __forceinline__ void store_split4(float* y, long long m, int n, int Cout, const float (&v)[4], bool relu, float s, bool hi_only) {
    uint4 t = relu ? make_uint4(__float_as_uint(fmaxf(v[0], 0.f)), __float_as_uint(fmaxf(v[1], 0.f)), __float_as_uint(fmaxf(v[2], 0.f)), __float_as_uint(fmaxf(v[3], 0.f)))
                   : make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    const uint4 sp = hi_only ? split_f16x1(t, s) : split_f16x3(t, s);
    char* row = reinterpret_cast<char*>(y) + (m * Cout + (n & ~31)) * 4;
    *reinterpret_cast<uint2*>(row + 2 * (n & 31)) = make_uint2(sp.x, sp.y);
    *reinterpret_cast<uint2*>(row + 64 + 2 * (n & 31)) = make_uint2(sp.z, sp.w);
}

// 16-bit storage kinds (8 values per 16-byte vector)
template <typename H> struct OutVecHalf {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const H* p, float (&v)[8]) { unpack(*reinterpret_cast<const uint4*>(p), v); }
    static __device__ __forceinline__ void unpack(const uint4 t, float (&v)[8]) {
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) unpack2<H>(u[e], v[2 * e], v[2 * e + 1]);
    }
    static __device__ __forceinline__ void store(H* p, const float (&v)[8]) {
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = Half<H>::pack2(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(u[0], u[1], u[2], u[3]);
    }
    // round, then ReLU on the packed pairs (rounding is monotonic and sign-preserving: same result as ReLU in fp32 first)
    static __device__ __forceinline__ void store_act(H* p, const float (&v)[8], bool relu) {
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = relu ? Half<H>::pack2_relu(v[2 * e], v[2 * e + 1]) : Half<H>::pack2(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(u[0], u[1], u[2], u[3]);
    }
};
template <> struct OutVec<bf16_t> : OutVecHalf<bf16_t> {};
template <> struct OutVec<f16s_t> : OutVecHalf<f16s_t> {};


// pre-activation BN (+ReLU) applied to one 16-byte chunk of input channels starting at channel c
template <typename TI>
__device__ __forceinline__ uint4 prologue(uint4 v, const float* ps, const float* pb, int c, bool relu);
template <>
__device__ __forceinline__ uint4 prologue<float>(uint4 v, const float* ps, const float* pb, int c, bool relu) {
    float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = fmaf(f[e], ps[c + e], pb[c + e]);
        if (relu) f[e] = fmaxf(f[e], 0.f);
    }
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <>
__device__ __forceinline__ uint4 prologue<f16x3_t>(uint4 v, const float* ps, const float* pb, int c, bool relu) { return prologue<float>(v, ps, pb, c, relu); }
template <>
__device__ __forceinline__ uint4 prologue<f16x1_t>(uint4 v, const float* ps, const float* pb, int c, bool relu) { return prologue<float>(v, ps, pb, c, relu); }
template <>
__device__ __forceinline__ uint4 prologue<f16x3p_t>(uint4 v, const float*, const float*, int, bool) { return v; }      // (never instantiated with PRE)
template <>
__device__ __forceinline__ uint4 prologue<f16x1p_t>(uint4 v, const float*, const float*, int, bool) { return v; }
template <typename H>
__device__ __forceinline__ uint4 prologue_half(uint4 v, const float* ps, const float* pb, int c, bool relu) {
    uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack2<H>(u[e], lo, hi);
        lo = fmaf(lo, ps[c + 2 * e], pb[c + 2 * e]);
        hi = fmaf(hi, ps[c + 2 * e + 1], pb[c + 2 * e + 1]);
        if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
        u[e] = Half<H>::pack2(lo, hi);
    }
    return make_uint4(u[0], u[1], u[2], u[3]);
}
template <>
__device__ __forceinline__ uint4 prologue<bf16_t>(uint4 v, const float* ps, const float* pb, int c, bool relu) { return prologue_half<bf16_t>(v, ps, pb, c, relu); }
template <>
__device__ __forceinline__ uint4 prologue<f16s_t>(uint4 v, const float* ps, const float* pb, int c, bool relu) { return prologue_half<f16s_t>(v, ps, pb, c, relu); }

// Column statistics of the output tile (values fmaf(acc, scale, shift): what the epilogue stores when there is no residual / activation) over
// the tile's valid rows, two passes over the accumulators (sum -> tile mean -> sum of squared deviations), combined across the two lane
// halves by a cross-lane move and across the WM waves of a column through `red` (LDS, 2 * WM * BN floats, free at the epilogue's start) in wave
// order: deterministic.  Ends with a barrier: `red` may be overwritten by the staging that follows.
template <int MI, int NJ, int WM, int WN>
__device__ __forceinline__ void tile_col_stats(const ConvArgs& a, f32x16 (&acc)[MI][NJ], float* red, int m0, int n0, int wm, int wn, int lane) {
    constexpr int BM = 32 * MI * WM, BN = 32 * NJ * WN;
    const int nvalid = min(BM, a.M - m0), tm = m0 / BM, l32 = lane & 31, rbase = wm * MI * 32 + 4 * (lane >> 5);
    float sc[NJ], sh[NJ], mean[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cl = wn * NJ * 32 + j * 32 + l32, n = n0 + cl;
        sc[j] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        sh[j] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rbase + i * 32 + (r & 3) + 8 * (r >> 2) < nvalid) t += fmaf(acc[i][j][r], sc[j], sh[j]);
        t += __shfl_xor(t, 32);
        if (lane < 32) red[wm * BN + cl] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cl = wn * NJ * 32 + j * 32 + l32;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += red[w * BN + cl];
        mean[j] = t / (float)nvalid;
        if (wm == 0 && lane < 32 && n0 + cl < a.Cout) a.st_p1[(long long)tm * a.Cout + n0 + cl] = t;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rbase + i * 32 + (r & 3) + 8 * (r >> 2) < nvalid) { const float d = fmaf(acc[i][j][r], sc[j], sh[j]) - mean[j]; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 32);
        if (lane < 32) red[(WM + wm) * BN + cl] = q;
    }
    __syncthreads();
    if (wm == 0 && lane < 32) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cl = wn * NJ * 32 + j * 32 + l32;
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) q += red[(WM + w) * BN + cl];
            if (n0 + cl < a.Cout) a.st_p2[(long long)tm * a.Cout + n0 + cl] = q;
        }
    }
    __syncthreads();
}

// BatchNorm-backward chunk partials in the fp32 epilogues (ConvArgs::bs_*).  A thread of the store loop owns 4 consecutive columns (fixed) and a
// strided set of rows: it keeps the two sums of its columns in registers, the threads of a column group are then added in LDS in thread order
// (deterministic) and one thread per column writes the tile's partial.
struct BnBwdAcc { float t1[4], t2[4], mu[4], rs[4], g[4], be[4]; };
__device__ __forceinline__ void bn_bwd_acc_init(const ConvArgs& a, int n, BnBwdAcc& s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s.t1[e] = s.t2[e] = 0.f; s.mu[e] = 0.f; s.rs[e] = 0.f; s.g[e] = 1.f; s.be[e] = 0.f; }
    if (n + 3 < a.Cout) {
        const float4 m = *reinterpret_cast<const float4*>(a.bs_mu + n), k = *reinterpret_cast<const float4*>(a.bs_rs + n);
        s.mu[0] = m.x; s.mu[1] = m.y; s.mu[2] = m.z; s.mu[3] = m.w; s.rs[0] = k.x; s.rs[1] = k.y; s.rs[2] = k.z; s.rs[3] = k.w;
        if (a.bs_w) { const float4 t = *reinterpret_cast<const float4*>(a.bs_w + n); s.g[0] = t.x; s.g[1] = t.y; s.g[2] = t.z; s.g[3] = t.w; }
        if (a.bs_b) { const float4 t = *reinterpret_cast<const float4*>(a.bs_b + n); s.be[0] = t.x; s.be[1] = t.y; s.be[2] = t.z; s.be[3] = t.w; }
    }
}
__device__ __forceinline__ void bn_bwd_acc_add(const ConvArgs& a, long long m, int n, const float (&v)[4], BnBwdAcc& s) {
    const float4 zz = *reinterpret_cast<const float4*>(a.bs_z + m * a.Cout + n);
    const float z[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (z[e] - s.mu[e]) * s.rs[e];
        float gm = v[e];
        if (a.bs_relu && !(fmaf(xh, s.g[e], s.be[e]) > 0.f)) gm = 0.f;          // (bn_value of train_ops.hip: the mask the BatchNorm backward re-computes)
        s.t1[e] += gm;
        s.t2[e] = fmaf(gm, xh, s.t2[e]);
    }
}
template <int NT, int BN>
__device__ __forceinline__ void bn_bwd_acc_finish(const ConvArgs& a, float* red, int tid, int tm, int n0, const BnBwdAcc& s) {
    constexpr int CPR = BN / 4, G = NT / CPR;
    static_assert(NT % CPR == 0, "a thread's columns must not change between its rows");
    __syncthreads();                                       // every thread is done reading the staged tile: red may overwrite it
    const int grp = tid / CPR, cc = (tid - grp * CPR) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[grp * BN + cc + e] = s.t1[e]; red[(G + grp) * BN + cc + e] = s.t2[e]; }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q) { u1 += red[q * BN + c]; u2 += red[(G + q) * BN + c]; }
        if (n0 + c < a.Cout) { a.bs_p1[(long long)tm * a.Cout + n0 + c] = u1; a.bs_p2[(long long)tm * a.Cout + n0 + c] = u2; }
    }
}

// Epilogue of the 8-wave kernels (conv_pipe.hip, bonefuse.hip): tile of (32*MI*WM) x (32*NJ*WN), wave (wm, wn).
// scale/shift in registers -> fp32 tile in LDS -> 16-byte row segments (+ residual, ReLU) to HBM (conv.hip's epilogue)
template <typename TO, int MI, int NJ, int WM, int WN>
__device__ __forceinline__ void epilogue_tile(const ConvArgs& a, f32x16 (&acc)[MI][NJ], char* smem, int m0, int n0, int wm, int wn,
                                         int tid, int lane) {
    constexpr int NT = 64 * WM * WN, BM = 32 * MI * WM, BN = 32 * NJ * WN;
    TO* __restrict__ y = (TO*)a.y;
    const TO* __restrict__ res = (const TO*)a.res;
    const bool relu = (a.flags & 1) != 0;
    float* st = reinterpret_cast<float*>(smem);
    if constexpr (std::is_same<TO, float>::value) {
        if (a.st_p1) tile_col_stats<MI, NJ, WM, WN>(a, acc, st, m0, n0, wm, wn, lane);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cl = wn * NJ * 32 + j * 32 + (lane & 31);
        const int n = n0 + cl;
        const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        const float sh = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * MI * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                st[rl * BN + cl] = fmaf(acc[i][j][r], sc, sh);
            }
    }
    __syncthreads();
    constexpr int VN = OutVec<TO>::N, CPR = BN / VN;
    BnBwdAcc bs;
    const bool bs_on = std::is_same<TO, float>::value && a.bs_p1 != nullptr;
    if constexpr (std::is_same<TO, float>::value) {
        if (bs_on) bn_bwd_acc_init(a, n0 + (tid % CPR) * VN, bs);
    }
    for (int c = tid; c < BM * CPR; c += NT) {
        const int rl = c / CPR, cc = (c - rl * CPR) * VN;
        const int m = m0 + rl, n = n0 + cc;
        if (m >= a.M || n >= a.Cout) continue;
        float v[VN];
        const float4* sp = reinterpret_cast<const float4*>(st + rl * BN + cc);
#pragma unroll
        for (int q = 0; q < VN / 4; ++q) { const float4 t = sp[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
        if (res) {
            float rv[VN];
            OutVec<TO>::load(res + (long long)m * a.res_cs + a.res_co + n, rv);
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] += rv[e];
        }
        if constexpr (std::is_same<TO, float>::value) {
            if (a.mask) {
                const float4 mk = *reinterpret_cast<const float4*>(a.mask + (long long)m * a.out_cs + a.out_co + n);
                v[0] = mk.x > 0.f ? v[0] : 0.f; v[1] = mk.y > 0.f ? v[1] : 0.f; v[2] = mk.z > 0.f ? v[2] : 0.f; v[3] = mk.w > 0.f ? v[3] : 0.f;
            }
            if (bs_on) bn_bwd_acc_add(a, m, n, v, bs);
            if (a.out_split_scale > 0.f) { store_split4(y, m, n, a.Cout, v, relu, a.out_split_scale, a.out_split_hi_only != 0); continue; }
        }
        OutVec<TO>::store_act(y + (long long)m * a.out_cs + a.out_co + n, v, relu);
    }
    if constexpr (std::is_same<TO, float>::value) {
        if (bs_on) bn_bwd_acc_finish<NT, BN>(a, st, tid, m0 / BM, n0, bs);
    }
}

// Tile order of a launch (flags bit 8).  The XCD-aware remap hands every XCD one contiguous range of nwg / 8 tiles.  Row-major (tn fastest) makes
// that range whole ROWS of tiles: the XCD streams every weight row through its 4 MB L2 but only its own pixels; column-major makes it whole
// COLUMNS: every pixel, its own weight rows.  The order with fewer unique operand bytes per XCD is taken (column-major only when it saves a
// fifth): the attention conv at B = 64 (16 x 16 tiles, 75 MB of weights, 17 MB of activations) goes from 77 MB to 26 MB per XCD.
constexpr int CONV_COL_MAJOR = 8;
inline void choose_tile_order(ConvArgs& a, int es) {
    a.flags &= ~CONV_COL_MAJOR;
    static const int forced = getenv("DIR_TILE_ORDER") ? atoi(getenv("DIR_TILE_ORDER")) : -1;      // tuning aid: 0 = always row-major, 1 = always column-major
    if (a.bbox || a.tiles_m <= 0 || a.tiles_n <= 0 || forced == 0) return;
    if (forced == 1) { a.flags |= CONV_COL_MAJOR; return; }
    const double nwg = (double)a.tiles_m * a.tiles_n, per = (nwg + 7.0) / 8.0;
    const double abytes = (double)a.B * a.H * a.W * a.Cin * es + (a.x2 ? (double)a.x2_bytes : 0.0), wbytes = (double)a.Cout * a.K * es;
    auto frac = [](double num, double den) { const double f = num / den; return f < 1.0 ? f : 1.0; };
    auto ceil_div = [](double x, double y) { const long long q = (long long)(x / y); return (double)(q * y < x ? q + 1 : q); };
    const double cost_row = abytes * frac(ceil_div(per, a.tiles_n), a.tiles_m) + wbytes * frac(per, a.tiles_n);
    const double cost_col = wbytes * frac(ceil_div(per, a.tiles_m), a.tiles_n) + abytes * frac(per, a.tiles_m);
    if (cost_col < 0.8 * cost_row) a.flags |= CONV_COL_MAJOR;
    static const int log = getenv("DIR_ORDER_LOG") ? atoi(getenv("DIR_ORDER_LOG")) : 0;          // tuning aid: one line per launch
    if (log)
        fprintf(stderr, "tile_order M=%d N=%d K=%d tiles %dx%d %s: per-XCD unique MB row-major %.1f col-major %.1f, x8 = %.1f MB against operands %.1f MB\n", a.M, a.Cout, a.K,
                a.tiles_m, a.tiles_n, (a.flags & CONV_COL_MAJOR) ? "COL" : "row", cost_row / 1e6, cost_col / 1e6,
                8.0 * ((a.flags & CONV_COL_MAJOR) ? cost_col : cost_row) / 1e6, (abytes + wbytes) / 1e6);
}
// (tm, tn) of workgroup `bid` (after the XCD remap)
__device__ __forceinline__ void tile_of(const ConvArgs& a, int bid, int& tm, int& tn) {
    if (a.flags & CONV_COL_MAJOR) { tn = bid / a.tiles_m; tm = bid - tn * a.tiles_m; }
    else { tm = bid / a.tiles_n; tn = bid - tm * a.tiles_n; }
}

// conv_big.hip (DIR_CONV_VARIANT 11, the 256 x 256 block tile): returns true if it took the launch
bool launch_conv_big(const ConvArgs& a, bool out_f32, hipStream_t s, bool f16 = false);
// conv_pipe.hip: returns true if it took the launch (bf16 input, no pre-activation, long reduction, enough tiles)
bool launch_conv_pipe(const ConvArgs& a, bool out_f32, int num_cu, hipStream_t s, int xm = 0, bool f16 = false);      // xm 3 / 1: both operands pre-split f16 (F16X3P / F16X1P); f16: DIR_DT_F16 storage instead of bf16

}  // namespace convk
}  // namespace dir
