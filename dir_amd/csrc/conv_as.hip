// dir_conv2d_as_forward: the convolutions on SMALL maps (8x8, 16x16, 32x32: ResNet layer3 / layer4, the decoder's first stage) as an
// ACTIVATION-STATIONARY kernel (round 6, VERDICT r5 item 2: "the 33 launches with M <= 16 384 are 58 % of the convolution time at ~0.10 of the
// MFMA peak -- change the structure").
//   models/backbone/resnet.py:120-140     Bottleneck conv1 / conv2 (3x3) / conv3 (+ identity) at 16x16 and 8x8
//   models/backbone/hourglass.py:55-70    Residual conv2 (3x3) and the 1x1 layers without pre-activation
//   models/dir.py:227-241                 the attention / head convolutions on c4
// What the tiled implicit-GEMM kernels (conv.hip, conv_pipe.hip) do on these layers: one 128 x 128 tile per CU and launch, every K-slab of BOTH
// operands through an LDS ring (two or three barriers per 64 channels), a pipeline that has no second tile to overlap its fill and its epilogue
// with -- 39 - 64 us for 7.7 us of MFMA work (profiles/r05_h_bench_line_one_in_flight.txt).  Here:
//   * a workgroup owns 32 PB pixels of ONE image (PB x 32 / W whole rows) x 128 A output channels.  Its whole input patch -- every input channel,
//     the 3x3 halo included, zeros outside the image -- goes global -> LDS ONCE by DMA (16 x 16 x 256 @ 3x3: 57 KB; 8 x 8 x 512: 102 KB) and stays;
//   * the weights never touch LDS: the GEMM is D[channel][pixel] (weights = MFMA A operand), wave w owns channels [32 A w, 32 A (w + 1)) of the
//     workgroup's slice, and the host packs every wave's fragments in exactly the order its MFMAs consume them (dir_amd/engine.py::pack_as_weights),
//     so the K loop is: one coalesced 1 KB load per fragment into a register ring NSTG - 1 steps ahead, ds_read_b128 of the pixels' channels,
//     MFMA -- NO barrier and NO LDS write between the first and the last MFMA of the workgroup;
//   * per 32x32x16 MFMA a wave reads 1 / A KB from LDS and 1 / PB KB from L2 (the tiled kernels' 64 x 64 wave tiles: 1 KB from LDS, and every byte of
//     it was first written there), so neither port is the limit at A = PB = 2;
//   * epilogue as conv.hip's: fmaf(acc, scale, shift) staged as fp32 in LDS (the patch is dead by then), + residual, round, ReLU, 16-byte stores.
// K order and k-slot assignment are conv.hip's (64-channel slab outer, taps inner; MFMA ks of a slab multiplies channels 8 ks .. + 8 in lanes 0-31 and
// 32 + 8 ks .. + 8 in lanes 32-63; taps outside the image multiply zeros, as the tiled kernels' hardware-zeroed halo does): the same fp32 chains, so
// the outputs are bit-identical and the engine's autotune may pick this kernel per layer (DIR_CONV_VARIANT 25 .. 28).
#include "conv_common.h"

namespace dir {
namespace {

using convk::bf16_t;
using convk::f16s_t;
using convk::f32x16;
using convk::Half;
using convk::OutVec;

constexpr unsigned OOB = 0x80000000u;
constexpr int AS_THR = 256;

struct AsArgs {
    const void* x; const uint4* w; const float* scale; const float* shift; const void* res; void* y;
    int B, H, W, logW, Cin, in_cs, in_co, Cout, out_cs, out_co, res_cs, res_co;
    int TR;                            // image rows per workgroup tile (32 PB / W)
    int tiles_per_img, ntile, nslice, nsteps, relu, xcd_map;
    int KC, nchunk, csteps, chunk_bytes;   // patch chunk: channels resident at a time (Cin when the whole patch fits), chunks, steps per chunk, LDS bytes of one ring buffer
    int q, PW, NP, ninstr;             // 16-byte chunks per patch position (KC / 8); patch width; patch positions; 1 KB DMA instructions per patch chunk
    unsigned mg_q, sh_q, mg_pw, sh_pw; // magic divisors (conv_common.h: magic_u31) of q + 1 and PW
    unsigned x_bytes;
    long long* stamps;                 // DIR_STAMPS=conv_as (tuning aid, else NULL): phase times of workgroup 0
};

// A: 32-channel blocks per wave (workgroup = 128 A output channels); PB: 32-pixel blocks per workgroup; K3: 3x3 / pad 1 (else 1x1);
// NSTG: depth of the weight ring in steps (a step = one (64-channel slab, tap) = 4 k-steps = 4 A fragments per wave)
// CHK: the patch does not fit LDS as a whole (the attention convolution: 100 positions x 2048 channels = 411 KB): it is brought in KC channels at a
// time into a ring of TWO buffers -- chunk c + 1's DMA is issued (from source offsets kept in registers: one add per instruction) while chunk c is
// multiplied; one wait + barrier per chunk.  The K order is unchanged (slab outer, taps inner: a chunk is KC / 64 whole slabs).
template <typename H, int A, int PB, bool K3, int NSTG, bool CHK = false>
__global__ __launch_bounds__(AS_THR, (A * PB <= 4 && NSTG <= 3 && !CHK) ? 2 : 1) void conv_as_kernel(AsArgs a) {
    convk::half_kernel_init<H>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWG = 128 * A, TPX = 32 * PB, NTAP = K3 ? 9 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, h = lane >> 5;
    int nstamp = 0;
    auto stamp = [&]() {            // workgroup 0: slots 0.., the last workgroup: slots 6.. (all relative to workgroup 0's first stamp when printed)
        if (a.stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) a.stamps[(blockIdx.x == 0 ? 0 : 6) + nstamp++] = (long long)__builtin_amdgcn_s_memtime();
    };
    stamp();

    // ---- (pixel tile, output-channel slice) of this workgroup.  Workgroup b runs on XCD b % 8 (observed dispatch): the slices of one pixel tile are
    //      given to ONE XCD, so the patch is fetched from HBM / the Infinity Cache once and re-read through that XCD's L2
    int tile, slice;
    {
        const int bid = blockIdx.x;
        if (a.xcd_map) {
            const int xcd = bid & 7, j = bid >> 3, tl = j / a.nslice;
            slice = j - tl * a.nslice;
            tile = tl * 8 + xcd;
        } else {
            tile = bid / a.nslice;
            slice = bid - tile * a.nslice;
        }
    }
    const int img = tile / a.tiles_per_img, trow = (tile - img * a.tiles_per_img) * a.TR;

    // ---- the patch: global -> LDS by DMA.  Position p (patch row p / PW, column p % PW) occupies PITCH = 2 Cin + 16 bytes from p * PITCH: the
    //      16-byte skew puts the SAME channel chunk of 16 consecutive positions into sixteen different 16-byte bank groups, and it keeps every
    //      read address ADDITIVE in (position, slab, k-step): one v_add per 32 pixels and step, the k-steps as instruction offsets.  (The first
    //      version kept rows unpadded and XOR-swizzled the chunk index, as the tiled kernels do for their 128-byte rows: ~14 VALU instructions per
    //      32 pixels and step beside 8 MFMAs, and a key that ignored ds_read_b128's lane groups -- SQ_LDS_BANK_CONFLICT was 47 - 74 % of the
    //      LDS-active cycles, profiles/r06_a_fwd_sq_counters.txt.)  The DMA sees the padded patch as a linear run of 16-byte slots, 64 per
    //      instruction: slot s = (position s / (q + 1), chunk s % (q + 1)); the pad chunk and positions outside the image fetch out of range = zeros.
    const int CB = a.KC * 2, PITCH = CB + 16;
    const convk::i32x4 xd = {(int)(unsigned)(unsigned long long)a.x, (int)(unsigned)((unsigned long long)a.x >> 32), (int)a.x_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    constexpr int MAXI = CHK ? 20 : 1;             // CHK: DMA instructions per wave and chunk whose source offsets are kept (launcher: ninstr <= 4 MAXI)
    unsigned voff0[MAXI];
    auto slot_voff = [&](int i) -> unsigned {     // source byte offset of lane's slot of instruction i (chunk 0), or OOB
        const int slot = i * 64 + lane;
        const int pos = convk::div_magic(slot, a.mg_q, a.sh_q), ch = slot - pos * (a.q + 1);
        const int pr = convk::div_magic(pos, a.mg_pw, a.sh_pw), pc = pos - pr * a.PW;
        const int iy = trow + pr - (K3 ? 1 : 0), ix = pc - (K3 ? 1 : 0);
        const bool ok = pos < a.NP && ch < a.q && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        // (measured and dropped, round 6: advancing (position, chunk, row, column) with adds and carries instead of dividing every slot made the
        //  prologue SLOWER, 3.9 -> 5.3 k ticks: the carried chain serialises what are otherwise independent iterations.  DMA issue alone: 1.4 - 2.6 k.)
        return ok ? (unsigned)((((img * a.H + iy) * a.W + ix) * a.in_cs + a.in_co) * 2 + (ch << 4)) : OOB;
    };
    if constexpr (CHK) {
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            voff0[j] = wave + 4 * j < a.ninstr ? slot_voff(wave + 4 * j) : OOB;
            if (wave + 4 * j < a.ninstr) convk::lds_dma16_m0(xd, lds0 + (unsigned)(wave + 4 * j) * 1024u, voff0[j], 0);          // chunk 0 -> buffer 0
        }
    } else {
        for (int i = wave; i < a.ninstr; i += 4) convk::lds_dma16_m0(xd, lds0 + (unsigned)i * 1024u, slot_voff(i), 0);
    }
    auto chunk_dma = [&](int c, int buf) {        // CHK: patch chunk c (channels [c KC, (c + 1) KC)) -> ring buffer buf
        if constexpr (CHK) {
            const unsigned coff = (unsigned)(c * CB), dst = lds0 + (unsigned)(buf * a.chunk_bytes);
#pragma unroll
            for (int j = 0; j < MAXI; ++j)
                if (wave + 4 * j < a.ninstr) convk::lds_dma16_m0(xd, dst + (unsigned)(wave + 4 * j) * 1024u, voff0[j] == OOB ? OOB : voff0[j] + coff, 0);
        }
    };

    stamp();                           // patch DMA issued
    // ---- the weight stream of this wave: fragment (step, ks, cb) at w[(((slice * 4 + wave) * nsteps + step) * 4 + ks) * A + cb][lane]
    const uint4* wb = a.w + ((long long)(slice * 4 + wave) * a.nsteps) * (4 * A * 64) + lane;
    uint4 wr[NSTG][4][A];
    auto wload = [&](auto S, int step) {
        constexpr int s = decltype(S)::value;
#if defined(DIR_AS_DBG_W)             // investigation aid (never in the product build): every step re-reads step 0's fragments (L1-resident after the first)
        const uint4* p = wb + (long long)(step & 0) * (4 * A * 64);
#else
        const uint4* p = wb + (long long)min(step, a.nsteps - 1) * (4 * A * 64);      // unconditional (clamped): a load in a branch costs a full wait at the join
#endif
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int cb = 0; cb < A; ++cb) wr[s][ks][cb] = p[(ks * A + cb) * 64];
    };
    [&]<int... S>(std::integer_sequence<int, S...>) { (wload(std::integral_constant<int, S>{}, S), ...); }(std::make_integer_sequence<int, NSTG - 1>{});

    f32x16 acc[A][PB];
#pragma unroll
    for (int cb = 0; cb < A; ++cb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;

    // ---- which pixel a lane holds.  ds_read_b128 is served in four NON-contiguous 16-lane groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
    //      same + 32 (MI355X_MICROARCH.md, LDS) -- one LDS cycle per group when its 16 addresses fall into 16 different 16-byte bank groups.  With
    //      the skewed pitch that holds for 16 CONSECUTIVE patch positions, so each lane group is given one 16-pixel run of a row (W >= 16; two rows
    //      of a 1x1 tile at W = 8) -- or, for the 3x3 patch of an 8x8 map (PW = 10), the two rows r and r + 4, whose positions are 40 apart =
    //      8 bank groups.  The permutation is free: MFMA column n of a pixel block is whatever pixel lane n's B operand came from, and the epilogue
    //      stages lane n's results at that pixel.  pix[pb]: pixel index in the tile (row-major); posb[pb]: LDS address of its tap (0, 0), slab 0,
    //      this lane half's chunk (4 h) of k-step 0.
    int pix[PB], posb[PB];
    {
        const bool g1 = ((l32 >= 4) & (l32 < 12)) | ((l32 >= 16) & (l32 < 20)) | (l32 >= 28);
        const int rk = g1 ? (l32 < 12 ? l32 - 4 : l32 < 20 ? l32 - 8 : l32 - 16) : (l32 < 4 ? l32 : l32 < 16 ? l32 - 8 : l32 - 12);      // rank inside the group
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int seg = 2 * pb + (g1 ? 1 : 0);
            int p = 16 * seg + rk;
            if (K3 && a.W == 8) p = (rk < 8 ? seg : seg + 4) * 8 + (rk & 7);          // (the tile is the whole 8 x 8 image: PB = 2, segments 0 .. 3)
            pix[pb] = p;
            posb[pb] = ((p >> a.logW) * a.PW + (p & (a.W - 1))) * PITCH + 64 * h;
        }
    }

    // The pixels' channels (MFMA B operands) of a step are read from LDS a whole STEP ahead of the MFMAs that use them: with one wave per SIMD
    // (A x PB >= 4: 256 workgroups or fewer) nothing else covers a ds_read_b128's latency, and reading one k-step ahead left the matrix pipe idle
    // ~45 % of the K loop (phase stamps, profiles/r06_as_phase_stamps.txt: 933 ticks per step for 512 of MFMAs).
    uint4 bv[2][4][PB];
    const char* pbuf = smem;                      // the ring buffer of the chunk being multiplied
    auto bread = [&](auto Par, int cs_, int tap_) {
        constexpr int par = decltype(Par)::value;
        const int ky = K3 ? (tap_ * 11) >> 5 : 0;
        const int soff = (K3 ? ky * a.PW + (tap_ - 3 * ky) : 0) * PITCH + cs_ * 128;      // wave-uniform: tap shift + slab
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#if defined(DIR_AS_DBG_B)             // investigation aid: the same reads from a fixed kilobyte (what the loop costs without the gather)
            const char* rp = smem + (lane << 4) + 0 * soff;
#else
            const char* rp = pbuf + posb[pb] + soff;
#endif
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bv[par][ks][pb] = *reinterpret_cast<const uint4*>(rp + 16 * ks);
        }
    };
    constexpr int UNR = (NSTG % 2 == 0) ? NSTG : 2 * NSTG;          // steps per unrolled iteration: ring stage and operand parity both compile-time
    int cs = 0, tap = 0;                                            // (slab of the chunk, tap) of the step whose operands are being read: one ahead of the MFMAs
    int sbase = 0;                                                  // global index of the chunk's first step (the weight stream does not know about chunks)
    // One group of UNR steps.  GUARD = false: every step of the group exists -- the body is ONE basic block, and the scheduler is told to deal the
    // step's memory instructions out between its MFMAs (one weight fragment load and one operand read per A x PB / 2 ... MFMAs) instead of
    // issuing eight loads and eight reads in a row: a 64-lane 16-byte load occupies the wave's issue for about half an MFMA, and with one wave per
    // SIMD eight of them back to back leave the matrix pipe idle for three (DIR_AS_NO_SCHED: the compiler's own order, A/B aid).
    auto group = [&](auto Guard, int s0) {
        constexpr bool GUARD = decltype(Guard)::value;
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                 const int step = s0 + S;
                 constexpr int stage = S % NSTG, par = S & 1;
                 wload(std::integral_constant<int, (stage + NSTG - 1) % NSTG>{}, sbase + step + NSTG - 1);
                 {   // next step's (slab, tap), branch-free; past the end: the last step's operands again (nobody uses them)
                     const bool adv = step + 1 < a.csteps, wrap = tap + 1 == NTAP;
                     const int ntap = wrap ? 0 : tap + 1, ncs = wrap ? cs + 1 : cs;
                     tap = adv ? ntap : tap;
                     cs = adv ? ncs : cs;
                 }
                 bread(std::integral_constant<int, par ^ 1>{}, cs, tap);
                 if (!GUARD || step < a.csteps) {
#pragma unroll
                     for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                         for (int cb = 0; cb < A; ++cb)
#pragma unroll
                             for (int pb = 0; pb < PB; ++pb) acc[cb][pb] = Half<H>::mfma32(wr[stage][ks][cb], bv[par][ks][pb], acc[cb][pb]);
                 }
#if !defined(DIR_AS_NO_SCHED)
                 if constexpr (!GUARD) {
                     constexpr int NMEM = 4 * A > 4 * PB ? 4 * A : 4 * PB, MPM = 4 * A * PB / NMEM;          // memory slots per step; MFMAs between them
#pragma unroll
                     for (int i = 0; i < NMEM; ++i) {
                         __builtin_amdgcn_sched_group_barrier(0x008, MPM, 0);                                 // MFMA
                         if (i < 4 * A) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    // one weight fragment load (VMEM read)
                         if (i < 4 * PB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   // one operand read (DS read)
                     }
                 }
#endif
             }()),
             ...);
        }(std::make_integer_sequence<int, UNR>{});
    };
    for (int c = 0; c < a.nchunk; ++c) {
        convk::wait_vmcnt<0>();        // this wave's share of chunk c has landed (and the ring stages in flight with it)
        __syncthreads();               // ... everybody's; and every wave is done with the buffer chunk c + 1 goes into
        if (c == 0) stamp();           // patch (chunk 0) + first ring stages landed
        if (CHK && c + 1 < a.nchunk) chunk_dma(c + 1, (c + 1) & 1);
        pbuf = smem + (CHK ? (c & 1) * a.chunk_bytes : 0);
        sbase = c * a.csteps;
        cs = 0; tap = 0;
        bread(std::integral_constant<int, 0>{}, 0, 0);
        int s0 = 0;
        for (; s0 + UNR <= a.csteps; s0 += UNR) group(std::false_type{}, s0);
        if (s0 < a.csteps) group(std::true_type{}, s0);
    }

    // ---- epilogue (conv.hip's arithmetic): fmaf(acc, scale, shift) as fp32 through LDS, then per 16-byte output chunk + residual, round, ReLU
    stamp();                           // K loop done (this wave)
    __syncthreads();                   // every wave is done with the patch
    constexpr int SP = NWG * 4 + 16;   // bytes per staged pixel (the 16-byte skew spreads a wave's 32 pixels over the bank groups)
#pragma unroll
    for (int cb = 0; cb < A; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = (wave * A + cb) * 32 + 8 * q + 4 * h, n = slice * NWG + cl;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.scale) sc = *reinterpret_cast<const float4*>(a.scale + n);
            if (a.shift) sh = *reinterpret_cast<const float4*>(a.shift + n);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
                *reinterpret_cast<float4*>(smem + pix[pb] * SP + cl * 4) =
                    make_float4(fmaf(acc[cb][pb][4 * q], sc.x, sh.x), fmaf(acc[cb][pb][4 * q + 1], sc.y, sh.y),
                                fmaf(acc[cb][pb][4 * q + 2], sc.z, sh.z), fmaf(acc[cb][pb][4 * q + 3], sc.w, sh.w));
        }
    __syncthreads();
    stamp();                           // tile staged
    {
        constexpr int CPR = NWG / 8;   // 16-byte output chunks per pixel
        H* __restrict__ y = (H*)a.y;
        const H* __restrict__ res = (const H*)a.res;
        const long long m0 = ((long long)img * a.H + trow) * a.W;          // the tile's pixels are consecutive in NHW order (whole rows)
        const bool relu = a.relu != 0;
#pragma unroll 4
        for (int c = tid; c < TPX * CPR; c += AS_THR) {
            const int px = c / CPR, cc = c - px * CPR;
            float v[8];
            const float4 t0 = *reinterpret_cast<const float4*>(smem + px * SP + cc * 32), t1 = *reinterpret_cast<const float4*>(smem + px * SP + cc * 32 + 16);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            const int n = slice * NWG + cc * 8;
            if (res) {
                float rv[8];
                OutVec<H>::load(res + (m0 + px) * a.res_cs + a.res_co + n, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            OutVec<H>::store_act(y + (m0 + px) * a.out_cs + a.out_co + n, v, relu);
        }
    }
    stamp();
}

template <typename H, int A, int PB, bool K3, int NSTG, bool CHK = false>
int launch_as(const AsArgs& a, size_t lds, hipStream_t s) {
    static bool attr_set = false;      // (per instantiation)
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv_as_kernel<H, A, PB, K3, NSTG, CHK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("dir_conv2d_as_forward: cannot raise the dynamic LDS limit");
            return DIR_E_LAUNCH;
        }
        attr_set = true;
    }
    AsArgs b = a;
    b.stamps = stamps_begin("conv_as");
    DIR_LAUNCH((conv_as_kernel<H, A, PB, K3, NSTG, CHK>), dim3(a.ntile * a.nslice), dim3(AS_THR), lds, s, b);
    stamps_end("conv_as", b.stamps, s);
    return check_launch("dir_conv2d_as_forward");
}


// How a layer's patch is held: whole (KC = Cin, one buffer) when it fits 160 KB beside nothing else, otherwise -- for the one shape built with a
// chunk ring, (A, PB) = (4, 2) -- in the largest KC in {1024, 512, 256, 128} whose TWO buffers fit, whose steps per chunk are a multiple of the
// unrolled group (ring stage and operand parity stay compile-time across chunks: NSTG = 4 -> 4 steps) and whose DMA instructions per wave fit the
// registers that keep their source offsets (<= 20).  Returns false when the layer cannot be served.
struct AsPlan { int KC, nchunk, csteps, ninstr, chunk_bytes; size_t lds; bool chunked; };
bool as_plan(const dir_conv_desc* d, int A, int PB, AsPlan& p) {
    const bool k3 = d->kh == 3;
    const int tpx = 32 * PB, TR = tpx / d->W, PW = k3 ? d->W + 2 : d->W, NP = (k3 ? TR + 2 : TR) * PW, ntap = k3 ? 9 : 1;
    const size_t stage = (size_t)tpx * (128 * A * 4 + 16);
    auto bytes = [&](int kc) { return (size_t)(((long long)NP * (kc / 8 + 1) + 63) / 64) * 1024; };
    if (bytes(d->Cin) <= 160 * 1024 && stage <= 160 * 1024) {
        p.KC = d->Cin; p.nchunk = 1; p.csteps = ntap * d->Cin / 64; p.ninstr = (int)(bytes(d->Cin) / 1024); p.chunk_bytes = (int)bytes(d->Cin);
        p.lds = bytes(d->Cin) > stage ? bytes(d->Cin) : stage; p.chunked = false;
        return true;
    }
    if (!(A == 4 && PB == 2) || stage > 160 * 1024) return false;
    for (int kc = 1024; kc >= 128; kc >>= 1) {
        if (d->Cin % kc || d->Cin / kc < 2) continue;
        const size_t b = bytes(kc);
        if (2 * b > 160 * 1024 || (ntap * kc / 64) % 4 || b / 1024 > 80) continue;
        p.KC = kc; p.nchunk = d->Cin / kc; p.csteps = ntap * kc / 64; p.ninstr = (int)(b / 1024); p.chunk_bytes = (int)b;
        p.lds = 2 * b > stage ? 2 * b : stage; p.chunked = true;
        return true;
    }
    return false;
}

}  // namespace
}  // namespace dir

extern "C" int dir_conv2d_as_supported(const dir_conv_desc* d, int blocks_per_wave, int pixel_blocks) {
    if (!d) return 0;
    const int A = blocks_per_wave, PB = pixel_blocks;
    if (!((A == 2 && PB == 2) || (A == 2 && PB == 4) || (A == 4 && PB == 2) || (A == 1 && PB == 2))) return 0;      // the built shapes
    if (!((d->in_dtype == DIR_DT_BF16 || d->in_dtype == DIR_DT_F16) && d->out_dtype == d->in_dtype)) return 0;
    const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1, k1 = d->kh == 1 && d->kw == 1 && d->pad == 0;
    if (!(k3 || k1) || d->stride != 1 || d->B <= 0) return 0;
    if (d->Ho && (d->Ho != d->H || d->Wo != d->W)) return 0;
    if (!(d->W == 8 || d->W == 16 || d->W == 32) || d->H <= 0) return 0;
    const int tpx = 32 * PB;
    if (tpx % d->W || (d->H * d->W) % tpx) return 0;
    if (d->Cin <= 0 || d->Cin % 64 || d->Cout <= 0 || d->Cout % (128 * A)) return 0;
    const int in_cs = d->in_cstride ? d->in_cstride : d->Cin, out_cs = d->out_cstride ? d->out_cstride : d->Cout;
    if (in_cs % 8 || d->in_coff % 8 || out_cs % 8 || d->out_coff % 8 || d->res_cstride % 8 || d->res_coff % 8) return 0;
    dir::AsPlan plan;
    if (!dir::as_plan(d, A, PB, plan)) return 0;                                      // the patch (or two chunks of it) and the staged tile must fit 160 KB
    if ((long long)d->B * d->H * d->W * in_cs * 2 >= (1ll << 31)) return 0;           // 32-bit buffer offsets, and OOB must be out of range
    return 1;
}

extern "C" int dir_conv2d_as_forward(const dir_conv_desc* d, const void* x, const void* w_as, const float* scale, const float* shift,
                                     const void* residual, void* y, int blocks_per_wave, int pixel_blocks, void* stream) {
    using namespace dir;
    DIR_REQUIRE(d && x && w_as && y, "dir_conv2d_as_forward: null pointer");
    DIR_REQUIRE(dir_conv2d_as_supported(d, blocks_per_wave, pixel_blocks), "dir_conv2d_as_forward: layer not supported (16-bit storage, 1x1 / 3x3 pad 1 stride 1, "
                "W in {8, 16, 32}, whole rows per tile, Cin %% 64 == 0, Cout %% (128 A) == 0, patch (or, for (4, 2), two chunks of it) <= 160 KB)");
    const int A = blocks_per_wave, PB = pixel_blocks;
    const bool k3 = d->kh == 3, f16 = d->in_dtype == DIR_DT_F16;
    AsArgs a;
    a.x = x; a.w = (const uint4*)w_as; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.logW = d->W == 8 ? 3 : d->W == 16 ? 4 : 5;
    a.Cin = d->Cin; a.in_cs = d->in_cstride ? d->in_cstride : d->Cin; a.in_co = d->in_coff;
    a.Cout = d->Cout; a.out_cs = d->out_cstride ? d->out_cstride : d->Cout; a.out_co = d->out_coff;
    a.res_cs = d->res_cstride ? d->res_cstride : d->Cout; a.res_co = d->res_coff;
    a.TR = 32 * PB / d->W;
    a.tiles_per_img = d->H * d->W / (32 * PB); a.ntile = d->B * a.tiles_per_img; a.nslice = d->Cout / (128 * A);
    a.nsteps = (k3 ? 9 : 1) * d->Cin / 64; a.relu = (d->flags & DIR_CONV_RELU) != 0;
    a.xcd_map = a.ntile % 8 == 0;
    AsPlan plan;
    DIR_REQUIRE(as_plan(d, A, PB, plan), "dir_conv2d_as_forward: no LDS plan for this layer");
    a.KC = plan.KC; a.nchunk = plan.nchunk; a.csteps = plan.csteps; a.chunk_bytes = plan.chunk_bytes; a.ninstr = plan.ninstr;
    a.q = a.KC / 8; a.PW = k3 ? d->W + 2 : d->W; a.NP = (k3 ? a.TR + 2 : a.TR) * a.PW;
    convk::magic_u31((unsigned)(a.q + 1), &a.mg_q, &a.sh_q);
    convk::magic_u31((unsigned)a.PW, &a.mg_pw, &a.sh_pw);
    a.x_bytes = (unsigned)((long long)d->B * d->H * d->W * a.in_cs * 2);
    a.stamps = nullptr;
    const size_t lds = plan.lds;
    hipStream_t s = (hipStream_t)stream;
    if (plan.chunked) {                // (A, PB) = (4, 2) on a ring of two patch chunks, weight ring of 4 steps
        if (k3) { if (f16) return launch_as<convk::f16s_t, 4, 2, true, 4, true>(a, lds, s); return launch_as<convk::bf16_t, 4, 2, true, 4, true>(a, lds, s); }
        if (f16) return launch_as<convk::f16s_t, 4, 2, false, 4, true>(a, lds, s);
        return launch_as<convk::bf16_t, 4, 2, false, 4, true>(a, lds, s);
    }
    // ring depth: 3 steps where two workgroups fit a CU (<= 256 registers, <= 80 KB), else 4
    const bool two = A * PB <= 4 && lds <= 80 * 1024;
#define DIR_AS3(A_, PB_, NS_) do { if (k3) { if (f16) return launch_as<convk::f16s_t, A_, PB_, true, NS_>(a, lds, s); return launch_as<convk::bf16_t, A_, PB_, true, NS_>(a, lds, s); } \
                                   if (f16) return launch_as<convk::f16s_t, A_, PB_, false, NS_>(a, lds, s); return launch_as<convk::bf16_t, A_, PB_, false, NS_>(a, lds, s); } while (0)
#define DIR_AS(A_, PB_) do { if (A_ * PB_ <= 4 && two) DIR_AS3(A_, PB_, 3); else DIR_AS3(A_, PB_, 4); } while (0)
    if (A == 2 && PB == 2) DIR_AS(2, 2);
    if (A == 2 && PB == 4) DIR_AS(2, 4);
    if (A == 4 && PB == 2) DIR_AS(4, 2);
    if (A == 1 && PB == 2) DIR_AS(1, 2);
#undef DIR_AS
#undef DIR_AS3
    DIR_REQUIRE(false, "dir_conv2d_as_forward: unsupported (A, PB) = (%d, %d)", A, PB);
}
