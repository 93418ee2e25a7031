// Backward of the factorised bone fusion (bonefuse.hip) for the training step: Joint2BoneFeature.bone_proj (models/dir.py:132-174) +
// fusion[0] (3x3 conv 2560 -> 256, models/dir.py:57-62) differentiated WITHOUT the [B,S,S,2560] bone map and its gradient map.
//
// Forward (bonefuse.hip):   y[b,p,n] = sum_tap sum_e Wgt[b, p + tap, e] * G[b, tap, e, n],   e = 2*(hand*20 + bone) + end in [0, 80)
//                           G[b, tap, e, n] = sum_c f_e[b][c] * W[tap][hb][c][n],  f_e = emb row of the bone's parent (end 0) / child (end 1)
// Backward from gy [B,S,S,256]:
//   (1) g G[b,tap,e,n]  = sum_p Wgt[b, p + tap, e] * gy[b,p,n]                      K = pixels
//   (2) g Wgt[b,q,e]    = sum_tap sum_n gy[b, q - tap, n] * G[b,tap,e,n]            K = 9 * 256   -> g uv (bone_proj's weight formulas)
//   (3) g f_e[b][c]     = sum_tap sum_n g G[b,tap,e,n] * W[tap][hb][c][n]           -> g emb (index_select backward: bones -> joints)
//   (4) g W[tap][hb][c][n] = sum_b sum_end f_e[b][c] * g G[b,tap,e,n]
// 24 GFLOP at 32 images and S = 32 instead of the 3 x 1.9 TFLOP of the materialised convolution's forward / data / weight gradients.
//
// All four are exact-fp32 GEMMs on dir_gemm_f32 / dir_gemm_f32_grouped (train_ops.hip) over operands laid out so that every tap is a POINTER SHIFT:
//   pixels live on the zero-bordered grid (S+2) x (S+2), flattened, with S+3 zero rows before and after (`R` rows per sample), so
//   row q + (ky-1)*(S+2) + (kx-1) is the tap's neighbour of row q for every q, and the border rows carry the convolution's zero padding;
//   G and g G are kept as [tap][e][b][n], so (tap, hand-bone) selects one contiguous [2B (end, b)][256] block for (3) and (4).
// The kernels of this file only build those operands (bone weights with bone_proj's own arithmetic: bit-identical mask) and finish
// (2) -> g uv and (3) -> g emb per joint.  Deterministic: fixed summation orders, no atomics.
#include "bone_common.h"
#include "dir_common.h"

namespace dir {
namespace {

using dir::bone::kChild;
using dir::bone::kParent;

constexpr int NE = 80, NCOUT = 256, NTAP = 9;

__host__ __device__ inline int pad_w(int S) { return S + 2; }
__host__ __device__ inline int pad_pp(int S) { return (S + 2) * (S + 2); }
__host__ __device__ inline int pad_mg(int S) { return S + 3; }
__host__ __device__ inline int pad_rows(int S) { return pad_pp(S) + 2 * pad_mg(S); }

// ---- Wgt on the padded grid: [B][R][80] (word (hb) = float2 (m*wa, m*wb)), zero outside the image
struct WgtArgs { const float* uv[2]; float* wgt; int B, S; float distance; };
constexpr int WROWS = 32;
__global__ __launch_bounds__(256) void bone_wgt_pad_kernel(WgtArgs a) {
    __shared__ float s_uv[84];
    __shared__ float s_bone[40 * 6];
    const int S = a.S, PW = pad_w(S), PP = pad_pp(S), MG = pad_mg(S), R = pad_rows(S);
    const int b = blockIdx.y, r0 = blockIdx.x * WROWS, tid = threadIdx.x;
    if (tid < 84) {
#pragma clang fp contract(off)
        const int hand = tid / 42, r = tid - hand * 42;
        const float v = a.uv[hand][(long long)b * 42 + r];
        s_uv[tid] = (v + 1.f) / 2.f * (float)S;                      // models/dir.py:150
    }
    __syncthreads();
    if (tid < 40) {
        const int hand = tid / 20, bone = tid - hand * 20;
        const float* uv = s_uv + hand * 42;
        const int pa = kParent[bone], ch = kChild[bone];
        float dx, dy;
        dir::bone::bone_dir(uv[2 * pa], uv[2 * pa + 1], uv[2 * ch], uv[2 * ch + 1], dx, dy);
        float* sb = s_bone + 6 * tid;
        sb[0] = uv[2 * pa]; sb[1] = uv[2 * pa + 1]; sb[2] = uv[2 * ch]; sb[3] = uv[2 * ch + 1]; sb[4] = dx; sb[5] = dy;
    }
    __syncthreads();
    float2* out = reinterpret_cast<float2*>(a.wgt + (long long)b * R * NE);
    for (int i = tid; i < WROWS * 40; i += 256) {
        const int rl = i / 40, hb = i - rl * 40, r = r0 + rl;
        if (r >= R) break;
        float2 word = make_float2(0.f, 0.f);
        const int q = r - MG;
        if (q >= 0 && q < PP) {
            const int py = q / PW, px = q - py * PW, iy = py - 1, ix = px - 1;
            if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
                const float* sb = s_bone + 6 * hb;
                float wa, wb;
                if (dir::bone::bone_weights_fast((float)ix + 0.5f, (float)iy + 0.5f, sb[0], sb[1], sb[2], sb[3], sb[4], sb[5], a.distance, wa, wb))
                    word = make_float2(wa, wb);
            }
        }
        out[(long long)r * 40 + hb] = word;
    }
}

// ---- gy [B][S*S][256] -> the padded grid [B][R][256]
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ gy, float* __restrict__ out, int S) {
    const int PW = pad_w(S), PP = pad_pp(S), MG = pad_mg(S), R = pad_rows(S);
    const int b = blockIdx.y, r = blockIdx.x * 4 + (threadIdx.x >> 6), c4 = threadIdx.x & 63;
    if (r >= R) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int q = r - MG;
    if (q >= 0 && q < PP) {
        const int py = q / PW, px = q - py * PW, iy = py - 1, ix = px - 1;
        if (iy >= 0 && iy < S && ix >= 0 && ix < S)
            v = reinterpret_cast<const float4*>(gy + ((long long)b * S * S + iy * S + ix) * NCOUT)[c4];
    }
    reinterpret_cast<float4*>(out + ((long long)b * R + r) * NCOUT)[c4] = v;
}

// ---- G of dir_bone_fusion_prepare (exact fp32: float2 (end 0, end 1) at [b][tap][hb][n]) -> [tap][2 hb + end][b][n]
__global__ __launch_bounds__(256) void g_transpose_kernel(const float2* __restrict__ g, float* __restrict__ gt, int B) {
    const int z = blockIdx.x, b = z / (NTAP * 40), th = z - b * (NTAP * 40), n = threadIdx.x;   // th = tap * 40 + hb
    const float2 v = g[(long long)z * NCOUT + n];
    float* o = gt + (((long long)th * 2) * B + b) * NCOUT + n;
    o[0] = v.x;
    o[(long long)B * NCOUT] = v.y;
}

// ---- f_e gathered for (4): [tap][hb][end][b][c] (nine copies: one GEMM launch with a uniform batch stride over (tap, hb))
__global__ __launch_bounds__(256) void emb_gather_kernel(const float* __restrict__ emb, float* __restrict__ out, int B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, n = (long long)NTAP * 40 * 2 * B * 64;
    if (i >= n) return;
    const int c = (int)(i & 63);
    long long t = i >> 6;
    const int b = (int)(t % B); t /= B;
    const int end = (int)(t & 1); t >>= 1;
    const int hb = (int)(t % 40), hand = hb / 20, bone = hb - hand * 20;
    const int j = hand * 21 + (end ? kChild[bone] : kParent[bone]);
    out[i] = emb[((long long)b * 42 + j) * 64 + c];
}

// ---- (2) -> g uv of a bone's two ends: one wave per (sample, hand-bone), lanes = pixels; bone_bwd_kernel's formulas (train_spatial.hip)
struct UvArgs { const float* uv[2]; const float* dwgt; float* g_bone_uv; int B, S; float distance; };
__global__ __launch_bounds__(64) void bone_uv_bwd_kernel(UvArgs a) {
#pragma clang fp contract(off)
    const int hb = blockIdx.x % 40, b = blockIdx.x / 40, hand = hb / 20, k = hb - hand * 20, lane = threadIdx.x;
    const int S = a.S, PW = pad_w(S), PP = pad_pp(S), ja = kParent[k], jb = kChild[k];
    const float* uv = a.uv[hand] + (long long)b * 42;
    const float Ax = (uv[2 * ja] + 1.f) / 2.f * S, Ay = (uv[2 * ja + 1] + 1.f) / 2.f * S;
    const float Bx = (uv[2 * jb] + 1.f) / 2.f * S, By = (uv[2 * jb + 1] + 1.f) / 2.f * S;
    const float* gw = a.dwgt + ((long long)b * NE + 2 * hb) * PP;             // rows (hb, end 0), (hb, end 1) of dwgt [B][80][PP]
    float gAx = 0.f, gAy = 0.f, gBx = 0.f, gBy = 0.f;
    for (int p = lane; p < S * S; p += 64) {
        const int y = p / S, x = p - y * S, q = (y + 1) * PW + x + 1;
        const float2 g = make_float2(gw[q], gw[PP + q]);                         // requested before the mask is known (no dependent round trip)
        const float px = x + 0.5f, py = y + 0.5f;
        float wa, wb; bool in;
        dir::bone::bone_weights(px, py, Ax, Ay, Bx, By, a.distance, wa, wb, in);
        if (!in) continue;
        const float dax = px - Ax + 1e-6f, day = py - Ay + 1e-6f, dbx = px - Bx + 1e-6f, dby = py - By + 1e-6f;
        const float da = sqrtf(dax * dax + day * day), db = sqrtf(dbx * dbx + dby * dby), sum = da + db;
        const float gda = (g.y - g.x) * db / (sum * sum), gdb = (g.x - g.y) * da / (sum * sum);
        gAx -= gda * dax / da; gAy -= gda * day / da;
        gBx -= gdb * dbx / db; gBy -= gdb * dby / db;
    }
    gAx = wave_sum(gAx); gAy = wave_sum(gAy); gBx = wave_sum(gBx); gBy = wave_sum(gBy);
    if (lane == 0) {
        float* gu = a.g_bone_uv + ((long long)b * 40 + hb) * 4;
        gu[0] = gAx * S / 2.f; gu[1] = gAy * S / 2.f; gu[2] = gBx * S / 2.f; gu[3] = gBy * S / 2.f;
    }
}

// ---- bones -> joints: g emb[b][hand*21 + j][c] = sum over the bone ends at joint j (bone order) of sum_tap part[tap*40 + hb][end*B + b][c]
//      and g uv[hand][b][j] from g_bone_uv.  One wave per (sample, hand, joint).
struct JointArgs { const float* part; const float* g_bone_uv; float* g_emb; float* g_uv[2]; int B; };
__global__ __launch_bounds__(64) void bone_joint_scatter_kernel(JointArgs a) {
    const int j = blockIdx.x % 21, hand = (blockIdx.x / 21) & 1, b = blockIdx.x / 42, c = threadIdx.x, B = a.B;
    float s = 0.f, ux = 0.f, uy = 0.f;
    for (int k = 0; k < 20; ++k)
#pragma unroll
        for (int end = 0; end < 2; ++end) {
            if ((end ? kChild[k] : kParent[k]) != j) continue;
            const int hb = hand * 20 + k;
            float v[NTAP];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) v[t] = a.part[(((long long)(t * 40 + hb) * 2 + end) * B + b) * 64 + c];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) s += v[t];
            const float* gu = a.g_bone_uv + ((long long)b * 40 + hb) * 4 + 2 * end;
            ux += gu[0]; uy += gu[1];
        }
    a.g_emb[((long long)b * 42 + hand * 21 + j) * 64 + c] = s;
    if (c == 0 && a.g_uv[hand]) { a.g_uv[hand][(long long)b * 42 + 2 * j] = ux; a.g_uv[hand][(long long)b * 42 + 2 * j + 1] = uy; }
}

struct Ws {
    float *wgt, *gyp, *gt, *dgt, *dwgt, *embe, *part, *buv;
    long long total;
};
inline Ws carve(float* base, int B, int S) {
    Ws w{};
    long long o = 0;
    auto take = [&](long long n) { float* p = base ? base + o : nullptr; o += (n + 63) / 64 * 64; return p; };
    const long long R = pad_rows(S), PP = pad_pp(S);
    w.wgt = take((long long)B * R * NE);
    w.gyp = take((long long)B * R * NCOUT);
    w.gt = take((long long)NTAP * NE * B * NCOUT);
    w.dgt = take((long long)NTAP * NE * B * NCOUT);
    w.dwgt = take((long long)B * PP * NE);
    w.embe = take((long long)NTAP * 40 * 2 * B * 64);
    w.part = take((long long)NTAP * 40 * 2 * B * 64);
    w.buv = take((long long)B * 40 * 4);
    w.total = o * 4;
    return w;
}

}  // namespace
}  // namespace dir

extern "C" long long dir_bone_fusion_backward_workspace_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return -1;
    return dir::carve(nullptr, B, S).total;
}

extern "C" int dir_bone_fusion_backward(const float* w_g, const float* emb, const float* uv_left, const float* uv_right, const void* g_scratch,
                                        const float* gy, float distance, float* g_w_g, float* g_emb, float* g_uv_left, float* g_uv_right,
                                        void* workspace, long long workspace_bytes, int B, int S, void* stream) {
    using namespace dir;
    DIR_REQUIRE(w_g && emb && uv_left && uv_right && g_scratch && gy && g_w_g && g_emb && workspace, "dir_bone_fusion_backward: null pointer");
    DIR_REQUIRE(B > 0 && S > 0 && S <= 256, "dir_bone_fusion_backward: B=%d S=%d", B, S);
    DIR_REQUIRE(workspace_bytes >= dir_bone_fusion_backward_workspace_bytes(B, S), "dir_bone_fusion_backward: workspace too small (dir_bone_fusion_backward_workspace_bytes)");
    DIR_REQUIRE((((uintptr_t)workspace | (uintptr_t)gy) & 15) == 0, "dir_bone_fusion_backward: gy and workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const Ws w = carve((float*)workspace, B, S);
    const int PW = pad_w(S), PP = pad_pp(S), MG = pad_mg(S), R = pad_rows(S);

    WgtArgs wa{};
    wa.uv[0] = uv_left; wa.uv[1] = uv_right; wa.wgt = w.wgt; wa.B = B; wa.S = S; wa.distance = distance;
    DIR_LAUNCH(bone_wgt_pad_kernel, dim3((R + WROWS - 1) / WROWS, B), dim3(256), 0, s, wa);
    DIR_LAUNCH(pad_rows_kernel, dim3((R + 3) / 4, B), dim3(256), 0, s, gy, w.gyp, S);
    DIR_LAUNCH(g_transpose_kernel, dim3(B * NTAP * 40), dim3(256), 0, s, (const float2*)g_scratch, w.gt, B);
    {
        const long long n = (long long)NTAP * 40 * 2 * B * 64;
        DIR_LAUNCH(emb_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, emb, w.embe, B);
    }
    int rc = check_launch("dir_bone_fusion_backward (operands)");
    if (rc != DIR_OK) return rc;

    const long long tapblk = (long long)NE * B * NCOUT;              // one tap of gt / dgt
    // tap (ky, kx) reads row q + off, off = (ky - 1) * PW + (kx - 1): a displacement of ky * PW + kx rows from the tap (0, 0) pointer
    // (2) dwgt[b][e][q] = sum_tap gt[tap][e][b][:] . gy_pad[b][q - off][:]   -- the nine taps summed inside one workgroup; 80 rows (bone ends)
    //     x pixels, so that the rows fill the short-output tile exactly and g uv reads its pixels contiguously
    {
        dir_gemm_desc d{NE, PP, NCOUT, B * NCOUT, NCOUT, PP, 0, 1, 0, B, NCOUT, (long long)R * NCOUT, (long long)NE * PP};
        dir_gemm_groups g{3, 3, 1, 0, 3 * tapblk, tapblk, -(long long)PW * NCOUT, -(long long)NCOUT, 0, 0};
        rc = dir_gemm_f32_grouped(&d, &g, w.gt, w.gyp + (long long)(MG + PW + 1) * NCOUT, nullptr, w.dwgt, stream);
        if (rc != DIR_OK) return rc;
    }
    // (1) dgt[tap][e][b][n] = sum_q wgt_pad[b][q + off][e] * gy_pad[b][q][n]   -- nine products per sample, one launch
    {
        dir_gemm_desc d{NE, NCOUT, PP, NE, NCOUT, B * NCOUT, 1, 0, 0, B, (long long)R * NE, (long long)R * NCOUT, NCOUT};
        dir_gemm_groups g{3, 3, 0, 0, (long long)PW * NE, NE, 0, 0, 3 * tapblk, tapblk};
        rc = dir_gemm_f32_grouped(&d, &g, w.wgt + (long long)(MG - PW - 1) * NE, w.gyp + (long long)MG * NCOUT, nullptr, w.dgt, stream);
        if (rc != DIR_OK) return rc;
    }
    // (3) part[(tap, hb)][(end, b)][c] = dgt block [2B][256] . w_g[tap][hb][c][:]^T
    dir_gemm_desc d3{2 * B, 64, NCOUT, NCOUT, NCOUT, 64, 0, 1, 0, NTAP * 40, 2ll * B * NCOUT, 64ll * NCOUT, 2ll * B * 64};
    rc = dir_gemm_f32(&d3, w.dgt, w_g, nullptr, w.part, stream);
    if (rc != DIR_OK) return rc;
    // (4) g_w_g[tap][hb][c][n] = embE block^T [64][2B] . dgt block [2B][256]
    dir_gemm_desc d4{64, NCOUT, 2 * B, 64, NCOUT, NCOUT, 1, 0, 0, NTAP * 40, 2ll * B * 64, 2ll * B * NCOUT, 64ll * NCOUT};
    rc = dir_gemm_f32(&d4, w.embe, w.dgt, nullptr, g_w_g, stream);
    if (rc != DIR_OK) return rc;

    UvArgs ua{};
    ua.uv[0] = uv_left; ua.uv[1] = uv_right; ua.dwgt = w.dwgt; ua.g_bone_uv = w.buv; ua.B = B; ua.S = S; ua.distance = distance;
    DIR_LAUNCH(bone_uv_bwd_kernel, dim3(B * 40), dim3(64), 0, s, ua);
    JointArgs ja{};
    ja.part = w.part; ja.g_bone_uv = w.buv; ja.g_emb = g_emb; ja.g_uv[0] = g_uv_left; ja.g_uv[1] = g_uv_right; ja.B = B;
    DIR_LAUNCH(bone_joint_scatter_kernel, dim3(B * 42), dim3(64), 0, s, ja);
    return check_launch("dir_bone_fusion_backward");
}
