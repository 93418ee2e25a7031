// Backward of the fused MANO layer + weak-perspective projection (dir_mano_forward): the gradient of
//   <g_verts, verts> + <g_joints, joints> + <g_joint_uv, joint_uv> + <g_mesh_uv, mesh_uv>
// w.r.t. the 64-vector a regressor predicts per hand (pose 51 = 6D root | 45 PCA, betas 10, cam 3; models/dir.py:352-363) --
// what torch autograd computes through manopth/manopth/manolayer.py:110-270 (+ rodrigues_layer.py:15-54, rot6d.py:26-60,
// tensutils.py:6-42) and utils/utils.py:47-63 in the reference's training step (train.py:66-70).  First link of the backward
// pass behind dir_stage_losses_backward.
//
// One 256-thread workgroup per (sample, hand).  The forward is recomputed in LDS (pose maths, chain, posed and skinned vertices:
// cheaper than round-tripping 30 KB of intermediates per (sample, hand) through HBM), then reversed stage by stage:
//   projection / centring -> joint routing (chain joints, fingertip vertices) -> skinning (g v_posed per vertex; g A'_k as 192
//   reductions over the 778 vertices) -> A' -> A -> kinematic chain (one thread per finger, root partials combined afterwards)
//   -> pose blend shapes (135 wave-level dot products over the k-major table) -> Rodrigues-via-quaternion and robust-6D chain
//   rules -> PCA and shape blend (10 wave-level dot products + the folded joint regressor).
// Reductions run in a fixed order (no atomics): results are deterministic.  fp32 throughout, like the forward.
#include "dir_common.h"

namespace {

constexpr int NV = 778, NV3 = 2334, NV3P = 2336, NJ = 16, BT = 256;

__constant__ int kReorderJb[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
__constant__ int kTipsb[2][5] = {{745, 317, 444, 556, 673}, {745, 317, 445, 556, 673}};

struct ManoBwdHand {
    dir_mano_tables t;
    const float* pose; int pose_stride;
    const float* betas; int betas_stride;
    const float* cam; int cam_stride;
    const float* g_verts; const float* g_joints; const float* g_joint_uv; const float* g_mesh_uv;
    float* g_pose; int g_pose_stride;
    float* g_betas; int g_betas_stride;
    float* g_cam; int g_cam_stride;
};
struct ManoBwdArgs { ManoBwdHand h[2]; };

__device__ __forceinline__ void normalize3b(float& x, float& y, float& z, float& mag) {
#pragma clang fp contract(off)
    mag = fmaxf(sqrtf(x * x + y * y + z * z), 1e-8f);
    x /= mag; y /= mag; z /= mag;
}
// n = v / max(|v|, 1e-8): g v = (g - n (n . g)) / |v|   (the clamp branch has zero measure on the path)
__device__ __forceinline__ void normalize3_bwd(float nx, float ny, float nz, float mag, float& gx, float& gy, float& gz) {
    const float d = nx * gx + ny * gy + nz * gz;
    gx = (gx - nx * d) / mag; gy = (gy - ny * d) / mag; gz = (gz - nz * d) / mag;
}

__global__ __launch_bounds__(BT) void mano_backward_kernel(ManoBwdArgs args) {
    const ManoBwdHand& a = args.h[blockIdx.y];
    __shared__ float s_v[NV3P];          // v_posed
    __shared__ float s_vert[NV3P];       // skinned vertices (before centring)
    __shared__ float s_gv[NV3P];         // g of the skinned vertices
    __shared__ float s_gvp[NV3P];        // g of v_posed (= g of v_shaped)
    __shared__ float s_pose[51], s_beta[10], s_cam[3];
    __shared__ float s_full[45], s_rot[135], s_pm[135], s_root[9], s_J[48];
    __shared__ float s_A[NJ * 12];
    __shared__ __attribute__((aligned(16))) float s_A2[NJ * 12];
    __shared__ float s_jtr[63], s_c[3];
    __shared__ float s_gj[63], s_gAt[48], s_gA2[NJ * 12], s_gA[NJ * 12], s_gJ[48], s_gR[135], s_gfull[45];
    __shared__ float s_part[5][15];      // per finger: g A_0 (12) | g J_0 (3) contributions
    __shared__ float s_red[4][6], s_sum[6], s_groot[9];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int center = a.t.center_idx;

    // ================================================================ forward (as mano_forward_kernel, one vertex part)
    if (tid < 51) s_pose[tid] = a.pose[(size_t)b * a.pose_stride + tid];
    if (tid >= 64 && tid < 74) s_beta[tid - 64] = a.betas[(size_t)b * a.betas_stride + tid - 64];
    if (tid >= 128 && tid < 131) s_cam[tid - 128] = a.cam ? a.cam[(size_t)b * a.cam_stride + tid - 128] : 0.f;
    __syncthreads();
    if (tid < 45) {
        float acc = 0.f;
        for (int k = 0; k < 45; ++k) acc = fmaf(s_pose[6 + k], a.t.comps[k * 45 + tid], acc);
        s_full[tid] = a.t.hands_mean[tid] + acc;
    }
    for (int i = tid; i < NV3; i += BT) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(a.t.shapedirs_t[k * NV3P + i], s_beta[k], acc);
        s_v[i] = acc + a.t.v_template[i];
    }
    if (tid == 64) {
#pragma clang fp contract(off)
        float x0 = s_pose[0], x1 = s_pose[1], x2 = s_pose[2], y0 = s_pose[3], y1 = s_pose[4], y2 = s_pose[5], mg;
        normalize3b(x0, x1, x2, mg);
        normalize3b(y0, y1, y2, mg);
        float m0 = x0 + y0, m1 = x1 + y1, m2 = x2 + y2;
        float o0 = x0 - y0, o1 = x1 - y1, o2 = x2 - y2;
        normalize3b(m0, m1, m2, mg);
        normalize3b(o0, o1, o2, mg);
        x0 = m0 + o0; x1 = m1 + o1; x2 = m2 + o2;
        y0 = m0 - o0; y1 = m1 - o1; y2 = m2 - o2;
        normalize3b(x0, x1, x2, mg);
        normalize3b(y0, y1, y2, mg);
        float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        normalize3b(z0, z1, z2, mg);
        s_root[0] = x0; s_root[1] = y0; s_root[2] = z0;
        s_root[3] = x1; s_root[4] = y1; s_root[5] = z1;
        s_root[6] = x2; s_root[7] = y2; s_root[8] = z2;
    }
    __syncthreads();
    if (tid < 15) {
#pragma clang fp contract(off)
        float vx = s_full[3 * tid], vy = s_full[3 * tid + 1], vz = s_full[3 * tid + 2];
        float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
        float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        float ax = vx / angle, ay = vy / angle, az = vz / angle;
        float half = angle * 0.5f;
        float w = cosf(half), sn = sinf(half);
        float x = sn * ax, y = sn * ay, z = sn * az;
        float qn = sqrtf(w * w + x * x + y * y + z * z);
        w /= qn; x /= qn; y /= qn; z /= qn;
        float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
        float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        float* R = s_rot + 9 * tid;
        R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
        R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
        R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
#pragma unroll
        for (int e = 0; e < 9; ++e) s_pm[9 * tid + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    if (tid >= 128 && tid < 128 + NJ * 3) {
        const int o = tid - 128;
        float acc = a.t.j_template[o];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(a.t.j_shapedirs[o * 10 + k], s_beta[k], acc);
        s_J[o] = acc;
    }
    __syncthreads();
    for (int c4 = tid; c4 < NV3P / 4; c4 += BT) {
        const float4* pd = reinterpret_cast<const float4*>(a.t.posedirs_t) + c4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 9
        for (int k = 0; k < 135; ++k) {
            const float4 p = pd[k * (NV3P / 4)];
            const float w = s_pm[k];
            acc.x = fmaf(p.x, w, acc.x); acc.y = fmaf(p.y, w, acc.y); acc.z = fmaf(p.z, w, acc.z); acc.w = fmaf(p.w, w, acc.w);
        }
        const int i = 4 * c4;
        s_v[i] += acc.x; s_v[i + 1] += acc.y;
        if (i + 2 < NV3) { s_v[i + 2] += acc.z; s_v[i + 3] += acc.w; }
    }
    if (tid < 5) {
        float A[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            A[4 * r + 0] = s_root[3 * r + 0]; A[4 * r + 1] = s_root[3 * r + 1]; A[4 * r + 2] = s_root[3 * r + 2];
            A[4 * r + 3] = s_J[r];
        }
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) s_A[e] = A[e];
        }
        int parent = 0;
        for (int l = 0; l < 3; ++l) {
            const int j = 1 + 3 * tid + l;
            const float* R = s_rot + 9 * (j - 1);
            const float t0 = s_J[3 * j] - s_J[3 * parent], t1 = s_J[3 * j + 1] - s_J[3 * parent + 1], t2 = s_J[3 * j + 2] - s_J[3 * parent + 2];
            float N[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float a0 = A[4 * r], a1 = A[4 * r + 1], a2 = A[4 * r + 2], a3 = A[4 * r + 3];
                N[4 * r + 0] = a0 * R[0] + a1 * R[3] + a2 * R[6];
                N[4 * r + 1] = a0 * R[1] + a1 * R[4] + a2 * R[7];
                N[4 * r + 2] = a0 * R[2] + a1 * R[5] + a2 * R[8];
                N[4 * r + 3] = a0 * t0 + a1 * t1 + a2 * t2 + a3;
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) { A[e] = N[e]; s_A[12 * j + e] = N[e]; }
            parent = j;
        }
    }
    __syncthreads();
    if (tid < NJ) {
        const float* A = s_A + 12 * tid;
        const float j0 = s_J[3 * tid], j1 = s_J[3 * tid + 1], j2 = s_J[3 * tid + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            s_A2[12 * tid + 4 * r + 0] = A[4 * r + 0];
            s_A2[12 * tid + 4 * r + 1] = A[4 * r + 1];
            s_A2[12 * tid + 4 * r + 2] = A[4 * r + 2];
            s_A2[12 * tid + 4 * r + 3] = A[4 * r + 3] - (A[4 * r] * j0 + A[4 * r + 1] * j1 + A[4 * r + 2] * j2);
        }
    }
    __syncthreads();
    auto blend = [&](int v, float (&T)[12]) {            // T = sum_k w[v][k] A'[k]
        const float4* wp = reinterpret_cast<const float4*>(a.t.weights + 16 * v);
        float w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t4 = wp[q]; w[4 * q] = t4.x; w[4 * q + 1] = t4.y; w[4 * q + 2] = t4.z; w[4 * q + 3] = t4.w; }
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4* ak = reinterpret_cast<const float4*>(s_A2 + 12 * k);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 t4 = ak[r];
                T[4 * r] = fmaf(t4.x, w[k], T[4 * r]); T[4 * r + 1] = fmaf(t4.y, w[k], T[4 * r + 1]);
                T[4 * r + 2] = fmaf(t4.z, w[k], T[4 * r + 2]); T[4 * r + 3] = fmaf(t4.w, w[k], T[4 * r + 3]);
            }
        }
    };
    for (int v = tid; v < NV; v += BT) {
        float T[12];
        blend(v, T);
        const float x = s_v[3 * v], y = s_v[3 * v + 1], z = s_v[3 * v + 2];
        s_vert[3 * v + 0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        s_vert[3 * v + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        s_vert[3 * v + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
    __syncthreads();
    if (tid < 21) {
        const int src = kReorderJb[tid];
        float x, y, z;
        if (src < 16) { x = s_A[12 * src + 3]; y = s_A[12 * src + 7]; z = s_A[12 * src + 11]; }
        else { const int v = kTipsb[a.t.side][src - 16]; x = s_vert[3 * v]; y = s_vert[3 * v + 1]; z = s_vert[3 * v + 2]; }
        s_jtr[3 * tid] = x; s_jtr[3 * tid + 1] = y; s_jtr[3 * tid + 2] = z;
    }
    __syncthreads();
    if (tid < 3) s_c[tid] = center >= 0 ? s_jtr[3 * center + tid] : 0.f;
    __syncthreads();

    // ================================================================ backward
    // ---- projection (uv = s (p - c).xy + t) and centring: g of every position, sum of them (-> g c), g s, g t
    const float sc = s_cam[0];
    float red[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // sum g.x, g.y, g.z | g s | g tx, g ty
    for (int v = tid; v < NV; v += BT) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (a.g_verts) { const float* g = a.g_verts + ((size_t)b * NV + v) * 3; gx = g[0]; gy = g[1]; gz = g[2]; }
        if (a.g_mesh_uv && a.cam) {
            const float* g = a.g_mesh_uv + ((size_t)b * NV + v) * 2;
            red[3] += g[0] * (s_vert[3 * v] - s_c[0]) + g[1] * (s_vert[3 * v + 1] - s_c[1]);
            red[4] += g[0]; red[5] += g[1];
            gx = fmaf(sc, g[0], gx); gy = fmaf(sc, g[1], gy);
        }
        s_gv[3 * v] = gx; s_gv[3 * v + 1] = gy; s_gv[3 * v + 2] = gz;
        red[0] += gx; red[1] += gy; red[2] += gz;
    }
    if (tid < 21) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (a.g_joints) { const float* g = a.g_joints + ((size_t)b * 21 + tid) * 3; gx = g[0]; gy = g[1]; gz = g[2]; }
        if (a.g_joint_uv && a.cam) {
            const float* g = a.g_joint_uv + ((size_t)b * 21 + tid) * 2;
            red[3] += g[0] * (s_jtr[3 * tid] - s_c[0]) + g[1] * (s_jtr[3 * tid + 1] - s_c[1]);
            red[4] += g[0]; red[5] += g[1];
            gx = fmaf(sc, g[0], gx); gy = fmaf(sc, g[1], gy);
        }
        s_gj[3 * tid] = gx; s_gj[3 * tid + 1] = gy; s_gj[3 * tid + 2] = gz;
        red[0] += gx; red[1] += gy; red[2] += gz;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) { const float s = dir::wave_sum(red[e]); if (lane == 0) s_red[wave][e] = s; }
    if (tid < 48) { s_gAt[tid] = 0.f; }
    __syncthreads();
    if (tid < 6) s_sum[tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    __syncthreads();
    if (tid < 3 && center >= 0) s_gj[3 * center + tid] -= s_sum[tid];           // c = jtr[center] is subtracted from every output position
    __syncthreads();
    if (tid < 21) {          // joints back to their sources: chain joint translations, or fingertip vertices
        const int src = kReorderJb[tid];
        if (src < 16) { s_gAt[3 * src] = s_gj[3 * tid]; s_gAt[3 * src + 1] = s_gj[3 * tid + 1]; s_gAt[3 * src + 2] = s_gj[3 * tid + 2]; }
        else {
            const int v = kTipsb[a.t.side][src - 16];
            s_gv[3 * v] += s_gj[3 * tid]; s_gv[3 * v + 1] += s_gj[3 * tid + 1]; s_gv[3 * v + 2] += s_gj[3 * tid + 2];
        }
    }
    __syncthreads();
    // ---- skinning: vert = T.R v_posed + T.t  ->  g v_posed = T.R^T g vert
    for (int v = tid; v < NV; v += BT) {
        float T[12];
        blend(v, T);
        const float gx = s_gv[3 * v], gy = s_gv[3 * v + 1], gz = s_gv[3 * v + 2];
        s_gvp[3 * v + 0] = T[0] * gx + T[4] * gy + T[8] * gz;
        s_gvp[3 * v + 1] = T[1] * gx + T[5] * gy + T[9] * gz;
        s_gvp[3 * v + 2] = T[2] * gx + T[6] * gy + T[10] * gz;
    }
    if (tid < 2) s_gvp[NV3 + tid] = 0.f;                 // padding of the 2336-float rows
    // ---- g A'[k][r][c] = sum_v w[v][k] g vert[v][r] (c < 3 ? v_posed[v][c] : 1): 192 reductions over the vertices
    if (tid < NJ * 12) {
        const int k = tid / 12, e = tid - 12 * k, r = e >> 2, c = e & 3;
        float acc = 0.f;
        int v = 0;
        for (; v + 8 <= NV; v += 8) {                      // eight skinning weights in flight, accumulated in vertex order (the same sum): the loop
            float w[8];                                    // was 778 dependent global loads per thread, most of this kernel's 233 us
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = a.t.weights[16 * (v + u) + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(w[u], s_gv[3 * (v + u) + r] * (c < 3 ? s_v[3 * (v + u) + c] : 1.f), acc);
        }
        for (; v < NV; ++v) acc = fmaf(a.t.weights[16 * v + k], s_gv[3 * v + r] * (c < 3 ? s_v[3 * v + c] : 1.f), acc);
        s_gA2[tid] = acc;
    }
    __syncthreads();
    // ---- A' = [A.R | A.t - A.R J]  ->  g A.R = g A'.R - g A'.t (x) J ; g A.t = g A'.t (+ the joint's own gradient) ; g J = -A.R^T g A'.t
    if (tid < NJ) {
        const float* A = s_A + 12 * tid;
        const float* G = s_gA2 + 12 * tid;
        const float j0 = s_J[3 * tid], j1 = s_J[3 * tid + 1], j2 = s_J[3 * tid + 2];
        const float t0 = G[3], t1 = G[7], t2 = G[11];
        const float tt[3] = {t0, t1, t2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            s_gA[12 * tid + 4 * r + 0] = G[4 * r + 0] - tt[r] * j0;
            s_gA[12 * tid + 4 * r + 1] = G[4 * r + 1] - tt[r] * j1;
            s_gA[12 * tid + 4 * r + 2] = G[4 * r + 2] - tt[r] * j2;
            s_gA[12 * tid + 4 * r + 3] = tt[r] + s_gAt[3 * tid + r];
        }
        s_gJ[3 * tid + 0] = -(A[0] * t0 + A[4] * t1 + A[8] * t2);
        s_gJ[3 * tid + 1] = -(A[1] * t0 + A[5] * t1 + A[9] * t2);
        s_gJ[3 * tid + 2] = -(A[2] * t0 + A[6] * t1 + A[10] * t2);
    }
    __syncthreads();
    // ---- kinematic chain, leaf to root: A_j = A_p [R_j | d_j], d_j = J_j - J_p
    if (tid < 5) {
        float carry[12];                                 // gradient flowing into A_p from its child inside this finger
#pragma unroll
        for (int e = 0; e < 12; ++e) carry[e] = 0.f;
        for (int l = 2; l >= 0; --l) {
            const int j = 1 + 3 * tid + l, p = l == 0 ? 0 : j - 1;
            float G[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) G[e] = s_gA[12 * j + e] + carry[e];
            const float* Ap = s_A + 12 * p;
            const float* R = s_rot + 9 * (j - 1);
            const float d0 = s_J[3 * j] - s_J[3 * p], d1 = s_J[3 * j + 1] - s_J[3 * p + 1], d2 = s_J[3 * j + 2] - s_J[3 * p + 2];
            // g R_j = A_p.R^T G.R
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    s_gR[9 * (j - 1) + 3 * r + c] = Ap[r] * G[c] + Ap[4 + r] * G[4 + c] + Ap[8 + r] * G[8 + c];
            // g d_j = A_p.R^T G.t
            const float gd0 = Ap[0] * G[3] + Ap[4] * G[7] + Ap[8] * G[11];
            const float gd1 = Ap[1] * G[3] + Ap[5] * G[7] + Ap[9] * G[11];
            const float gd2 = Ap[2] * G[3] + Ap[6] * G[7] + Ap[10] * G[11];
            s_gJ[3 * j] += gd0; s_gJ[3 * j + 1] += gd1; s_gJ[3 * j + 2] += gd2;
            // into the parent: g A_p.R = G.R R_j^T + G.t (x) d_j ; g A_p.t = G.t
            float P[12];
            const float dd[3] = {d0, d1, d2};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    P[4 * r + c] = G[4 * r] * R[3 * c] + G[4 * r + 1] * R[3 * c + 1] + G[4 * r + 2] * R[3 * c + 2] + G[4 * r + 3] * dd[c];
                P[4 * r + 3] = G[4 * r + 3];
            }
            if (l > 0) {
                s_gJ[3 * p] -= gd0; s_gJ[3 * p + 1] -= gd1; s_gJ[3 * p + 2] -= gd2;     // p = j - 1: this finger's own joint
#pragma unroll
                for (int e = 0; e < 12; ++e) carry[e] = P[e];
            } else {
#pragma unroll
                for (int e = 0; e < 12; ++e) s_part[tid][e] = P[e];
                s_part[tid][12] = -gd0; s_part[tid][13] = -gd1; s_part[tid][14] = -gd2;
            }
        }
    }
    __syncthreads();
    if (tid < 15) {          // root: A_0 = [R_root | J_0]; fixed order over the five fingers
        float acc = tid < 12 ? s_gA[tid] : 0.f;
        for (int f = 0; f < 5; ++f) acc += s_part[f][tid];
        if (tid < 12) s_gA[tid] = acc; else s_gJ[tid - 12] += acc;
    }
    __syncthreads();
    if (tid < 3) s_gJ[tid] += s_gA[4 * tid + 3];
    if (tid >= 64 && tid < 73) { const int e = tid - 64; s_groot[e] = s_gA[4 * (e / 3) + (e % 3)]; }
    // ---- pose blend shapes: g pm[k] = <posedirs_t[k], g v_posed>  (wave w: rows w, w + 4, ...), added to g R (pm = R - I)
    for (int k = wave; k < 135; k += BT / 64) {
        const float* row = a.t.posedirs_t + (size_t)k * NV3P;
        float acc = 0.f;
        int i = lane;
        for (; i + 448 < NV3; i += 512) {                  // eight table loads in flight, accumulated in order (the same sum)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = row[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(t[u], s_gvp[i + 64 * u], acc);
        }
        for (; i < NV3; i += 64) acc = fmaf(row[i], s_gvp[i], acc);
        acc = dir::wave_sum(acc);
        if (lane == 0) s_pm[k] = acc;                     // (s_pm re-used: the forward value is not needed any more)
    }
    __syncthreads();
    // ---- Rodrigues (via quaternion) chain rule per joint; robust 6D chain rule for the root
    if (tid < 15) {
        float g[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) g[e] = s_gR[9 * tid + e] + s_pm[9 * tid + e];
        const float vx = s_full[3 * tid], vy = s_full[3 * tid + 1], vz = s_full[3 * tid + 2];
        const float ex = vx + 1e-8f, ey = vy + 1e-8f, ez = vz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float ax = vx / angle, ay = vy / angle, az = vz / angle;
        const float half = angle * 0.5f;
        const float cw = cosf(half), sn = sinf(half);
        const float ux = sn * ax, uy = sn * ay, uz = sn * az;
        const float qn = sqrtf(cw * cw + ux * ux + uy * uy + uz * uz);
        const float w = cw / qn, x = ux / qn, y = uy / qn, z = uz / qn;
        // R(q) -> q
        float gw = 2 * w * (g[0] + g[4] + g[8]) + 2 * (-z * g[1] + y * g[2] + z * g[3] - x * g[5] - y * g[6] + x * g[7]);
        float gx = 2 * x * (g[0] - g[4] - g[8]) + 2 * (y * g[1] + z * g[2] + y * g[3] - w * g[5] + z * g[6] + w * g[7]);
        float gy = 2 * y * (-g[0] + g[4] - g[8]) + 2 * (x * g[1] + w * g[2] + x * g[3] + z * g[5] - w * g[6] + z * g[7]);
        float gz = 2 * z * (-g[0] - g[4] + g[8]) + 2 * (-w * g[1] + x * g[2] + w * g[3] + y * g[5] + x * g[6] + y * g[7]);
        // q = u / |u|
        const float dq = w * gw + x * gx + y * gy + z * gz;
        gw = (gw - w * dq) / qn; gx = (gx - x * dq) / qn; gy = (gy - y * dq) / qn; gz = (gz - z * dq) / qn;
        // u = (cos(half), sin(half) a)
        const float gsn = ax * gx + ay * gy + az * gz;
        const float gax = sn * gx, gay = sn * gy, gaz = sn * gz;
        const float ghalf = -sn * gw + cw * gsn;
        // a = v / angle, half = angle / 2, angle = |v + 1e-8|
        const float gangle = 0.5f * ghalf - (vx * gax + vy * gay + vz * gaz) / (angle * angle);
        s_gfull[3 * tid + 0] = gax / angle + gangle * ex / angle;
        s_gfull[3 * tid + 1] = gay / angle + gangle * ey / angle;
        s_gfull[3 * tid + 2] = gaz / angle + gangle * ez / angle;
    }
    if (tid == 64 && a.g_pose) {
        // forward again, keeping every normalisation's output and magnitude
        float xr0 = s_pose[0], xr1 = s_pose[1], xr2 = s_pose[2], yr0 = s_pose[3], yr1 = s_pose[4], yr2 = s_pose[5];
        float xh0 = xr0, xh1 = xr1, xh2 = xr2, yh0 = yr0, yh1 = yr1, yh2 = yr2, mxh, myh, mm, mo, mX, mY, mZ;
        normalize3b(xh0, xh1, xh2, mxh);
        normalize3b(yh0, yh1, yh2, myh);
        float m0 = xh0 + yh0, m1 = xh1 + yh1, m2 = xh2 + yh2, o0 = xh0 - yh0, o1 = xh1 - yh1, o2 = xh2 - yh2;
        normalize3b(m0, m1, m2, mm);
        normalize3b(o0, o1, o2, mo);
        float X0 = m0 + o0, X1 = m1 + o1, X2 = m2 + o2, Y0 = m0 - o0, Y1 = m1 - o1, Y2 = m2 - o2;
        normalize3b(X0, X1, X2, mX);
        normalize3b(Y0, Y1, Y2, mY);
        float Z0 = X1 * Y2 - X2 * Y1, Z1 = X2 * Y0 - X0 * Y2, Z2 = X0 * Y1 - X1 * Y0;
        normalize3b(Z0, Z1, Z2, mZ);
        // g of the columns (row-major matrix with columns X, Y, Z)
        float gX0 = s_groot[0], gY0 = s_groot[1], gZ0 = s_groot[2], gX1 = s_groot[3], gY1 = s_groot[4], gZ1 = s_groot[5],
              gX2 = s_groot[6], gY2 = s_groot[7], gZ2 = s_groot[8];
        normalize3_bwd(Z0, Z1, Z2, mZ, gZ0, gZ1, gZ2);                      // Z = n(X x Y)
        // c = X x Y: g X += Y x g c ; g Y += g c x X
        gX0 += Y1 * gZ2 - Y2 * gZ1; gX1 += Y2 * gZ0 - Y0 * gZ2; gX2 += Y0 * gZ1 - Y1 * gZ0;
        gY0 += gZ1 * X2 - gZ2 * X1; gY1 += gZ2 * X0 - gZ0 * X2; gY2 += gZ0 * X1 - gZ1 * X0;
        normalize3_bwd(X0, X1, X2, mX, gX0, gX1, gX2);                      // X = n(m + o)
        normalize3_bwd(Y0, Y1, Y2, mY, gY0, gY1, gY2);                      // Y = n(m - o)
        float gm0 = gX0 + gY0, gm1 = gX1 + gY1, gm2 = gX2 + gY2, go0 = gX0 - gY0, go1 = gX1 - gY1, go2 = gX2 - gY2;
        normalize3_bwd(m0, m1, m2, mm, gm0, gm1, gm2);                      // m = n(x^ + y^)
        normalize3_bwd(o0, o1, o2, mo, go0, go1, go2);                      // o = n(x^ - y^)
        float gxh0 = gm0 + go0, gxh1 = gm1 + go1, gxh2 = gm2 + go2, gyh0 = gm0 - go0, gyh1 = gm1 - go1, gyh2 = gm2 - go2;
        normalize3_bwd(xh0, xh1, xh2, mxh, gxh0, gxh1, gxh2);
        normalize3_bwd(yh0, yh1, yh2, myh, gyh0, gyh1, gyh2);
        float* gp = a.g_pose + (size_t)b * a.g_pose_stride;
        gp[0] = gxh0; gp[1] = gxh1; gp[2] = gxh2; gp[3] = gyh0; gp[4] = gyh1; gp[5] = gyh2;
    }
    // ---- shape blend: g beta[k] = <shapedirs_t[k], g v_shaped> + <j_shapedirs[:, k], g J>   (g v_shaped = g v_posed)
    for (int k = wave; k < 10; k += BT / 64) {
        const float* row = a.t.shapedirs_t + (size_t)k * NV3P;
        float acc = 0.f;
        int i = lane;
        for (; i + 448 < NV3; i += 512) {                  // eight table loads in flight, accumulated in order (the same sum)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = row[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(t[u], s_gvp[i + 64 * u], acc);
        }
        for (; i < NV3; i += 64) acc = fmaf(row[i], s_gvp[i], acc);
        if (lane < 48) acc = fmaf(a.t.j_shapedirs[lane * 10 + k], s_gJ[lane], acc);
        acc = dir::wave_sum(acc);
        if (lane == 0 && a.g_betas) a.g_betas[(size_t)b * a.g_betas_stride + k] = acc;
    }
    __syncthreads();
    // ---- PCA: full = mean + pca . comps  ->  g pca[k] = <comps[k], g full>
    if (tid < 45 && a.g_pose) {
        float acc = 0.f;
        for (int t = 0; t < 45; ++t) acc = fmaf(a.t.comps[tid * 45 + t], s_gfull[t], acc);
        a.g_pose[(size_t)b * a.g_pose_stride + 6 + tid] = acc;
    }
    if (tid >= 64 && tid < 67 && a.g_cam) a.g_cam[(size_t)b * a.g_cam_stride + tid - 64] = a.cam ? s_sum[3 + tid - 64] : 0.f;
}

}  // namespace

extern "C" int dir_mano_backward_pair(const dir_mano_tables* tables_lr, const float* const* pose_lr, int pose_stride,
                                      const float* const* betas_lr, int betas_stride, const float* const* cam_lr, int cam_stride,
                                      const float* const* g_verts_lr, const float* const* g_joints_lr, const float* const* g_joint_uv_lr,
                                      const float* const* g_mesh_uv_lr, float* const* g_pose_lr, int g_pose_stride, float* const* g_betas_lr,
                                      int g_betas_stride, float* const* g_cam_lr, int g_cam_stride, int hands, int B, void* stream) {
    using namespace dir;
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(B > 0 && (hands == 1 || hands == 2) && tables_lr && pose_lr && betas_lr && g_pose_lr && g_betas_lr, "dir_mano_backward_pair: bad arguments");
    ManoBwdArgs a;
    for (int h = 0; h < hands; ++h) {
        const dir_mano_tables& t = tables_lr[h];
        DIR_REQUIRE(t.shapedirs_t && t.posedirs_t && t.v_template && t.j_template && t.j_shapedirs && t.weights && t.hands_mean && t.comps,
                    "dir_mano_backward_pair: null table");
        DIR_REQUIRE(!t.root_palm && (t.side == 0 || t.side == 1) && t.center_idx >= -1 && t.center_idx < 21, "dir_mano_backward_pair: unsupported table flags");
        DIR_REQUIRE(pose_lr[h] && betas_lr[h] && g_pose_lr[h] && g_betas_lr[h] && pose_stride >= 51 && betas_stride >= 10 && g_pose_stride >= 51 && g_betas_stride >= 10,
                    "dir_mano_backward_pair: bad pointers / strides");
        const float* cam = cam_lr ? cam_lr[h] : nullptr;
        DIR_REQUIRE(!cam || (cam_stride >= 3 && (!g_cam_lr || !g_cam_lr[h] || g_cam_stride >= 3)), "dir_mano_backward_pair: bad cam strides");
        a.h[h] = ManoBwdHand{t, pose_lr[h], pose_stride, betas_lr[h], betas_stride, cam, cam_stride,
                             g_verts_lr ? g_verts_lr[h] : nullptr, g_joints_lr ? g_joints_lr[h] : nullptr,
                             g_joint_uv_lr ? g_joint_uv_lr[h] : nullptr, g_mesh_uv_lr ? g_mesh_uv_lr[h] : nullptr,
                             g_pose_lr[h], g_pose_stride, g_betas_lr[h], g_betas_stride, g_cam_lr ? g_cam_lr[h] : nullptr, g_cam_stride};
    }
    if (hands == 1) a.h[1] = a.h[0];
    DIR_LAUNCH(mano_backward_kernel, dim3(B, hands), dim3(BT), 0, (hipStream_t)stream, a);
    return check_launch("dir_mano_backward_pair");
}
