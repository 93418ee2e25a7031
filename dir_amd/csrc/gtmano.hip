// the reference's ground-truth MANO layer, models/manolayer.py:251-323 (ManoLayer.forward), as one launch.
// This is the formulation dataset/interhand.py:130-149 runs per sample on the CPU inside DataLoader workers: rotation-matrix
// root, PCA pose through the classic Rodrigues formula (:32-48), serial 16-joint SE(3) chain whose transforms already
// contain the rest-pose removal (t = (I - R) j), fingertips 745/317/444/556/673 for BOTH hands, optional centring, scale,
// translation and the `new_skel` knuckle override.  One 640-thread workgroup per sample; tables in the dir_mano_tables
// packing of mano.hip (k-major, 2336-float rows; joint regression folded into j_template / j_shapedirs).
#include "dir_common.h"

namespace {

constexpr int NV = 778, NV3 = 2334, NV3P = 2336, NJ = 16, NTHR = 640;
__constant__ int kOrder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};   // :111-116
__constant__ int kTipV[5] = {745, 317, 444, 556, 673};                                                        // :297

struct GtArgs {
    dir_mano_tables t;
    const float* root; const float* pose; const float* shape; const float* trans; const float* scale;
    float* verts; float* joints;
    int ncomps;        // > 0: PCA coefficients [B][ncomps]; 0: rotation matrices [B][15][9]
    int center_idx, new_skel;
};

__global__ __launch_bounds__(NTHR) void gt_mano_kernel(GtArgs a) {
    __shared__ float s_v[NV3P];
    __shared__ float s_in[45], s_beta[10], s_axis[45];
    __shared__ float s_rot[16 * 9];     // [0] = root, [1..15] = articulated joints
    __shared__ float s_pm[135];
    __shared__ float s_J[NJ * 3];
    __shared__ float s_M[NJ * 12];      // SE3_j (top three rows), models/manolayer.py:275-284
    __shared__ float s_jt[21 * 3];
    __shared__ float s_c[3];
    const int b = blockIdx.x, tid = threadIdx.x;

    if (a.ncomps > 0) { if (tid < a.ncomps) s_in[tid] = a.pose[(size_t)b * a.ncomps + tid]; }
    else if (tid < 135) s_rot[9 + tid] = a.pose[(size_t)b * 135 + tid];
    if (tid >= 192 && tid < 201) s_rot[tid - 192] = a.root[(size_t)b * 9 + tid - 192];
    if (tid >= 256 && tid < 266) s_beta[tid - 256] = a.shape[(size_t)b * 10 + tid - 256];
    __syncthreads();

    // ---- pca2axis (:161-164) ; shape blend (:265-266) ; joints of the shaped template (:268, folded regression)
    if (a.ncomps > 0 && tid < 45) {
        float acc = 0.f;
        for (int k = 0; k < a.ncomps; ++k) acc = fmaf(s_in[k], a.t.comps[k * 45 + tid], acc);
        s_axis[tid] = acc + a.t.hands_mean[tid];
    }
    for (int i = tid; i < NV3; i += NTHR) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(a.t.shapedirs_t[k * NV3P + i], s_beta[k], acc);
        s_v[i] = a.t.v_template[i] + acc;
    }
    if (tid >= 576 && tid < 576 + NJ * 3) {
        const int o = tid - 576;
        float acc = a.t.j_template[o];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc = fmaf(a.t.j_shapedirs[o * 10 + k], s_beta[k], acc);
        s_J[o] = acc;
    }
    __syncthreads();

    // ---- rodrigues_batch (:32-48): R = I + sin(angle) L + (1 - cos(angle)) L L, angle = |axis| + 1e-8
    if (a.ncomps > 0 && tid < 15) {
#pragma clang fp contract(off)
        const float vx = s_axis[3 * tid], vy = s_axis[3 * tid + 1], vz = s_axis[3 * tid + 2];
        const float angle = sqrtf(vx * vx + vy * vy + vz * vz) + 1e-8f;
        const float x = vx / angle, y = vy / angle, z = vz / angle;
        const float sn = sinf(angle), oc = 1.f - cosf(angle);
        const float L[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
        float* R = s_rot + 9 * (tid + 1);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float ll = L[3 * r] * L[c] + L[3 * r + 1] * L[3 + c] + L[3 * r + 2] * L[6 + c];
                R[3 * r + c] = (r == c ? 1.f : 0.f) + sn * L[3 * r + c] + oc * ll;
            }
    }
    __syncthreads();
    if (tid < 135) s_pm[tid] = s_rot[9 + tid] - ((tid % 9 == 0 || tid % 9 == 4 || tid % 9 == 8) ? 1.f : 0.f);   // :270-271
    __syncthreads();

    // ---- pose blend shapes (:272-273)
    if (tid < NV3P / 4) {
        const float4* pd = reinterpret_cast<const float4*>(a.t.posedirs_t) + tid;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 9
        for (int k = 0; k < 135; ++k) {
            const float4 p = pd[k * (NV3P / 4)];
            const float w = s_pm[k];
            acc.x = fmaf(p.x, w, acc.x); acc.y = fmaf(p.y, w, acc.y); acc.z = fmaf(p.z, w, acc.z); acc.w = fmaf(p.w, w, acc.w);
        }
        const int i = 4 * tid;
        s_v[i] += acc.x; s_v[i + 1] += acc.y;
        if (i + 2 < NV3) { s_v[i + 2] += acc.z; s_v[i + 3] += acc.w; }
    }
    // ---- serial chain (:275-284): SE3_j[i] = SE3_j[parent] . [R_i | (I - R_i) j_i]; finger f owns joints 1+3f .. 3+3f
    if (tid < 5) {
        float A[12];
        {
            const float* R = s_rot;
            const float j0 = s_J[0], j1 = s_J[1], j2 = s_J[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                A[4 * r] = R[3 * r]; A[4 * r + 1] = R[3 * r + 1]; A[4 * r + 2] = R[3 * r + 2];
                A[4 * r + 3] = (r == 0 ? j0 : r == 1 ? j1 : j2) - (R[3 * r] * j0 + R[3 * r + 1] * j1 + R[3 * r + 2] * j2);
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) s_M[e] = A[e];
        }
        for (int l = 0; l < 3; ++l) {
            const int j = 1 + 3 * tid + l;
            const float* R = s_rot + 9 * j;
            const float p0 = s_J[3 * j], p1 = s_J[3 * j + 1], p2 = s_J[3 * j + 2];
            float t[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) t[r] = (r == 0 ? p0 : r == 1 ? p1 : p2) - (R[3 * r] * p0 + R[3 * r + 1] * p1 + R[3 * r + 2] * p2);
            float N[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float a0 = A[4 * r], a1 = A[4 * r + 1], a2 = A[4 * r + 2], a3 = A[4 * r + 3];
                N[4 * r + 0] = a0 * R[0] + a1 * R[3] + a2 * R[6];
                N[4 * r + 1] = a0 * R[1] + a1 * R[4] + a2 * R[7];
                N[4 * r + 2] = a0 * R[2] + a1 * R[5] + a2 * R[8];
                N[4 * r + 3] = a0 * t[0] + a1 * t[1] + a2 * t[2] + a3;
            }
            // j_withoutTips[j] = SE3_apply(SE3_j[parent], j_tpose[j])  (:286-289) -- A still holds the parent's transform
#pragma unroll
            for (int r = 0; r < 3; ++r) s_jt[3 * j + r] = A[4 * r] * p0 + A[4 * r + 1] * p1 + A[4 * r + 2] * p2 + A[4 * r + 3];
#pragma unroll
            for (int e = 0; e < 12; ++e) { A[e] = N[e]; s_M[12 * j + e] = N[e]; }
        }
    }
    if (tid >= 64 && tid < 67) s_jt[tid - 64] = s_J[tid - 64];
    __syncthreads();

    // ---- skinning (:292-295): SE3_v = weights . SE3_j ; v = R_v v_tpose + t_v
    for (int v = tid; v < NV; v += NTHR) {
        const float4* wp = reinterpret_cast<const float4*>(a.t.weights + 16 * v);
        float w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t4 = wp[q];
            w[4 * q] = t4.x; w[4 * q + 1] = t4.y; w[4 * q + 2] = t4.z; w[4 * q + 3] = t4.w;
        }
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = fmaf(s_M[12 * k + e], w[k], T[e]);
        const float x = s_v[3 * v], y = s_v[3 * v + 1], z = s_v[3 * v + 2];
        s_v[3 * v + 0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        s_v[3 * v + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        s_v[3 * v + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
    __syncthreads();

    // ---- joints: chain joints + fingertip vertices, reordered (:297-300); centre / scale / trans (:302-315); new_skel (:317-321)
    __shared__ float s_j21[21 * 3];
    if (tid < 63) {
        const int j = tid / 3, c = tid - 3 * j, src = kOrder[j];
        s_j21[tid] = src < 16 ? s_jt[3 * src + c] : s_v[3 * kTipV[src - 16] + c];
    }
    __syncthreads();
    if (tid < 3) s_c[tid] = a.center_idx >= 0 ? s_j21[3 * a.center_idx + tid] : 0.f;
    __syncthreads();
    const float sc = a.scale ? a.scale[b] : 1.f;
    const bool has_sc = a.scale != nullptr, has_c = a.center_idx >= 0;
    auto finish = [&](float v, int c) {
        if (has_c) v = v - s_c[c];
        if (has_sc) v = v * sc;
        if (a.trans) v = v + a.trans[(size_t)b * 3 + c];
        return v;
    };
    float* vout = a.verts + (size_t)b * NV3;
    for (int i = tid; i < NV3; i += NTHR) {
        const float v = finish(s_v[i], i % 3);
        s_v[i] = v;
        vout[i] = v;
    }
    __syncthreads();
    if (tid < 63) {
        const int j = tid / 3, c = tid - 3 * j;
        float v = finish(s_j21[tid], c);
        if (a.new_skel) {
            if (j == 5) v = (s_v[3 * 63 + c] + s_v[3 * 144 + c]) / 2;
            else if (j == 9) v = (s_v[3 * 271 + c] + s_v[3 * 220 + c]) / 2;
            else if (j == 13) v = (s_v[3 * 148 + c] + s_v[3 * 290 + c]) / 2;
            else if (j == 17) v = (s_v[3 * 770 + c] + s_v[3 * 83 + c]) / 2;
        }
        a.joints[(size_t)b * 63 + tid] = v;
    }
}

}  // namespace

extern "C" int dir_gt_mano_forward(const dir_mano_tables* t, const float* root_rotation, const float* pose, int ncomps,
                                   const float* shape, const float* trans, const float* scale, int center_idx,
                                   int new_skel, float* verts, float* joints, int B, void* stream) {
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(B > 0, "dir_gt_mano_forward: bad B");
    DIR_REQUIRE(t && root_rotation && pose && shape && verts && joints, "dir_gt_mano_forward: null pointer");
    DIR_REQUIRE(t->shapedirs_t && t->posedirs_t && t->v_template && t->j_template && t->j_shapedirs && t->weights &&
                    t->hands_mean && t->comps, "dir_gt_mano_forward: null table");
    DIR_REQUIRE(ncomps >= 0 && ncomps <= 45, "dir_gt_mano_forward: ncomps=%d not in [0,45] (0 = rotation-matrix pose)", ncomps);
    DIR_REQUIRE(center_idx >= -1 && center_idx < 21, "dir_gt_mano_forward: center_idx out of range");
    DIR_REQUIRE(((uintptr_t)t->posedirs_t & 15) == 0 && ((uintptr_t)t->weights & 15) == 0,
                "dir_gt_mano_forward: posedirs_t / weights must be 16-byte aligned");
    GtArgs a{*t, root_rotation, pose, shape, trans, scale, verts, joints, ncomps, center_idx, new_skel};
    DIR_LAUNCH(gt_mano_kernel, dim3(B), dim3(NTHR), 0, (hipStream_t)stream, a);
    return dir::check_launch("dir_gt_mano_forward");
}
