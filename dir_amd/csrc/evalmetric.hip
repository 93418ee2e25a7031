// the evaluation-metric maths of the reference's apps/eval.py:151-241 as one launch per batch.
// One workgroup per sample handles both hands: vertices are staged once in LDS (37 KB), the 2 x 2 x 21 joint
// regressions (apps/eval.py:43-44) run one (hand, joint) per wave pass with fp64 accumulation, and the alignment /
// error maths then follows the reference's fp32 operation order.  HBM-bound: ~62 KB per sample in + out; the two
// [21,778] regressors are shared by every workgroup and stay in L2.
#include "dir_common.h"

namespace dir {
namespace {

constexpr int NV = 778, NJ = 21, EV_THREADS = 256;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// xyz2uvd (apps/eval.py:82-85) for one point: (p @ cam^T)[:2] / (p @ cam^T)[2]
__device__ __forceinline__ void project(const float* __restrict__ cam, float x, float y, float z, float& u, float& v) {
    const float a = fmaf(cam[2], z, fmaf(cam[1], y, cam[0] * x));
    const float b = fmaf(cam[5], z, fmaf(cam[4], y, cam[3] * x));
    const float c = fmaf(cam[8], z, fmaf(cam[7], y, cam[6] * x));
    u = a / c;
    v = b / c;
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }

__global__ __launch_bounds__(256) void joint_regress_kernel(const float* __restrict__ jr, const float* __restrict__ verts,
                                                             float* __restrict__ joints) {
    __shared__ float s_v[NV * 3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* v = verts + (size_t)b * NV * 3;
    for (int i = tid; i < NV * 3; i += 256) s_v[i] = v[i];
    __syncthreads();
    for (int j = wave; j < NJ; j += 4) {
        double ax = 0, ay = 0, az = 0;
        for (int k = lane; k < NV; k += 64) {
            const double w = jr[j * NV + k];
            ax = fma(w, (double)s_v[k * 3], ax);
            ay = fma(w, (double)s_v[k * 3 + 1], ay);
            az = fma(w, (double)s_v[k * 3 + 2], az);
        }
        ax = wave_sum_d(ax), ay = wave_sum_d(ay), az = wave_sum_d(az);
        if (lane == 0) {
            float* o = joints + ((size_t)b * NJ + j) * 3;
            o[0] = (float)ax, o[1] = (float)ay, o[2] = (float)az;
        }
    }
}

__global__ __launch_bounds__(EV_THREADS) void eval_metrics_kernel(dir_eval_inputs in, dir_eval_outputs out, int root_joint,
                                                                   int use_scale) {
    __shared__ float s_vg[2][NV * 3], s_vp[2][NV * 3];
    __shared__ float s_jg[2][NJ][3], s_jp[2][NJ][3];       // regressed joints: GT (camera space), prediction (model space)
    __shared__ float s_scale[2], s_cam[9];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int h = 0; h < 2; ++h) {
        const float* g = in.verts_gt[h] + (size_t)b * NV * 3;
        const float* p = in.verts_pd[h] + (size_t)b * NV * 3;
        for (int i = tid; i < NV * 3; i += EV_THREADS) s_vg[h][i] = g[i], s_vp[h][i] = p[i];
    }
    if (tid < 9) s_cam[tid] = in.cam[(size_t)b * 9 + tid];
    __syncthreads();

    // apps/eval.py:151-152,173-174: joints = Jr @ verts, 42 (hand, joint) tasks over the 4 waves
    for (int t = wave; t < 2 * NJ; t += EV_THREADS / 64) {
        const int h = t / NJ, j = t - h * NJ;
        const float* w = in.jr[h] + j * NV;
        double a[6] = {0, 0, 0, 0, 0, 0};
        for (int k = lane; k < NV; k += 64) {
            const double wk = w[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a[c] = fma(wk, (double)s_vg[h][k * 3 + c], a[c]);
                a[3 + c] = fma(wk, (double)s_vp[h][k * 3 + c], a[3 + c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) a[c] = wave_sum_d(a[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s_jg[h][j][c] = (float)a[c], s_jp[h][j][c] = (float)a[3 + c];
        }
    }
    __syncthreads();

    // :160-161,178-185: scale = |J9 - J0|_gt / |J9 - J0|_pred
    if (tid < 2) {
        const int h = tid;
        const float lg = norm3(s_jg[h][9][0] - s_jg[h][0][0], s_jg[h][9][1] - s_jg[h][0][1], s_jg[h][9][2] - s_jg[h][0][2]);
        const float lp = norm3(s_jp[h][9][0] - s_jp[h][0][0], s_jp[h][9][1] - s_jp[h][0][1], s_jp[h][9][2] - s_jp[h][0][2]);
        s_scale[h] = use_scale ? lg / lp : 1.0f;
    }
    __syncthreads();

    // joints (:187-196, 214-215, 225-231)
    if (tid < 2 * NJ) {
        const int h = tid / NJ, j = tid - h * NJ;
        const float* rg = s_jg[h][root_joint];
        const float* rp = s_jp[h][root_joint];
        const float sc = s_scale[h];
        float pa[3], gr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pa[c] = (s_jp[h][j][c] - rp[c]) * sc;
            gr[c] = s_jg[h][j][c] - rg[c];
        }
        const size_t o = ((size_t)b * 2 + h) * NJ + j;
        if (out.joint_err) out.joint_err[o] = norm3(pa[0] - gr[0], pa[1] - gr[1], pa[2] - gr[2]);
        if (out.joints_pd) out.joints_pd[o * 3] = pa[0], out.joints_pd[o * 3 + 1] = pa[1], out.joints_pd[o * 3 + 2] = pa[2];
        if (out.joints_gt) out.joints_gt[o * 3] = gr[0], out.joints_gt[o * 3 + 1] = gr[1], out.joints_gt[o * 3 + 2] = gr[2];
        if (out.joint2d_err) {
            float u0, v0, u1, v1;
            project(s_cam, s_jg[h][j][0], s_jg[h][j][1], s_jg[h][j][2], u0, v0);               // :153-154
            project(s_cam, pa[0] + rg[0], pa[1] + rg[1], pa[2] + rg[2], u1, v1);                 // :214-215
            out.joint2d_err[o] = sqrtf(fmaf(v1 - v0, v1 - v0, (u1 - u0) * (u1 - u0)));
        }
    }

    // vertices (:188,190,204-217)
    for (int i = tid; i < 2 * NV; i += EV_THREADS) {
        const int h = i >= NV, k = i - h * NV;
        const float* rg = s_jg[h][root_joint];
        const float* rp = s_jp[h][root_joint];
        const float sc = s_scale[h];
        float pa[3], gr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pa[c] = (s_vp[h][k * 3 + c] - rp[c]) * sc;
            gr[c] = s_vg[h][k * 3 + c] - rg[c];
        }
        const size_t o = ((size_t)b * 2 + h) * NV + k;
        if (out.vert_err) out.vert_err[o] = norm3(pa[0] - gr[0], pa[1] - gr[1], pa[2] - gr[2]);
        if (out.vert2d_err) {
            float u, v;
            project(s_cam, pa[0] + rg[0], pa[1] + rg[1], pa[2] + rg[2], u, v);                   // :212-213
            const float2 g2 = reinterpret_cast<const float2*>(in.verts2d_gt[h])[(size_t)b * NV + k];
            out.vert2d_err[o] = sqrtf(fmaf(v - g2.y, v - g2.y, (u - g2.x) * (u - g2.x)));
        }
    }

    // relative root (:156,170,233-239)
    if (tid == 0 && out.root_err) {
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gt_off = s_jg[1][root_joint][c] - s_jg[0][root_joint][c];
            float rel = in.pd_offset[(size_t)b * 3 + c] * 0.15f;
            if (root_joint != 0) rel = (s_jp[1][root_joint][c] + rel) - s_jp[0][root_joint][c];
            d[c] = gt_off - rel;
        }
        out.root_err[b] = norm3(d[0], d[1], d[2]);
    }
}

}  // namespace
}  // namespace dir

extern "C" int dir_joint_regress_forward(const float* jr, const float* verts, float* joints, int B, void* stream) {
    DIR_REQUIRE(B >= 0, "dir_joint_regress_forward: B=%d", B);
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(jr && verts && joints, "dir_joint_regress_forward: null pointer");
    DIR_LAUNCH(dir::joint_regress_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, jr, verts, joints);
    return dir::check_launch("dir_joint_regress_forward");
}

extern "C" int dir_eval_metrics_forward(const dir_eval_inputs* in, const dir_eval_outputs* out, int B, int root_joint,
                                        int use_scale, void* stream) {
    DIR_REQUIRE(in && out, "dir_eval_metrics_forward: null descriptor");
    DIR_REQUIRE(B >= 0, "dir_eval_metrics_forward: B=%d", B);
    DIR_REQUIRE(root_joint >= 0 && root_joint < 21, "dir_eval_metrics_forward: root_joint=%d not in [0,21)", root_joint);
    if (B == 0) return DIR_OK;
    for (int h = 0; h < 2; ++h)
        DIR_REQUIRE(in->verts_pd[h] && in->verts_gt[h] && in->jr[h], "dir_eval_metrics_forward: null input (hand %d)", h);
    DIR_REQUIRE(in->cam && in->pd_offset, "dir_eval_metrics_forward: null cam / pd_offset");
    DIR_REQUIRE(!out->vert2d_err || (in->verts2d_gt[0] && in->verts2d_gt[1]),
                "dir_eval_metrics_forward: vert2d_err requested without verts2d_gt");
    DIR_LAUNCH(dir::eval_metrics_kernel, dim3(B), dim3(dir::EV_THREADS), 0, (hipStream_t)stream, *in, *out, root_joint, use_scale);
    return dir::check_launch("dir_eval_metrics_forward");
}
