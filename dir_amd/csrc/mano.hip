// Fused MANO forward for gfx950: PCA -> Rodrigues (via quaternion) -> robust 6D root rotation -> shape
// and pose blend shapes -> 3-level kinematic chain -> LBS -> fingertips -> joint reorder -> root
// centring -> weak-perspective projection.  One 640-thread workgroup per (sample, hand) -- both hands of a stage go
// in ONE launch -- every intermediate lives in LDS (~11 KB); tables are read k-major with 16-byte loads (rows padded
// to 2336 floats) and stay L2 resident across the batch.
//
// Replaces manopth/manopth/manolayer.py:110-270 (+ rodrigues_layer.py:15-54, rot6d.py:26-60,
// tensutils.py:6-42) and utils/utils.py:47-63 of the reference: ~4040 ATen op calls and a host sync
// per call there, one launch here.
#include "dir_common.h"

#include <stdlib.h>

namespace {

constexpr int NV = 778, NV3 = 2334, NV3P = 2336, NJ = 16, NTHR = 640;   // table rows padded to 2336 floats (16-B aligned)

__constant__ int kReorderJ[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
__constant__ int kTips[2][5] = {{745, 317, 444, 556, 673}, {745, 317, 445, 556, 673}};

struct ManoHand {
    dir_mano_tables t;
    const float* pose; int pose_stride;
    const float* betas; int betas_stride;
    const float* cam; int cam_stride;
    float* verts; float* joints; float* joint_uv; float* mesh_uv; int32_t* flags;
};
struct ManoArgs { ManoHand h[2]; long long* stamps; int B, hands, parts; };   // stamps: DIR_STAMPS=mano (tuning aid, else NULL)

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
    // rot6d.py:54-60: v / max(|v|, 1e-8).  No FMA contraction: the robust-6D construction is
    // ill-conditioned for near-parallel inputs (x - y tiny), so follow the reference's op sequence exactly.
#pragma clang fp contract(off)
    float m = fmaxf(sqrtf(x * x + y * y + z * z), 1e-8f);
    x /= m; y /= m; z /= m;
}

// XCD-aware 1-D grid: a (hand, vertex part) combination lives on ONE XCD (8 / (hands * parts) XCDs each, samples dealt among them), so
// each XCD's L2 holds only that combination's slice of the blend-shape tables -- with hands = 2, parts = 4 exactly one slice per XCD.
// (The former (B, hands, parts) grid put every combination on every XCD: 24 MB of table traffic per launch for 2.8 MB of tables.)
// parts = vertex parts: with 4 parts every workgroup owns 196 vertices (588 floats, 16-byte aligned) of one (sample, hand),
// recomputes the cheap pose / chain maths, and streams only its quarter of the blend-shape tables -- the 1.3 MB posedirs
// read per (sample, hand) is spread over four CUs instead of one.  blockDim.x = NTHR (1 part) or PTHR (4 parts).
constexpr int PART_V = 196, PTHR = 256;   // >= PART_V threads: the skinning loop is one pass per part

// THREADS = NTHR (one part) or PTHR (four vertex parts): the launch bound sets the register budget -- under the 640-thread bound the
// 4-part launches (256 threads) were compiled down to 168 VGPRs and spilled (scratch traffic inside the skinning loop)
// SPW = samples per workgroup (same hand, same vertex part): every table value (shape / pose blend shapes, skinning weights) is loaded
// ONCE and applied to SPW samples -- at B = 64 the launch is bound by the L2 -> CU traffic of the per-workgroup table streams (512
// workgroups x 315 KB), which SPW divides; the per-sample arithmetic and its order are unchanged (bit-identical results).
template <int THREADS, int SPW>
__global__ __launch_bounds__(THREADS) void mano_forward_kernel(ManoArgs args) {
    const int combos = args.hands * args.parts, rep = 8 / combos;      // combos in {1, 2, 4, 8}
    const int xcd = blockIdx.x & 7, combo = xcd / rep;
    const int bg = (blockIdx.x >> 3) * rep + (xcd - combo * rep);      // sample group
    if (bg * SPW >= args.B) return;
    const int hand = combo / args.parts, part = combo - hand * args.parts;
    const ManoHand& a = args.h[hand];
    const int nthr = blockDim.x;
    const int v_lo = args.parts > 1 ? part * PART_V : 0;
    const int v_hi = args.parts > 1 ? min(NV, v_lo + PART_V) : NV;
    const int f_lo = 3 * v_lo, f_hi = 3 * v_hi;          // float range [f_lo, f_hi) of the flattened vertex array
    __shared__ float s_v[SPW][NV3P];         // v_shaped -> v_posed -> skinned vertices (in place)
    __shared__ float s_pose[SPW][51], s_beta[SPW][10], s_cam[SPW][3];
    __shared__ float s_full[SPW][45];        // axis-angle of the 15 articulated joints
    __shared__ float s_rot[SPW][15 * 9];     // rotation matrices (row major)
    __shared__ float s_pm[SPW][135];         // pose map = R - I
    __shared__ float s_root[SPW][9];
    __shared__ float s_J[SPW][NJ * 3];
    __shared__ float s_A[SPW][NJ * 12];      // global transforms (top 3 rows), th_j joint order
    __shared__ __attribute__((aligned(16))) float s_A2[SPW][NJ * 12];     // with the rest-pose joint removed: A' = A - pack(A.[J;0])
    __shared__ float s_jtr[SPW][21 * 3];
    __shared__ float s_c[SPW][3];

    const int tid = threadIdx.x;
    int nstamp = 0;
    auto stamp = [&]() { if (args.stamps && blockIdx.x == 0 && tid == 0) args.stamps[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    // sample s of the group: index bs(s); a group's tail beyond B re-computes sample B - 1 and writes nothing
    auto bs = [&](int s_) { return min(bg * SPW + s_, args.B - 1); };
    auto live = [&](int s_) { return bg * SPW + s_ < args.B; };

#pragma unroll
    for (int s_ = 0; s_ < SPW; ++s_) {
        const size_t b = (size_t)bs(s_);
        if (tid < 51) s_pose[s_][tid] = a.pose[b * a.pose_stride + tid];
        if (tid >= 64 && tid < 74) s_beta[s_][tid - 64] = a.betas[b * a.betas_stride + tid - 64];
        if (tid >= 128 && tid < 131) s_cam[s_][tid - 128] = a.cam ? a.cam[b * a.cam_stride + tid - 128] : 0.f;
    }
    __syncthreads(); stamp();

    // ---- PCA coefficients -> axis angle (manolayer.py:131-144) ; shape blend (manolayer.py:180-182)
#pragma unroll
    for (int s_ = 0; s_ < SPW; ++s_)
        if (tid < 45) {
            float acc = 0.f;
            for (int k = 0; k < 45; ++k) acc = fmaf(s_pose[s_][6 + k], a.t.comps[k * 45 + tid], acc);
            s_full[s_][tid] = a.t.hands_mean[tid] + acc;
        }
    for (int i = f_lo + tid; i < f_hi; i += nthr) {
        float acc[SPW];
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) acc[s_] = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float sd = a.t.shapedirs_t[k * NV3P + i];
#pragma unroll
            for (int s_ = 0; s_ < SPW; ++s_) acc[s_] = fmaf(sd, s_beta[s_][k], acc[s_]);
        }
        const float vt = a.t.v_template[i];
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) s_v[s_][i] = acc[s_] + vt;
    }
    if (tid >= 64 && tid < 64 + SPW) {
        // robust 6D -> rotation (rot6d.py:26-51): columns (x', y', z)
#pragma clang fp contract(off)
        const int s_ = tid - 64;
        float x0 = s_pose[s_][0], x1 = s_pose[s_][1], x2 = s_pose[s_][2], y0 = s_pose[s_][3], y1 = s_pose[s_][4], y2 = s_pose[s_][5];
        normalize3(x0, x1, x2);
        normalize3(y0, y1, y2);
        float m0 = x0 + y0, m1 = x1 + y1, m2 = x2 + y2;
        float o0 = x0 - y0, o1 = x1 - y1, o2 = x2 - y2;
        normalize3(m0, m1, m2);
        normalize3(o0, o1, o2);
        x0 = m0 + o0; x1 = m1 + o1; x2 = m2 + o2;
        y0 = m0 - o0; y1 = m1 - o1; y2 = m2 - o2;
        normalize3(x0, x1, x2);
        normalize3(y0, y1, y2);
        float z0 = x1 * y2 - x2 * y1, z1 = x2 * y0 - x0 * y2, z2 = x0 * y1 - x1 * y0;
        normalize3(z0, z1, z2);
        float* R = s_root[s_];
        R[0] = x0; R[1] = y0; R[2] = z0;
        R[3] = x1; R[4] = y1; R[5] = z1;
        R[6] = x2; R[7] = y2; R[8] = z2;
        if (a.flags && live(s_)) {
            float det = x0 * (y1 * z2 - z1 * y2) - y0 * (x1 * z2 - z1 * x2) + z0 * (x1 * y2 - y1 * x2);
            a.flags[bs(s_)] = det < 0.f ? 1 : 0;
        }
    }
    __syncthreads(); stamp();

    // ---- Rodrigues via quaternion (rodrigues_layer.py:43-54, 15-40)
    if (tid < 15 * SPW) {
#pragma clang fp contract(off)
        const int s_ = tid / 15, jt = tid - 15 * s_;
        float vx = s_full[s_][3 * jt], vy = s_full[s_][3 * jt + 1], vz = s_full[s_][3 * jt + 2];
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
        float* R = s_rot[s_] + 9 * jt;
        R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
        R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
        R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
#pragma unroll
        for (int e = 0; e < 9; ++e) s_pm[s_][9 * jt + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
    }
    // ---- joint regression from the shaped template (manolayer.py:183).  J = Jreg (v_template + shapedirs beta) is
    //      linear in beta: j_template = Jreg v_template [16,3] and j_shapedirs = Jreg shapedirs [16,3,10] are folded once
    //      at pack time (fp64), replacing 48 dot products of length 778 per sample by 48 of length 10.
    if (tid >= 128 && tid < 128 + NJ * 3) {
        const int o = tid - 128;
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            float acc = a.t.j_template[o];
#pragma unroll
            for (int k = 0; k < 10; ++k) acc = fmaf(a.t.j_shapedirs[o * 10 + k], s_beta[s_][k], acc);
            s_J[s_][o] = acc;
        }
    }
    __syncthreads(); stamp();

    // ---- pose blend shapes (manolayer.py:186-187): one float4 column of the k-major table per thread, 27 loads in flight
    if (f_lo / 4 + tid < (f_hi + 3) / 4) {
        const int c4 = f_lo / 4 + tid;                    // float4 column (f_lo is a multiple of 12)
        const float4* pd = reinterpret_cast<const float4*>(a.t.posedirs_t) + c4;
        float4 acc[SPW];
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) acc[s_] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 27
        for (int k = 0; k < 135; ++k) {
            const float4 p = pd[k * (NV3P / 4)];
#pragma unroll
            for (int s_ = 0; s_ < SPW; ++s_) {
                const float w = s_pm[s_][k];
                acc[s_].x = fmaf(p.x, w, acc[s_].x); acc[s_].y = fmaf(p.y, w, acc[s_].y);
                acc[s_].z = fmaf(p.z, w, acc[s_].z); acc[s_].w = fmaf(p.w, w, acc[s_].w);
            }
        }
        const int i = 4 * c4;
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            s_v[s_][i] += acc[s_].x; s_v[s_][i + 1] += acc[s_].y;
            if (i + 2 < NV3) { s_v[s_][i + 2] += acc[s_].z; s_v[s_][i + 3] += acc[s_].w; }
        }
    }
    // ---- kinematic chain (manolayer.py:192-229): finger f owns joints 1+3f, 2+3f, 3+3f
    if (tid < 5 * SPW) {
        const int s_ = tid / 5, fg = tid - 5 * s_;
        float A[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            A[4 * r + 0] = s_root[s_][3 * r + 0]; A[4 * r + 1] = s_root[s_][3 * r + 1]; A[4 * r + 2] = s_root[s_][3 * r + 2];
            A[4 * r + 3] = s_J[s_][r];
        }
        if (fg == 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) s_A[s_][e] = A[e];
        }
        int parent = 0;
        for (int l = 0; l < 3; ++l) {
            const int j = 1 + 3 * fg + l;
            const float* R = s_rot[s_] + 9 * (j - 1);
            const float t0 = s_J[s_][3 * j] - s_J[s_][3 * parent], t1 = s_J[s_][3 * j + 1] - s_J[s_][3 * parent + 1],
                        t2 = s_J[s_][3 * j + 2] - s_J[s_][3 * parent + 2];
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
            for (int e = 0; e < 12; ++e) { A[e] = N[e]; s_A[s_][12 * j + e] = N[e]; }
            parent = j;
        }
    }
    __syncthreads(); stamp();
    if (tid < NJ * SPW) {   // A' = A - pack(A.[J;0])  (manolayer.py:231-234)
        const int s_ = tid / NJ, jt = tid - NJ * s_;
        const float* A = s_A[s_] + 12 * jt;
        const float j0 = s_J[s_][3 * jt], j1 = s_J[s_][3 * jt + 1], j2 = s_J[s_][3 * jt + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            s_A2[s_][12 * jt + 4 * r + 0] = A[4 * r + 0];
            s_A2[s_][12 * jt + 4 * r + 1] = A[4 * r + 1];
            s_A2[s_][12 * jt + 4 * r + 2] = A[4 * r + 2];
            s_A2[s_][12 * jt + 4 * r + 3] = A[4 * r + 3] - (A[4 * r] * j0 + A[4 * r + 1] * j1 + A[4 * r + 2] * j2);
        }
    }
    __syncthreads(); stamp();

    // ---- linear blend skinning (manolayer.py:236-246): T = sum_k w[v][k] A'[k]; vert = T.[v_posed;1]
    for (int v = v_lo + tid; v < v_hi; v += nthr) {
        const float4* wp = reinterpret_cast<const float4*>(a.t.weights + 16 * v);
        float w[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 t4 = wp[q];
            w[4 * q] = t4.x; w[4 * q + 1] = t4.y; w[4 * q + 2] = t4.z; w[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int s_ = 0; s_ < SPW; ++s_) {
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {                       // A'[k] as three 16-byte LDS reads (was 12 scalar reads; same fmaf order)
                const float4* ak = reinterpret_cast<const float4*>(s_A2[s_] + 12 * k);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float4 t4 = ak[r];
                    T[4 * r] = fmaf(t4.x, w[k], T[4 * r]); T[4 * r + 1] = fmaf(t4.y, w[k], T[4 * r + 1]);
                    T[4 * r + 2] = fmaf(t4.z, w[k], T[4 * r + 2]); T[4 * r + 3] = fmaf(t4.w, w[k], T[4 * r + 3]);
                }
            }
            const float x = s_v[s_][3 * v], y = s_v[s_][3 * v + 1], z = s_v[s_][3 * v + 2];
            s_v[s_][3 * v + 0] = T[0] * x + T[1] * y + T[2] * z + T[3];
            s_v[s_][3 * v + 1] = T[4] * x + T[5] * y + T[6] * z + T[7];
            s_v[s_][3 * v + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
        }
    }
    __syncthreads(); stamp();

    // ---- joints: 16 chain joints + 5 fingertip vertices, reordered (manolayer.py:247-259)
    if (tid < 21 * SPW) {
        const int s_ = tid / 21, jt = tid - 21 * s_;
        const int src = kReorderJ[jt];
        const float* sv = s_v[s_];
        float x, y, z;
        if (src == 0 && a.t.root_palm) {
            x = (sv[3 * 95] + sv[3 * 22]) / 2; y = (sv[3 * 95 + 1] + sv[3 * 22 + 1]) / 2;
            z = (sv[3 * 95 + 2] + sv[3 * 22 + 2]) / 2;
        } else if (src < 16) { x = s_A[s_][12 * src + 3]; y = s_A[s_][12 * src + 7]; z = s_A[s_][12 * src + 11]; }
        else { const int v = kTips[a.t.side][src - 16]; x = sv[3 * v]; y = sv[3 * v + 1]; z = sv[3 * v + 2]; }
        s_jtr[s_][3 * jt] = x; s_jtr[s_][3 * jt + 1] = y; s_jtr[s_][3 * jt + 2] = z;
    }
    __syncthreads(); stamp();
    if (tid < 3 * SPW) { const int s_ = tid / 3, c = tid - 3 * s_; s_c[s_][c] = a.t.center_idx >= 0 ? s_jtr[s_][3 * a.t.center_idx + c] : 0.f; }   // manolayer.py:261-265
    __syncthreads(); stamp();

    // with vertex parts: part 0 writes the 16 chain joints, a fingertip joint is written by the part that owns its vertex
    auto owns_joint = [&](int j) {
        if (args.parts == 1) return true;
        const int src = kReorderJ[j];
        const int v = src < 16 ? 0 : kTips[a.t.side][src - 16];
        return v >= v_lo && v < v_hi;
    };
#pragma unroll
    for (int s_ = 0; s_ < SPW; ++s_) {
        if (!live(s_)) continue;
        const size_t b = (size_t)bs(s_);
        const float sc = s_cam[s_][0], tx = s_cam[s_][1], ty = s_cam[s_][2];
        float* vout = a.verts + b * NV3;
        for (int i = f_lo + tid; i < f_hi; i += nthr) vout[i] = s_v[s_][i] - s_c[s_][i % 3];
        if (tid < 63 && owns_joint(tid / 3)) a.joints[b * 63 + tid] = s_jtr[s_][tid] - s_c[s_][tid % 3];
        if (a.cam) {   // utils/utils.py:47-63: uv = s * xy + t
            if (a.joint_uv && tid >= 64 && tid < 64 + 42) {
                const int i = tid - 64, j = i >> 1, c = i & 1;
                if (owns_joint(j)) a.joint_uv[b * 42 + i] = sc * (s_jtr[s_][3 * j + c] - s_c[s_][c]) + (c ? ty : tx);
            }
            if (a.mesh_uv) {
                float* mo = a.mesh_uv + b * NV * 2;
                for (int i = 2 * v_lo + tid; i < 2 * v_hi; i += nthr) {
                    const int v = i >> 1, c = i & 1;
                    mo[i] = sc * (s_v[s_][3 * v + c] - s_c[s_][c]) + (c ? ty : tx);
                }
            }
        }
    }
    stamp();
}

}  // namespace

// Four vertex parts per (sample, hand) unless the centre joint needs vertices of another part (a fingertip centre or the
// root_palm wrist) or the batch alone fills the chip.
static void launch_mano(const ManoArgs& a0, int B, int hands, hipStream_t s) {
    ManoArgs a = a0;
    a.stamps = dir::stamps_begin("mano");
    bool split = B * hands < 1024;
    for (int h = 0; h < hands; ++h) {
        const dir_mano_tables& t = a.h[h].t;
        const int c = t.center_idx;
        if (t.root_palm || (c >= 0 && (c == 4 || c == 8 || c == 12 || c == 16 || c == 20))) split = false;
    }
    a.B = B; a.hands = hands; a.parts = split ? 4 : 1;
    const int rep = 8 / (hands * a.parts);
    // samples per workgroup (table values shared): DIR_MANO_SPW = 1 | 2 | 4 (tuning aid).  Default 1: with the tables hot in L2 two samples per
    // workgroup are 22 % faster (21.9 -> 17.0 us, back-to-back launches), but inside a forward the tables come from the Infinity Cache and
    // the launch is bound by loads in flight per CU -- 24 us either way, 32 us with two samples and the shallower unroll (measured in-engine)
    static const int spw_env = getenv("DIR_MANO_SPW") ? atoi(getenv("DIR_MANO_SPW")) : 0;
    const int spw = !split ? 1 : spw_env == 2 || spw_env == 4 ? spw_env : 1;
    const int groups = (B + spw - 1) / spw;
    const dim3 grid(8 * ((groups + rep - 1) / rep));
    if (!split) DIR_LAUNCH((mano_forward_kernel<NTHR, 1>), grid, dim3(NTHR), 0, s, a);
    else if (spw == 4) DIR_LAUNCH((mano_forward_kernel<PTHR, 4>), grid, dim3(PTHR), 0, s, a);
    else if (spw == 2) DIR_LAUNCH((mano_forward_kernel<PTHR, 2>), grid, dim3(PTHR), 0, s, a);
    else DIR_LAUNCH((mano_forward_kernel<PTHR, 1>), grid, dim3(PTHR), 0, s, a);
    dir::stamps_end("mano", a.stamps, s);
}

static int check_hand(const dir_mano_tables* t, const float* pose, int pose_stride, const float* betas,
                      int betas_stride, const float* cam, int cam_stride, float* verts, float* joints) {
    DIR_REQUIRE(t && pose && betas && verts && joints, "dir_mano_forward: null pointer");
    DIR_REQUIRE(t->shapedirs_t && t->posedirs_t && t->v_template && t->j_template && t->j_shapedirs && t->weights &&
                    t->hands_mean && t->comps, "dir_mano_forward: null table");
    DIR_REQUIRE(pose_stride >= 51 && betas_stride >= 10, "dir_mano_forward: bad stride");
    DIR_REQUIRE(t->side == 0 || t->side == 1, "dir_mano_forward: side must be 0 (right) or 1 (left)");
    DIR_REQUIRE(t->center_idx >= -1 && t->center_idx < 21, "dir_mano_forward: center_idx out of range");
    DIR_REQUIRE(cam == nullptr || cam_stride >= 3, "dir_mano_forward: bad cam stride");
    DIR_REQUIRE(((uintptr_t)t->posedirs_t & 15) == 0 && ((uintptr_t)t->weights & 15) == 0,
                "dir_mano_forward: posedirs_t / weights must be 16-byte aligned");
    return DIR_OK;
}

extern "C" int dir_mano_forward(const dir_mano_tables* t, const float* pose, int pose_stride, const float* betas,
                                int betas_stride, const float* cam, int cam_stride, float* verts, float* joints,
                                float* joint_uv, float* mesh_uv, int32_t* flags_out, int B, void* stream) {
    if (B == 0) return DIR_OK;   /* empty batch: nothing to do, pointers may be null */
    DIR_REQUIRE(B > 0, "dir_mano_forward: bad B");
    int rc = check_hand(t, pose, pose_stride, betas, betas_stride, cam, cam_stride, verts, joints);
    if (rc) return rc;
    ManoArgs a;
    a.h[0] = ManoHand{*t, pose, pose_stride, betas, betas_stride, cam, cam_stride, verts, joints, joint_uv, mesh_uv, flags_out};
    a.h[1] = a.h[0];
    launch_mano(a, B, 1, (hipStream_t)stream);
    return dir::check_launch("dir_mano_forward");
}

extern "C" int dir_mano_forward_pair(const dir_mano_tables* tables_lr, const float* const* pose_lr, int pose_stride,
                                     const float* const* betas_lr, int betas_stride, const float* const* cam_lr,
                                     int cam_stride, float* const* verts_lr, float* const* joints_lr,
                                     float* const* joint_uv_lr, float* const* mesh_uv_lr, int32_t* const* flags_lr, int B, void* stream) {
    if (B == 0) return DIR_OK;
    DIR_REQUIRE(B > 0 && tables_lr && pose_lr && betas_lr && verts_lr && joints_lr, "dir_mano_forward_pair: bad arguments");
    ManoArgs a;
    for (int h = 0; h < 2; ++h) {
        const float* cam = cam_lr ? cam_lr[h] : nullptr;
        int rc = check_hand(&tables_lr[h], pose_lr[h], pose_stride, betas_lr[h], betas_stride, cam, cam_stride, verts_lr[h],
                            joints_lr[h]);
        if (rc) return rc;
        a.h[h] = ManoHand{tables_lr[h], pose_lr[h], pose_stride, betas_lr[h], betas_stride, cam, cam_stride, verts_lr[h],
                          joints_lr[h], joint_uv_lr ? joint_uv_lr[h] : nullptr, mesh_uv_lr ? mesh_uv_lr[h] : nullptr, flags_lr ? flags_lr[h] : nullptr};
    }
    launch_mano(a, B, 2, (hipStream_t)stream);
    return dir::check_launch("dir_mano_forward_pair");
}
