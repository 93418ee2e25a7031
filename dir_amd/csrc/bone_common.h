// Bone geometry shared by bone_proj (spatial.hip) and the factorised bone-fusion kernels (bonefuse.hip).
#pragma once
#include "dir_common.h"

namespace dir {
namespace bone {

// models/dir.py:86-87 (Joint2BoneFeature.parent / .child)
__constant__ const int kParent[20] = {0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19};
__constant__ const int kChild[20] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20};

// correctly rounded hypot (products of two floats are exact in double): ATen's CPU kernel (Sleef hypotf_u05) and
// glibc are correctly rounded, the device libm's hypotf is not, and pixels lying exactly on the capsule boundary
// flip on a 1-ulp difference.
__device__ __forceinline__ float hypot_cr(float x, float y) {
    return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}

// point-segment distance exactly as lineseg_dists (models/dir.py:132-144): same fp32 op sequence, no FMA
// contraction, so the `distance < threshold` mask is bit-identical to the reference's.
__device__ __forceinline__ void bone_weights(float px, float py, float ax, float ay, float bx, float by, float thr,
                                             float& wa, float& wb, bool& inside) {
#pragma clang fp contract(off)
    const float dbx = bx - ax, dby = by - ay;
    const float len = hypot_cr(dbx, dby);
    const float dx = dbx / len, dy = dby / len;
    const float s = (ax - px) * dx + (ay - py) * dy;
    const float t = (px - bx) * dx + (py - by) * dy;
    const float h = fmaxf(fmaxf(s, t), 0.f);
    const float dpx = px - ax, dpy = py - ay;
    const float c = dpx * dy - dpy * dx;
    const float dist = hypot_cr(h, c);
    inside = dist < thr;                                  // NaN (zero-length bone) -> false, like torch.lt
    // F.pairwise_distance(p, a): || p - a + 1e-6 ||_2  (models/dir.py:164-167)
    const float eax = px - ax + 1e-6f, eay = py - ay + 1e-6f;
    const float ebx = px - bx + 1e-6f, eby = py - by + 1e-6f;
    const float da = sqrtf(eax * eax + eay * eay), db = sqrtf(ebx * ebx + eby * eby);
    wa = 1.f - da / (da + db);
    wb = 1.f - db / (da + db);
}

// The same arithmetic split into its per-bone part (unit direction: one correctly rounded hypot + two divisions, pixel independent)
// and its per-pixel part with an exact early-out: hypot_cr(h, c) >= max(|h|, |c|) (correct rounding is monotonic), so a pixel with
// max(h, |c|) >= thr is outside the capsule without evaluating the double-precision square root -- ~90 % of the patch.  Every
// value that is produced is bit-identical to bone_weights'.
__device__ __forceinline__ void bone_dir(float ax, float ay, float bx, float by, float& dx, float& dy) {
#pragma clang fp contract(off)
    const float dbx = bx - ax, dby = by - ay;
    const float len = hypot_cr(dbx, dby);
    dx = dbx / len;
    dy = dby / len;
}
__device__ __forceinline__ bool bone_weights_fast(float px, float py, float ax, float ay, float bx, float by, float dx, float dy, float thr,
                                                  float& wa, float& wb) {
#pragma clang fp contract(off)
    const float s = (ax - px) * dx + (ay - py) * dy;
    const float t = (px - bx) * dx + (py - by) * dy;
    const float h = fmaxf(fmaxf(s, t), 0.f);
    const float dpx = px - ax, dpy = py - ay;
    const float c = dpx * dy - dpy * dx;
    if (fmaxf(h, fabsf(c)) >= thr) return false;          // provably outside (NaN falls through to the exact test)
    if (!(hypot_cr(h, c) < thr)) return false;
    const float eax = px - ax + 1e-6f, eay = py - ay + 1e-6f;
    const float ebx = px - bx + 1e-6f, eby = py - by + 1e-6f;
    const float da = sqrtf(eax * eax + eay * eay), db = sqrtf(ebx * ebx + eby * eby);
    wa = 1.f - da / (da + db);
    wb = 1.f - db / (da + db);
    return true;
}

}  // namespace bone
}  // namespace dir
