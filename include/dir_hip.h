/*
 * libdir_hip.so -- C ABI of the MI355X (gfx950) implementation of DIR's iterative-refinement hot path.
 *
 * The reference (PengfeiRen96/DIR) has no FFI: its boundary is Python nn.Module classes.
 * This header is therefore build-defined.  Each entry point names the reference function it replaces
 * (file:line relative to the reference tree); the Python modules under dir_amd/ that mirror the
 * reference classes are thin ctypes callers of exactly these symbols (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises, nothing is allocated (graph-capture safe);
 *   - tensors are dense row-major float32 unless stated; feature maps are NHWC ("channels last");
 *   - return value: 0 = OK, <0 = DIR_E_* ; dir_last_error() gives a message for the calling thread;
 *   - no entry point throws or aborts.
 */
#ifndef DIR_HIP_H
#define DIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIR_OK 0
#define DIR_E_INVALID (-1)  /* bad argument (null pointer, unsupported shape) */
#define DIR_E_LAUNCH (-2)   /* hipLaunchKernel / HIP runtime error          */
#define DIR_E_NODEVICE (-3) /* no gfx950 device visible                     */

#define DIR_ABI_VERSION 1

int dir_abi_version(void);
const char* dir_last_error(void);
/* number of visible HIP devices, and the gcnArchName of device 0 copied into buf (host). */
int dir_device_info(char* arch_host, int arch_len, int* num_cu_host);

/* ------------------------------------------------------------------------------------------------
 * a8 + a9: MANO forward + weak-perspective projection
 * replaces manopth/manopth/manolayer.py:110-270 (ManoLayer.forward in the configuration of
 * models/dir.py:221-224: 6D root rotation "robust" variant rot6d.py:26-51, 45 PCA pose components,
 * Rodrigues via quaternion rodrigues_layer.py:43-54, LBS, fingertips, joint reorder, root centring)
 * and utils/utils.py:47-63 (projection_batch_xy).
 *
 * Tables (float32, device).  Layouts are k-major so the blend-shape contractions read coalesced:
 *   shapedirs_t [10][2334]   = th_shapedirs[778,3,10]  transposed   (2334 = 778*3, index v*3+c)
 *   posedirs_t  [135][2334]  = th_posedirs[778,3,135]  transposed
 *   v_template  [2334]
 *   j_regressor [16][778]
 *   weights     [778][16]
 *   hands_mean  [45]
 *   comps       [45][45]      th_selected_comps (row k = PCA component k)
 */
typedef struct dir_mano_tables {
    const float* shapedirs_t;
    const float* posedirs_t;
    const float* v_template;
    const float* j_regressor;
    const float* weights;
    const float* hands_mean;
    const float* comps;
    int32_t side;       /* 0 = right (tip vertex 444), 1 = left (tip vertex 445): manolayer.py:249-252 */
    int32_t center_idx; /* joint (after reorder) subtracted from verts and joints; -1 = none          */
    int32_t root_palm;  /* !=0: joint 0 = (v95 + v22)/2 instead of the wrist (manolayer.py:253-255)   */
} dir_mano_tables;

/* pose  : B rows of 51 floats (6D root | 45 PCA coeffs), row stride pose_stride floats
 * betas : B rows of 10 floats, row stride betas_stride
 * cam   : optional (NULL ok) B rows of 3 floats (scale, tx, ty), row stride cam_stride
 * verts [B,778,3], joints [B,21,3] metres.  joint_uv [B,21,2] / mesh_uv [B,778,2] optional,
 * written only when cam != NULL.  flags_out (optional, int32[B]): bit0 set when the 6D root matrix
 * has det < 0 (the reference raises AssertionError there, rot6d.py:50; the wrapper re-raises it). */
int dir_mano_forward(const dir_mano_tables* tables_host, const float* pose, int pose_stride,
                     const float* betas, int betas_stride, const float* cam, int cam_stride,
                     float* verts, float* joints, float* joint_uv, float* mesh_uv, int32_t* flags_out,
                     int B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a1 / a2 / a3 / a11: 2-D convolution as an implicit GEMM on the matrix cores
 * replaces every nn.Conv2d (+ the BatchNorm / bias / ReLU / residual add around it) on the path:
 * models/backbone/resnet.py:120-140,243-255 (Bottleneck, stem excluded), models/backbone/hourglass.py:10-30,
 * 55-70 (Conv, pre-activation Residual), models/dir.py:57-62 (fusion), :227-241 (attention), :404-420 (heads).
 *
 * x  : NHWC activations, B*H*W pixels of in_cstride channels; channels [in_coff, in_coff+Cin) are read
 * w  : [Cout][kh][kw][Cin] in in_dtype  (= torch weight.permute(0,2,3,1))
 * y  : NHWC, out_cstride channels per pixel, channels [out_coff, out_coff+Cout) are written
 *      y = relu?( acc * scale[n] + shift[n] + residual[m][n] )        (scale/shift/residual optional)
 * pre_scale/pre_shift [Cin] (optional): the input is first mapped x -> relu?(x*pre_scale + pre_shift)
 *      (eval-mode BatchNorm+ReLU in front of the conv; zero padding is applied AFTER this map)
 * dtype: in_dtype f32 computes with v_mfma_f32_32x32x2_f32 (exact fp32), bf16 with v_mfma_f32_32x32x16_bf16
 *      (fp32 accumulate).  Cin must be a multiple of 32 (f32) / 64 (bf16); *_cstride = 0 means dense.
 */
#define DIR_DT_F32 0
#define DIR_DT_BF16 1
#define DIR_CONV_RELU 1
#define DIR_CONV_PRE_RELU 2

typedef struct dir_conv_desc {
    int32_t B, H, W;
    int32_t Cin, in_cstride, in_coff;
    int32_t Cout, out_cstride, out_coff;
    int32_t res_cstride, res_coff;
    int32_t kh, kw, stride, pad;
    int32_t in_dtype, out_dtype; /* DIR_DT_*; residual is read in out_dtype */
    int32_t flags;               /* DIR_CONV_* */
} dir_conv_desc;

int dir_conv2d_forward(const dir_conv_desc* desc_host, const void* x, const void* w, const float* scale,
                       const float* shift, const float* pre_scale, const float* pre_shift,
                       const void* residual, void* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIR_HIP_H */
