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

#define DIR_ABI_VERSION 37

int dir_abi_version(void);
const char* dir_last_error(void);
/* number of visible HIP devices, and the gcnArchName of device 0 copied into buf (host). */
int dir_device_info(char* arch_host, int arch_len, int* num_cu_host);
/* Measurement aid (bench.py's live roofline): the library notes the name of every kernel it launches for the calling thread.
 * dir_launch_log_reset() clears the note pad; dir_launch_log_get() copies the names launched since then into buf_host as a
 * comma-separated list (kernel names as rocprofv3 reports them, template arguments dropped) and returns how many there were. */
void dir_launch_log_reset(void);
int dir_launch_log_get(char* buf_host, int len);
/* as if `times` launches of kernel `name` had been noted: the counter dir_launch_log_get() returns SATURATES at INT_MAX (a serving process
 * never resets it), only the first 32 names since the last reset are kept.  Used by the CPU tests; launches nothing. */
void dir_launch_log_note(const char* name, long long times);
/* Measurement aid (bench.py `roofline.measured_ceilings`; nothing in the reference, nothing in the product path): ONE launch of a ceiling probe
 * on `stream`.  mode 0: bf16 MFMA loop on pseudo-random operands held in registers (16 x v_mfma_f32_32x32x16_bf16 per wave and iteration, 8 waves
 * per CU, every CU), `iters` iterations; mode 1: 16-byte loads streaming `bytes` of `buf` (larger than the Infinity Cache to price HBM).  `buf`: any
 * device buffer >= 64 bytes (mode 0 only needs a sink); mode 2 (round 5): the first half of `buf` copied to the second half (a streaming kernel reads AND
 * writes: the read-only loop under-reports the HBM ceiling); mode 3: mode 0 on v_mfma_f32_32x32x16_f16 with pseudo-random f16 operands.  Returns the
 * FLOPs (modes 0, 3) / bytes read (mode 1) / bytes read + written (mode 2) of the launch, negative = error code.  Round 6: mode 2's `iters` selects the
 * copy shape (0 = the shipped one); mode 4: every workgroup (2 per CU) streams the SAME first `bytes` of buf `iters` times -- the aggregate L2 -> CU
 * rate at which a small-map convolution's weights reach all CUs; mode 5: ds_read_b128 from a 64 KB LDS tile, 8 waves per CU, `iters` passes of 16
 * reads per lane.  Both return bytes delivered. */
long long dir_probe_launch(int mode, void* buf, long long bytes, int iters, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a8 + a9: MANO forward + weak-perspective projection
 * replaces manopth/manopth/manolayer.py:110-270 (ManoLayer.forward in the configuration of
 * models/dir.py:221-224: 6D root rotation "robust" variant rot6d.py:26-51, 45 PCA pose components,
 * Rodrigues via quaternion rodrigues_layer.py:43-54, LBS, fingertips, joint reorder, root centring)
 * and utils/utils.py:47-63 (projection_batch_xy).
 *
 * Tables (float32, device).  Layouts are k-major so the blend-shape contractions read coalesced:
 *   shapedirs_t [10][2336]   = th_shapedirs[778,3,10]  transposed, rows zero-padded 2334 -> 2336 (index v*3+c)
 *   posedirs_t  [135][2336]  = th_posedirs[778,3,135]  transposed, rows zero-padded (16-byte aligned rows)
 *   v_template  [2334]
 *   j_template  [16][3]      = th_J_regressor @ th_v_template            (folded once, in fp64)
 *   j_shapedirs [16][3][10]  = th_J_regressor @ th_shapedirs              (J is linear in beta)
 *   weights     [778][16]
 *   hands_mean  [45]
 *   comps       [45][45]      th_selected_comps (row k = PCA component k)
 */
typedef struct dir_mano_tables {
    const float* shapedirs_t;
    const float* posedirs_t;
    const float* v_template;
    const float* j_template;
    const float* j_shapedirs;
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

/* Both hands of one stage in ONE launch (grid = B x 2).  tables_lr[2] = {left, right}; the *_lr arguments are HOST
 * arrays of two device pointers.  cam_lr / joint_uv_lr / mesh_uv_lr / flags_lr may be NULL (mesh_uv_lr: pd_mesh_uv_* [B,778,2], which only
 * the training loss reads, models/dir.py:278-280,574-575); flags_lr[h] = int32[B] as flags_out above
 * (the reflection check of rot6d.py:50 that DIR.forward turns into the reference's AssertionError). */
int dir_mano_forward_pair(const dir_mano_tables* tables_lr_host, const float* const* pose_lr_host, int pose_stride,
                          const float* const* betas_lr_host, int betas_stride, const float* const* cam_lr_host,
                          int cam_stride, float* const* verts_lr_host, float* const* joints_lr_host,
                          float* const* joint_uv_lr_host, float* const* mesh_uv_lr_host, int32_t* const* flags_lr_host, int B,
                          void* stream);

/* SURVEY 8f rank 2, backward pass, first link behind the loss gradients: the gradient of
 *   <g_verts, verts> + <g_joints, joints> + <g_joint_uv, joint_uv> + <g_mesh_uv, mesh_uv>
 * w.r.t. pose [B,51], betas [B,10] and cam [B,3] of dir_mano_forward -- what torch autograd computes through
 * manopth/manopth/manolayer.py:110-270 and utils/utils.py:47-63 in the reference's training step (train.py:66-70).
 * The *_lr arguments are HOST arrays of `hands` (1 or 2) device pointers; any g_* input may be NULL (or hold NULL) = no
 * contribution.  Inputs as for the forward (the forward is recomputed, nothing has to be saved).  Deterministic (no atomics). */
int dir_mano_backward_pair(const dir_mano_tables* tables_lr_host, const float* const* pose_lr_host, int pose_stride,
                           const float* const* betas_lr_host, int betas_stride, const float* const* cam_lr_host, int cam_stride,
                           const float* const* g_verts_lr_host, const float* const* g_joints_lr_host,
                           const float* const* g_joint_uv_lr_host, const float* const* g_mesh_uv_lr_host,
                           float* const* g_pose_lr_host, int g_pose_stride, float* const* g_betas_lr_host, int g_betas_stride,
                           float* const* g_cam_lr_host, int g_cam_stride, int hands, int B, void* stream);

/* Backward of RegressorOffset's three Linears (models/dir.py:339-351).  w_left / w_right [64][1408], w_offset [3][2691]: the
 * nn.Linear weights in their own layout; tok [B,42,64] = the STE head output (tokens 0..20 left); prev_* the previous stage's
 * (detached) mano_para [B,64] / offset [B,3]; g_para_* [B,64], g_offset [B,3]: gradients w.r.t. the Linears' outputs.
 * Outputs (gw_* / gb_* in the parameters' layout, written not accumulated; all six or none) and g_tok [B,42,64] (optional). */
int dir_regress_backward(const float* w_left, const float* w_right, const float* w_offset, const float* tok,
                         const float* prev_para_left, const float* prev_para_right, const float* prev_offset,
                         const float* g_para_left, const float* g_para_right, const float* g_offset,
                         float* gw_left, float* gb_left, float* gw_right, float* gb_right, float* gw_offset, float* gb_offset,
                         float* g_tok, int B, void* stream);

/* ------------------------------------------------------------------------------------------------
 * building blocks of the backward pass over the joint-token path (train.py:66-70 runs torch autograd through
 * transformer/mixSTE.py, SemGCN/p_graph_conv.py and the Conv1d / Linear / LayerNorm / BatchNorm1d modules of models/dir.py:19-130).
 * Exact fp32, deterministic (fixed summation orders, no atomics); composed on the host by dir_amd/train/.
 */
/* C[b] (+)= op(A[b]) op(B[b]) (+ bias[n]):  op(X) = X or X^T; row-major, leading dimensions in elements, `batch` problems
 * `stride_*` elements apart.  trans_a = 0: A is [M][K]; 1: A is [K][M].  trans_b = 0: B is [K][N]; 1: B is [N][K] (nn.Linear weight).
 * accumulate != 0: C += product + bias (a residual branch lands on its trunk in place).  v_mfma_f32_16x16x4_f32: exact fp32
 * products, k ascending.  64 x 64 output tiles, or 32 x 32 when those would be <= 128 workgroups (the joint-token path's products: a wave
 * then owns one 16 x 16 block and the per-step LDS -> MFMA chain is a quarter as long); the k order per element, and so the bits, are the same
 * (DIR_GEMM_SMALL=0 keeps the 64 x 64 tiles). */
typedef struct dir_gemm_desc {
    int32_t M, N, K, lda, ldb, ldc, trans_a, trans_b, accumulate, batch;
    int64_t stride_a, stride_b, stride_c;
} dir_gemm_desc;
int dir_gemm_f32(const dir_gemm_desc* desc_host, const float* A, const float* B, const float* bias, float* C, void* stream);
/* dir_gemm_f32 over an ny x nx grid of GROUPS: group (gy, gx) displaces A, B and C by gy * *_y + gx * *_x elements -- a convolution tap
 * as a pointer shift over a zero-bordered pixel grid (csrc/bonefuse_bwd.hip).  reduce = 0: every (batch entry, group) its own product in
 * ONE launch (batch * ny * nx <= 65535); reduce = 1: C = sum over the groups (in group order, inside one workgroup) of the displaced
 * products; c_y / c_x are ignored.  batch == 1 and one group: the same bits as dir_gemm_f32. */
typedef struct dir_gemm_groups {
    int32_t ny, nx, reduce, reserved;
    int64_t a_y, a_x, b_y, b_x, c_y, c_x;
} dir_gemm_groups;
int dir_gemm_f32_grouped(const dir_gemm_desc* desc_host, const dir_gemm_groups* groups_host, const float* A, const float* B, const float* bias,
                         float* C, void* stream);
/* the same product (batch == 1) with the reduction cut into chunks of 64, one workgroup per (tile, chunk), partial tiles summed in chunk order by a
 * second launch (deterministic): for tall reductions with a small output -- the Linear weight gradients of the token path (K = rows = B * 21 .. 42).
 * Not bit-identical to dir_gemm_f32 (another summation order); workspace of dir_gemm_f32_splitk_workspace_bytes(d) bytes. */
long long dir_gemm_f32_splitk_workspace_bytes(const dir_gemm_desc* desc_host);
int dir_gemm_f32_splitk(const dir_gemm_desc* desc_host, const float* A, const float* B, const float* bias, float* C, float* workspace,
                        long long workspace_bytes, void* stream);
/* out[n] (+)= sum_r x[r][n]: bias gradients.  R > 512: 256-row chunk partials added in chunk order; workspace of
 * dir_colsum_workspace_bytes(R, N) bytes (0 for R <= 512). */
long long dir_colsum_workspace_bytes(int R, int N);
int dir_colsum_f32(const float* x, float* out, int R, int N, int ld, int accumulate, float* workspace, long long workspace_bytes, void* stream);
/* nn.LayerNorm over the last dimension of x [R][C] (C <= 256); mean / rstd [R] are saved for the backward */
int dir_layernorm_forward(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int R, int C, float eps, void* stream);
int dir_layernorm_backward(const float* gy, const float* x, const float* w, const float* mean, const float* rstd, float* gx, float* gw,
                           float* gb, int R, int C, int accumulate_x, int accumulate_wb, void* stream);
/* nn.GELU() (exact erf form; transformer/mixSTE.py:12,27) */
int dir_gelu_forward(const float* x, float* y, long long n, void* stream);
int dir_gelu_backward(const float* gy, const float* x, float* gx, long long n, void* stream);
/* Attention.forward without the two Linears (transformer/mixSTE.py:76-97): qkv [B][T][3][H][32] -> out [B][T][H*32] =
 * softmax(q k^T * scale) v per (sample, head); probs [B][H][T][T] is saved for the backward (may be NULL in inference).  T <= 64. */
int dir_attention_forward(const float* qkv, float* probs, float* out, int B, int T, int H, float scale, void* stream);
int dir_attention_backward(const float* qkv, const float* probs, const float* gout, float* gqkv, int B, int T, int H, float scale, void* stream);
/* BatchNorm in TRAINING mode over x [R][C] with row stride ld (channels last: R = samples x positions): batch mean and biased
 * variance, y = (x - mean) * rstd * w + b, running statistics updated with `momentum` and the unbiased variance (torch semantics);
 * save_mean / save_rstd [C] feed the backward, which returns g x (optional), g w, g b (optional).  relu != 0 fuses the nn.ReLU that follows
 * the BatchNorm on the path (models/backbone/resnet.py:122-131, hourglass.py:58-66): the forward writes max(y, 0) and the backward masks gy
 * where the re-computed y is not positive (same expression, same mask as the forward's; it therefore also needs b).  residual (forward,
 * optional, [R][C] with the same row stride): added before the ReLU -- the tail of a ResNet bottleneck, relu(bn3(.) + identity)
 * (models/backbone/resnet.py:136-140); the backward of THAT form masks with the saved output instead (dir_relu_backward) and passes relu = 0.  R <= 512: one
 * thread per channel walks the rows in order, no workspace.  Larger R (BatchNorm2d over feature maps): the column reductions are cut
 * into 256-row chunks whose partials are combined in chunk order (deterministic; the variance as sum_k [M2_k + n_k (mean_k - mean)^2] / R,
 * one pass over HBM); workspace of dir_bn_train_workspace_bytes(R, C).
 * R > 512, C % 4 == 0 (round 5), OPTIONAL: one launch each way instead of three -- a persistent grid (every workgroup resident) walks the same
 * chunks; the workgroup whose chunk reaches a 64-channel group's counter last combines that group's partials (in chunk order: the same bits as
 * the three launches), the others wait on the group's flag and then normalise the chunks they read.  The counters live in library-owned device
 * words, one block per stream (allocated at the first call outside a stream capture; no block -> the three launches run), and are left zero by
 * every launch.  A workgroup that waits longer than 4 s gives up and raises the error word dir_bn_one_launch_status() reports (0 | 1; it
 * synchronises the device and clears the words; the step never calls it).  OFF by default: measured 3 % slower per training step than the three
 * launches at 32 images (profiles/r05_bn_one_launch_ab.txt); DIR_BN_ONE_LAUNCH=1 in the environment or dir_bn_one_launch_enable(1) (returns the
 * previous setting) turns it on. */
long long dir_bn_train_workspace_bytes(int R, int C);
int dir_bn_one_launch_status(void);
int dir_bn_one_launch_enable(int on);
int dir_bn_train_forward(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd, float* running_mean,
                         float* running_var, int R, int C, int ld, float eps, float momentum, int relu, const float* residual,
                         float* workspace, long long workspace_bytes, void* stream);
int dir_bn_train_backward(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* gx,
                          float* gw, float* gb, int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream);
/* dir_bn_train_backward whose first pass (the two column sums) is replaced by chunk partials a data-gradient convolution's epilogue formed
 * (dir_conv2d_forward_ex): p1 / p2 [chunks][C]; workspace: 2 C floats.  Same combine (chunk order) and apply kernels. */
int dir_bn_train_backward_from_partials(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                                        const float* p1, const float* p2, int chunks, float* gx, float* gw, float* gb, int R, int C, int ld, int relu,
                                        float* workspace, long long workspace_bytes, void* stream);
/* round 5 -- BatchNorm2d (training mode) whose output feeds exactly ONE convolution (Bottleneck bn1 / bn2, models/backbone/resnet.py:125-131;
 * every BatchNorm of the pre-activation Residual, models/backbone/hourglass.py:60-67): only the statistics are formed here (the same kernels and bits
 * as dir_bn_train_forward: save_mean, save_rstd, running statistics) plus pre_scale = w rstd and pre_shift = b - mean w rstd [C]; the normalised
 * map is never written -- the consuming convolution applies act(x pre_scale + pre_shift) where it reads x (dir_conv2d_forward's pre_scale /
 * pre_shift + DIR_CONV_PRE_RELU, dir_split_f16_forward's, and dir_conv2d_wgrad_f16x3_pre for its weight gradient).  dir_bn_train_backward is
 * unchanged (it re-computes the ReLU mask from x).  R > 512 rows, C % 4 == 0 (the Python step takes it for C % 32 == 0: whole reduction slabs of the consumer); workspace: dir_bn_train_workspace_bytes(R, C). */
int dir_bn_train_stats(const float* x, const float* w, const float* b, float* save_mean, float* save_rstd, float* pre_scale, float* pre_shift,
                       float* running_mean, float* running_var, int R, int C, int ld, float eps, float momentum, float* workspace,
                       long long workspace_bytes, void* stream);
/* ... from chunk partials the PRODUCING convolution's epilogue formed (dir_conv2d_forward_stats: the map is not read again for its statistics):
 * p1 / p2 [ceil(R / chunk_rows)][C] = per chunk the column sums and the sums of squared deviations from the chunk's own mean; combined in chunk
 * order like dir_bn_train_forward's (deterministic).  pre_scale / pre_shift may both be NULL (a BatchNorm that is applied by dir_bn_train_apply).
 * cap_rows: how many rows of C floats p1 and p2 each have room for; with more than 256 chunks and cap_rows >= chunks + ceil(chunks / 32) the chunks
 * are first pooled in groups of 32 behind the chunk rows (one more launch, both short) -- the buffers are scratch after the call. */
int dir_bn_train_stats_from_partials(float* p1, float* p2, int chunk_rows, int cap_rows, const float* w, const float* b, float* save_mean, float* save_rstd,
                                     float* pre_scale, float* pre_shift, float* running_mean, float* running_var, int R, int C, float eps, float momentum,
                                     void* stream);
/* the normalisation alone from statistics already formed: y = act(BatchNorm(x) + residual), the third launch of dir_bn_train_forward (a Bottleneck's
 * bn3 + identity + ReLU, models/backbone/resnet.py:133-140, whose statistics came out of conv3's epilogue).  C % 4 == 0, 16-byte aligned. */
int dir_bn_train_apply(const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* y, int R, int C, int ld,
                       int relu, const float* residual, void* stream);
/* dir_bn_train_stats_from_partials (without the affine) + dir_bn_train_apply in one call: dir_bn_train_forward whose statistics pass is replaced by the
 * producing convolution's chunk partials. */
int dir_bn_train_forward_from_partials(const float* x, float* p1, float* p2, int chunk_rows, int cap_rows, const float* w, const float* b, float* y,
                                       float* save_mean, float* save_rstd, float* running_mean, float* running_var, int R, int C, int ld, float eps,
                                       float momentum, int relu, const float* residual, void* stream);
/* BatchNorm with FROZEN statistics inside a training pass (a BatchNorm module put in .eval() under model.train(): torch then normalises with the
 * running statistics and leaves them alone -- torch/nn/modules/batchnorm.py; the reference never freezes them, train.py:64; this form exists
 * because the reference's whole-step gradient is only reproducible (to 4e-5) with it: tests/golden G20e): y = (x - running_mean) / sqrt(running_var
 * + eps) * w + b (+ residual, ReLU as in the training-mode entry points); backward: g w, g b as there, g x = gy w rstd.  save_mean / save_rstd [C]
 * are written by the forward for the backward.  workspace (backward): dir_bn_frozen_workspace_bytes(R, C). */
long long dir_bn_frozen_workspace_bytes(int R, int C);
int dir_bn_frozen_forward(const float* x, const float* w, const float* b, float* y, float* save_mean, float* save_rstd, const float* running_mean,
                          const float* running_var, int R, int C, int ld, float eps, int relu, const float* residual, void* stream);
int dir_bn_frozen_backward(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* gx,
                           float* gw, float* gb, int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream);
/* SyncBN building blocks (round 5; SURVEY.md 8e: the reference trains its batch of 64 on one GPU, config.py:13-15 -- data parallelism over 8 GPUs x 32
 * images changes the BatchNorm batch unless the statistics are pooled; torch.nn.SyncBatchNorm semantics).  The library computes, the caller moves
 * the 2 C (+ 4) floats between the ranks (dir_amd/train/ops.py: all-gather of the parts forward, all-reduce of the sums backward):
 *   dir_bn_sync_local_stats     part [2 C] = this rank's (mean | M2 = sum (x - mean)^2) over its R rows        (the chunked one-pass statistics of
 *                               dir_bn_train_forward);  the caller appends its row count: parts [W][2 C + 4] = (mean | M2 | rows, 0, 0, 0) per rank
 *   dir_bn_sync_combine         pooled mean / BIASED variance [C] from the gathered parts (Chan's exact formula, rank order: identical on every
 *                               rank); running statistics updated with the unbiased variance over the pooled count.  The forward is then
 *                               dir_bn_frozen_forward(x, w, b, y, save_mean, save_rstd, mean, var, ...) -- normalise with GIVEN statistics.
 *   dir_bn_sync_backward_sums   sums [2 C] = this rank's (sum g | sum g xhat), g = gy under the ReLU mask: g b and g w of THIS rank (the data-parallel
 *                               gradient exchange averages them like every other parameter gradient)
 *   dir_bn_sync_backward_apply  g x = w rstd (g - S1 / n - xhat S2 / n) with the sums pooled over all ranks (all-reduce SUM) and n = rows_pooled
 * C and ld multiples of 4, 16-byte aligned pointers; workspace: dir_bn_sync_workspace_bytes(R, C). */
long long dir_bn_sync_workspace_bytes(int R, int C);
int dir_bn_sync_local_stats(const float* x, float* part, int R, int C, int ld, float* workspace, long long workspace_bytes, void* stream);
int dir_bn_sync_combine(const float* parts, int world, int C, float* mean, float* var, float* running_mean, float* running_var, float momentum, void* stream);
int dir_bn_sync_backward_sums(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd, float* sums,
                              int R, int C, int ld, int relu, float* workspace, long long workspace_bytes, void* stream);
int dir_bn_sync_backward_apply(const float* gy, const float* x, const float* w, const float* b, const float* save_mean, const float* save_rstd,
                               const float* sums_pooled, float* gx, int R, float rows_pooled, int C, int ld, int relu, void* stream);
int dir_relu_forward(const float* x, float* y, long long n, void* stream);
int dir_relu_backward(const float* gy, const float* y, float* gx, long long n, void* stream);   /* g x = y > 0 ? g y : 0 */
/* Nearest-neighbour upsampling by a power-of-two factor in training form (the fuse layers of an HRNet module, y_i = relu(sum_j f_ij(x_j)) with
 * f_ij = upsample(bn(conv1x1(x_j))) for j > i; no reference counterpart: dir_amd/models/backbone/hrnet.py): dst [B][h f][w f][C] += up(src [B][h][w][C]),
 * and the backward gx [B][h][w][C] = the sum of gy over each f x f block.  NHWC fp32, C % 4 == 0, 16-byte aligned. */
int dir_upsample_nearest_add_f32(const float* src, float* dst, int B, int h, int w, int C, int factor, void* stream);
int dir_upsample_nearest_backward_f32(const float* gy, float* gx, int B, int h, int w, int C, int factor, void* stream);
/* PGraphConv's adjacency (SemGCN/p_graph_conv.py:43-50): A_1 [21][21] = row-softmax of the hand-skeleton mask filled with e_1 [40]
 * (row-major nonzero order); backward: g e_1 from g z [B,21,128] (gradient of the layer's pre-BatchNorm output) and h1 = x W_1 [B,21,128]
 * (g A_1[j][k] = sum_b <g z[b][j], h1[b][k]> on the edges, then the softmax chain rule).  scratch40: 40 floats. */
/* ImgFeature2JointFeature's sampler in training form (models/dir.py:197-198): F.grid_sample(bilinear, zeros, align_corners False) of
 * feat NHWC fp32 [B,S,S,C] at uv [B,21,2] -> rows [B*21][C]; backward w.r.t. feat only (uv arrives detached, models/dir.py:447-453):
 * g feat += sum over `hands` samplers (g_rows_h / uv_h: host arrays of device pointers); zero_first != 0 clears g feat before.
 * Deterministic: the taps of a (sample, channel) are applied in a fixed order by one thread. */
int dir_grid_rows_forward(const float* feat_nhwc, const float* uv, float* rows, int B, int S, int C, void* stream);
int dir_grid_rows_backward(const float* const* g_rows_h_host, const float* const* uv_h_host, int hands, float* g_feat_nhwc, int B, int S, int C,
                           int zero_first, void* stream);
/* Backward pass, image half: the spatial operators between the convolutions (fp32 NHWC, deterministic gather forms).
 * dir_maxpool3x3s2_backward: nn.MaxPool2d(3,2,1) (models/backbone/resnet.py:247): g x from x [B,H,W,C] and g y [B,Ho,Wo,C]; the gradient of a
 *   window goes to its FIRST maximum in (ky, kx) order, as ATen's max_pool2d_with_indices picks it.
 * dir_upsample2x_bilinear_backward: nn.Upsample(2, bilinear) (models/dir.py:392,398): g x [B,H,W,C] from g y [B,2H,2W,*] (channel slice
 *   gy_coff .. +C of rows of gy_cstride floats: the upsampled map is one half of a concatenation).
 * dir_attn_pool_forward / _backward: InitRegressor's pooling (models/dir.py:263-270): attn = sigmoid(logit [B,HW]);
 *   pooled [B,C] = sum_p feat[p,c] attn[p] / (sum_p attn[p] + 1e-8); mean [B,C] = feat.mean over the pixels.  Backward: g feat (written, or
 *   added with accumulate != 0) from g pooled and g mean (either may be NULL), g logit [B,HW] (optional).
 * dir_bone_proj_backward: Joint2BoneFeature.bone_proj (models/dir.py:146-174) for `hands` hands: from g img [B,S,S,*] (channel
 *   (hand*20 + bone)*64 + c at offset img_coff of rows of img_cstride floats) the gradients w.r.t. the re-embedded joint features
 *   emb [B,42,64] (-> g_emb [B,42,64]) and the joint uv [B,21,2] per hand (-> g_uv_lr, optional).  The capsule mask and the weights are
 *   recomputed with the forward's own arithmetic; scratch: dir_bone_proj_backward_scratch_bytes(B, hands). */
int dir_maxpool3x3s2_backward(const float* x, const float* gy, float* gx, int B, int H, int W, int C, void* stream);
int dir_upsample2x_bilinear_backward(const float* gy, float* gx, int B, int H, int W, int C, int gy_cstride, int gy_coff, void* stream);
int dir_attn_pool_forward(const float* feat, const float* logit, float* attn, float* pooled, float* mean, int B, int HW, int C, void* stream);
int dir_attn_pool_backward(const float* feat, const float* attn, const float* pooled, const float* g_pooled, const float* g_mean, float* g_feat,
                           float* g_logit, int B, int HW, int C, int accumulate, void* stream);
long long dir_bone_proj_backward_scratch_bytes(int B, int hands);
int dir_bone_proj_backward(const float* const* uv_lr_host, const float* emb, const float* g_img, int img_cstride, int img_coff, float distance,
                           float* g_emb, float* const* g_uv_lr_host, float* scratch, int B, int S, int hands, void* stream);
/* dst += alpha * src, n floats (gradient accumulation of the modules a stage runs once per hand: global_pos_emb, proj_feat_emb,
 * models/dir.py:106-107,118-119). */
int dir_axpy_f32(float* dst, const float* src, long long n, float alpha, void* stream);
/* dst_t += alpha * src_t for `count` tensors (host arrays of device pointers and element counts) in count / 40 launches: the gradients of a
 * training step into the flat all-reduce bucket (dir_amd/train/step.py::add_grads) */
int dir_axpy_multi_f32(float* const* dst_host, const float* const* src_host, const long long* n_host, int count, float alpha, void* stream);
/* Joint2BoneFeature's token inputs (models/dir.py:97-98,106-107): pos = xyz / 0.15, gpos_left = xyz_left / 0.15 - offset / 2,
 * gpos_right = xyz_right / 0.15 + offset / 2; xyz [B,21,3], offset [B,3], outputs [B*21,3]. */
int dir_stage_positions(const float* xyz_left, const float* xyz_right, const float* offset, float* pos_left, float* pos_right,
                        float* gpos_left, float* gpos_right, int B, void* stream);
int dir_pgcn_adjacency_forward(const float* e1, float* A, void* stream);
int dir_pgcn_adjacency_backward(const float* e1, const float* gz, const float* h1, float* scratch40, float* g_e1, int B, void* stream);

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
#define DIR_DT_U8 2 /* input images only (dir_stem_pool_forward) */
/* dir_conv2d_* only, as in_dtype (out_dtype = DIR_DT_F32): "split precision".  Tensors are fp32 in memory; every product a*w is
 * evaluated on the f16 matrix cores as hi(a)*hi(w) + lo(a)*hi(w) + hi(a)*lo(w) with hi(x) = f16(x), lo(x) = f16(x - hi(x)) and fp32
 * accumulation: 22 significant bits per operand (error ~2^-22 |a w| per product, below the fp32 accumulation noise of any K >= 8
 * reduction) at 3 instead of 16 matrix-core cycles per product of the exact fp32 path.  This is the mode that meets the 1e-4 mm parity
 * budget at several times the fp32 mode's speed.  The caller passes the WEIGHTS already split:
 *   w = f16 [Cout][kh][kw][Cin/32][2][32]   (per 32-channel slab: 32 hi values, then the 32 lo values; the same bytes / addressing as
 *                                            the fp32 tensor [Cout][kh][kw][Cin]; for dir_conv2d_dual_* the rows are [kh*kw*Cin | Cin2])
 * of the weights PRE-SCALED per output channel n by a power of two p_n chosen so that max |w_n| p_n lies in [2^12, 2^13) (their lo
 * parts are then normal f16 numbers), with 1 / p_n folded into scale[n] (dir_amd/functional.py::pack_f16x3_weights).  The ACTIVATIONS
 * are split inside the kernel after multiplication by dir_conv_desc.in_scale, a power of two the caller picks per layer so that the
 * layer's typical maximum lands near 2^9 (dir_amd/engine.py::DirEngine.calibrate): 64x headroom below the f16 maximum 65504 (beyond it
 * values saturate), full 22-bit accuracy down to max / 4096, below that an absolute floor of 2^-25 / in_scale per element (f16 denormal
 * lo parts, which the matrix cores honour: tools/ubench_f16_denorm.hip).  Cin % 32 == 0 as for fp32. */
#define DIR_DT_F16X3 3
/* the same operands and packing with the hi parts only: ONE f16 MFMA per product (operands rounded to f16, fp32 accumulation) -- the
 * arithmetic of torch.autocast(float16) on fp32 tensors, 8x finer than bf16; the "fp16 MFMA path" of BASELINE config 5 */
#define DIR_DT_F16X1 4
/* F16X3 / F16X1 with the ACTIVATIONS pre-split as well: x is what dir_split_f16_forward wrote ([pixel][Cin/32][32 hi | 32 lo] f16, the
 * bytes and addressing of an fp32 tensor with in_cstride = Cin, in_coff = 0), already multiplied by in_scale and through the
 * pre-activation (so pre_scale / pre_shift must be NULL here and in_scale is ignored; 1 / in_scale still belongs in scale[]).  Both
 * operands then travel global -> LDS by DMA and nothing is converted per output tile: the form for layers with a long reduction or many
 * output-channel tiles. */
#define DIR_DT_F16X3P 5
#define DIR_DT_F16X1P 6
/* f16 STORAGE (round 5): feature maps and convolution weights held as IEEE binary16 -- byte for byte the layouts, entry points and kernels of
 * DIR_DT_BF16 (64-channel K-slabs, LDS-DMA operand path, v_mfma_f32_32x32x16_f16 instead of _bf16, fp32 accumulation and epilogue), with an
 * 11-bit significand per stored value instead of 8.  Outputs are rounded to nearest even after clamping to +-65504 (no inf is ever stored).
 * Accepted wherever DIR_DT_BF16 is, for the feature-map / convolution-weight side; the token path's weight_dtype / w_dtype stay F32 | BF16. */
#define DIR_DT_F16 7
#define DIR_CONV_RELU 1
#define DIR_CONV_PRE_RELU 2
/* Optional kernel choice in bits 8..15 of dir_conv_desc.flags (0 = the library's per-layer heuristic).  Every variant
 * computes the same fp32 accumulation order (bit-identical results); a variant that does not apply to the layer
 * (alignment, pre-activation, grid) silently falls back to the heuristic.  dir_amd.engine times them per layer.
 *   1..4  : 4-wave kernel (conv.hip), tile 128x128 | 128x64 | 64x128 | 64x64 ; +16: 3-buffer DMA ring
 *   8..10 : 8-wave pipelined kernel (conv_pipe.hip), tile 256x128 | 128x128 | 256x64
 *   11    : 8-wave 256x256 tile, two-deep 64 KB slab ring (conv_big.hip; bf16 operands, Cout > 128, no pre-activation / second source)
 *   12..14: the same tiles with halo reuse (stride-1 kh x kw layers whose tile is a rectangle of image rows)
 *   15    : the 8-wave pipelined kernel on a 128x64 tile (32x32 wave tiles): the small-M layers (8x8 / 16x16 maps)
 *   22, 23: dir_conv1x1_stream_forward only: 64- / 32-pixel workgroups instead of 128 (the 16x16 / 8x8 stages, whose 128-pixel grid does not
 *           cover the CUs); (19 = 64x128 on the ring: not in the product build, see conv.hip)
 *   25..28: dir_conv2d_as_forward only (activation-stationary kernel for the small maps): (A, PB) = (2,2) | (2,4) | (4,2) | (1,2) */
#define DIR_CONV_VARIANT(v) (((v) & 0xff) << 8)

typedef struct dir_conv_desc {
    int32_t B, H, W;
    int32_t Cin, in_cstride, in_coff;
    int32_t Cout, out_cstride, out_coff;
    int32_t res_cstride, res_coff;
    int32_t kh, kw, stride, pad;
    int32_t in_dtype, out_dtype; /* DIR_DT_*; residual is read in out_dtype */
    int32_t flags;               /* DIR_CONV_* */
    int32_t Ho, Wo;              /* 0 = (H + 2*pad - kh)/stride + 1 ; set explicitly for the pre-padded stem image */
    float in_scale;              /* DIR_DT_F16X3 only (0 = 1): a power of two the activations (after the pre-activation, both sources of a
                                    dual convolution) are multiplied by before the f16 hi / lo split, to centre them in the f16 range;
                                    the caller folds 1 / in_scale into scale[].  Scaled values are clamped to +-65504 (no inf / nan). */
    float out_split_scale;       /* 0: y as declared by out_dtype.  != 0 (fp32 outputs that are whole tensors, Cout % 32 == 0): y is written as the pre-split
                                    operand of the NEXT convolution (DIR_DT_F16X3P / F16X1P) instead -- act(result) * |out_split_scale| as f16 hi | lo
                                    slabs (negative: hi only), the same bytes as the fp32 tensor, so that consumer needs no dir_split_f16_forward pass.
                                    |out_split_scale| = the consumer's in_scale. */
} dir_conv_desc;

int dir_conv2d_forward(const dir_conv_desc* desc_host, const void* x, const void* w, const float* scale,
                       const float* shift, const float* pre_scale, const float* pre_shift,
                       const void* residual, void* y, void* stream);
/* round 5 -- dir_conv2d_forward (fp32 output, 16-byte aligned rows) with an output MASK: y = mask > 0 ? (conv(x) + residual) : 0, mask an fp32 tensor laid
 * out like y.  The training step's use: the data gradient of a Bottleneck's conv1 is the gradient of the PREVIOUS block's output, whose ReLU backward
 * (`out = relu(...)`, models/backbone/resnet.py:140) is this mask with that block's stored output -- applied where the gradient is written instead of
 * in a pass of its own over (gradient, output). */
int dir_conv2d_forward_masked(const dir_conv_desc* desc, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                              const float* pre_shift, const void* residual, const float* mask, void* y, void* stream);
/* round 5 -- the data-gradient convolutions of the training step: dir_conv2d_forward (fp32 output) + residual + output mask (dir_conv2d_forward_masked) + the
 * chunk partials of the BACKWARD pass of the BatchNorm (+ ReLU) whose output's gradient the convolution writes.  y [M][Cout] is d loss / d (BatchNorm
 * output); bn->z [M][Cout] the BatchNorm's input, mean / rstd / w / b [Cout] its saved statistics and affine (w, b may be NULL), relu != 0: the ReLU after it
 * (mask BatchNorm(z) > 0, re-computed like dir_bn_train_backward does).  p1 / p2 [ceil(M / rows)][Cout] (room for ceil(M / 64) x Cout each) receive per M tile
 * the sums of the masked gradient and of the masked gradient times xhat = (z - mean) rstd; *chunk_rows = rows per tile, 0 when the kernel that took the
 * launch does not form them.  Feed them to dir_bn_train_backward_from_partials.  nn.Conv2d <- nn.BatchNorm2d <- nn.ReLU under autograd
 * (models/backbone/resnet.py:125-140, hourglass.py:60-69). */
typedef struct dir_conv_bn_bwd {
    const float *z, *mean, *rstd, *w, *b;
    int32_t relu;
    float *p1, *p2;
} dir_conv_bn_bwd;
int dir_conv2d_forward_ex(const dir_conv_desc* desc, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                          const float* pre_shift, const void* residual, const float* mask, void* y, const dir_conv_bn_bwd* bn, int* chunk_rows, void* stream);
/* round 5 -- dir_conv2d_forward (no residual, no activation, whole fp32 output tensor) that ALSO forms the chunk partials of the training-mode
 * BatchNorm that follows the convolution (nn.Conv2d -> nn.BatchNorm2d, models/backbone/resnet.py:123-133, hourglass.py:14-27) from the output
 * tile while it is in registers: p1 / p2 [ceil(M / rows)][Cout] (room for ceil(M / 64) x Cout floats each), *chunk_rows = rows (the M tile of the
 * kernel that took the launch: 64 / 128 / 256), or 0 when that kernel does not form them -- then run dir_bn_train_stats on y.  Feed them to
 * dir_bn_train_stats_from_partials. */
int dir_conv2d_forward_stats(const dir_conv_desc* desc, const void* x, const void* w, const float* scale, const float* shift, const float* pre_scale,
                             const float* pre_shift, void* y, float* p1, float* p2, int* chunk_rows, void* stream);

/* fp32 NHWC channel slice x[pixel][in_cstride] (channels [in_coff, in_coff + C)) -> y = the pre-split operand of DIR_DT_F16X3P / F16X1P:
 * v = x (* pre_scale + pre_shift, ReLU if pre_relu: the pre-activation of hourglass.Residual) * in_scale, clamped to +-65504, stored per
 * 32-channel slab as 32 hi = f16(v) then 32 lo = f16(v - hi) (lo = 0 with hi_only).  y: pixels * C * 4 bytes.  C % 32 == 0. */
int dir_split_f16_forward(const float* x, void* y, long long pixels, int C, int in_cstride, int in_coff, const float* pre_scale,
                          const float* pre_shift, int pre_relu, float in_scale, int hi_only, void* stream);

/* The weight operand of DIR_DT_F16X3 (above) from fp32 rows w [N][K] (K % 32 == 0) in one launch, for weights that change every step
 * (training): packed = f16 [N][K/32][2][32], scale_out[n] = (scale_in ? scale_in[n] : 1) / p_n.  Same result as the host packing of
 * dir_amd/functional.py::pack_f16x3_weights (tests/test_gpu_f16x3.py). */
int dir_pack_f16x3_weights(const float* w, void* packed, float* scale_out, const float* scale_in, int N, int K, void* stream);

/* Every convolution weight of one training step in both DIR_DT_F16X3 operand forms, straight from the reference's OIHW parameters
 * (nn.Conv2d.weight, models/backbone/resnet.py:23-40, hourglass.py:14, models/dir.py:58-61) in ONE launch -- the weights change every
 * optimiser step (train.py:70), and packing them per convolution call cost ~600 launches per step.  `table` and `wg_start` live in DEVICE
 * memory.  Per entry: w [Cout][Cin][kh][kw] fp32;
 *   fwd   (may be NULL: skipped)  f16 [Cout][kh*kw*Cin/32][2][32]: rows of the OHWI matrix (k = tap * Cin + c), Cin % 32 == 0;
 *                                 fwd_scale[o] = fwd_inv_in / p_o       (what dir_pack_f16x3_weights returns, times 1 / in_scale)
 *   dgrad (may be NULL: skipped)  f16 [Cin][kh*kw*Cout32/32][2][32]: the transposed convolution's weights, row c, k = tap * Cout32 + o
 *                                 = w[o][c][kh*kw - 1 - tap] (taps flipped, Cout padded to Cout32 = ceil32(Cout) with zeros);
 *                                 dgrad_scale[c] = dgrad_inv_in / p_c.
 * kh * kw must be 1 or 9 and Cin % 4 == 0 (every convolution of the path but the 3-channel stem, which keeps its own packing).
 * wg_start[e] = first workgroup of entry e in the launch grid (workgroups of an entry: Cout if fwd -- one per row --, then Cin / 4 if dgrad --
 * one per four rows), wg_start[entries] = total_workgroups.
 * Bit-identical to dir_pack_f16x3_weights on the copied / flipped / padded matrices (tests/test_gpu_conv_bwd.py). */
typedef struct {
    const float* w; void* fwd; float* fwd_scale; void* dgrad; float* dgrad_scale;
    int32_t Cout, Cin, kh, kw;
    float fwd_inv_in, dgrad_inv_in;
} dir_train_weight;
int dir_train_pack_conv_weights(const dir_train_weight* table, const int* wg_start, int entries, int total_workgroups, void* stream);

/* dir_conv2d_forward with the reduction split over `splits` workgroups per output tile (bf16 -> bf16 layers whose M x Cout grid is too
 * small to fill 256 CUs at the benchmark batch: ResNet layer4 at 8x8, the decoder's 16x16 Residual blocks).  128x128 tiles; every
 * workgroup reduces a contiguous range of K-slabs and writes its raw fp32 partial tile to the workspace; the LAST one to arrive at the
 * tile's counter sums all partials in split order -- so the result does not depend on the arrival order -- and runs the usual epilogue
 * (scale / shift, residual, ReLU).  The summation order differs from the unsplit kernels': outputs agree to bf16 rounding, not bit for
 * bit.  workspace: dir_conv2d_splitk_workspace_bytes(d, splits) bytes, 16-byte aligned, its first 16 KiB ZERO before the first use (the
 * kernel leaves them zero); one workspace must not be shared by launches that can run concurrently.  splits == 1 = dir_conv2d_forward.
 * EXPERIMENTAL, off by default (DIR_SPLITK=1 / ConvOp.split): the cross-workgroup hand-over publishes the partial tiles as relaxed
 * agent-scope atomic stores that are drained (s_waitcnt vmcnt(0)) before a relaxed ticket increment -- sufficient on gfx950, where a
 * completed sc1 store is visible device-wide, but NOT a release / acquire pair in the HIP memory model (a formal release on the ticket
 * costs an L2 write-back per workgroup: +12 us per split, which erases the gain).  Measured no faster than the tiled kernels at the
 * benchmark batch; kept for small-batch latency experiments, covered by tests/test_gpu_splitk.py (incl. a many-iteration stress case). */
long long dir_conv2d_splitk_workspace_bytes(const dir_conv_desc* d, int splits);
int dir_conv2d_splitk_forward(const dir_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                              const float* pre_scale, const float* pre_shift, const void* residual, void* y, int splits,
                              void* workspace, long long workspace_bytes, void* stream);

/* d loss / d weight of nn.Conv2d in fp32 (train.py:68 runs autograd through every Conv2d of models/backbone/resnet.py,
 * models/backbone/hourglass.py and models/dir.py): d = the FORWARD geometry (in_cstride / in_coff address x, out_cstride / out_coff address
 * gy), x NHWC [B,H,W,*], gy NHWC [B,Ho,Wo,*], gw [Cout][kh][kw][Cin] (the layout dir_conv2d_forward reads); accumulate != 0 adds to gw.
 * The reduction over the B*Ho*Wo output pixels is cut into chunks whose partial tiles are added in chunk order (deterministic).
 * workspace: dir_conv2d_wgrad_workspace_bytes(d) bytes (0 = none needed unless accumulate, then the weight size).
 * The data gradient needs no kernel of its own: it is dir_conv2d_forward on gy with the flipped, transposed weights (zeros inserted
 * between the rows / columns of gy for stride 2), dir_amd/train/conv.py. */
long long dir_conv2d_wgrad_workspace_bytes(const dir_conv_desc* d);
int dir_conv2d_wgrad_f32(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                         long long workspace_bytes, void* stream);

/* The same weight gradient on the f16 matrix cores in split precision (the arithmetic of DIR_DT_F16X3: every product gy * x as hi*hi + lo*hi +
 * hi*lo of the operands' f16 hi / lo halves, fp32 accumulation; ~2^-22 per product).  x_scale / gy_scale: powers of two by which x / gy are
 * multiplied before the split (the largest |value| belongs near 2^9 .. 2^10: dir_amd.functional.pow2_in_scale; values saturate at the f16
 * maximum, never inf / nan); 1 / (x_scale * gy_scale) is applied to the result.  Same geometry descriptor, result layout, chunked
 * deterministic pixel reduction and accumulate semantics as dir_conv2d_wgrad_f32; channel counts, strides and offsets must be multiples of 4
 * (DIR_E_ARG otherwise: use dir_conv2d_wgrad_f32).  workspace: dir_conv2d_wgrad_f16x3_workspace_bytes(d). */
long long dir_conv2d_wgrad_f16x3_workspace_bytes(const dir_conv_desc* d);
int dir_conv2d_wgrad_f16x3(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                           long long workspace_bytes, float x_scale, float gy_scale, void* stream);
/* the same with a pre-activation on the x operand: x <- act(x pre_scale[c] + pre_shift[c]) per input channel (pre_relu != 0: max(., 0)) where the
 * kernel reads it, padding taps zero -- the weight gradient of a convolution whose input is a BatchNorm (+ ReLU) that was never materialised
 * (dir_bn_train_stats).  x_scale is the power-of-two scale of the ACTIVATED operand. */
int dir_conv2d_wgrad_f16x3_pre(const dir_conv_desc* d, const float* x, const float* gy, float* gw, int accumulate, float* workspace,
                               long long workspace_bytes, float x_scale, float gy_scale, const float* pre_scale, const float* pre_shift, int pre_relu,
                               void* stream);

/* A convolution with a SECOND source accumulated into the same output tile:
 *   y = epilogue( conv(x; kh x kw, stride, pad) + conv1x1(x2; stride2) )
 * -- ResNet's projection shortcut (`downsample`, models/backbone/resnet.py:137-140 and :117-119) folded into the block's last
 * convolution, so the identity tensor is neither written nor read back.  w: [Cout][kh*kw*Cin + Cin2] (the second source's
 * columns appended to every row) with both BatchNorm scales pre-multiplied into the rows; shift = the sum of the two folded
 * shifts; flags: DIR_CONV_RELU.  x2: NHWC [B, d2->H, d2->W, in_cstride], read at pixels (oy*stride, ox*stride). */
typedef struct dir_conv_src2 {
    int32_t H, W, Cin, in_cstride, in_coff, stride;
} dir_conv_src2;
int dir_conv2d_dual_forward(const dir_conv_desc* desc, const void* x, const dir_conv_src2* src2, const void* x2, const void* w,
                            const float* shift, void* y, void* stream);
/* the same with a per-output-channel scale applied to the SUM of both sources before the shift (needed by DIR_DT_F16X3, whose weight
 * rows carry a power-of-two prescale; scale may be NULL = dir_conv2d_dual_forward) */
int dir_conv2d_dual_scaled_forward(const dir_conv_desc* desc, const void* x, const dir_conv_src2* src2, const void* x2, const void* w,
                                   const float* scale, const float* shift, void* y, void* stream);

/* The HBM-bound 1x1 convolutions (bf16 in / out, stride 1) as a streaming kernel: small synchronous workgroups (128 pixels x
 * 128 | 256 output channels), several per CU, activations global -> registers -> LDS two K-chunks ahead, weights straight
 * into MFMA operand registers from a stream packed in consumption order.  Same mathematics as dir_conv2d_forward /
 * dir_conv2d_dual_forward for kh = kw = 1 (models/backbone/hourglass.py:55-70 conv1 with its pre-activation and conv3 + skip_layer;
 * models/backbone/resnet.py:117-140 conv1 / conv3 + projection shortcut): y = act(scale * (x . W1^T + x2 . W2^T) + shift), with
 * the optional pre-activation relu(x * pre_scale + pre_shift) on the FIRST source only.  No residual input.
 * desc: kh = kw = stride = 1, pad = 0, Cin % 64 == 0, Cout % 128 == 0, bf16; src2 / x2 optional (1x1, stride src2->stride).
 * w_stream: bf16 [Cout / NWG][4 waves][K / 64 chunks][4 k-steps][NCB][64 lanes][8] with NWG = 256, NCB = 2 when Cout % 256 == 0,
 * else NWG = 128, NCB = 1; lane l of fragment (chunk c, k-step ks, block cb) of wave w in N-chunk g holds
 *   W[g*NWG + (w*NCB + cb)*32 + (l & 31)][64*c + 16*ks + 8*(l >> 5) .. +8],   W = [W1 | W2] along K
 * (dir_amd/engine.py::pack_stream_weights). */
int dir_conv1x1_stream_forward(const dir_conv_desc* desc, const void* x, const dir_conv_src2* src2, const void* x2,
                               const void* w_stream, const float* scale, const float* shift, const float* pre_scale,
                               const float* pre_shift, void* y, void* stream);

/* The convolutions on SMALL maps (W = 8 | 16 | 32, stride 1, 1x1 or 3x3 / pad 1, 16-bit storage in = out) as an ACTIVATION-STATIONARY kernel
 * (conv_as.hip, round 6): a workgroup owns 32 * pixel_blocks pixels of one image (whole rows) x 128 * blocks_per_wave output channels, its whole input
 * patch (every input channel, halo included) goes global -> LDS once by DMA, the weights stream L2 -> MFMA operand registers from a buffer packed in
 * consumption order, and there is no barrier inside the reduction.  Same mathematics, K order and k-slot assignment as dir_conv2d_forward
 * (bit-identical outputs): y = act(scale * conv(x) + shift (+ residual)) -- models/backbone/resnet.py:120-140 (Bottleneck conv1 / conv2 / conv3 +
 * identity at 16x16 and 8x8), models/backbone/hourglass.py:55-70 (Residual conv2), models/dir.py:227-241.  No pre-activation, no second source.
 * (blocks_per_wave A, pixel_blocks PB) in {(2,2), (2,4), (4,2), (1,2)}; Cin % 64 == 0, Cout % (128 A) == 0, (H W) % (32 PB) == 0, (32 PB) % W == 0,
 * patch and staging <= 160 KB of LDS -- or, for (4, 2), a ring of TWO patch chunks of 128 .. 1024 channels (the attention convolution, 8x8x2048:
 * 8 chunks of 256 channels; same K order, bit-identical).  dir_conv2d_as_supported returns 1 when a layer qualifies.
 * w_as: 16-bit [Cout / (128 A)][4 waves][kh kw Cin / 64 steps][4 k-steps][A][64 lanes][8]; step s = (64-channel slab s / (kh kw), tap s % (kh kw)); lane l of
 * fragment (step, ks, cb) of wave w in slice g holds W[g 128 A + (w A + cb) 32 + (l & 31)][tap][64 slab + 8 ks + 32 (l >> 5) .. + 8] of the
 * [Cout][kh][kw][Cin] weights (dir_amd/engine.py::pack_as_weights). */
int dir_conv2d_as_supported(const dir_conv_desc* desc, int blocks_per_wave, int pixel_blocks);
int dir_conv2d_as_forward(const dir_conv_desc* desc, const void* x, const void* w_as, const float* scale, const float* shift,
                          const void* residual, void* y, int blocks_per_wave, int pixel_blocks, void* stream);

/* a11 with a10's sparsity: same as dir_conv2d_forward (no prologue), plus group_bbox int32 [B][Cin/64][4] = for every
 * image and every 64-channel input group the pixel box (ymin,ymax,xmin,xmax) outside which that group is exactly zero.
 * K-slabs (tap x group) that cannot touch an output tile are skipped: the sum only loses exact-zero products, so the
 * result is bit-identical to the dense call.  The rasterised bone features (models/dir.py:146-174) are ~93 % zeros and
 * one group == one (hand, bone).  Falls back to dense when (Ho*Wo) % 128 != 0 or Cin % 64 != 0. */
int dir_conv2d_sparse_forward(const dir_conv_desc* desc_host, const void* x, const void* w, const float* scale,
                              const float* shift, const void* residual, void* y, const int32_t* group_bbox,
                              void* stream);

/* ------------------------------------------------------------------------------------------------
 * HBM-bound spatial helpers (NHWC, dtype = DIR_DT_*; arithmetic in fp32)
 */
/* NCHW fp32 image [B,3,H,W] -> zero-padded NHWC4 [B,Hp,Wp,4] (RGB0) with `pad` blank pixels top/left, so that the
 * 7x7/s2 stem (models/backbone/resnet.py:176,244) becomes a kh=7,kw=1 implicit GEMM over contiguous pixel windows. */
int dir_stem_prep(const float* img_nchw, void* out, int B, int H, int W, int Hp, int Wp, int pad, int dtype,
                  void* stream);
/* The same staging as 2x2 space-to-depth blocks: out [B,Hs,Ws,16] (dtype), channel (dy*2+dx)*4 + c =
 * img[b][c][2Y-4+dy][2X-4+dx] (zero outside the image / for c = 3); Hs >= H/2+3, Ws >= W/2+3.  The 7x7/2 stem becomes
 * dir_conv2d_forward with kh=4, kw=1, stride=1, pad=0, Cin=64 (4 blocks x 16 channels), in_cstride=16, Ho=H/2, Wo=W/2 and
 * weights [Cout][4][1][64] (w[n][r][0][j*16 + (dy*2+dx)*4 + c] = conv1.weight[n, c, 2r+dy-1, 2j+dx-1], zero outside 0..6):
 * K = 256 per output instead of 448. */
int dir_stem_prep_s2d(const float* img_nchw, void* out, int B, int H, int W, int Hs, int Ws, int dtype, void* stream);
/* nn.MaxPool2d(3, 2, 1) (models/backbone/resnet.py:179,247): [B,H,W,C] -> [B,(H+1)/2,(W+1)/2,C] */
int dir_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear') (models/dir.py:392,398; align_corners=False): [B,H,W,C] ->
 * channels [out_coff, out_coff+C) of y [B,2H,2W,out_cstride] (out_cstride 0 = C) */
int dir_upsample2x_bilinear(const void* x, void* y, int B, int H, int W, int C, int out_cstride, int out_coff,
                            int dtype, void* stream);
/* f4 (HRNet fuse layers; no reference counterpart): acc [B,H,W,C] = act(acc + nearest_upsample(src [B,H/f,W/f,C], f)), in place, one term
 * of `y = y + fuse_layers[i][j](x[j])` at a time; factor 1 = a plain add; relu != 0 on the last term (the fuse layer's ReLU). */
int dir_add_upsampled(void* acc, const void* src, int B, int H, int W, int C, int factor, int relu, int dtype, void* stream);
/* the whole fuse row in one pass (round 5): out [B,H,W,C] = act(base + sum over t < nsrc (<= 4) of nearest_upsample(srcs[t] [B,H/f_t,W/f_t,C], f_t)),
 * summed in fp32 in source order and rounded once; out may be base.  `y = sum_j fuse_layers[i][j](x[j])` + ReLU of an HRNet module (no reference
 * counterpart): 2 passes over the row's map instead of 2 per term + a copy. */
int dir_fuse_sum(void* out, const void* base, const void* const* srcs, const int* factors, int nsrc, int B, int H, int W, int C, int relu, int dtype,
                 void* stream);

/* InitRegressor tail (models/dir.py:263-270): 1x1 conv Ch->1 + sigmoid attention, attention-weighted pooling
 * of c4 (+1e-8), plain mean, Linear C->64 (left, right) and C->3 (offset). */
typedef struct dir_init_head_params {
    const float* attn_w[2]; /* [Ch]    attention_{left,right}.3.weight */
    float attn_b[2];        /*         attention_{left,right}.3.bias   */
    const float* mano_wt;   /* [C][128] k-major: column o < 64 = mano_left.weight[o, :], o >= 64 = mano_right.weight[o-64, :] */
    const float* mano_b[2]; /* [64]                                    */
    const float* off_w;     /* [3][C]  offset.weight                   */
    const float* off_b;     /* [3]                                     */
} dir_init_head_params;
/* c4 [B,HW,C]; h_left/h_right = relu(bn(conv3x3(c4))) of each attention branch (dtype): Ch channels per pixel, pixels
 * h_cstride elements apart (0 = Ch) -- both branches may live in one [B,HW,2*Ch] buffer produced by a single N=2*Ch
 * convolution; outputs fp32: para_left/right [B,64], offset [B,3]. */
int dir_init_head_forward(const dir_init_head_params* params_host, const void* c4, const void* h_left,
                          const void* h_right, int h_cstride, float* para_left, float* para_right, float* offset,
                          int B, int HW, int C, int Ch, int dtype, void* stream);

/* a10: Joint2BoneFeature.bone_proj + lineseg_dists (models/dir.py:132-174) for BOTH hands.
 * uv_left/right [B,21,2] in [-1,1]; emb [B,42,64] (tokens 0..20 left, 21..41 right);
 * out NHWC [B,S,S,2560] (channel = hand*1280 + bone*64 + c == torch.cat((left,right),1) of models/dir.py:122); may be
 * NULL when only vis_nchw is wanted (dir_bone_fusion_forward covers the convolution);
 * vis_nchw (optional) fp32 [B,1280,S,S] = left + right (vis_img_feat / proj_feat, models/dir.py:128,481).
 * The capsule mask `distance < thr` follows the reference's fp32 op order exactly (bit-exact support).
 * group_bbox (optional) int32 [B][40][4]: conservative pixel box (ymin,ymax,xmin,xmax; empty when min > max) outside
 * which channel group hand*20+bone (64 channels) of `out` is exactly zero -- input of dir_conv2d_sparse_forward. */
int dir_bone_proj_forward(const float* uv_left, const float* uv_right, const float* emb, void* out, float* vis_nchw,
                          int32_t* group_bbox, int B, int S, float distance, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Joint-token operators of a refinement stage (all fp32)
 */
/* Conv1d(k=1) -> BatchNorm1d(eval) -> ReLU -> Conv1d(k=1) on tokens (models/dir.py:31-56,180-185), weights k-major:
 *   hid = relu((w1t^T x) * s1 + b1)   with s1 = gamma/sqrt(var+eps), b1 = (conv_bias - mean)*s1 + beta
 *   out = w2t^T hid + b2 */
typedef struct dir_token_mlp {
    const float* w1t; /* [Cin][Cmid]  */
    const float* s1;  /* [Cmid]       */
    const float* b1;  /* [Cmid]       */
    const float* w2t; /* [Cmid][Cout] */
    const float* b2;  /* [Cout]       */
} dir_token_mlp;

/* a4 + a12: ImgFeature2JointFeature.forward (models/dir.py:197-200: F.grid_sample bilinear / zeros /
 * align_corners=False at the 21 joint uv, then Conv1d 256->128, BN, ReLU, Conv1d 128->128), pos_emb_{left,right}
 * (xyz/0.15, models/dir.py:97-98) and global_pos_emb (xyz/0.15 -/+ offset/2, models/dir.py:106-107), both hands.
 * feat NHWC [B,S,S,feat_cstride] (feat_dtype), channels [feat_coff, feat_coff+256) are sampled (feat_cstride 0 =
 * dense); uv [B,21,2]; xyz [B,21,3]; offset [B,3];
 * x0   [2][B][21][128] = pos_emb + img2joint (GCN input, models/dir.py:100-101), hand 0 = left
 * gpos [2][B][21][128] = global_pos_emb output. */
int dir_grid_tokens_forward(const void* feat, int feat_dtype, int S, int C, int feat_cstride, int feat_coff,
                            const float* uv_left,
                            const float* uv_right, const float* xyz_left, const float* xyz_right, const float* offset,
                            const dir_token_mlp* img2joint_lr_host, const dir_token_mlp* pos_emb_lr_host,
                            const dir_token_mlp* global_pos_emb_host, float* x0, float* gpos, int B, void* stream);

/* a5: one _GraphConv = PGraphConv + BatchNorm1d(eval) + ReLU (SemGCN/p_graph_conv.py:39-59, SemGCN/p_gcn.py:20-27) */
typedef struct dir_pgcn_layer {
    const float* W;        /* gconv.W: w_dtype DIR_DT_F32 -> fp32 [2][21][128 k][128 o] (reference layout, exact fp32 MFMA);
                              DIR_DT_BF16 -> bf16 [2][21][128 o][128 k] (transposed), bf16 MFMA with fp32 accumulation =
                              torch.autocast semantics for the two matmuls of SemGCN/p_graph_conv.py:47-48            */
    const float* e1;       /* [40]               gconv.e_1, row-major nonzero order of adj > 0    */
    const float* bias;     /* [128]              gconv.bias                                       */
    const float* bn_scale; /* [128]              gamma / sqrt(var + eps)                          */
    const float* bn_shift; /* [128]              beta - mean * bn_scale                           */
    int32_t relu;          /* 1: ReLU after BN (the _GraphConv of the network); 0: bare PGraphConv (scale 1, shift 0) */
    int32_t w_dtype;       /* DIR_DT_F32 | DIR_DT_BF16 (layout of W above)                                              */
} dir_pgcn_layer;
/* ResSimplePGCN.forward (SemGCN/p_gcn.py:71-73): x [B,21,128] -> out; `add` (optional, [B,21,128]) is added after the
 * last layer (global_pos_emb, models/dir.py:109-110); out rows are out_bstride floats apart so both hands can write
 * into one [B,42,128] token buffer.  e_0 is a dead parameter (softmax of a diagonal-only mask == I). scratch: 2*B*21*256
 * floats. */
int dir_pgcn_stack_forward(const dir_pgcn_layer* layers_host, int num_layers, const float* x, const float* add,
                           float* out, long long out_bstride, float* scratch, int B, void* stream);

/* both hands in one launch sequence: x_lr / add_lr [2][B][21][128] (hand 0 = left), tokens [B][42][128] (left tokens
 * 0..20, right 21..41), scratch 4*B*21*256 floats. */
int dir_pgcn_stack_forward_pair(const dir_pgcn_layer* layers_left_host, const dir_pgcn_layer* layers_right_host,
                                int num_layers, const float* x_lr, const float* add_lr, float* tokens, float* scratch,
                                int B, void* stream);

/* The same stack of both hands in ONE launch (round 4; replaces the five launches of dir_pgcn_stack_forward_pair, same arithmetic, bit-identical
 * tokens): one persistent workgroup per (hand, node, batch split) runs all layers of its node; layers are separated by per-node flags in
 * `sync_ws` (device memory, dir_pgcn_fused_sync_bytes() bytes, ZEROED ONCE by the caller and then left alone -- every launch raises the flags it
 * owns by one; one sync_ws per stream that may run this concurrently) instead of launches.  splits: batch splits per node (0 = the library's
 * choice; <= 8; DIR_PGCN_SPLITS overrides).  The word at byte offset dir_pgcn_fused_sync_bytes() - 16 becomes nonzero if a workgroup ever gave up
 * waiting (~1 s) for a neighbour -- the tokens of that launch are then invalid.  num_layers <= 4.  scratch as for the pair entry point. */
long long dir_pgcn_fused_sync_bytes(void);
int dir_pgcn_stack_forward_fused(const dir_pgcn_layer* layers_left_host, const dir_pgcn_layer* layers_right_host, int num_layers,
                                 const float* x_lr, const float* add_lr, float* tokens, float* scratch, void* sync_ws, int splits, int B,
                                 void* stream);

/* a6: STE.forward (transformer/mixSTE.py:194-205) on [B,42,128] -> [B,42,64].
 * weight_dtype DIR_DT_F32: the six Linear weights per block (*_wt) and head_wt are fp32, k-major ([in][out]); exact fp32.
 * weight_dtype DIR_DT_BF16: they are bf16 in nn.Linear's own [out][in] layout and the Linears run on the bf16 matrix cores
 * with fp32 accumulation -- torch.autocast(bfloat16) semantics (BASELINE config 4); LayerNorm, softmax, attention and the
 * residual stream stay fp32, biases and LayerNorm parameters are fp32 in both modes. */
typedef struct dir_ste_block {
    const float *ln1_w, *ln1_b, *qkv_wt, *qkv_b, *proj_wt, *proj_b, *ln2_w, *ln2_b, *fc1_wt, *fc1_b, *fc2_wt, *fc2_b;
} dir_ste_block;
typedef struct dir_ste_params {
    const float* pos_embed;  /* [42][128] */
    dir_ste_block blocks[3]; /* STEblocks[1..3]: block 0 is never executed (transformer/mixSTE.py:197) */
    int32_t num_blocks;      /* depth - 1 */
    const float *snorm_w, *snorm_b, *head_ln_w, *head_ln_b, *head_wt /* [128][64] */, *head_b;
    int32_t weight_dtype;    /* DIR_DT_F32 | DIR_DT_BF16 */
} dir_ste_params;
/* x_pos_out (optional): receives x + pos_embed, reproducing the reference's in-place `x += pos` on its input. */
int dir_ste_forward(const dir_ste_params* params_host, const float* x, float* x_pos_out, float* y, int B,
                    void* stream);

/* a7 + a12: RegressorOffset Linears (models/dir.py:342-351) + proj_feat_emb (models/dir.py:118-119).
 * tok [B,42,64]; prev_para_* [B,64] (detached previous mano_para); prev_offset [B,3];
 * para_* [B,64] = Linear(cat(tok_hand.flatten(), prev_para)), offset [B,3] = Linear(cat(tokL, tokR, prev_offset)),
 * emb [B,42,64] = proj_feat_emb(tok). */
typedef struct dir_regress_params {
    const float* mano_wt;   /* [1408][128] k-major: column o < 64 = regressor.mano_left.weight[o, :],
                               o >= 64 = regressor.mano_right.weight[o - 64, :]                       */
    const float* mano_b[2]; /* [64]                                          */
    const float* off_w;     /* [3][2691]  regressor.offset.weight            */
    const float* off_b;     /* [3]                                           */
    dir_token_mlp emb;      /* 64 -> 64 -> 64                                */
} dir_regress_params;
int dir_regress_forward(const dir_regress_params* params_host, const float* tok, const float* prev_para_left,
                        const float* prev_para_right, const float* prev_offset, float* para_left, float* para_right,
                        float* offset, float* emb, int B, void* stream);

/* a10 + a11 fused (bf16 throughput mode): Joint2BoneFeature.bone_proj (models/dir.py:132-174) + fusion[0..2] (3x3 conv
 * 2560 -> 256 + BatchNorm + ReLU, models/dir.py:57-62) without the [B,S,S,2560] bone map.  The rasterised operand is rank 2
 * per bone, so conv(img)[p,n] = sum_tap sum_e Wgt[p+tap, e] * G[tap, e, n] with e = (hand, bone, end) in [0,80),
 * G[tap,(hb,end),n] = sum_c f_end[hb][c] * W[n, hb*64+c, tap] and Wgt = mask * (wa | wb): K = 720 instead of 23040.
 * Same mathematics as dir_bone_proj_forward + dir_conv2d_forward, re-associated (so the fp32 parity mode keeps those).
 * Two launches so that the first can overlap the MANO layer (it needs the token features only):
 *   dir_bone_fusion_prepare : G for every sample from emb [B,42,64] (proj_feat_emb output) into `scratch`
 *                             (dir_bone_fusion_scratch_bytes(B) bytes of device memory);
 *   dir_bone_fusion_forward : uv_* [B,21,2] (pd_joint_uv) + scratch -> y, NHWC bf16 (fp32 with exact_f32) [B,S,S,out_cstride],
 *                             channels [out_coff, out_coff+256).  S in {16, 32, ...} with 256 % S == 0 and S*S % 256 == 0
 *                             (exact_f32: 128 % S == 0). */
typedef struct dir_bone_fusion_params {
    const float* w_g;   /* [9][40][64][256]: fusion.0.weight[n, hb*64 + c, ky, kx] at [ky*3+kx][hb][c][n], rounded to bf16 */
    const float* scale; /* [256] folded fusion.1 BatchNorm scale            */
    const float* shift; /* [256] folded BatchNorm shift (+ fusion.0 bias)   */
    int32_t exact_f32;  /* 0: bf16 operands (w_g rounded to bf16 by the caller, G and Wgt rounded to bf16, y bf16) -- the throughput mode;
                           1: everything fp32 on the exact fp32 matrix cores (w_g unrounded, y NHWC fp32) -- the parity modes: differs from
                           bone_proj + conv3x3 only by the association of the sum (fp32 rounding noise);
                           2 (round 5): as 0 with f16 instead of bf16 everywhere (w_g rounded to f16 by the caller, G / Wgt / y f16: DIR_DT_F16)      */
    float g_scale;      /* exact_f32 only.  0: the exact fp32 matrix cores.  A power of two > 0: split precision on the f16 matrix cores (the
                           arithmetic of DIR_DT_F16X3: hi*hi + lo*hi + hi*lo per product, ~2^-22) -- G is multiplied by g_scale before its f16
                           hi / lo split (the largest |G| belongs near 2^9 .. 2^10: DirEngine.calibrate; values saturate, never inf / nan)      */
} dir_bone_fusion_params;
size_t dir_bone_fusion_scratch_bytes(int B);
int dir_bone_fusion_prepare(const dir_bone_fusion_params* params_host, const float* emb, void* scratch, int B, void* stream);
int dir_bone_fusion_forward(const dir_bone_fusion_params* params_host, const float* uv_left, const float* uv_right,
                            const void* scratch, void* y, int B, int S, float distance, int out_cstride, int out_coff,
                            int relu, void* stream);
/* Backward of the factorised form for the training step (reference: torch autograd through models/dir.py:132-174 and the
 * first convolution of `fusion`, models/dir.py:57-62, run by train.py:66-70).  Forward of the training path = dir_bone_fusion_prepare +
 * dir_bone_fusion_forward with exact_f32 = 1, g_scale = 0, scale = NULL, shift = fusion.0.bias, relu = 0 (the raw convolution; BatchNorm
 * with batch statistics follows as its own step).  From gy [B,S,S,256] (NHWC fp32, contiguous):
 *   g_w_g  [9][40][64][256]   gradient of w_g (the caller permutes it back to fusion.0.weight's OIHW);
 *   g_emb  [B,42,64]          gradient of the re-embedded joint features (both the G path and index_select's backward);
 *   g_uv_* [B,21,2]           gradient of the stage's joint uv through the bone weights (optional, NULL to skip);
 * g_scratch = the G that dir_bone_fusion_prepare(exact_f32 = 1) wrote for this batch.  No [B,S,S,2560] map or gradient map exists:
 * four groups of exact-fp32 GEMMs (dir_gemm_f32) over zero-bordered pixel grids in which a tap is a pointer shift (csrc/bonefuse_bwd.hip).
 * Deterministic.  workspace: dir_bone_fusion_backward_workspace_bytes(B, S) bytes, 16-byte aligned. */
long long dir_bone_fusion_backward_workspace_bytes(int B, int S);
int dir_bone_fusion_backward(const float* w_g, const float* emb, const float* uv_left, const float* uv_right, const void* g_scratch,
                             const float* gy, float distance, float* g_w_g, float* g_emb, float* g_uv_left, float* g_uv_right,
                             void* workspace, long long workspace_bytes, int B, int S, void* stream);

/* ---- evaluation-metric maths of apps/eval.py --------------------------------------------------
 * f1a: Jr.__call__ (apps/eval.py:43-44): joints[B,21,3] = jr[21,778] @ verts[B,778,3].  `jr` is the Jr-processed
 * regressor (16 MANO rows + 5 one-hot fingertip rows, re-ordered; built on the host by dir_amd.apps.eval.Jr).
 * Dot products accumulate in fp64 and round once to fp32. */
int dir_joint_regress_forward(const float* jr, const float* verts, float* joints, int B, void* stream);

/* f1b: the per-batch body of the eval loop, apps/eval.py:151-241, for both hands (index 0 = left, 1 = right).
 * Inputs (device, fp32, contiguous): verts_pd[h] [B,778,3] = result[-1]['pd_mesh_xyz_*'] (:171-172), pd_offset [B,3]
 * (:170, multiplied by 0.15 inside), verts_gt[h] [B,778,3] = data[3]/data[5] (camera space), verts2d_gt[h] [B,778,2] =
 * data[7]/data[9], cam [B,3,3] = data[10], jr[h] [21,778].
 * Outputs (any may be NULL): joint_err/joint2d_err [B,2,21], vert_err/vert2d_err [B,2,778] (:192,204,217,225: L2 norms,
 * metres / pixels), joints_pd/joints_gt [B,2,21,3] (:195-196: aligned prediction, root-relative GT), root_err [B]
 * (:233-239).  root_joint = opt.root_joint (0 wrist | 9 middle MCP), use_scale = opt.scale (:180-185). */
typedef struct dir_eval_inputs {
    const float* verts_pd[2];
    const float* pd_offset;
    const float* verts_gt[2];
    const float* verts2d_gt[2];
    const float* cam;
    const float* jr[2];
} dir_eval_inputs;
typedef struct dir_eval_outputs {
    float *joint_err, *vert_err, *joint2d_err, *vert2d_err, *joints_pd, *joints_gt, *root_err;
} dir_eval_outputs;
int dir_eval_metrics_forward(const dir_eval_inputs* in_host, const dir_eval_outputs* out_host, int B, int root_joint,
                             int use_scale, void* stream);

/* f1c: the ground-truth MANO layer, models/manolayer.py:251-323 (ManoLayer.forward; rodrigues_batch :32-48), the
 * formulation dataset/interhand.py:130-149 uses to synthesise GT.  Tables in the dir_mano_tables packing (comps = the full
 * [45][45] hands_components, the first `ncomps` rows are used; side / root_palm are ignored: fingertip vertices are
 * 745,317,444,556,673 for both hands, models/manolayer.py:297).  root_rotation [B,3,3]; pose [B,ncomps] PCA coefficients
 * (use_pca) or, with ncomps = 0, [B,15,3,3] rotation matrices; shape [B,10]; trans [B,3] / scale [B] optional (NULL);
 * center_idx -1 = None; verts [B,778,3], joints [B,21,3] (metres). */
int dir_gt_mano_forward(const dir_mano_tables* tables_host, const float* root_rotation, const float* pose, int ncomps,
                        const float* shape, const float* trans, const float* scale, int center_idx, int new_skel,
                        float* verts, float* joints, int B, void* stream);

/* f3 (tensor side of the input pipeline): apps/eval.py:59-61 == dataset/interhand.py:223-225 --
 * uint8 BGR HWC [B,H,W,3] (what cv.imread / cv.resize hand over) -> RGB, / 255, (t - mean) / std -> fp32 NCHW [B,3,H,W],
 * in the reference's fp32 operation order (bit-identical to the torch CPU result).  mean / std: host pointers to 3 floats
 * (ImageNet: 0.485 0.456 0.406 / 0.229 0.224 0.225).  JPEG decode and resize stay on the host. */
int dir_image_normalize_forward(const uint8_t* img_bgr_hwc, float* out_nchw, const float* mean_host, const float* std_host,
                                int B, int H, int W, void* stream);
/* the same arithmetic fused into the space-to-depth stem staging (== dir_image_normalize_forward + dir_stem_prep_s2d,
 * bit for bit, without the fp32 image in HBM) */
int dir_stem_prep_s2d_u8(const uint8_t* img_bgr_hwc, void* out, const float* mean_host, const float* std_host, int B, int H,
                         int W, int Hs, int Ws, int dtype, void* stream);

/* a1 (stem), bf16 mode: models/backbone/resnet.py:244-247 -- conv1 7x7/s2/p3 (3->64) + bn1 (eval, folded scale / shift) + ReLU +
 * MaxPool2d(3,2,1) in ONE launch; with uint8 input also apps/eval.py:59-61 (== dir_image_normalize_forward, bit for bit).
 * img: DIR_DT_F32 -> fp32 NCHW [B,3,H,W] (normalised), DIR_DT_U8 -> uint8 BGR HWC [B,H,W,3] (mean / std: host pointers to 3
 * floats, else ignored).  w_packed: bf16 [64][7 ky][8 kx][4 c] = conv1.weight[n][c][ky][kx], zero for kx = 7 and c = 3.
 * y: bf16 NHWC [B,H/4,W/4,64].  H, W multiples of 4.  Same operand rounding as dir_stem_prep_s2d + dir_conv2d_forward +
 * dir_maxpool3x3s2 (bf16 operands, fp32 accumulation, bf16 conv output); only the summation order inside K differs. */
int dir_stem_pool_forward(const void* img, int img_dtype, const float* mean_host, const float* std_host, const void* w_packed,
                          const float* scale, const float* shift, void* y, int B, int H, int W, void* stream);
/* the same with the 16-bit storage kind of y and w_packed chosen by the caller: out_dtype DIR_DT_BF16 (what the entry point above passes) or
 * DIR_DT_F16 (round 5: f16 feature maps and weights, v_mfma_f32_16x16x32_f16) */
int dir_stem_pool_forward_dt(const void* img, int img_dtype, int out_dtype, const float* mean_host, const float* std_host, const void* w_packed,
                             const float* scale, const float* shift, void* y, int B, int H, int W, void* stream);

/* a1 (layer1 bottlenecks), bf16 mode: models/backbone/resnet.py:126-140 of block i -- conv2 3x3/s1/p1 (64->64) + bn2 + ReLU +
 * conv3 1x1 (64->256) + bn3 + identity + ReLU -- and, optionally, :122-124 of block i+1 -- conv1 1x1 (256->64) + bn1 + ReLU --
 * in one launch: the 64-channel intermediate and the next conv1's input never touch HBM (534 -> 334 MB per block at B = 64).
 * y1 [B,H,W,64] = ReLU(bn1(conv1(x))) of block i; residual [B,H,W,256] or NULL; out [B,H,W,256]; y1_next [B,H,W,n_next] or NULL
 * (then w1n / scale1n / shift1n are ignored); all bf16 NHWC.  H % 8 == 0, W % 16 == 0.  Weights bf16: w2 [64][3][3][64]
 * (dir_conv2d_forward's packing), w3 [256][64], w1n [64][256]; scale / shift = folded eval BatchNorm, fp32, device pointers.
 * Same rounding points as the unfused dir_conv2d_forward sequence (bf16 operands, fp32 accumulation, bf16 y2 / out). */
typedef struct dir_bneck_chain_params {
    const void* w2; const float* scale2; const float* shift2;
    const void* w3; const float* scale3; const float* shift3;
    const void* w1n; const float* scale1n; const float* shift1n;
    const void* wd;   /* projection shortcut (models/backbone/resnet.py:117-119), bf16 [256][64]: with x2 != NULL conv3's GEMM gets
                         64 more K from x2 [B,H,W,64] (the block input).  Both BatchNorm scales must then be folded into w3 / wd
                         rows, scale3 = 1 and shift3 = shift_bn3 + shift_bn_ds; `residual` must be NULL. */
    int32_t n_next;   /* output channels of the fused next conv1: 64 (next layer1 block) or 128 (layer2's first block); y1_next is
                         [B,H,W,n_next], w1n [n_next][256] */
    int32_t out_decimate; /* 0: out is [B,H,W,256].  1: only the pixels with even y and even x are written, as out [B,H/2,W/2,256] -- for the last
                         layer1 block, whose output is read by layer2's stride-2 projection shortcut alone (models/backbone/resnet.py:117-119;
                         its conv1 is the fused y1_next): three quarters of the 256-channel map never leave the CU.  The ResNet's c1 feature is
                         then not produced (models/dir.py never reads it: models/dir.py:437-483 use c2..c4) */
    int32_t dtype;    /* storage kind of every 16-bit tensor and weight of the call: 0 or DIR_DT_BF16 -> bf16, DIR_DT_F16 -> f16 (round 5) */
} dir_bneck_chain_params;
int dir_bottleneck_chain_forward(const dir_bneck_chain_params* p, const void* y1, const void* residual, const void* x2, void* out,
                                 void* y1_next, int B, int H, int W, void* stream);

/* a1, layer2 / layer3 geometry (planes P = 128 | 256): the 1x1 TAIL of a bottleneck and the 1x1 HEAD of the next one in one launch
 *   models/backbone/resnet.py:132-140   out = relu(bn3(conv3(y2)) + identity)            conv3: 1x1, P -> 4P
 *   models/backbone/resnet.py:122-124   y1_next = relu(bn1'(conv1'(out)))                conv1': 1x1, 4P -> n_next
 * bf16 NHWC: y2 [M][P], residual [M][4P] (the block input), out [M][4P], y1_next [M][n_next]; M = B*H*W pixels, a multiple of 64
 * (1x1 convolutions: the image geometry does not matter).  The block output is written once and never read back.
 * wstream: both weight matrices as bf16 MFMA A-operand fragments in the order the kernel's waves consume them (16 bytes per lane,
 * 1 KB per fragment), [4P/512 halves][8 waves][NBF + NCF fragments][64 lanes][8 bf16]:
 *   conv3 fragment f < NBF = P/8        channel block cb = f / (P/16), k-step ks = f % (P/16): lane l holds
 *                                       w3[half*512 + 64*wave + 32*cb + (l & 31)][16*ks + 8*(l >> 5) .. +8]
 *   conv1' fragment fc (n_next = 256)   NCF = 32: w1n[32*wave + (l & 31)][half*512 + 16*fc + 8*(l >> 5) .. +8]
 *   conv1' fragment fc (n_next = 128)   NCF = 16: w1n[16*wave + (l & 15)][half*512 + 32*fc + 8*(l >> 4) .. +8]
 * (dir_amd/engine.py::pack_tail_stream builds it).  scale / shift: the folded eval-mode BatchNorms, fp32. */
typedef struct dir_bneck_tail_params {
    const void* wstream;
    const float* scale3; const float* shift3;      /* [4P]     */
    const float* scale1n; const float* shift1n;    /* [n_next] */
    int32_t planes;   /* P: 128 | 256 */
    int32_t n_next;   /* 128 | 256 (P = 128), 256 (P = 256) */
    int32_t waves;    /* 8: the layout above (64-pixel tiles, one workgroup per CU).  4: the thin variant (32-pixel tiles, two
                         workgroups per CU, for M <~ 64 pixels x CUs): [halves][4 waves][P/4 + 32*(n_next/128) fragments][64][8] with
                         conv3 fragment f < P/4: cb = f / (P/16), ks = f % (P/16): w3[half*512 + 128*wave + 32*cb + (l&31)][16*ks + 8*(l>>5)..],
                         conv1' fragment fc: ks = fc / (n_next/128), cc = fc % (n_next/128):
                         w1n[(n_next/4)*wave + 32*cc + (l&31)][half*512 + 16*ks + 8*(l>>5)..] */
    int32_t dtype;    /* storage kind of the tensors and of the weight stream: 0 or DIR_DT_BF16 -> bf16, DIR_DT_F16 -> f16 (round 5) */
} dir_bneck_tail_params;
int dir_bottleneck_tail_forward(const dir_bneck_tail_params* p, const void* y2, const void* residual, void* out, void* y1_next,
                                long long M, void* stream);

/* a13 / 8f rank 2, forward half: the training objective, models/dir.py:542-594 (SmoothL1Loss models/loss.py:63-93, EdgeLengthLoss
 * :36-60, NormalVectorLoss :6-33, nn.CrossEntropyLoss(weight), lovasz_softmax models/lovasz_loss.py:155-202).  Forward values;
 * the gradients w.r.t. the predictions are dir_stage_losses_backward / dir_dense_losses_backward below.  All tensors fp32, device pointers, index 0 = left hand, 1 = right hand. */
typedef struct dir_loss_pred {      /* one entry of iter_outs (models/dir.py:519,571) */
    const float* joint_uv[2];       /* pd_joint_uv_*  [B,21,2] */
    const float* mesh_uv[2];        /* pd_mesh_uv_*   [B,778,2] (dir_mano_forward's optional output), or NULL: then computed here */
    const float* proj[2];           /* pd_proj_* [B,3] = (scale, tx, ty): mesh_uv = scale * mesh_xyz[..., :2] + t (utils/utils.py:47-63,
                                       models/dir.py:278-280,359-361); read only where mesh_uv is NULL */
    const float* joint_xyz[2];      /* pd_joint_xyz_* [B,21,3] metres */
    const float* mesh_xyz[2];       /* pd_mesh_xyz_*  [B,778,3] metres */
    const float* offset;            /* pd_offset      [B,3] */
} dir_loss_pred;
typedef struct dir_loss_target {    /* target / meta_info of models/dir.py:543-554 */
    const float* joint_2d[2];       /* joint_2d_* [B,21,c2]: the first two columns are used (:572) */
    const float* mesh_2d[2];        /* mesh_2d_*  [B,778,c2] */
    const float* joint_3d[2];       /* joint_3d_* [B,21,3] metres */
    const float* mesh_3d[2];        /* mesh_3d_*  [B,778,3] metres */
    const float* center[2];         /* center_*   [B,3] */
    const int32_t* faces[2];        /* ManoLayer.th_faces [n_faces,3] */
    int32_t c2, n_faces;
} dir_loss_target;
/* The 13 terms of one stage (models/dir.py:571-592) in this order: joint_left_uv, joint_right_uv, mesh_left_uv, mesh_right_uv,
 * joint_left_xyz, joint_right_xyz, mesh_left_xyz, mesh_right_xyz, edge_left, edge_right, normal_left (x0.1), normal_right (x0.1),
 * offset; SmoothL1 terms x coord_weight (the reference's 10).  scratch: B*13 doubles (device); out13: 13 floats (device). */
int dir_stage_losses_forward(const dir_loss_pred* pred_host, const dir_loss_target* gt_host, float coord_weight, double* scratch,
                             float* out13, int B, void* stream);
/* models/dir.py:562-569: seg (weighted cross entropy x0.1), dense (SmoothL1), lovasz (x0.1), each x dense_weight -> out3.
 * seg_logits / dense_pred [B,3,S,S]; gt_seg [B,1,H,W] labels 0..2 stored as floats (F.interpolate nearest -> .long());
 * gt_dense [B,3,H,W] (F.interpolate bilinear).  class_weight_host: 3 floats (0.1, 0.45, 0.45).  workspace: device bytes,
 * dir_dense_losses_workspace_bytes(B, S) of them (sort keys / values, both halves of the radix sort's ping-pong, and its digit histograms). */
long long dir_dense_losses_workspace_bytes(int B, int S);
int dir_dense_losses_forward(const float* seg_logits, const float* dense_pred, const float* gt_seg, const float* gt_dense,
                             const float* class_weight_host, float dense_weight, void* workspace, long long workspace_bytes,
                             float* out3, int B, int S, int H, int W, void* stream);

/* 8f rank 2, optimiser: train.py:227 `optim.AdamW(params, lr)` (torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01) as one
 * launch over flat fp32 buffers of n elements (16-byte aligned): decoupled weight decay, bias-corrected moments, torch's fp32
 * operation order.  step = 1-based update count (the value torch keeps in state['step'] AFTER this update). */
int dir_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, long long step, void* stream);

/* a13 backward, first step of the training backward pass (8f rank 2): gradients of (sum_k grad_out[k] * term_k) w.r.t. the predictions
 * the terms read -- what autograd gives through models/loss.py / models/lovasz_loss.py / nn.CrossEntropyLoss as models/dir.py:562-592
 * composes them.  grad_out: device pointer to the upstream gradients of the 13 (3) terms in forward order, or NULL for ones.
 * pd_mesh_uv is an independent input here (pred->mesh_uv must be given; the projection's chain rule is the caller's).
 * vert_face_offsets[h] [779] / vert_face_index[h] [3 n_faces]: CSR lists vertex -> (face * 3 + corner) of faces[h] (int32, device):
 * vertex gradients are summed in that order, without atomics (deterministic). */
typedef struct dir_loss_pred_grad {
    float* joint_uv[2]; float* mesh_uv[2]; float* joint_xyz[2]; float* mesh_xyz[2]; float* offset;     /* same shapes as dir_loss_pred */
} dir_loss_pred_grad;
int dir_stage_losses_backward(const dir_loss_pred* pred_host, const dir_loss_target* gt_host, float coord_weight, const float* grad_out13,
                              const int32_t* const* vert_face_offsets, const int32_t* const* vert_face_index,
                              const dir_loss_pred_grad* grads_host, int B, void* stream);
long long dir_dense_losses_backward_workspace_bytes(int B, int S);
/* grad_seg / grad_dense: [B,3,S,S] fp32 */
int dir_dense_losses_backward(const float* seg_logits, const float* dense_pred, const float* gt_seg, const float* gt_dense,
                              const float* class_weight_host, float dense_weight, const float* grad_out3, void* workspace,
                              long long workspace_bytes, float* grad_seg, float* grad_dense, int B, int S, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * f3, from files: the GPU half of the JPEG decode (round 5).  cv.imread of the reference's input pipeline (apps/eval.py:56,
 * dataset/interhand.py:223 over dataset/prepare_data.py:123-166's files) = libjpeg's default decode.  The host decodes the Huffman stream only
 * (include/dir_jpeg.h: dir_jpeg_decode_coefficients -> one record per image = dir_jpeg_header + quantised int16 coefficients); this entry point
 * takes a batch of records that sit `record_stride` bytes apart in device memory and writes the uint8 BGR frames [B,H,W,3] that
 * dir_stem_pool_forward(_dt) / dir_stem_prep_s2d_u8 / dir_image_normalize_forward read: dequantisation, the "islow" integer IDCT with its range
 * limit, "fancy" h2v2 / h2v1 chroma upsampling, 16-bit fixed-point YCbCr -> RGB -- integer arithmetic, bit-exact with libjpeg(-turbo)
 * (oracle/jpeg.py, pinned to Pillow's decoder).  Every record must describe an H x W image (4:2:0, 4:2:2, 4:4:4 or grayscale); a record that
 * does not sets *err_flag (device int32, zeroed by the caller) to 1 + its index and its frame is left untouched.
 * planes_scratch: B x round_up_16(dir_jpeg_planes_bytes(record_stride)) bytes of device memory. */
long long dir_jpeg_planes_bytes(long long record_bytes);
int dir_jpeg_decode_records(const void* records, long long record_stride, int B, int H, int W, void* planes_scratch, long long scratch_bytes, void* out_bgr,
                            int32_t* err_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIR_HIP_H */
