"""ctypes binding of libdir_hip.so (the C ABI declared in include/dir_hip.h).

There is NO fallback: if the library is missing or a call fails this raises.  torch is imported first so
that the HIP runtime PyTorch-ROCm already mapped (libamdhip64.so.7) is the one the library binds to; the
kernels then run on torch's current stream against torch-owned device memory.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DIR_LIB_PATH') or os.path.join(_HERE, 'lib', 'libdir_hip.so')      # DIR_LIB_PATH: investigation builds only (build.py)
_lib = None


class DirHipError(RuntimeError):
    pass


class ManoTables(C.Structure):
    _fields_ = [('shapedirs_t', C.c_void_p), ('posedirs_t', C.c_void_p), ('v_template', C.c_void_p),
                ('j_template', C.c_void_p), ('j_shapedirs', C.c_void_p), ('weights', C.c_void_p), ('hands_mean', C.c_void_p),
                ('comps', C.c_void_p), ('side', C.c_int32), ('center_idx', C.c_int32), ('root_palm', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'B', 'H', 'W', 'Cin', 'in_cstride', 'in_coff', 'Cout', 'out_cstride', 'out_coff', 'res_cstride', 'res_coff',
        'kh', 'kw', 'stride', 'pad', 'in_dtype', 'out_dtype', 'flags', 'Ho', 'Wo')] + [('in_scale', C.c_float), ('out_split_scale', C.c_float)]


class ConvBnBwd(C.Structure):          # dir_conv_bn_bwd
    _fields_ = [(n, C.c_void_p) for n in ('z', 'mean', 'rstd', 'w', 'b')] + [('relu', C.c_int32)] + [(n, C.c_void_p) for n in ('p1', 'p2')]


class ConvSrc2(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('H', 'W', 'Cin', 'in_cstride', 'in_coff', 'stride')]


class TrainWeight(C.Structure):          # dir_train_weight
    _fields_ = [(n, C.c_void_p) for n in ('w', 'fwd', 'fwd_scale', 'dgrad', 'dgrad_scale')] + [(n, C.c_int32) for n in ('Cout', 'Cin', 'kh', 'kw')] + [
        ('fwd_inv_in', C.c_float), ('dgrad_inv_in', C.c_float)]


class BneckChainParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w2', 'scale2', 'shift2', 'w3', 'scale3', 'shift3', 'w1n', 'scale1n', 'shift1n', 'wd')] + [
        ('n_next', C.c_int32), ('out_decimate', C.c_int32), ('dtype', C.c_int32)]


class BneckTailParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('wstream', 'scale3', 'shift3', 'scale1n', 'shift1n')] + [('planes', C.c_int32), ('n_next', C.c_int32), ('waves', C.c_int32), ('dtype', C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('M', 'N', 'K', 'lda', 'ldb', 'ldc', 'trans_a', 'trans_b', 'accumulate', 'batch')] + \
               [(n, C.c_int64) for n in ('stride_a', 'stride_b', 'stride_c')]


class GemmGroups(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('ny', 'nx', 'reduce', 'reserved')] + [(n, C.c_int64) for n in ('a_y', 'a_x', 'b_y', 'b_x', 'c_y', 'c_x')]


class TokenMlp(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w1t', 's1', 'b1', 'w2t', 'b2')]


class PgcnLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('W', 'e1', 'bias', 'bn_scale', 'bn_shift')] + [('relu', C.c_int32), ('w_dtype', C.c_int32)]


class SteBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('ln1_w', 'ln1_b', 'qkv_wt', 'qkv_b', 'proj_wt', 'proj_b', 'ln2_w', 'ln2_b',
                                          'fc1_wt', 'fc1_b', 'fc2_wt', 'fc2_b')]


class SteParams(C.Structure):
    _fields_ = [('pos_embed', C.c_void_p), ('blocks', SteBlock * 3), ('num_blocks', C.c_int32)] + \
               [(n, C.c_void_p) for n in ('snorm_w', 'snorm_b', 'head_ln_w', 'head_ln_b', 'head_wt', 'head_b')] + \
               [('weight_dtype', C.c_int32)]


class RegressParams(C.Structure):
    _fields_ = [('mano_wt', C.c_void_p), ('mano_b', C.c_void_p * 2), ('off_w', C.c_void_p), ('off_b', C.c_void_p),
                ('emb', TokenMlp)]


class InitHeadParams(C.Structure):
    _fields_ = [('attn_w', C.c_void_p * 2), ('attn_b', C.c_float * 2), ('mano_wt', C.c_void_p),
                ('mano_b', C.c_void_p * 2), ('off_w', C.c_void_p), ('off_b', C.c_void_p)]


class BoneFusionParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w_g', 'scale', 'shift')] + [('exact_f32', C.c_int32), ('g_scale', C.c_float)]


class EvalInputs(C.Structure):
    _fields_ = [('verts_pd', C.c_void_p * 2), ('pd_offset', C.c_void_p), ('verts_gt', C.c_void_p * 2),
                ('verts2d_gt', C.c_void_p * 2), ('cam', C.c_void_p), ('jr', C.c_void_p * 2)]


class LossPred(C.Structure):
    _fields_ = [('joint_uv', C.c_void_p * 2), ('mesh_uv', C.c_void_p * 2), ('proj', C.c_void_p * 2), ('joint_xyz', C.c_void_p * 2),
                ('mesh_xyz', C.c_void_p * 2), ('offset', C.c_void_p)]


class LossTarget(C.Structure):
    _fields_ = [('joint_2d', C.c_void_p * 2), ('mesh_2d', C.c_void_p * 2), ('joint_3d', C.c_void_p * 2),
                ('mesh_3d', C.c_void_p * 2), ('center', C.c_void_p * 2), ('faces', C.c_void_p * 2),
                ('c2', C.c_int32), ('n_faces', C.c_int32)]


class LossPredGrad(C.Structure):
    _fields_ = [('joint_uv', C.c_void_p * 2), ('mesh_uv', C.c_void_p * 2), ('joint_xyz', C.c_void_p * 2),
                ('mesh_xyz', C.c_void_p * 2), ('offset', C.c_void_p)]


class EvalOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('joint_err', 'vert_err', 'joint2d_err', 'vert2d_err', 'joints_pd', 'joints_gt',
                                          'root_err')]


ABI_VERSION = 37          # DIR_ABI_VERSION (include/dir_hip.h)
DT_F32, DT_BF16, DT_F16X3, DT_F16X1, DT_F16X3P, DT_F16X1P, DT_F16 = 0, 1, 3, 4, 5, 6, 7      # DT_F16: f16 STORAGE (round 5)
CONV_RELU, CONV_PRE_RELU = 1, 2

_p, _i = C.c_void_p, C.c_int
_SIGNATURES = {
    'dir_abi_version': (C.c_int, []),
    'dir_last_error': (C.c_char_p, []),
    'dir_device_info': (C.c_int, [C.c_char_p, _i, C.POINTER(C.c_int)]),
    'dir_launch_log_reset': (None, []),
    'dir_launch_log_get': (C.c_int, [C.c_char_p, _i]),
    'dir_launch_log_note': (None, [C.c_char_p, C.c_longlong]),
    'dir_probe_launch': (C.c_longlong, [_i, _p, C.c_longlong, _i, _p]),
    'dir_conv2d_forward': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'dir_conv2d_forward_stats': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(C.c_int), _p]),
    'dir_conv2d_forward_masked': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'dir_conv2d_forward_ex': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(ConvBnBwd), C.POINTER(C.c_int), _p]),
    'dir_add_upsampled': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'dir_fuse_sum': (C.c_int, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'dir_pack_f16x3_weights': (C.c_int, [_p, _p, _p, _p, _i, _i, _p]),
    'dir_train_pack_conv_weights': (C.c_int, [_p, _p, _i, _i, _p]),
    'dir_gemm_f32_splitk_workspace_bytes': (C.c_longlong, [C.POINTER(GemmDesc)]),
    'dir_gemm_f32_splitk': (C.c_int, [C.POINTER(GemmDesc), _p, _p, _p, _p, _p, C.c_longlong, _p]),
    'dir_axpy_multi_f32': (C.c_int, [_p, _p, _p, _i, C.c_float, _p]),
    'dir_split_f16_forward': (C.c_int, [_p, _p, C.c_longlong, _i, _i, _i, _p, _p, _i, C.c_float, _i, _p]),
    'dir_stem_prep': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'dir_stem_prep_s2d': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'dir_image_normalize_forward': (C.c_int, [_p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _i, _i, _p]),
    'dir_stem_prep_s2d_u8': (C.c_int, [_p, _p, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _i, _i, _i, _i, _i, _p]),
    'dir_adamw_step': (C.c_int, [_p, _p, _p, _p, C.c_longlong, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_longlong, _p]),
    'dir_stage_losses_backward': (C.c_int, [C.POINTER(LossPred), C.POINTER(LossTarget), C.c_float, _p, C.POINTER(C.c_void_p * 2),
                                            C.POINTER(C.c_void_p * 2), C.POINTER(LossPredGrad), _i, _p]),
    'dir_dense_losses_backward_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_dense_losses_backward': (C.c_int, [_p, _p, _p, _p, C.POINTER(C.c_float), C.c_float, _p, _p, C.c_longlong, _p, _p, _i, _i, _i, _i, _p]),
    'dir_stage_losses_forward': (C.c_int, [C.POINTER(LossPred), C.POINTER(LossTarget), C.c_float, _p, _p, _i, _p]),
    'dir_dense_losses_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_dense_losses_forward': (C.c_int, [_p, _p, _p, _p, C.POINTER(C.c_float), C.c_float, _p, C.c_longlong, _p, _i, _i, _i, _i, _p]),
    'dir_bottleneck_chain_forward': (C.c_int, [C.POINTER(BneckChainParams), _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'dir_bottleneck_tail_forward': (C.c_int, [C.POINTER(BneckTailParams), _p, _p, _p, _p, C.c_longlong, _p]),
    'dir_stem_pool_forward': (C.c_int, [_p, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _p, _p, _p, _i, _i, _i, _p]),
    'dir_stem_pool_forward_dt': (C.c_int, [_p, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _p, _p, _p, _i, _i, _i, _p]),
    'dir_maxpool3x3s2': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'dir_upsample2x_bilinear': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'dir_init_head_forward': (C.c_int, [C.POINTER(InitHeadParams), _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'dir_bone_proj_forward': (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, C.c_float, _i, _p]),
    'dir_conv2d_dual_forward': (C.c_int, [C.POINTER(ConvDesc), _p, C.POINTER(ConvSrc2), _p, _p, _p, _p, _p]),
    'dir_conv2d_dual_scaled_forward': (C.c_int, [C.POINTER(ConvDesc), _p, C.POINTER(ConvSrc2), _p, _p, _p, _p, _p, _p]),
    'dir_conv1x1_stream_forward': (C.c_int, [C.POINTER(ConvDesc), _p, C.POINTER(ConvSrc2), _p, _p, _p, _p, _p, _p, _p, _p]),
    'dir_conv2d_as_supported': (C.c_int, [C.POINTER(ConvDesc), _i, _i]),
    'dir_conv2d_as_forward': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    'dir_conv2d_sparse_forward': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p]),
    'dir_grid_tokens_forward': (C.c_int, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, C.POINTER(TokenMlp),
                                          C.POINTER(TokenMlp), C.POINTER(TokenMlp), _p, _p, _i, _p]),
    'dir_pgcn_stack_forward': (C.c_int, [C.POINTER(PgcnLayer), _i, _p, _p, _p, C.c_longlong, _p, _i, _p]),
    'dir_pgcn_stack_forward_pair': (C.c_int, [C.POINTER(PgcnLayer), C.POINTER(PgcnLayer), _i, _p, _p, _p, _p, _i, _p]),
    'dir_pgcn_fused_sync_bytes': (C.c_longlong, []),
    'dir_pgcn_stack_forward_fused': (C.c_int, [C.POINTER(PgcnLayer), C.POINTER(PgcnLayer), _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    'dir_ste_forward': (C.c_int, [C.POINTER(SteParams), _p, _p, _p, _i, _p]),
    'dir_regress_forward': (C.c_int, [C.POINTER(RegressParams), _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'dir_bone_fusion_scratch_bytes': (C.c_size_t, [_i]),
    'dir_bone_fusion_prepare': (C.c_int, [C.POINTER(BoneFusionParams), _p, _p, _i, _p]),
    'dir_bone_fusion_forward': (C.c_int, [C.POINTER(BoneFusionParams), _p, _p, _p, _p, _i, _i, C.c_float, _i, _i, _i, _p]),
    'dir_bone_fusion_backward_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_bone_fusion_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, C.c_float, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _p]),
    'dir_gt_mano_forward': (C.c_int, [C.POINTER(ManoTables), _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    'dir_joint_regress_forward': (C.c_int, [_p, _p, _p, _i, _p]),
    'dir_eval_metrics_forward': (C.c_int, [C.POINTER(EvalInputs), C.POINTER(EvalOutputs), _i, _i, _i, _p]),
    'dir_mano_forward_pair': (C.c_int, [C.POINTER(ManoTables), _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p]),
    'dir_mano_backward_pair': (C.c_int, [C.POINTER(ManoTables), _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    'dir_regress_backward': (C.c_int, [_p] * 17 + [_i, _p]),
    'dir_gemm_f32': (C.c_int, [C.POINTER(GemmDesc), _p, _p, _p, _p, _p]),
    'dir_gemm_f32_grouped': (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmGroups), _p, _p, _p, _p, _p]),
    'dir_colsum_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_colsum_f32': (C.c_int, [_p, _p, _i, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_layernorm_forward': (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, C.c_float, _p]),
    'dir_layernorm_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'dir_gelu_forward': (C.c_int, [_p, _p, C.c_longlong, _p]),
    'dir_gelu_backward': (C.c_int, [_p, _p, _p, C.c_longlong, _p]),
    'dir_attention_forward': (C.c_int, [_p, _p, _p, _i, _i, _i, C.c_float, _p]),
    'dir_attention_backward': (C.c_int, [_p, _p, _p, _p, _i, _i, _i, C.c_float, _p]),
    'dir_bn_train_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_bn_train_forward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, C.c_float, C.c_float, _i, _p, _p, C.c_longlong, _p]),
    'dir_bn_train_backward_from_partials': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_bn_train_stats': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, C.c_float, C.c_float, _p, C.c_longlong, _p]),
    'dir_bn_train_stats_from_partials': (C.c_int, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, C.c_float, C.c_float, _p]),
    'dir_bn_train_apply': (C.c_int, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    'dir_bn_train_forward_from_partials': (C.c_int, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, C.c_float, C.c_float, _i, _p, _p]),
    'dir_bn_one_launch_status': (C.c_int, []),
    'dir_upsample_nearest_add_f32': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'dir_upsample_nearest_backward_f32': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _p]),
    'dir_bn_one_launch_enable': (C.c_int, [_i]),
    'dir_bn_train_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_bn_frozen_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_bn_sync_workspace_bytes': (C.c_longlong, [_i, _i]),
    'dir_bn_sync_local_stats': (C.c_int, [_p, _p, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_bn_sync_combine': (C.c_int, [_p, _i, _i, _p, _p, _p, _p, C.c_float, _p]),
    'dir_bn_sync_backward_sums': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_bn_sync_backward_apply': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, C.c_float, _i, _i, _i, _p]),
    'dir_bn_frozen_forward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, C.c_float, _i, _p, _p]),
    'dir_bn_frozen_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, C.c_longlong, _p]),
    'dir_relu_forward': (C.c_int, [_p, _p, C.c_longlong, _p]),
    'dir_relu_backward': (C.c_int, [_p, _p, _p, C.c_longlong, _p]),
    'dir_pgcn_adjacency_forward': (C.c_int, [_p, _p, _p]),
    'dir_pgcn_adjacency_backward': (C.c_int, [_p, _p, _p, _p, _p, _i, _p]),
    'dir_conv2d_splitk_workspace_bytes': (C.c_longlong, [C.POINTER(ConvDesc), _i]),
    'dir_conv2d_splitk_forward': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, C.c_longlong, _p]),
    'dir_conv2d_wgrad_workspace_bytes': (C.c_longlong, [C.POINTER(ConvDesc)]),
    'dir_conv2d_wgrad_f32': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _i, _p, C.c_longlong, _p]),
    'dir_conv2d_wgrad_f16x3_workspace_bytes': (C.c_longlong, [C.POINTER(ConvDesc)]),
    'dir_conv2d_wgrad_f16x3': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _i, _p, C.c_longlong, C.c_float, C.c_float, _p]),
    'dir_conv2d_wgrad_f16x3_pre': (C.c_int, [C.POINTER(ConvDesc), _p, _p, _p, _i, _p, C.c_longlong, C.c_float, C.c_float, _p, _p, _i, _p]),
    'dir_maxpool3x3s2_backward': (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'dir_upsample2x_bilinear_backward': (C.c_int, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'dir_attn_pool_forward': (C.c_int, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'dir_attn_pool_backward': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    'dir_bone_proj_backward_scratch_bytes': (C.c_longlong, [_i, _i]),
    'dir_bone_proj_backward': (C.c_int, [_p, _p, _p, _i, _i, C.c_float, _p, _p, _p, _i, _i, _i, _p]),
    'dir_axpy_f32': (C.c_int, [_p, _p, C.c_longlong, C.c_float, _p]),
    'dir_stage_positions': (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'dir_grid_rows_forward': (C.c_int, [_p, _p, _p, _i, _i, _i, _p]),
    'dir_grid_rows_backward': (C.c_int, [_p, _p, _i, _p, _i, _i, _i, _i, _p]),
    'dir_mano_forward': (C.c_int, [C.POINTER(ManoTables), _p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _p]),
    'dir_jpeg_planes_bytes': (C.c_longlong, [C.c_longlong]),
    'dir_jpeg_decode_records': (C.c_int, [_p, C.c_longlong, _i, _i, _i, _p, C.c_longlong, _p, _p, _p]),
}


# ---- live per-launch measurement (bench.py roofline, DirEngine.autotune).  With PROFILE set to a list, lib() hands out a proxy that
# brackets every entry point that launches kernels with HIP events on the current stream and appends one record per call:
#   {'api': entry point, 'kernels': "name,name" as rocprofv3 reports them (dir_launch_log_get), 'e0' / 'e1': events,
#    + whatever the caller announced for this call with annotate(): 'flops', 'bytes' (algorithmic work), 'shape', 'op'}
PROFILE = None
_pending = {}
_NO_PROFILE = ('dir_conv2d_as_supported', 'dir_abi_version', 'dir_bn_one_launch_status', 'dir_bn_one_launch_enable', 'dir_last_error', 'dir_device_info', 'dir_launch_log_reset', 'dir_launch_log_get', 'dir_launch_log_note',
               'dir_bone_fusion_scratch_bytes', 'dir_dense_losses_workspace_bytes', 'dir_dense_losses_backward_workspace_bytes',
               'dir_gemm_f32_splitk_workspace_bytes', 'dir_bn_train_workspace_bytes', 'dir_bn_sync_workspace_bytes', 'dir_bn_frozen_workspace_bytes', 'dir_jpeg_planes_bytes', 'dir_colsum_workspace_bytes', 'dir_conv2d_wgrad_workspace_bytes', 'dir_conv2d_wgrad_f16x3_workspace_bytes')


def annotate(**kw):
    """algorithmic work / label of the NEXT library call (ignored unless PROFILE is a list)"""
    if PROFILE is not None:
        _pending.update(kw)


class _ProfLib(object):
    def __init__(self, l):
        self._l = l

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        if name in _NO_PROFILE:
            return fn
        import torch

        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._l.dir_launch_log_reset()
            e0.record()
            rc = fn(*args)
            e1.record()
            buf = C.create_string_buffer(1024)
            self._l.dir_launch_log_get(buf, 1024)
            rec = dict(_pending, api=name, kernels=buf.value.decode(), e0=e0, e1=e1)
            _pending.clear()
            PROFILE.append(rec)
            return rc
        return call


def lib():
    global _lib
    if _lib is not None and PROFILE is not None:
        return _ProfLib(_lib)
    if _lib is None:
        import torch  # noqa: F401  (maps torch's libamdhip64 first)
        if not os.path.exists(LIB_PATH):
            raise DirHipError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              '(or python dir_amd/build.py). There is no CPU fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if l.dir_abi_version() != ABI_VERSION:
            raise DirHipError('libdir_hip.so ABI version %d != %d' % (l.dir_abi_version(), ABI_VERSION))
        _lib = l
    return _ProfLib(_lib) if PROFILE is not None else _lib


def check(rc, what):
    if rc != 0:
        raise DirHipError('%s failed (%d): %s' % (what, rc, lib().dir_last_error().decode()))


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DirHipError('dir_amd kernels run on the GPU only: got a %s tensor (no CPU fallback exists)'
                              % t.device)


def f32c(t):
    """contiguous float32 view/copy (plumbing only)."""
    import torch
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
