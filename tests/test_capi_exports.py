"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/dir_hip.h declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'dir_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dir_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (same load order as the product path: torch's HIP runtime first)
    from dir_amd import build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert 'dir_mano_forward' in syms and len(syms) >= 4
    for s in syms:
        assert hasattr(lib, s), 'libdir_hip.so does not export ' + s
    lib.dir_abi_version.restype = ctypes.c_int
    hdr = open(os.path.join(ROOT, 'include', 'dir_hip.h')).read()
    declared = int(re.search(r'#define\s+DIR_ABI_VERSION\s+(\d+)', hdr).group(1))
    from dir_amd import _capi
    assert lib.dir_abi_version() == declared == _capi.ABI_VERSION      # header, library and ctypes binding agree


def test_binding_signatures_cover_header():
    from dir_amd import _capi
    assert sorted(_capi._SIGNATURES) == declared_symbols()


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from dir_amd import _capi
    from dir_amd.manopth.manolayer import ManoLayer
    m = ManoLayer(root_rot_mode='6D', use_pca=True, robust_rot=True, ncomps=45, center_idx=0, flat_hand_mean=False)
    with pytest.raises(_capi.DirHipError):
        m(torch.zeros(2, 51), torch.zeros(2, 10))
    with pytest.raises(NotImplementedError):
        ManoLayer(root_rot_mode='axisang', use_pca=True, ncomps=45)


def test_launch_log_counter_saturates_instead_of_overflowing():
    """the per-thread launch counter is only reset by profiling code; a serving process crosses 2^31 launches in days.  It
    must saturate (not wrap negative and index the 32-entry name table out of bounds)."""
    import torch  # noqa: F401
    from dir_amd import build
    lib = ctypes.CDLL(build.build(verbose=False))
    lib.dir_launch_log_note.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    lib.dir_launch_log_note.restype = None
    lib.dir_launch_log_get.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.dir_launch_log_get.restype = ctypes.c_int
    buf = ctypes.create_string_buffer(4096)
    lib.dir_launch_log_reset()
    lib.dir_launch_log_note(b'(alpha_kernel<1, 2>)', 3)
    assert lib.dir_launch_log_get(buf, 4096) == 3 and buf.value == b'alpha_kernel,alpha_kernel,alpha_kernel'
    lib.dir_launch_log_note(b'beta_kernel', 100)                      # past the 32-entry table: counted, names beyond 32 dropped
    assert lib.dir_launch_log_get(buf, 4096) == 103 and buf.value.count(b',') == 31
    lib.dir_launch_log_note(b'beta_kernel', 1 << 40)                  # far past INT_MAX
    n = lib.dir_launch_log_get(buf, 4096)
    assert n == 2 ** 31 - 1 and buf.value.count(b',') == 31
    lib.dir_launch_log_note(b'beta_kernel', 5)                        # stays saturated, table untouched
    assert lib.dir_launch_log_get(buf, 4096) == 2 ** 31 - 1
    lib.dir_launch_log_reset()
    assert lib.dir_launch_log_get(buf, 4096) == 0 and buf.value == b''


def test_bone_fusion_backward_host_side_contract():
    """round 4 entry points, host side only (no launch): the workspace size is a pure function of (B, S) that rejects nonsense, and the
    argument checks of dir_bone_fusion_backward / dir_gemm_f32_grouped fail with DIR_E_INVALID before anything touches a device"""
    import torch  # noqa: F401
    from dir_amd import _capi
    L = _capi.lib()
    assert L.dir_bone_fusion_backward_workspace_bytes(0, 32) == -1 and L.dir_bone_fusion_backward_workspace_bytes(4, 0) == -1
    n16, n32 = L.dir_bone_fusion_backward_workspace_bytes(32, 16), L.dir_bone_fusion_backward_workspace_bytes(32, 32)
    assert 0 < n16 < n32 < (1 << 30) and n32 % 256 == 0
    assert L.dir_bone_fusion_backward_workspace_bytes(64, 32) > n32
    # the padded grids alone: wgt [B][R][80] + gy [B][R][256] floats, R = (S+2)^2 + 2 (S+3)
    assert n32 >= 32 * ((34 * 34 + 70) * (80 + 256)) * 4
    rc = L.dir_bone_fusion_backward(None, None, None, None, None, None, 2.0, None, None, None, None, None, 0, 32, 32, None)
    assert rc != 0 and b'null pointer' in L.dir_last_error()
    d = _capi.GemmDesc(80, 64, 32, 80, 64, 64, 1, 0, 0, 70000, 0, 0, 0)
    g = _capi.GemmGroups(3, 3, 0, 0, 0, 0, 0, 0, 0, 0)
    one = ctypes.c_void_p(16)
    assert L.dir_gemm_f32_grouped(d, g, one, one, None, one, None) != 0 and b'groups' in L.dir_last_error()      # batch * groups > 65535


def test_round5_training_entry_points_host_side_contract():
    """round 5 entry points (BatchNorm split into statistics / apply, statistics and backward sums out of the convolution epilogues, the fused HRNet
    sum): argument checks fail with an error code and a message before anything touches a device"""
    import torch  # noqa: F401
    from dir_amd import _capi
    L = _capi.lib()
    one = ctypes.c_void_p(16)

    def bad(rc, word):
        assert rc != 0 and word in L.dir_last_error(), (rc, L.dir_last_error())
    bad(L.dir_fuse_sum(None, one, None, None, 0, 2, 8, 8, 64, 1, 1, None), b'bad args')
    bad(L.dir_fuse_sum(one, one, None, None, 5, 2, 8, 8, 64, 1, 1, None), b'at most 4')
    bad(L.dir_fuse_sum(one, one, None, None, 0, 2, 8, 8, 60, 1, 1, None), b'multiple of 8')
    # dir_bn_train_stats is for maps of more than 512 rows (smaller ones are one cooperative launch in dir_bn_train_forward)
    bad(L.dir_bn_train_stats(one, one, one, one, one, one, one, None, None, 512, 64, 64, 1e-5, 0.1, one, 1 << 20, None), b'512')
    bad(L.dir_bn_train_stats(one, one, one, one, one, one, one, None, None, 4096, 64, 64, 1e-5, 0.1, one, 8, None), b'workspace too small')
    bad(L.dir_bn_train_stats_from_partials(one, one, 128, 3, one, one, one, one, None, None, None, None, 4096, 64, 1e-5, 0.1, None), b'cap_rows')
    bad(L.dir_bn_train_stats_from_partials(one, one, 0, 64, one, one, one, one, None, None, None, None, 4096, 64, 1e-5, 0.1, None), b'bad arguments')
    bad(L.dir_bn_train_apply(one, one, one, one, one, one, 4096, 62, 62, 1, None, None), b'multiples of 4')
    bad(L.dir_bn_train_backward_from_partials(one, one, one, one, one, one, one, one, 32, one, one, one, 4096, 64, 64, 1, one, 8, None), b'workspace')
    d = _capi.ConvDesc(2, 16, 16, 64, 64, 0, 64, 64, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F32, _capi.DT_F32, 0, 0, 0, 1.0, 0.0)
    rows = ctypes.c_int(7)
    bad(L.dir_conv2d_forward_stats(d, one, one, None, None, None, None, one, None, one, ctypes.byref(rows), None), b'null pointer')
    dr = _capi.ConvDesc(2, 16, 16, 64, 64, 0, 64, 64, 0, 0, 0, 1, 1, 1, 0, _capi.DT_F32, _capi.DT_F32, 1, 0, 0, 1.0, 0.0)          # DIR_CONV_RELU: not the BatchNorm's input any more
    bad(L.dir_conv2d_forward_stats(dr, one, one, None, None, None, None, one, one, one, ctypes.byref(rows), None), b'without activation')
    bad(L.dir_conv2d_forward_masked(d, one, one, None, None, None, None, None, None, one, None), b'null mask')
    bb = _capi.ConvBnBwd(16, 16, 16, None, None, 1, 16, 16)
    bad(L.dir_conv2d_forward_ex(d, one, one, None, None, None, None, None, None, one, ctypes.byref(bb), None, None), b'chunk_rows')
    bad(L.dir_conv2d_wgrad_f16x3_pre(d, one, one, one, 0, None, 0, 1.0, 1.0, one, None, 1, None), b'go together')
