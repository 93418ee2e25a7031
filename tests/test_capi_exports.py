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
