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
