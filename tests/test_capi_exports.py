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
