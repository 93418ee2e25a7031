"""Builds dir_amd/lib/libdir_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Objects are rebuilt only when their source (or a header) is newer.
The library links against libamdhip64.so.7 by SONAME only: inside a PyTorch-ROCm process the loader
reuses the runtime torch already mapped, so kernels run on torch's streams and device pointers.
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build', 'obj')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdir_hip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++20', '-fPIC', '-Wall', '-Wno-unused-function',
         '-fno-gpu-rdc',
         # No packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Measured on MI355X ("Packed FP32
         # beside another kernel"): a wave executing them while waves of certain OTHER kernels (e.g. the 64x128 ring variant of the
         # MFMA convolution) are resident on the same CU gets wrong LOW-half results -- never alone, never in LDS / VGPR / memory
         # canaries, only when two streams' kernels really overlap.  With the feature off the same overlap is bit-exact, at no
         # measurable cost (the bf16 epilogues keep v_cvt_pk_bf16_f32 / v_pk_max_i16, which are other features).
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
if os.environ.get('DIR_PACKED_FP32') == '1':
    # investigation aid (tools/pkfp32_repro.hip, tools/aggressor_test.py): the library WITH packed-FP32 instructions, built beside the
    # product library as lib/libdir_hip_pk.so from its own object directory; never loaded unless DIR_LIB_PATH points at it
    FLAGS = [f for f in FLAGS if f not in ('-Xclang', '-target-feature', '-packed-fp32-ops')] + ['-DDIR_INVESTIGATE_RING_64x128=1']
    OBJ = os.path.join(HERE, 'build', 'obj_pk')
    LIB = os.path.join(LIBDIR, 'libdir_hip_pk.so')
FLAGS += os.environ.get('DIR_HIPCC_EXTRA', '').split()       # tuning / debugging aid (changing it needs --force)
if os.environ.get('DIR_BUILD_TAG'):
    # A/B aid: a SECOND library beside the product one, from its own object directory, e.g.
    #   DIR_BUILD_TAG=noclamp DIR_HIPCC_EXTRA=-DDIR_F16_NOCLAMP=1 python -m dir_amd.build      ->  lib/libdir_hip_noclamp.so
    # never loaded unless DIR_LIB_PATH points at it
    OBJ = os.path.join(HERE, 'build', 'obj_' + os.environ['DIR_BUILD_TAG'])
    LIB = os.path.join(LIBDIR, 'libdir_hip_%s.so' % os.environ['DIR_BUILD_TAG'])


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')
    return exe


def _newest_header():
    hs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + '.o')
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_header()):
        return obj, False
    cmd = [hipcc()] + FLAGS + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    err = '\n'.join(l for l in r.stderr.splitlines() if 'is not a recognized feature for this target' not in l)   # the host pass of the feature flag
    if err.strip():
        sys.stderr.write(err + '\n')
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose:
            print('[dir_amd.build] linked %s (%d objects, %d recompiled)' % (LIB, len(objs), sum(c for _, c in res)))
    elif verbose:
        print('[dir_amd.build] %s up to date' % LIB)
    if not os.environ.get('DIR_BUILD_TAG'):
        build_jpeg_host(force, verbose)
    return LIB


JPEG_SRC = os.path.join(CSRC, 'jpeg_huff.c')
JPEG_LIB = os.path.join(LIBDIR, 'libdir_jpeg.so')


def build_jpeg_host(force=False, verbose=True):
    """lib/libdir_jpeg.so: the HOST half of the from-files path (csrc/jpeg_huff.c: baseline-JPEG entropy decode, plain C, no GPU runtime -- the
    decode worker processes load it), compiled with gcc"""
    os.makedirs(LIBDIR, exist_ok=True)
    hdr = os.path.join(HERE, '..', 'include', 'dir_jpeg.h')
    if not force and os.path.exists(JPEG_LIB) and os.path.getmtime(JPEG_LIB) > max(os.path.getmtime(JPEG_SRC), os.path.getmtime(hdr)):
        return JPEG_LIB
    cc = shutil.which('gcc') or shutil.which('cc') or hipcc()
    r = subprocess.run([cc, '-O3', '-std=c11', '-Wall', '-Wextra', '-fPIC', '-shared', JPEG_SRC, '-o', JPEG_LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('building libdir_jpeg.so failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('[dir_amd.build] built %s' % JPEG_LIB)
    return JPEG_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    build_jpeg_host(force='--force' in sys.argv)
