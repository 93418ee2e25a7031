"""Every convolution kernel variant, forced through the C ABI on the whole conv / token parity suite.

dir_conv2d_forward picks a kernel per layer (conv.hip 4-wave tiles; conv_pipe.hip 8-wave pipelined 256x128 | 128x128 |
256x64 tiles; the opt-in halo-reuse kernel for stride-1 3x3).  The automatic choice only exercises the shapes of the DIR
layers; here each variant is forced on ALL the parity cases (tails, strides, channel slices, residuals, sparse-K, tiny K)
by re-running the suites in a subprocess with the library's tuning environment variables."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ['tests/test_gpu_conv.py', 'tests/test_gpu_tokens.py']


def _run(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider'] + SUITES, cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, 'variant %s failed:\n%s\n%s' % (env, r.stdout[-3000:], r.stderr[-2000:])
    assert ' passed' in r.stdout


@pytest.mark.parametrize('shape', ['1', '2', '3'])
def test_pipelined_kernel_forced(shape):
    _run({'DIR_PIPE': shape, 'DIR_PIPE_MIN_NK': '1'})


@pytest.mark.parametrize('shape', ['1', '2', '3'])
def test_halo_reuse_kernel_forced(shape):
    _run({'DIR_PIPE': shape, 'DIR_PIPE_MIN_NK': '1', 'DIR_PATCH': '1'})


def test_four_wave_kernel_only():
    _run({'DIR_PIPE': '0'})


def test_small_tiles_forced():
    _run({'DIR_PIPE': '0', 'DIR_FORCE_M64': '1', 'DIR_FORCE_N64': '1', 'DIR_RING_MIN_NK': '1'})


@pytest.mark.parametrize('pipe', ['0', '1', '2'])
def test_column_major_tile_order_forced(pipe):
    """DIR_TILE_ORDER=1: every launch maps workgroups to tiles column-major (the order the library picks itself only where the weights outweigh
    the activations per XCD, conv_common.h: choose_tile_order) -- ragged tiles in both directions, channel slices, residuals, second sources."""
    _run({'DIR_TILE_ORDER': '1', 'DIR_PIPE': pipe, 'DIR_PIPE_MIN_NK': '1'})
