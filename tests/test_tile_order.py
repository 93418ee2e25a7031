"""conv_common.h: choose_tile_order -- the per-launch choice between row-major and column-major tile order by unique operand bytes per XCD --
probed on the host (a tiny program compiled with hipcc; nothing runs on a device): the shapes of DIR.forward at B = 64 that must go
column-major (weights outweigh activations) and those that must not."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(rows, tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'tile_order_host')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'dir_amd', 'csrc'),
                           '-o', exe, os.path.join(ROOT, 'tests', 'helpers', 'tile_order_host.hip')], stderr=subprocess.DEVNULL)
    env = dict(os.environ)
    env.pop('DIR_TILE_ORDER', None)
    r = subprocess.run([exe], input='\n'.join(' '.join(str(v) for v in row) for row in rows) + '\n', capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 0, r.stderr
    return [int(x) for x in r.stdout.split()]


def test_tile_order_choice_for_the_forward_shapes(tmp_path):
    rows = [
        # B  H   W   Cin   Cout  K      tiles_m tiles_n es      (bf16: es = 2)
        (64, 8, 8, 2048, 2048, 18432, 16, 16, 2),       # attention conv, 256 x 128 tiles: 75 MB of weights against 17 MB of activations -> column-major
        (64, 8, 8, 512, 512, 4608, 16, 4, 2),           # layer4 3x3 on 256 x 128 tiles (halo reuse): column-major
        (64, 32, 32, 256, 256, 2304, 256, 1, 2),        # conv_final on the 256 x 256 tile: one column, nothing to choose -> row-major
        (64, 64, 64, 64, 256, 64, 2048, 2, 2),          # layer1 1x1: activations dominate -> row-major
        (64, 16, 16, 256, 1024, 256, 128, 8, 2),        # layer3 conv3: row-major
        (64, 8, 8, 2048, 2048, 18432, 16, 16, 4),       # the same attention conv with fp32-sized operands (split precision): column-major
    ]
    assert _probe(rows, tmp_path) == [1, 1, 0, 0, 0, 1]


def test_tile_order_switch_forces_an_order(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    exe = str(tmp_path / 'tile_order_host')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'dir_amd', 'csrc'),
                           '-o', exe, os.path.join(ROOT, 'tests', 'helpers', 'tile_order_host.hip')], stderr=subprocess.DEVNULL)
    line = '64 64 64 64 256 64 2048 2 2\n64 8 8 2048 2048 18432 16 16 2\n'
    for forced, want in (('0', [0, 0]), ('1', [1, 1])):
        r = subprocess.run([exe], input=line, capture_output=True, text=True, env=dict(os.environ, DIR_TILE_ORDER=forced), timeout=60)
        assert [int(x) for x in r.stdout.split()] == want
