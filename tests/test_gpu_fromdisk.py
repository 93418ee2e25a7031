"""GPU: the evaluation loop from files (dir_amd.apps.eval.evaluate_from_disk / main == apps/eval.py:88-306 of the reference) on a synthetic
split in the reference's on-disk layout: decode ring -> uint8 frames -> two forwards in flight -> GT MANO + metrics on the GPU must give
exactly what the plain one-batch-at-a-time path (DirEngine.forward on the normalised float image + EvalMetrics.update) gives."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers'))
from fake_split import write_split  # noqa: E402

from conftest import GOLDEN  # noqa: E402
from dir_amd import synth  # noqa: E402
from dir_amd.apps import dataset as DS  # noqa: E402
from dir_amd.apps import eval as EV  # noqa: E402
from dir_amd.engine import DirEngine  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def state():
    with open(os.path.join(GOLDEN, 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}


def test_evaluate_from_disk_equals_the_plain_loop(tmp_path, state):
    n, bs = 11, 4                                        # three batches, the last one ragged
    write_split(str(tmp_path), n, seed=3)
    eng = DirEngine(state, dtype=torch.bfloat16)
    mano = DS.gt_layers_from_checkpoint(state)
    jreg = {s: EV.Jr(mano[s].J_regressor) for s in ('left', 'right')}
    m, rate = EV.evaluate_from_disk(eng, str(tmp_path), jreg, mano, bs=bs, root_joint=0, scale=True, workers=2)
    assert rate['images'] == n and rate['images_per_sec'] > 0
    # the plain loop: decode on the host, normalise with the reference's three statements (oracle), one forward at a time
    from oracle import image_prep as IP
    ds = DS.InterHandSplit(str(tmp_path))
    ref = EV.EvalMetrics(jreg, 0, True, 3)
    for b0 in range(0, n, bs):
        idx = list(range(b0, min(n, b0 + bs)))
        frames = np.stack([ds.frame(i) for i in idx])
        x = torch.from_numpy(IP.normalize_u8_bgr(frames)).cuda()
        outs = eng.forward(x)
        annos = torch.from_numpy(np.stack([ds.anno(i) for i in idx])).cuda()
        ref.update([{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in outs[:3]], (None, None) + DS.gt_batch(mano, annos))
    a, b = m.arrays(), ref.arrays()
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k          # same kernels on the same pixels: bit-identical
    assert a['joints_loss_left'].shape == (n, 21) and np.isfinite(a['root_loss']).all()
    # the prepared uint8 split (no JPEG decode in the loop): the same frames, so the same numbers bit for bit
    DS.write_u8_shards(str(tmp_path), 'test', shard_size=4, workers=2)
    m8, rate8 = EV.evaluate_from_disk(eng, str(tmp_path), jreg, mano, bs=bs, root_joint=0, scale=True, workers=2, source='u8')
    a8 = m8.arrays()
    assert rate8['images'] == n and all(np.array_equal(a8[k], b[k]) for k in b)


def test_command_line(tmp_path, state, capsys):
    """python -m dir_amd.apps.eval --model --data_path --bs --root_joint (apps/eval.py:88-94): report + the twelve text files"""
    write_split(str(tmp_path / 'data'), 6, seed=5)
    ck = tmp_path / 'DIR.pth'
    torch.save({'net': state, 'last_epoch': 0}, str(ck))
    out = tmp_path / 'result'
    m = EV.main(['--model', str(ck), '--data_path', str(tmp_path / 'data'), '--bs', '4', '--root_joint', '9', '--workers', '2',
                 '--result_dir', str(out)])
    text = capsys.readouterr().out
    assert 'joint mean error:' in text and 'root error:' in text and 'images/s from files' in text
    assert sorted(os.listdir(out)) == sorted(['left_joint.txt', 'right_joint.txt', 'joint_left_error.txt', 'joint_right_error.txt',
                                              'mesh_left_error.txt', 'mesh_right_error.txt', 'joint_2d_left_error.txt', 'joint_2d_right_error.txt',
                                              'mesh_2d_left_error.txt', 'mesh_2d_right_error.txt', 'root_loss.txt', 'volume.txt'])
    assert np.loadtxt(str(out / 'root_loss.txt')).shape == (6,) and m.root_joint == 9
