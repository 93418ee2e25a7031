"""N > 1 on real GPUs (skipped on 1-GPU boxes): two ranks over RCCL shard a batch through DirEngine and gather it back."""
import json
import os

import pytest
import torch


@pytest.mark.gpu
def test_two_ranks_shard_and_gather_bit_identical(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    from dir_amd import dist as D
    out = tmp_path / 'two.json'
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'two_rank_engine.py')
    assert D.spawn_ranks([script, str(out)], 2, timeout=900) == 0
    got = json.loads(out.read_text())
    assert got == {'world': 2, 'bit_identical': True}
