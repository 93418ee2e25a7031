"""N > 1 on real GPUs (skipped on 1-GPU boxes): two ranks over RCCL shard a batch through DirEngine and gather it back."""
import json
import os

import numpy as np
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


@pytest.mark.gpu
def test_two_rank_data_parallel_train_step(tmp_path):
    """BASELINE config 3's set-up in small: two ranks, each its own batch, gradients averaged in FlatAdamW.flat_grad by
    dist.average_gradients, one AdamW launch each.  (gloo group so that it also runs with both ranks on ONE GPU; RCCL on a multi-GPU node
    is the same call.)  Both ranks end with bit-identical parameters; they equal a single process stepping on the hand-averaged
    gradients of the two batches (AdamW's first step moves every weight by ~lr: agreement to a small fraction of that)."""
    from dir_amd import dist as D
    out = tmp_path / 'dp.json'
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'two_rank_train.py')
    assert D.spawn_ranks([script, str(out)], 2, timeout=900) == 0
    got = json.loads(out.read_text())
    assert got['world'] == 2 and got['same_across_ranks'] is True
    assert got['bucketed_equals_unbucketed'] is True and got['buckets'] >= 4, got      # gradient exchange overlapped with the backward: same bits
    assert got['max_abs_diff_to_single_process'] < 0.02 * got['lr'], got


@pytest.mark.gpu
def test_rccl_allreduce_beside_replayed_forwards(tmp_path):
    """RCCL's all-reduce kernels (not built with this library's no-packed-FP32 flag) on the process group's stream while
    forwards replay on the slot's stream -- both sides reproduce their stand-alone results.  RCCL refuses two ranks on one device, so this
    needs two GPUs (the driver's multi-GPU tier); on a 1-GPU box the foreign-kernel soak of tests/test_gpu_soak.py is what runs."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    from dir_amd import dist as D
    out = tmp_path / 'soak.json'
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'two_rank_rccl_soak.py')
    assert D.spawn_ranks([script, str(out)], 2, timeout=900) == 0
    assert json.loads(out.read_text()) == {'world': 2, 'forward_bit_identical': True, 'allreduce_exact': True}


@pytest.mark.gpu
def test_training_steps_at_config_3_per_gpu_batch(tmp_path):
    """BASELINE configs[3]'s per-GPU workload as a test (not only a bench line): batch 32, 256x256, the whole network -- training-mode forward,
    42-term objective, backward, flat bucket, AdamW -- four steps through tools/bench_train.py in its own process: the objective is finite and
    falls monotonically on the fixed synthetic batch (lr 1e-5), and a step stays far inside the round-2 time (0.088 s)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'bench_train.py'), '32', '3'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('batch 32')][0]
    times = [float(t) for t in re.search(r'train step ([\d. ]+) s', line).group(1).split()]
    obj = [float(t) for t in re.search(r'objective on rank 0 ([\d.\- >]+);', line).group(1).split(' -> ')]
    assert len(obj) == 4 and all(np.isfinite(obj)) and all(b < a for a, b in zip(obj, obj[1:])), obj
    assert min(times) < 0.08, times


@pytest.mark.gpu
def test_bench_multi_rank_code_path_over_gloo(tmp_path):
    """bench.py's N > 1 path has never met more than one GPU (the driver's 8-GPU tier is the first to run it): rendezvous on 127.0.0.1, the
    barriers around the timed regions, the max over ranks, rank 0 deciding the kernel table for everybody, ONE line from rank 0, everyone
    leaving the process group together.  DIR_BENCH_BACKEND=gloo runs exactly that control flow with two ranks on whatever GPUs the box has
    (sharing one here, which RCCL would refuse): the line must say n_gpus 2, weak scaling, a whole-job value of two ranks' images, and its backend."""
    from dir_amd import dist as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'detail.json'
    env = dict(os.environ, DIR_BENCH_BACKEND='gloo')
    rc = D.spawn_ranks([os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--repeats', '2', '--no-cpu-baseline', '--no-fp32-mode',
                        '--no-train', '--no-proj-feat-variant', '--no-power', '--no-config5', '--no-ceiling-probe', '--no-time-table-pass',
                        '--detail-out', str(out)], 2, env=env, timeout=900)
    assert rc == 0
    d = json.loads(out.read_text())
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 3 and d['config']['world_size_observed'] == 2
    assert 'gloo' in d['config']['backend'] and d['config']['batch_per_gpu'] == 64
    assert abs(d['value'] - 2 * 64 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']          # whole-job: both ranks' images over the slowest rank's time
    assert d['roofline'] is not None and d['cpu_baseline'] is None


@pytest.mark.gpu
def test_sync_batchnorm_two_ranks_equal_the_single_process_batch(tmp_path):
    """SURVEY.md 8e / VERDICT r4 item 8: opt-in SyncBN (dir_amd.train.ops.sync_batchnorm).  The reference normalises its 64 images on ONE GPU
    (config.py:13-15); two ranks holding uneven halves of a batch pool (mean | M2 | rows) forward and (sum g | sum g xhat) backward and must
    reproduce the single-process BatchNorm over the whole batch: y, saved / running statistics and g x to fp32 rounding, g w / g b as the sum of
    the ranks' local ones; the pooled statistics are the same bytes on both ranks.  gloo group (both ranks share this box's GPU; RCCL on a
    multi-GPU node is the same call) -- unmeasured on multi-GPU hardware."""
    from dir_amd import dist as D
    out = tmp_path / 'syncbn.json'
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers', 'two_rank_syncbn.py')
    assert D.spawn_ranks([script, str(out)], 2, timeout=900) == 0
    got = json.loads(out.read_text())
    assert got['world'] == 2 and len(got['cases']) == 3
    for name, d in got['cases'].items():
        assert d['ranks_agree'] == 0.0, (name, d)
        for k in ('y', 'gx', 'gw', 'gb', 'running_mean', 'running_var', 'save_mean', 'save_rstd'):
            assert d[k] < 2e-5, (name, k, d)
