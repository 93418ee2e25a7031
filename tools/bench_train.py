"""(Under `python -m torch.distributed.run --nproc-per-node N tools/bench_train.py`: one rank per GPU, each with its own synthetic batch,
gradients averaged by dist.average_gradients over RCCL -- BASELINE config 3's data-parallel set-up; rank 0 prints the aggregate.)
Time of one whole-network training step (dir_amd.train.step.train_step: training-mode forward, 42-term objective, backward, flat gradient
bucket, AdamW) at BASELINE config 3's per-GPU batch (32), synthetic data, 1 GPU.  This path is correctness-first (not tuned); the number is a
baseline for the rounds that tune it.  usage: [BACKBONE=hrnet_w48 [EXTRA_STAGES=2]] bench_train.py [batch] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd import _capi, synth  # noqa: E402
from dir_amd.optim import FlatAdamW  # noqa: E402
from dir_amd.train import step as TSTEP  # noqa: E402

from dir_amd import dist as D  # noqa: E402
local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
rank, world, _ = D.init_from_env('nccl', torch.device('cuda', local))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if os.environ.get('BACKBONE', 'resnet50') == 'hrnet_w48':        # BASELINE configs[4]: HRNet-W48 + init + 4 refinement stages (no reference counterpart)
    from dir_amd.models.dir import DIR
    shapes = {k: tuple(v.shape) for k, v in DIR(21, 'unused', 0, backbone='hrnet_w48', extra_stages=int(os.environ.get('EXTRA_STAGES', '2'))).state_dict().items()}
    sd = synth.synth_state_dict(shapes, 1234, cond=True)
else:
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth.synth_state_dict(shapes, 1234)
is_buf = lambda k: any(t in k for t in ('running_', 'num_batches', 'mano_layer', 'img_gird', 'seg_loss.weight'))  # noqa: E731
params = {k: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).cuda()) for k, v in sd.items() if not is_buf(k)}
buffers = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items() if is_buf(k) and 'num_batches' not in k}
opt = FlatAdamW(list(params.values()), lr=1e-5)
opt.set_inactive(TSTEP.inactive_parameters(params))
rng = np.random.RandomState(rank)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
img = dv(synth.synth_input('train.img.%d' % rank, (B, 3, 256, 256), 1234))
target, meta = {}, {}
for s in ('left', 'right'):
    target['joint_2d_' + s] = dv(rng.uniform(-1, 1, (B, 21, 3)).astype(np.float32))
    target['mesh_2d_' + s] = dv(rng.uniform(-1, 1, (B, 778, 3)).astype(np.float32))
    target['joint_3d_' + s] = dv(rng.normal(0, 0.05, (B, 21, 3)).astype(np.float32))
    target['mesh_3d_' + s] = dv(rng.normal(0, 0.05, (B, 778, 3)).astype(np.float32))
    meta['center_' + s] = dv(rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32))
target['seg'] = dv(rng.randint(0, 3, (B, 1, 256, 256)).astype(np.float32))
target['dense'] = dv(rng.rand(B, 3, 256, 256).astype(np.float32))
faces = tuple(dv(synth.loss_faces(s, 1234).astype(np.int64)) for s in ('left', 'right'))
times, totals = [], []
for i in range(steps + 1):
    torch.cuda.synchronize()
    t0 = time.time()
    _capi.PROFILE = [] if i == steps else None
    loss = TSTEP.train_step(params, buffers, img, target, meta, faces, opt)
    torch.cuda.synchronize()
    times.append(time.time() - t0)
    totals.append(sum(float(v) for v in loss.values()))
prof, _capi.PROFILE = _capi.PROFILE, None
if rank == 0:
    print('batch %d x %d rank(s): train step %s s (first includes allocator warm-up); objective on rank 0 %s; %.0f images/s aggregate'
          % (B, world, ' '.join('%.3f' % t for t in times), ' -> '.join('%.3f' % t for t in totals), B * world / min(times)))
    print('peak memory %.1f GB' % (torch.cuda.max_memory_allocated() / 2 ** 30))
    agg = {}
    for r in prof or []:
        k = (r['api'], r.get('shape', ''))
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r['e0'].elapsed_time(r['e1']); a[2] += r.get('flops', 0.0)
    tot = sum(a[1] for a in agg.values())
    print('library calls of the last step: %.1f ms of events in %d calls; top by time:' % (tot, sum(a[0] for a in agg.values())))
    for (api, shape), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', '28'))]:
        print('  %-34s %-52s x%-4d %8.3f ms %8.1f TFLOP/s' % (api, shape, n, ms, fl / ms / 1e9 if ms > 0 else 0))
if world > 1:
    flat = opt.flat_param.clone()
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(flat, opt.flat_param), 'ranks diverged: the averaged gradients / AdamW steps are not identical across ranks'
    torch.distributed.destroy_process_group()
