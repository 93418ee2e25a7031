"""Rank script for tests/test_gpu_multi.py::test_two_rank_data_parallel_train_step: two ranks (gloo group: it also reduces CUDA tensors, so the
test runs on a one-GPU box; on a multi-GPU node each rank takes its own device) each run dir_amd.train.step.train_step on their own batch --
the gradients meet in dist.average_gradients(FlatAdamW.flat_grad) -- and must end with IDENTICAL parameters, equal to what one process
gets from the mean of the two ranks' gradients."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dir_amd import dist as D  # noqa: E402
from dir_amd import synth  # noqa: E402
from dir_amd.optim import FlatAdamW  # noqa: E402
from dir_amd.train import conv as TC  # noqa: E402
from dir_amd.train import net as TN  # noqa: E402
from dir_amd.train import step as TSTEP  # noqa: E402

local = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
rank, world, _ = D.init_from_env('gloo', dev)
with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
    shapes = {k: tuple(v) for k, v in json.load(f).items()}
sd = synth.synth_state_dict(shapes, 1234)
is_buf = lambda k: any(t in k for t in ('running_', 'num_batches', 'mano_layer', 'img_gird', 'seg_loss.weight'))  # noqa: E731


def fresh():
    params = {k: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).to(dev)) for k, v in sd.items() if not is_buf(k)}
    buffers = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sd.items() if is_buf(k) and 'num_batches' not in k}
    opt = FlatAdamW(list(params.values()), lr=1e-5)
    opt.set_inactive(TSTEP.inactive_parameters(params))
    return params, buffers, opt


def batch(r, B=2):
    rng = np.random.RandomState(100 + r)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    img = dv(synth.synth_input('dp.img.%d' % r, (B, 3, 256, 256), 1234))
    target, meta = {}, {}
    for s in ('left', 'right'):
        target['joint_2d_' + s] = dv(rng.uniform(-1, 1, (B, 21, 3)).astype(np.float32))
        target['mesh_2d_' + s] = dv(rng.uniform(-1, 1, (B, 778, 3)).astype(np.float32))
        target['joint_3d_' + s] = dv(rng.normal(0, 0.05, (B, 21, 3)).astype(np.float32))
        target['mesh_3d_' + s] = dv(rng.normal(0, 0.05, (B, 778, 3)).astype(np.float32))
        meta['center_' + s] = dv(rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32))
    target['seg'] = dv(rng.randint(0, 3, (B, 1, 256, 256)).astype(np.float32))
    target['dense'] = dv(rng.rand(B, 3, 256, 256).astype(np.float32))
    return img, target, meta


faces = tuple(torch.from_numpy(synth.loss_faces(s, 1234).astype(np.int64)).to(dev) for s in ('left', 'right'))
params, buffers, opt = fresh()
img, target, meta = batch(rank)
TSTEP.train_step(params, buffers, img, target, meta, faces, opt)          # forward, backward with the bucketed all-reduce (mean) overlapped, AdamW
mine = opt.flat_param.clone()
# the round-2 path (one exchange after the whole backward) from the same starting point: bit-identical parameters
params_u, buffers_u, opt_u = fresh()
TSTEP.train_step(params_u, buffers_u, img, target, meta, faces, opt_u, overlap_allreduce=False)
bucketed_equals_unbucketed = bool(torch.equal(opt_u.flat_param, mine))
n_buckets = len(opt._bucketer.buckets)
other = mine.clone()
torch.distributed.broadcast(other, src=0)
same_across_ranks = bool(torch.equal(mine, other))
result = {'world': world, 'same_across_ranks': same_across_ranks, 'bucketed_equals_unbucketed': bucketed_equals_unbucketed, 'buckets': n_buckets}
if rank == 0:
    # one process: both ranks' gradients from the same starting point, averaged by hand, one AdamW step
    p1, b1, o1 = fresh()
    acc = torch.zeros_like(o1.flat_grad)
    for r in range(world):
        P = {k: v.data for k, v in p1.items()}
        P.update({k: v.clone() for k, v in b1.items()})
        i2, t2, m2 = batch(r)
        TC.reset_scales()                                   # the split-precision operand scales are calibrated on a rank's own first batch: do the same here
        outs, ctx = TN.forward(P, i2)
        G = TN.backward(P, ctx, outs, t2, m2, faces)
        o1.zero_grad()
        TSTEP.add_grads(p1, '', G)
        acc += o1.flat_grad
    o1.flat_grad.copy_(acc / world)
    o1.step()
    d = float((o1.flat_param - mine).abs().max())
    result.update(max_abs_diff_to_single_process=d, lr=1e-5)
torch.distributed.barrier()
gathered = [None] * world
torch.distributed.all_gather_object(gathered, same_across_ranks)
if rank == 0:
    result['same_across_ranks'] = all(gathered)
    with open(sys.argv[1], 'w') as f:
        json.dump(result, f)
torch.distributed.destroy_process_group()
