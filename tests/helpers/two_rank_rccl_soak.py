"""Rank script for tests/test_gpu_multi.py::test_rccl_allreduce_beside_replayed_forwards (needs 2 GPUs): every rank replays a captured
64-image forward on its own stream while an RCCL all-reduce of a 64 MB buffer runs on the process group's stream, round after round.  RCCL's
kernels are somebody else's code (free to use packed-FP32 instructions); both sides must reproduce their stand-alone results: the forward bit
for bit, the all-reduce exactly (integers stored as floats)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dir_amd import dist as D  # noqa: E402
from dir_amd import synth  # noqa: E402
from dir_amd.engine import DirEngine, ForwardPipeline  # noqa: E402

local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
rank, world, _ = D.init_from_env('nccl', dev)
with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
    shapes = {k: tuple(v) for k, v in json.load(f).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
eng = DirEngine(sd, dtype=torch.bfloat16, device=dev)
gen = torch.Generator(device=dev).manual_seed(5 + rank)
img = torch.randn(64, 3, 256, 256, device=dev, generator=gen)
pipe = ForwardPipeline(eng, [img])
pipe.launch(0)
o = pipe.wait(0)
keys = ('pd_mesh_xyz_left', 'pd_joint_uv_right', 'pd_offset')
ref = [o[s][k].clone() for s in range(3) for k in keys] + [o[3]['seg'].clone()]
base = torch.arange(1 << 24, device=dev, dtype=torch.float32) % 1024 + rank          # small integers: the sum over ranks is exact in fp32
want = sum((torch.arange(1 << 24, device=dev, dtype=torch.float32) % 1024 + r) for r in range(world))
ok_fwd, ok_red = True, True
for rnd in range(6):
    pipe.launch(0)
    works, bufs = [], []
    for _ in range(6):
        b = base.clone()
        bufs.append(b)
        works.append(torch.distributed.all_reduce(b, async_op=True))
    o = pipe.wait(0)
    for w in works:
        w.wait()
    torch.cuda.synchronize(dev)
    got = [o[s][k] for s in range(3) for k in keys] + [o[3]['seg']]
    ok_fwd = ok_fwd and all(torch.equal(x, y) for x, y in zip(got, ref))
    ok_red = ok_red and all(torch.equal(b, want) for b in bufs)
flags = torch.tensor([int(ok_fwd), int(ok_red)], device=dev)
torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MIN)
if rank == 0:
    with open(sys.argv[1], 'w') as f:
        json.dump({'world': world, 'forward_bit_identical': bool(flags[0].item()), 'allreduce_exact': bool(flags[1].item())}, f)
torch.distributed.destroy_process_group()
