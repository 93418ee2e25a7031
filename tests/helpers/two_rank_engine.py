"""Rank script for tests/test_gpu_multi.py: 2 ranks (one per GPU, RCCL) each run the hot path on their contiguous shard of a batch
of 8 synthetic images and all-gather the results (dir_amd.dist.gather_shards); rank 0 also runs the whole batch on its GPU and
demands bit-identical outputs (images are independent: models/dir.py:513-540 has no cross-sample op in eval mode)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dir_amd import dist as D  # noqa: E402
from dir_amd import synth  # noqa: E402
from dir_amd.engine import DirEngine  # noqa: E402

local = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
rank, world, _ = D.init_from_env('nccl', dev)
with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
    shapes = {k: tuple(v) for k, v in json.load(f).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
N = 8
img = torch.from_numpy(synth.synth_input('multi.img', (N, 3, 256, 256))).to(dev)
eng = DirEngine(sd, dtype=torch.bfloat16, device=dev)
a, b = D.shard_range(N, rank, world)
outs = eng.forward(img[a:b].contiguous())
keys = ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_uv_left', 'pd_offset')
gathered = {(i, k): D.gather_shards(outs[i][k].contiguous(), N) for i in range(3) for k in keys}
gathered[(3, 'seg')] = D.gather_shards(outs[3]['seg'].contiguous(), N)
D.barrier(dev)
ok = True
if rank == 0:
    whole = eng.forward(img)
    for (i, k), t in gathered.items():
        ok = ok and torch.equal(t, whole[i][k].contiguous())
    with open(sys.argv[1], 'w') as f:
        json.dump({'world': torch.distributed.get_world_size(), 'bit_identical': bool(ok)}, f)
torch.distributed.destroy_process_group()
