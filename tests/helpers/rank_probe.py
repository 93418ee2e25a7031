"""Rank script for tests/test_dist_gloo.py::test_spawn_ranks_launcher: launched by dir_amd.dist.spawn_ranks (torch.distributed.run),
CPU + gloo.  Every rank contributes rank + 1 to a SUM and its rank to a gathered shard; rank 0 writes what it saw."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dir_amd import dist as D  # noqa: E402

rank, world, local = D.init_from_env('gloo')
D.barrier()
total = D.sum_over_ranks(rank + 1)
a, b = D.shard_range(10, rank, world)
full = D.gather_shards(torch.arange(a, b, dtype=torch.float32).reshape(-1, 1), 10)
if rank == 0:
    with open(sys.argv[1], 'w') as f:
        json.dump({'world': world, 'sum': total, 'gathered': full[:, 0].tolist(), 'dist_world': torch.distributed.get_world_size()}, f)
torch.distributed.destroy_process_group()
