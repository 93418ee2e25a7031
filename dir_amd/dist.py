"""Multi-GPU sharding of the DIR hot path: images are independent in eval mode (BatchNorm uses running
statistics, no cross-sample op), so N GPUs = N processes that each run the whole path on their own shard of the batch
with replicated weights and NO data-path collective.  The only collectives are control-plane: a barrier around timed
regions, a MAX over ranks of the elapsed time, and an all-gather of per-image results for evaluation.

Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" on CPU (used by the tests).  Rendezvous on 127.0.0.1."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """(rank, world, local_rank).  Initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n, rank, world):
    """contiguous shard [start, stop) of n independent images for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def barrier(device=None):
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_shards(local, n_total):
    """all-gather per-image results (first dim = images of this rank's shard, contiguous shards in rank order) back
    into the full [n_total, ...] tensor on every rank.  Shards may differ in length by one (padded for the collective)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local
    world = dist.get_world_size()
    longest = (n_total + world - 1) // world
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r, p in enumerate(parts):
        a, b = shard_range(n_total, r, world)
        out.append(p[:b - a])
    return torch.cat(out, 0)


def _collective_view(t):
    """gloo (the CPU test backend, and DIR_BENCH_BACKEND=gloo on a box with fewer GPUs than ranks) reduces host memory: a device tensor goes
    through a host copy there; RCCL takes the device tensor itself"""
    if t.is_cuda and dist.get_backend() != 'nccl':
        return t.cpu(), True
    return t, False


def all_gather_rows(t):
    """[n] -> [world, n]: every rank's vector, in rank order, on every rank (SyncBN forward: the per-channel (mean | M2 | rows) parts).
    One process: t[None]."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t.unsqueeze(0)
    world = dist.get_world_size()
    src, staged = _collective_view(t.contiguous())
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    out = torch.stack(parts, 0)
    return out.to(t.device) if staged else out


def all_reduce_sum(t):
    """sum over the ranks, returned as a new tensor on t's device (SyncBN backward: the pooled (sum g | sum g xhat)).  One process: t."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    src, staged = _collective_view(t.contiguous().clone())
    dist.all_reduce(src, op=dist.ReduceOp.SUM)
    return src.to(t.device) if staged else src


def average_gradients(flat_grad, bucket_elems=64 << 20):
    """Data-parallel gradient exchange of the reference's training set-up (one all-reduce of the 92.7 M gradients per
    step, divided by the world size): `flat_grad` is dir_amd.optim.FlatAdamW.flat_grad, already one contiguous buffer, reduced in
    place in `bucket_elems`-element pieces (256 MB of fp32 each: large enough for RCCL's ring / direct algorithms to run at link
    rate over xGMI, small enough that a following piece can be issued while the previous one completes).  No-op for one process."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return flat_grad
    world = dist.get_world_size()
    flat = flat_grad.view(-1)
    avg = dist.get_backend() == 'nccl'                    # RCCL averages inside the collective; gloo (CPU tests) sums, then one division
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    works = [dist.all_reduce(flat[i:i + bucket_elems], op=op, async_op=True) for i in range(0, flat.numel(), bucket_elems)]
    for w in works:
        w.wait()
    if not avg:
        flat.div_(world)
    return flat_grad


class GradientBucketer(object):
    """The same exchange as average_gradients, overlapped with the backward pass: the flat gradient buffer is cut
    into buckets at parameter boundaries, walking the parameters from the LAST to the first -- the order in which the backward pass
    finishes them (heads, decoder stages, InitRegressor, backbone layer4 .. stem) -- and a bucket's all-reduce is issued the moment its
    last gradient has been written (`mark_ready`), asynchronously: on RCCL the collective runs on the process group's own stream
    behind an event of the compute stream, so it overlaps the rest of the backward; `finish()` issues whatever is left (parameters that
    never receive a gradient) and waits.  Every element is reduced exactly once, by one all-reduce over the same ranks as in
    average_gradients: with two ranks the result is bit-identical to the unbucketed path (tests/test_dist_gloo.py).

        b = GradientBucketer(opt.flat_grad, opt.offsets, [p.numel() for p in opt.params])      # once
        b.begin(); ...backward...: b.mark_ready([parameter indices whose .grad is final]); ...; b.finish()"""

    def __init__(self, flat_grad, offsets, numels, bucket_elems=16 << 20):
        self.flat = flat_grad.view(-1)
        n = len(offsets)
        ends = list(offsets[1:]) + [self.flat.numel()]
        self.buckets, self.bucket_of = [], [0] * n           # bucket = [start, end, parameter indices]; built from the last parameter down
        cur = None
        for i in range(n - 1, -1, -1):
            if cur is None:
                cur = [offsets[i], ends[i], []]
            cur[0] = offsets[i]
            cur[2].append(i)
            self.bucket_of[i] = len(self.buckets)
            if cur[1] - cur[0] >= bucket_elems:
                self.buckets.append(cur)
                cur = None
        if cur is not None:
            self.buckets.append(cur)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.begin()

    def begin(self):
        self.pending = [set(b[2]) for b in self.buckets]
        self.works, self.issued = [], [False] * len(self.buckets)

    def _issue(self, bi):
        if self.issued[bi]:
            return
        self.issued[bi] = True
        if self.world > 1:
            avg = dist.get_backend() == 'nccl'
            a, e = self.buckets[bi][0], self.buckets[bi][1]
            self.works.append(dist.all_reduce(self.flat[a:e], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True))

    def mark_ready(self, indices):
        """the gradients of these parameters (indices into the optimiser's parameter list) are final"""
        for i in indices:
            bi = self.bucket_of[i]
            self.pending[bi].discard(i)
            if not self.pending[bi]:
                self._issue(bi)

    def finish(self):
        for bi in range(len(self.buckets)):
            self._issue(bi)
        for w in self.works:
            w.wait()
        if self.world > 1 and dist.get_backend() != 'nccl':
            self.flat.div_(self.world)                        # gloo sums; same single division as average_gradients
        self.works = []


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script_argv, nproc, env=None, timeout=None):
    """Run `python <script_argv...>` as `nproc` ranks of ONE node through torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 with a free port) and return its exit code: the same command line the driver uses for N > 1, so that a bare
    `python bench.py --gpus N` produces N ranks by itself.  Every rank sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    import subprocess
    import sys
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC only on this stack (RCCL needs it)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nproc)),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port())] + list(script_argv)
    return subprocess.run(cmd, env=e, timeout=timeout).returncode
