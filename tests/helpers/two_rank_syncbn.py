"""Two ranks (gloo; both may share one GPU) run SyncBN -- dir_amd.train.ops.sync_batchnorm -- on the two halves of a batch; rank 0 also runs the
ordinary training-mode BatchNorm over the WHOLE batch in one process and compares: outputs, saved statistics, running statistics, g x, and the
sum of the two ranks' g w / g b (the reference's batch of 64 lives on one GPU, config.py:13-15; torch.nn.SyncBatchNorm semantics)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dir_amd import dist as D            # noqa: E402
from dir_amd.train import ops as O       # noqa: E402


def main(out_path):
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    rank, world, _ = D.init_from_env('gloo')
    res = {'world': world}
    worst = {}
    for case, (R_each, C, relu, with_res) in enumerate([((700, 900), 64, True, False), ((1024, 1024), 256, False, True), ((130, 70), 128, True, True)]):
        g = torch.Generator(device='cpu').manual_seed(50 + case)
        R = sum(R_each)
        x = (torch.randn(R, C, generator=g) * 2 + 0.5).to(dev)
        gy = torch.randn(R, C, generator=g).to(dev)
        w, b = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
        resid = torch.randn(R, C, generator=g).to(dev) if with_res else None
        a0, a1 = (0, R_each[0]) if rank == 0 else (R_each[0], R)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        with O.sync_batchnorm():
            y, st = O.bn_train_fwd(x[a0:a1].contiguous(), w, b, rm, rv, relu=relu, residual=None if resid is None else resid[a0:a1].contiguous())
            gx, gw, gb = O.bn_train_bwd(gy[a0:a1].contiguous(), x[a0:a1].contiguous(), w, st, b=b, relu=relu and not with_res)
        assert len(st) == 3 and st[2] == float(R)
        # gather the ranks' pieces on rank 0 (through the host: gloo)
        ys = D.gather_shards(y.cpu(), R) if R_each[0] == R_each[1] else None
        pieces = [torch.empty(0)] * world
        objs = [None] * world
        torch.distributed.all_gather_object(objs, {'y': y.cpu(), 'gx': gx.cpu(), 'gw': gw.cpu(), 'gb': gb.cpu(), 'rm': rm.cpu(), 'rv': rv.cpu(), 'sm': st[0].cpu(), 'sr': st[1].cpu()})
        if rank == 0:
            rm1, rv1 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            y1, st1 = O.bn_train_fwd(x, w, b, rm1, rv1, relu=relu, residual=resid)
            gx1, gw1, gb1 = O.bn_train_bwd(gy, x, w, st1, b=b, relu=relu and not with_res)
            cat = lambda k: torch.cat([o[k] for o in objs], 0)       # noqa: E731
            rel = lambda a, b_: float((a - b_).abs().max() / b_.abs().max().clamp_min(1e-30))     # noqa: E731
            d = {'y': rel(cat('y'), y1.cpu()), 'gx': rel(cat('gx'), gx1.cpu()), 'gw': rel(objs[0]['gw'] + objs[1]['gw'], gw1.cpu()),
                 'gb': rel(objs[0]['gb'] + objs[1]['gb'], gb1.cpu()), 'running_mean': rel(objs[0]['rm'], rm1.cpu()), 'running_var': rel(objs[0]['rv'], rv1.cpu()),
                 'save_mean': rel(objs[0]['sm'], st1[0].cpu()), 'save_rstd': rel(objs[0]['sr'], st1[1].cpu()),
                 'ranks_agree': float(max((objs[0][k] - objs[1][k]).abs().max() for k in ('rm', 'rv', 'sm', 'sr')))}
            if ys is not None:
                d['gather_shards_y'] = rel(ys, y1.cpu())
            worst['case%d' % case] = d
    if rank == 0:
        res['cases'] = worst
        with open(out_path, 'w') as f:
            json.dump(res, f)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
