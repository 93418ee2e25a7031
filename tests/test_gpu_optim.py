"""SURVEY 8f rank 2, optimiser half: dir_adamw_step / dir_amd.optim against torch.optim.AdamW + CosineAnnealingLR (what train.py:227-230
constructs; run on CPU tensors as the checker) and against the numpy oracle.  fp32 in torch's operation order: 2e-7 relative."""
import os

import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import optim as DO
from oracle import optim as OO

pytestmark = pytest.mark.gpu
SHAPES = [(64, 3, 7, 7), (64,), (5, 3), (1,), (21, 128, 128), (1027,)]


def make_params(rng, device):
    return [torch.nn.Parameter(torch.from_numpy(rng.normal(0, 0.1, s).astype(np.float32)).to(device)) for s in SHAPES]


def test_flat_adamw_matches_torch_and_oracle():
    rng = np.random.RandomState(1)
    ref_p = make_params(rng, 'cpu')
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref = torch.optim.AdamW([{'params': ref_p, 'initial_lr': 1e-3}], 1e-3)
    ref_s = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=5, eta_min=0)
    our = DO.FlatAdamW([{'params': our_p, 'initial_lr': 1e-3}], 1e-3)
    our_s = DO.CosineAnnealingLR(our, T_max=5, eta_min=0)
    assert all(p.data_ptr() >= our.flat_param.data_ptr() for p in our_p) and all(p.data_ptr() % 16 == 0 for p in our_p)
    om = [np.zeros(s, np.float32) for s in SHAPES]
    ov = [np.zeros(s, np.float32) for s in SHAPES]
    op = [p.detach().numpy().copy() for p in ref_p]
    for step in range(1, 8):
        assert abs(our.param_groups[0]['lr'] - ref.param_groups[0]['lr']) < 1e-12
        lr = ref.param_groups[0]['lr']
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            g = (rng.normal(0, 1, a.shape) * 10.0 ** rng.randint(-4, 1)).astype(np.float32)
            a.grad = torch.from_numpy(g.copy())
            b.grad.copy_(torch.from_numpy(g))                   # .grad is a view of the flat gradient buffer
            op[i], om[i], ov[i] = OO.adamw_step(op[i], g, om[i], ov[i], step, lr)
        ref.step(); our.step()
        ref_s.step(); our_s.step()
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            assert relerr(b.detach().cpu().numpy(), a.detach().numpy()) < 2e-7, (step, i)
            assert relerr(b.detach().cpu().numpy(), op[i]) < 2e-7, (step, i)
    # the padding between slots never moves
    pad = torch.ones(our.numel, dtype=torch.bool)
    for p, o in zip(our_p, our.offsets):
        pad[o:o + p.numel()] = False
    assert float(our.flat_param.cpu()[pad].abs().sum()) == 0.0


def test_state_dict_interchange_with_torch(tmp_path):
    """a torch.optim.AdamW state loads into FlatAdamW and the other way round; both continue identically"""
    rng = np.random.RandomState(2)
    ref_p = make_params(rng, 'cpu')
    ref = torch.optim.AdamW([{'params': ref_p, 'initial_lr': 2e-3}], 2e-3)
    for _ in range(3):
        for a in ref_p:
            a.grad = torch.from_numpy(rng.normal(0, 1, a.shape).astype(np.float32))
        ref.step()
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    our = DO.FlatAdamW([{'params': our_p, 'initial_lr': 2e-3}], 2e-3)
    our.load_state_dict(ref.state_dict())
    assert our.step_count == 3
    for _ in range(2):
        for a, b in zip(ref_p, our_p):
            g = rng.normal(0, 1, a.shape).astype(np.float32)
            a.grad = torch.from_numpy(g.copy()); b.grad.copy_(torch.from_numpy(g))
        ref.step(); our.step()
    for a, b in zip(ref_p, our_p):
        assert relerr(b.detach().cpu().numpy(), a.detach().numpy()) < 2e-7
    # ... and back: a fresh torch optimiser resumes from our state dict
    sd = our.state_dict()
    assert set(sd) == {'state', 'param_groups'} and sd['param_groups'][0]['params'] == list(range(len(SHAPES)))
    back_p = [torch.nn.Parameter(p.detach().cpu().clone()) for p in our_p]
    back = torch.optim.AdamW([{'params': back_p, 'initial_lr': 2e-3}], 2e-3)
    back.load_state_dict({'state': {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd['state'].items()},
                          'param_groups': sd['param_groups']})
    for a, b, c in zip(ref_p, our_p, back_p):
        g = rng.normal(0, 1, a.shape).astype(np.float32)
        a.grad = torch.from_numpy(g.copy()); b.grad.copy_(torch.from_numpy(g)); c.grad = torch.from_numpy(g.copy())
    ref.step(); our.step(); back.step()
    for a, b, c in zip(ref_p, our_p, back_p):
        assert relerr(c.detach().numpy(), a.detach().numpy()) < 2e-7 and relerr(b.detach().cpu().numpy(), a.detach().numpy()) < 2e-7


def test_checkpoint_round_trip(tmp_path):
    """train.py:127-149: {'net', 'optimizer', 'schedule', 'last_epoch'}"""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.Linear(5, 3)).cuda()
    opt = DO.FlatAdamW([{'params': net.parameters(), 'initial_lr': 1e-3}], 1e-3)
    sch = DO.CosineAnnealingLR(opt, T_max=10)
    for _ in range(3):
        opt.flat_grad.normal_()
        opt.step(); sch.step()
    path = os.path.join(str(tmp_path), 'ck.pth')
    DO.save_checkpoint(path, net, opt, sch, 2)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) == {'net', 'optimizer', 'schedule', 'last_epoch'} and ck['last_epoch'] == 2
    net2 = torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.Linear(5, 3)).cuda()
    opt2 = DO.FlatAdamW([{'params': net2.parameters(), 'initial_lr': 1e-3}], 1e-3)
    sch2 = DO.CosineAnnealingLR(opt2, T_max=10)
    assert DO.load_checkpoint(path, net2, opt2, sch2) == 3
    assert opt2.param_groups[0]['lr'] == opt.param_groups[0]['lr'] and opt2.step_count == 3
    g = torch.randn(opt.numel, device='cuda')
    opt.flat_grad.copy_(g); opt2.flat_grad.copy_(g)
    opt.step(); opt2.step()
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)


def test_adamw_argument_errors():
    from dir_amd import _capi
    p = torch.zeros(16, device='cuda')
    L = _capi.lib()
    assert L.dir_adamw_step(_capi.ptr(p), _capi.ptr(p), _capi.ptr(p), None, 16, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, None) != 0
    assert L.dir_adamw_step(_capi.ptr(p), _capi.ptr(p), _capi.ptr(p), _capi.ptr(p), 16, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0, None) != 0
    with pytest.raises(_capi.DirHipError):
        DO.FlatAdamW([torch.nn.Parameter(torch.zeros(4))])       # CPU parameters: no fallback
