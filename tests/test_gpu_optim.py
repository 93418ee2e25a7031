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


def test_step_bumps_versions_and_skips_inactive_parameters():
    """(1) the kernel writes parameters through raw pointers: step() must bump every parameter's version counter like torch's
    in-place ops do, or caches keyed on (data_ptr, _version) -- DIR.engine() -- keep serving stale weights.  (2) parameters that
    never get a gradient (torch: `p.grad is None`, e.g. the never-executed STE block 0) take no weight decay and have no state,
    exactly like torch.optim.AdamW; (3) detached gradients are an error, not a silent decay-only step."""
    rng = np.random.RandomState(3)
    ref_p = make_params(rng, 'cpu')
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    dead = (1, 4)                                                   # indices that never receive a gradient
    ref = torch.optim.AdamW(ref_p, 1e-2)
    our = DO.FlatAdamW(our_p, 1e-2)
    our.set_inactive([our_p[i] for i in dead])
    v0 = [p._version for p in our_p]
    for _ in range(3):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i in dead:
                continue
            g = rng.normal(0, 1, a.shape).astype(np.float32)
            a.grad = torch.from_numpy(g.copy())
            b.grad.copy_(torch.from_numpy(g))
        ref.step(); our.step()
    assert all(p._version > v for p, v in zip(our_p, v0))
    for i, (a, b) in enumerate(zip(ref_p, our_p)):
        assert relerr(b.detach().cpu().numpy(), a.detach().numpy()) < 2e-7, i       # dead ones: untouched on both sides
    assert sorted(our.state_dict()['state']) == sorted(ref.state_dict()['state']) == [0, 2, 3, 5]
    our_p[0].grad = None                                            # what model.zero_grad(set_to_none=True) does
    with pytest.raises(DO._capi.DirHipError):
        our.step()


def test_dir_module_sees_an_optimiser_step():
    """train.py:87 validates after optimiser steps: DIR.forward must run the UPDATED weights"""
    import json
    from conftest import GOLDEN
    from dir_amd import synth
    from dir_amd.models.dir import DIR
    with open(os.path.join(GOLDEN, 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    net = DIR(21, './misc/mano', 0)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    img = torch.from_numpy(synth.synth_input('dir.img', (1, 3, 256, 256), 1234)).cuda()
    a = net({'img': img}, None, None)[0][2]['pd_mesh_xyz_left'].clone()
    opt = DO.FlatAdamW([p for p in net.parameters() if p.requires_grad], 1e-2)
    opt.flat_grad.fill_(1.0)
    opt.step()
    b = net({'img': img}, None, None)[0][2]['pd_mesh_xyz_left'].clone()
    assert not torch.equal(a, b)
    c = net({'img': img}, None, None)[0][2]['pd_mesh_xyz_left']
    assert torch.equal(b, c)                                        # and the re-packed engine is cached again
