"""Host-side mirror of the optimiser side of the reference's train.py, on libdir_hip.so.

  FlatAdamW            train.py:227   optim.AdamW([{'params': model.parameters(), 'initial_lr': lr}], lr): same defaults
                                       (betas (0.9, 0.999), eps 1e-8, weight_decay 0.01), same `state_dict()` layout, so a
                                       reference checkpoint's 'optimizer' entry loads and a saved one resumes under torch
  CosineAnnealingLR    train.py:229-230 optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min=0) (closed form)
  save_checkpoint / load_checkpoint    train.py:127-149  {'net', 'optimizer', 'schedule', 'last_epoch'}
The parameters are moved into ONE contiguous fp32 buffer (each tensor becomes a view of it), gradients into a second one
(`.grad` of each parameter is a view), the two Adam moments are flat as well: `step()` is a single dir_adamw_step launch over
all of them, and `flat_grad` is the bucket a data-parallel all-reduce would reduce in one call.
The gradient views are filled by dir_amd.train (train/step.py::train_step, or `.backward()` on the losses of the mirror DIR in
training mode).
"""
import math

import torch

from . import _capi


class FlatAdamW(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        groups = list(params)
        if groups and isinstance(groups[0], dict):            # the reference passes one param group with 'initial_lr'
            assert len(groups) == 1, 'one parameter group (train.py:227)'
            self.extra = {k: v for k, v in groups[0].items() if k != 'params'}
            groups = list(groups[0]['params'])
        else:
            self.extra = {}
        self.params = [p for p in groups]
        assert self.params, 'no parameters'
        dev = self.params[0].device
        _capi.require_cuda(*self.params)
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
        # 16-byte aligned slots so that every view keeps vector alignment
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev)
        self.flat_grad = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat_param[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = self.flat_param[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, **self.extra)]      # schedulers write param_groups[0]['lr']
        self.step_count = 0
        self.active = [True] * len(self.params)
        self._ranges = [(0, n)]

    def set_inactive(self, params):
        """Parameters that never receive a gradient (torch.optim.AdamW skips `p.grad is None`: no state, no weight decay -- in the
        reference backbone.fc.* and interaction.STEblocks.0.*, never executed; NOT PGraphConv's e_0, whose gradient is a zero TENSOR there and
        which therefore decays -- dir_amd.train.step.inactive_parameters() is the authoritative list):
        step() leaves their slots alone and state_dict() omits them, as torch does."""
        ids = {id(p) for p in params}
        self.active = [a and id(p) not in ids for a, p in zip(self.active, self.params)]
        self._ranges, cur = [], None
        ends = self.offsets[1:] + [self.numel]
        for a, o, e in zip(self.active, self.offsets, ends):
            if a:
                cur = (cur[0], e) if cur is not None and cur[1] == o else (o, e)
                if self._ranges and self._ranges[-1][0] == cur[0]:
                    self._ranges[-1] = cur
                else:
                    self._ranges.append(cur)
            else:
                cur = None

    def zero_grad(self, set_to_none=False):
        # the gradients live in flat_grad (the all-reduce bucket): they are zeroed in place, never detached
        self.flat_grad.zero_()

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                raise _capi.DirHipError('FlatAdamW: a parameter\'s .grad no longer aliases flat_grad (zero_grad(set_to_none=True) on the '
                                        'model, or .grad reassigned): use FlatAdamW.zero_grad()')
        with torch.cuda.device(self.flat_param.device):
            for a, b in self._ranges:
                off = lambda t: _capi.C.c_void_p(t.data_ptr() + 4 * a)       # noqa: E731
                _capi.check(_capi.lib().dir_adamw_step(off(self.flat_param), off(self.flat_grad), off(self.exp_avg), off(self.exp_avg_sq),
                                                       b - a, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                                       float(g['weight_decay']), self.step_count, _capi.stream_ptr()), 'dir_adamw_step')
        # the kernel writes through raw pointers: bump every parameter's version counter as an in-place torch op would, so that
        # caches keyed on (data_ptr, _version) -- DIR.engine()'s packed weights -- see the update
        torch._C._increment_version(self.params)

    # ---- torch.optim.Optimizer.state_dict() layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..]}]}
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                if not self.active[i]:
                    continue
                sl = slice(o, o + p.numel())
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.exp_avg[sl].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[sl].view(p.shape).clone()}
        g = dict(self.param_groups[0])
        g.setdefault('amsgrad', False)
        g['params'] = list(range(len(self.params)))
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        assert len(groups) == 1 and len(groups[0]['params']) == len(self.params), 'parameter count mismatch'
        g = {k: v for k, v in groups[0].items() if k != 'params'}
        g['betas'] = tuple(g['betas'])
        assert not g.get('amsgrad', False), 'amsgrad is not built'
        self.param_groups = [g]
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            steps.add(int(float(st['step'])))
            sl = slice(o, o + p.numel())
            self.exp_avg[sl].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[sl].copy_(st['exp_avg_sq'].reshape(-1))
        assert len(steps) <= 1, 'per-parameter step counts differ: not a state this optimiser can continue'
        self.step_count = steps.pop() if steps else 0


class CosineAnnealingLR(object):
    """optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min) as train.py:229-230,84 uses it: step() once per epoch,
    lr = eta_min + (base_lr - eta_min) (1 + cos(pi epoch / T_max)) / 2 (torch's closed form)."""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.optimizer, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lrs = [g.get('initial_lr', g['lr']) for g in optimizer.param_groups]
        for g, b in zip(optimizer.param_groups, self.base_lrs):
            g.setdefault('initial_lr', b)
        self.last_epoch = last_epoch
        self.step()

    def get_lr(self):
        return [self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2 for b in self.base_lrs]

    def get_last_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g['lr'] = lr

    def state_dict(self):
        return {'T_max': self.T_max, 'eta_min': self.eta_min, 'base_lrs': list(self.base_lrs), 'last_epoch': self.last_epoch,
                '_step_count': self.last_epoch + 1, '_last_lr': self.get_last_lr()}

    def load_state_dict(self, sd):
        self.T_max, self.eta_min = sd['T_max'], sd['eta_min']
        self.base_lrs, self.last_epoch = list(sd['base_lrs']), sd['last_epoch']
        for g, lr in zip(self.optimizer.param_groups, sd.get('_last_lr', self.get_lr())):
            g['lr'] = lr


def save_checkpoint(path, model, optimizer, schedule, epoch):
    """train.py:137-149"""
    torch.save({'net': model.state_dict(), 'optimizer': optimizer.state_dict(), 'schedule': schedule.state_dict(),
                'last_epoch': epoch}, path)


def load_checkpoint(path, model, optimizer, schedule):
    """train.py:127-135 -> start_epoch"""
    ck = torch.load(path, map_location='cpu', weights_only=False)
    model.load_state_dict(ck['net'])
    optimizer.load_state_dict(ck['optimizer'])
    schedule.load_state_dict(ck['schedule'])
    return ck['last_epoch'] + 1
