"""Drop-in for SemGCN/p_gcn.py:7-27,63-73 (_GraphConv, ResSimplePGCN): the 4-layer P-GCN stack of the network in
one C-ABI call (one launch per layer; BN+ReLU folded into the next layer's staging prologue).  In .train() mode the stack runs through
dir_amd/train/pgcn.py behind one autograd node (batch statistics, gradients for every parameter and the input)."""
import torch
import torch.nn as nn

from .. import _capi
from .p_graph_conv import PGraphConv


def _bn_fold(bn):
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
    b = bn.bias.detach().double() - bn.running_mean.double() * s
    return s.float().contiguous(), b.float().contiguous()


class _GraphConv(nn.Module):
    def __init__(self, adj, input_dim, output_dim, p_dropout=None):
        super().__init__()
        if p_dropout is not None:
            raise NotImplementedError('dropout is not used on the DIR path')
        self.gconv = PGraphConv(input_dim, output_dim, adj)
        self.bn = nn.BatchNorm1d(output_dim)
        self.relu = nn.ReLU()
        self.dropout = None

    def forward(self, x):
        return _run_stack([self], x)


def _train_stack(layers, x):
    """training mode (batch-statistics BatchNorm1d, running statistics updated, autograd through ONE node): dir_amd/train/pgcn.py's forward /
    backward -- the kernels of the whole-network training step -- behind torch.autograd (the reference's module is
    trainable: SemGCN/p_gcn.py:20-27 under train.py:64-70)"""
    from ..train import autograd as AG
    from ..train import pgcn as TP
    params, buffers = {}, {}
    for i, l in enumerate(layers):
        p = 'gconv_layers.%d.' % i
        params.update({p + 'gconv.W': l.gconv.W, p + 'gconv.e_0': l.gconv.e_0, p + 'gconv.e_1': l.gconv.e_1, p + 'bn.weight': l.bn.weight, p + 'bn.bias': l.bn.bias})
        if l.gconv.bias is not None:
            params[p + 'gconv.bias'] = l.gconv.bias
        buffers.update({p + 'bn.running_mean': l.bn.running_mean, p + 'bn.running_var': l.bn.running_var})
        if l.bn.num_batches_tracked is not None:
            l.bn.num_batches_tracked += 1
    bn0 = layers[0].bn
    mom = 0.1 if bn0.momentum is None else bn0.momentum

    def fwd(P, xx):
        y, ctx = TP.pgcn_forward(P, _capi.f32c(xx), num_layers=len(layers), momentum=mom, eps=bn0.eps)
        return (y,), ctx

    def bwd(P, ctx, gy):
        gx, G = TP.pgcn_backward(P, ctx, gy)
        return (gx,), G
    with torch.cuda.device(x.device):
        return AG.run(fwd, bwd, [x], params, buffers)[0]


def _run_stack(layers, x):
    if any(l.training for l in layers):
        if not all(l.training for l in layers):
            raise NotImplementedError('dir_amd P-GCN stack: all layers in .train() or all in .eval()')
        _capi.require_cuda(x)
        return _train_stack(layers, x)
    _capi.require_cuda(x)
    x = _capi.f32c(x.detach())
    B, keep = x.shape[0], []
    arr = (_capi.PgcnLayer * len(layers))()
    for i, l in enumerate(layers):
        s, b = _bn_fold(l.bn)
        arr[i] = l.gconv.c_layer(s, b, True, keep)
    out = torch.empty(B, 21, 128, device=x.device)
    scratch = torch.empty(2, B, 21, 256, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dir_pgcn_stack_forward(arr, len(layers), _capi.ptr(x), None, _capi.ptr(out), 21 * 128,
                                                       _capi.ptr(scratch), B, _capi.stream_ptr()),
                    'dir_pgcn_stack_forward')
    return out


class ResSimplePGCN(nn.Module):
    def __init__(self, adj, hidden_dim, num_layers=4):
        super().__init__()
        self.gconv_layers = nn.Sequential(*[_GraphConv(adj, hidden_dim, hidden_dim) for _ in range(num_layers)])

    def forward(self, x):
        return _run_stack(list(self.gconv_layers), x)
