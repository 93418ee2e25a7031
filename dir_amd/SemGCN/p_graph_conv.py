"""Drop-in for SemGCN/p_graph_conv.py:9-62 (PGraphConv): same parameters (W [2,J,Cin,Cout], e_0, e_1, bias) and
forward signature, computed by dir_pgcn_stack_forward (dir_amd/csrc/tokens.hip).  The HIP kernel is specialised to
what the network instantiates: the 21-joint hand skeleton graph and 128 -> 128 features (models/dir.py:22-28)."""
import math

import torch
import torch.nn as nn

from .. import _capi
from .utils import adj_mx_from_edges, get_sketch_setting


def _check_hand_graph(adj, in_features, out_features):
    ref = adj_mx_from_edges(21, get_sketch_setting(), sparse=False, eye=False) > 0
    if adj.shape != (21, 21) or not torch.equal(adj.cpu() > 0, ref) or in_features != 128 or out_features != 128:
        raise NotImplementedError('dir_amd PGraphConv is built for the hand-skeleton graph (21 nodes, 20 bones) with '
                                  '128 -> 128 features, the only configuration on the DIR path')


class PGraphConv(nn.Module):
    def __init__(self, in_features, out_features, adj, bias=True):
        super().__init__()
        _check_hand_graph(adj, in_features, out_features)
        self.in_features, self.out_features = in_features, out_features
        self.W = nn.Parameter(torch.zeros(size=(2, adj.size(0), in_features, out_features), dtype=torch.float))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.adj_0 = torch.eye(adj.size(0), dtype=torch.float)
        self.m_0 = (self.adj_0 > 0)
        self.e_0 = nn.Parameter(torch.ones(1, int(self.m_0.sum()), dtype=torch.float))   # dead parameter (A_0 == I)
        self.adj_1 = adj
        self.m_1 = (self.adj_1 > 0)
        self.e_1 = nn.Parameter(torch.ones(1, int(self.m_1.sum()), dtype=torch.float))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features, dtype=torch.float))
            stdv = 1. / math.sqrt(self.W.size(1))
            self.bias.data.uniform_(-stdv, stdv)
        else:
            self.register_parameter('bias', None)

    def c_layer(self, bn_scale, bn_shift, relu, keep):
        dev = self.W.device
        bias = self.bias if self.bias is not None else torch.zeros(128, device=dev)
        t = [_capi.f32c(self.W.detach()), _capi.f32c(self.e_1.detach().reshape(-1)), _capi.f32c(bias.detach()),
             bn_scale, bn_shift]
        keep.append(t)
        return _capi.PgcnLayer(*(x.data_ptr() for x in t), int(relu))

    def _train_forward(self, input):
        """.train(): SemGCN/p_graph_conv.py:39-59 with autograd (one node: dir_amd/train/pgcn.py gconv_forward / gconv_backward)"""
        from ..train import autograd as AG
        from ..train import pgcn as TP
        params = {'W': self.W, 'e_0': self.e_0, 'e_1': self.e_1}
        if self.bias is not None:
            params['bias'] = self.bias

        def fwd(P, xx):
            z, saved = TP.gconv_forward(P, '', _capi.f32c(xx))
            return (z,), saved

        def bwd(P, saved, gz):
            G = {}
            gx = TP.gconv_backward(P, '', saved, gz.reshape(-1, 128), G)
            return (gx,), G
        with torch.cuda.device(input.device):
            return AG.run(fwd, bwd, [input], params)[0]

    def forward(self, input):
        _capi.require_cuda(input, self.W)
        if self.training and torch.is_grad_enabled():
            return self._train_forward(input)
        x = _capi.f32c(input.detach())
        B = x.shape[0]
        keep = []
        one, zero = torch.ones(128, device=x.device), torch.zeros(128, device=x.device)
        layers = (_capi.PgcnLayer * 1)(self.c_layer(one, zero, False, keep))
        out = torch.empty(B, 21, 128, device=x.device)
        scratch = torch.empty(2, B, 21, 256, device=x.device)
        with torch.cuda.device(x.device):
            _capi.check(_capi.lib().dir_pgcn_stack_forward(layers, 1, _capi.ptr(x), None, _capi.ptr(out), 21 * 128,
                                                           _capi.ptr(scratch), B, _capi.stream_ptr()),
                        'dir_pgcn_stack_forward')
        return out

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
