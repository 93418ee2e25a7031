"""Drop-in for transformer/mixSTE.py:11-205 (Mlp, Attention, Block, STE): identical parameter tree / state-dict keys,
STE.forward runs as ONE HIP kernel launch (dir_ste_forward, dir_amd/csrc/ste.hip).  Like the reference, STE.forward
adds spatial_pos_embed to its input IN PLACE and never executes STEblocks[0] (transformer/mixSTE.py:196-197)."""
from functools import partial

import torch
import torch.nn as nn

from .. import _capi
from ..engine import pack_ste


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., comb=False, vis=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.comb, self.vis = comb, vis


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., attention=Attention, qkv_bias=False, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, comb=False, vis=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, comb=comb, vis=vis)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


class STE(nn.Module):
    def __init__(self, num_joints=17, in_chans=32, out_dim=32, depth=4, num_heads=4, mlp_ratio=2., qkv_bias=True,
                 qk_scale=None, norm_layer=None):
        super().__init__()
        if (num_joints, in_chans, out_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale) != (42, 128, 64, 4, 4, 2., True, None):
            raise NotImplementedError('dir_amd STE is built for the DIR configuration STE(42, 128, 64, depth=4) '
                                      '(models/dir.py:50)')
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, in_chans))
        self.block_depth = depth
        self.STEblocks = nn.ModuleList([Block(dim=in_chans, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                              qk_scale=qk_scale, norm_layer=norm_layer) for _ in range(depth)])
        self.spatial_norm = norm_layer(in_chans)
        self.head = nn.Sequential(nn.LayerNorm(in_chans), nn.Linear(in_chans, out_dim))

    def _train_forward(self, x):
        """.train(): transformer/mixSTE.py:194-205 with autograd -- dir_amd/train/ste.py's forward / backward (the whole-network training step's
        kernels) behind one node; x is updated in place (x += spatial_pos_embed, mixSTE.py:196) and marked dirty, STEblocks[0] gets no gradient
        like in the reference (it is never executed, mixSTE.py:197)"""
        from ..train import autograd as AG
        from ..train import ops as O
        from ..train import ste as TS
        params = dict(self.named_parameters())

        def fwd(P, xx):
            y, ctx = TS.ste_forward(P, xx, depth=self.block_depth)          # (xx IS x's storage: updated in place)
            return (y,), ctx

        def bwd(P, ctx, gy, g_dirty=None):
            gx, G = TS.ste_backward(P, ctx, gy)
            if g_dirty is not None:            # later uses of the updated x: d(x + pos)/dx = I, d/dpos = sum over the batch
                O.axpy(gx, g_dirty)
                O.axpy(G['spatial_pos_embed'].view(-1), O.colsum(g_dirty.view(g_dirty.shape[0], -1)))
            return (gx,), G
        with torch.cuda.device(x.device):
            return AG.run(fwd, bwd, [x], params, dirty_first=True)[0]

    def forward(self, x):
        _capi.require_cuda(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise _capi.DirHipError('STE.forward expects a contiguous float32 [B,42,128] tensor (it is updated in place)')
        if self.training and torch.is_grad_enabled():
            return self._train_forward(x)
        b = x.shape[0]
        keep = []
        sd = {'ste.' + k: v.detach() for k, v in self.state_dict().items()}
        P = pack_ste(sd, 'ste', keep, self.block_depth)
        y = torch.empty(b, 42, 64, device=x.device)
        import ctypes as C
        with torch.cuda.device(x.device):
            _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(x), _capi.ptr(x), _capi.ptr(y), b,
                                                    _capi.stream_ptr()), 'dir_ste_forward')
        return y.view(b, 42, -1)
