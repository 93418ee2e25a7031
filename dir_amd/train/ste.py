"""STE (transformer/mixSTE.py:194-205) in training form: forward that keeps what the backward needs, and the backward -- the
gradients torch autograd produces in the reference's training step (train.py:66-70), from libdir_hip.so kernels (dir_amd/train/ops.py).

    y, ctx = ste_forward(P, x)          P: {state-dict key -> fp32 cuda tensor} of one STE ('spatial_pos_embed', 'STEblocks.1.norm1.weight', ...)
    gx, grads = ste_backward(P, ctx, gy) grads: {state-dict key -> gradient}; block 0 is never executed (mixSTE.py:197): its parameters
                                         get no gradient, exactly as in the reference (torch leaves .grad = None)
"""
import torch

from . import ops as O

HEADS, EPS_BLOCK, EPS_HEAD = 4, 1e-6, 1e-5          # mixSTE.py:160,177,190


def ste_forward(P, x, depth=4):
    """x [B,T,C] fp32 (modified in place by the positional embedding, like the reference's `x += pos`) -> (y [B,T,out], ctx)"""
    B, T, C = x.shape
    R = B * T
    scale = (C // HEADS) ** -0.5
    # x += pos: the broadcast add is a rank-1 "GEMM" onto the trunk: ones[B*T... done as B rows of one [1, T*C] matrix
    ones = torch.ones(B, 1, device=x.device)
    O.gemm(ones, P['spatial_pos_embed'].reshape(1, T * C), out=x.view(B, T * C), accumulate=True)
    cur = x.view(R, C)
    ctx = {'B': B, 'T': T, 'C': C, 'blocks': [], 'depth': depth}
    for i in range(1, depth):
        p = 'STEblocks.%d.' % i
        a, st1 = O.layernorm_fwd(cur, P[p + 'norm1.weight'], P[p + 'norm1.bias'], EPS_BLOCK)
        qkv = O.linear_fwd(a, P[p + 'attn.qkv.weight'], P[p + 'attn.qkv.bias'])
        o, probs = O.attention_fwd(qkv, B, T, HEADS, scale)
        x1 = cur.clone()
        O.linear_fwd(o, P[p + 'attn.proj.weight'], P[p + 'attn.proj.bias'], out=x1, accumulate=True)           # x + proj(attn)
        m, st2 = O.layernorm_fwd(x1, P[p + 'norm2.weight'], P[p + 'norm2.bias'], EPS_BLOCK)
        h = O.linear_fwd(m, P[p + 'mlp.fc1.weight'], P[p + 'mlp.fc1.bias'])
        g = O.gelu_fwd(h)
        x2 = x1.clone()
        O.linear_fwd(g, P[p + 'mlp.fc2.weight'], P[p + 'mlp.fc2.bias'], out=x2, accumulate=True)               # x + mlp
        x3, st3 = O.layernorm_fwd(x2, P['spatial_norm.weight'], P['spatial_norm.bias'], EPS_BLOCK)
        ctx['blocks'].append(dict(x0=cur, a=a, st1=st1, qkv=qkv, probs=probs, o=o, x1=x1, m=m, st2=st2, h=h, g=g, x2=x2, st3=st3))
        cur = x3
    hn, sth = O.layernorm_fwd(cur, P['head.0.weight'], P['head.0.bias'], EPS_HEAD)
    y = O.linear_fwd(hn, P['head.1.weight'], P['head.1.bias'])
    ctx.update(xl=cur, hn=hn, sth=sth, scale=scale)
    return y.view(B, T, -1), ctx


def ste_backward(P, ctx, gy):
    """gy [B,T,out] -> (g x [B,T,C], {key: gradient})"""
    B, T, C, scale = ctx['B'], ctx['T'], ctx['C'], ctx['scale']
    R = B * T
    G = {}
    gy = gy.reshape(R, -1).contiguous()
    ghn, G['head.1.weight'], G['head.1.bias'] = O.linear_bwd(gy, ctx['hn'], P['head.1.weight'])
    gx, G['head.0.weight'], G['head.0.bias'] = O.layernorm_bwd(ghn, ctx['xl'], P['head.0.weight'], ctx['sth'])
    first_sn = True
    for i in range(ctx['depth'] - 1, 0, -1):
        p, s = 'STEblocks.%d.' % i, ctx['blocks'][i - 1]
        # spatial_norm (shared by the three blocks: its parameter gradients accumulate)
        gsw, gsb = G.get('spatial_norm.weight'), G.get('spatial_norm.bias')
        gx2, gsw, gsb = O.layernorm_bwd(gx, s['x2'], P['spatial_norm.weight'], s['st3'], gw=gsw, gb=gsb, accumulate_wb=not first_sn)
        G['spatial_norm.weight'], G['spatial_norm.bias'], first_sn = gsw, gsb, False
        # x2 = x1 + fc2(gelu(fc1(LN2(x1))))
        gg, G[p + 'mlp.fc2.weight'], G[p + 'mlp.fc2.bias'] = O.linear_bwd(gx2, s['g'], P[p + 'mlp.fc2.weight'])
        gh = O.gelu_bwd(gg, s['h'])
        gm, G[p + 'mlp.fc1.weight'], G[p + 'mlp.fc1.bias'] = O.linear_bwd(gh, s['m'], P[p + 'mlp.fc1.weight'])
        gx1, G[p + 'norm2.weight'], G[p + 'norm2.bias'] = O.layernorm_bwd(gm, s['x1'], P[p + 'norm2.weight'], s['st2'], gx=gx2, accumulate_x=True)
        # x1 = x0 + proj(attention(qkv(LN1(x0))))
        go, G[p + 'attn.proj.weight'], G[p + 'attn.proj.bias'] = O.linear_bwd(gx1, s['o'], P[p + 'attn.proj.weight'])
        gqkv = O.attention_bwd(s['qkv'], s['probs'], go, B, T, HEADS, scale)
        ga, G[p + 'attn.qkv.weight'], G[p + 'attn.qkv.bias'] = O.linear_bwd(gqkv, s['a'], P[p + 'attn.qkv.weight'])
        gx, G[p + 'norm1.weight'], G[p + 'norm1.bias'] = O.layernorm_bwd(ga, s['x0'], P[p + 'norm1.weight'], s['st1'], gx=gx1, accumulate_x=True)
    G['spatial_pos_embed'] = O.colsum(gx.view(B, T * C)).view(1, T, C)
    return gx.view(B, T, C), G
