"""ResSimplePGCN (SemGCN/p_gcn.py:64-73: four _GraphConv = PGraphConv -> BatchNorm1d -> ReLU) in TRAINING form -- batch-statistics
BatchNorm, as `train.py` runs it -- forward that keeps what the backward needs, and the backward, from libdir_hip.so kernels.

    y, ctx = pgcn_forward(P, x)            P: {'gconv_layers.0.gconv.W', '.gconv.e_0', '.gconv.e_1', '.gconv.bias', '.bn.weight', '.bn.bias',
                                               '.bn.running_mean', '.bn.running_var', ...} fp32 cuda tensors; running statistics are updated
    gx, grads = pgcn_backward(P, ctx, gy)  grads for W, e_0 (identically zero: a one-entry softmax row, SemGCN/p_graph_conv.py:46,49), e_1,
                                           bias, bn.weight, bn.bias of every layer
PGraphConv.forward (SemGCN/p_graph_conv.py:39-59): h0 = x W_0[j], h1 = x W_1[j] per node, z = h0 + A_1 h1 + bias.
"""
import torch

from . import ops as O

NJ, C = 21, 128
_ZEROS = {}


def _zeros_like(t):
    """a shared all-zero gradient (read-only by contract: gradients in G are only ever read) instead of a fill launch per layer and hand"""
    key = (tuple(t.shape), t.device)
    z = _ZEROS.get(key)
    if z is None:
        z = torch.zeros_like(t)
        if not torch.cuda.is_current_stream_capturing():   # a tensor first made inside a capture lives in that graph's pool: never cache that one
            _ZEROS[key] = z
    return z                                               # shared and READ-ONLY by convention (tests/test_gpu_pgcn_bwd.py checks it stays zero)


def gconv_forward(P, p, cur):
    """PGraphConv.forward (SemGCN/p_graph_conv.py:39-59) of the layer whose parameters are P[p + 'W' | 'e_1' | 'bias']: cur [B,21,128]
    -> (z [B,21,128], saved for gconv_backward)"""
    B = cur.shape[0]
    W = P[p + 'W']
    z, h1 = torch.empty(B, NJ, C, device=cur.device), torch.empty(B, NJ, C, device=cur.device)      # h0 is written straight into z
    for k, h in enumerate((z, h1)):                  # per node j: [B,128] (pitch 21*128, offset 128 j) x W_k[j] [128,128]
        O.gemm_strided(cur, W, h, B, C, C, NJ * C, C, NJ * C, batch=NJ, sa=C, sb=C * C, sc=C, b_off=k * NJ * C * C)
    A1 = O.pgcn_adjacency(P[p + 'e_1'].reshape(-1).contiguous())
    # z[b] += A_1 h1[b] + bias: per sample [21,21] x [21,128]
    O.gemm_strided(A1, h1, z, NJ, C, NJ, NJ, C, C, batch=B, sa=0, sb=NJ * C, sc=NJ * C, bias=P.get(p + 'bias'), accumulate=True)
    return z, dict(x=cur, h1=h1, A1=A1)


def gconv_backward(P, p, s, gz, G):
    """gz [B*21,128] -> g x [B,21,128]; G[p + 'W' | 'e_0' | 'e_1' | 'bias'] filled"""
    B = s['x'].shape[0]
    if (p + 'bias') in P and P[p + 'bias'] is not None:
        G[p + 'bias'] = O.colsum(gz)
    gz3 = gz.view(B, NJ, C)
    e1 = P[p + 'e_1'].reshape(-1).contiguous()
    G[p + 'e_1'] = O.pgcn_adjacency_bwd(e1, gz3, s['h1']).view_as(P[p + 'e_1'])
    G[p + 'e_0'] = _zeros_like(P[p + 'e_0'])
    gh1 = torch.empty(B, NJ, C, device=gz.device)          # g h1[b] = A_1^T g z[b]
    O.gemm_strided(s['A1'], gz3, gh1, NJ, C, NJ, NJ, C, C, ta=True, batch=B, sa=0, sb=NJ * C, sc=NJ * C)
    W = P[p + 'W']
    gW = torch.empty_like(W)
    gx = torch.empty(B, NJ, C, device=gz.device)
    for k, gh in enumerate((gz3, gh1)):
        # g W_k[j] = x_j^T g h_k[:, j]  ([128 in, B] x [B, 128 out]);  g x_j (+)= g h_k[:, j] W_k[j]^T
        O.gemm_strided(s['x'], gh, gW, C, C, B, NJ * C, NJ * C, C, ta=True, batch=NJ, sa=C, sb=C, sc=C * C, c_off=k * NJ * C * C)
        O.gemm_strided(gh, W, gx, B, C, C, NJ * C, C, NJ * C, tb=True, batch=NJ, sa=C, sb=C * C, sc=C, b_off=k * NJ * C * C, accumulate=k == 1)
    G[p + 'W'] = gW
    return gx


def pgcn_forward(P, x, num_layers=4, momentum=0.1, eps=1e-5):
    B = x.shape[0]
    ctx = {'B': B, 'layers': []}
    cur = x.contiguous()
    for l in range(num_layers):
        p = 'gconv_layers.%d.' % l
        z, s = gconv_forward(P, p + 'gconv.', cur)
        # BatchNorm1d + ReLU in one launch; the backward re-computes the mask from z (dir_bn_train_backward(relu))
        y, st = O.bn_train_fwd(z.view(B * NJ, C), P[p + 'bn.weight'], P[p + 'bn.bias'], P.get(p + 'bn.running_mean'), P.get(p + 'bn.running_var'), eps, momentum, relu=True)
        s.update(z=z, st=st, y=y)
        ctx['layers'].append(s)
        cur = y.view(B, NJ, C)
    return cur, ctx


def pgcn_backward(P, ctx, gy):
    B = ctx['B']
    G = {}
    g = gy.contiguous().view(B * NJ, C)
    for l in range(len(ctx['layers']) - 1, -1, -1):
        p, s = 'gconv_layers.%d.' % l, ctx['layers'][l]
        gz, G[p + 'bn.weight'], G[p + 'bn.bias'] = O.bn_train_bwd(g.contiguous(), s['z'].view(B * NJ, C), P[p + 'bn.weight'], s['st'], b=P[p + 'bn.bias'], relu=True)
        g = gconv_backward(P, p + 'gconv.', s, gz, G).view(B * NJ, C)
    return g.view(B, NJ, C), G
