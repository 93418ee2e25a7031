"""The token half of one refinement stage (Joint2BoneFeature.forward, models/dir.py:86-116) in TRAINING form -- batch-statistics
BatchNorm in the five token MLPs and the two P-GCNs, exactly what `train.py` runs -- forward that keeps what the backward needs, and
the backward, every arithmetic step a libdir_hip.so kernel (dir_amd/train/ops.py, dir_mano_*, dir_regress_backward).

    out, ctx = stage_tokens_forward(P, mano_tables_lr, feat_nhwc, prev)
        P      {state-dict key relative to the projecter ('img2joint_left.filters.0.weight', 'gcn_left.gconv_layers.0.gconv.W',
               'interaction.STEblocks.1.norm1.weight', 'regressor.mano_left.weight', ...) -> fp32 cuda tensor}; BatchNorm running
               statistics in P are updated like nn.BatchNorm1d does in train mode
        feat   fusion_feat NHWC fp32 [B,S,S,C]
        prev   the previous stage's outputs, all entering detached (models/dir.py:447-453): 'pd_joint_xyz_left/right' [B,21,3],
               'pd_joint_uv_left/right' [B,21,2], 'pd_mano_para_left/right' [B,64], 'pd_offset' [B,3]
        out    'pd_offset', 'pd_mano_para_*', 'pd_mesh_xyz_*', 'pd_joint_xyz_*', 'pd_joint_uv_*', 'pd_mesh_uv_*' and 'joint_feat' [B,42,64] (the STE output the image half re-embeds,
               models/dir.py:118-119)
    g_feat, grads = stage_tokens_backward(P, mano_tables_lr, ctx, cot, g_joint_feat=None)
        cot    cotangents of out: any of 'pd_offset' [B,3], 'pd_mano_para_left/right' [B,64], 'pd_mesh_xyz_*', 'pd_joint_xyz_*',
               'pd_joint_uv_*', 'pd_mesh_uv_*'  (what dir_stage_losses_backward produces, include/dir_hip.h)
        g_joint_feat  extra cotangent of the STE output from the image half (proj_feat_emb -> bone_proj), added before the STE backward
        g_feat NHWC [B,S,S,C]: the gradient the two samplers send into fusion_feat;  grads {key -> gradient} for every parameter on the path
The cuts: prev enters detached, so nothing flows to the previous stage; uv is not differentiated (models/dir.py:197).
"""
import torch

from . import ops as O
from . import pgcn as PG
from . import ste as ST
from .. import engine as E
from .. import functional as F

NJ, EMB, TOK = 21, 128, 64
SIDES = ('left', 'right')


# ---------------------------------------------------------------------------------------------------------------- Conv1d-BN-ReLU-Conv1d
def _w2(P, key):
    w = P[key]
    return w.reshape(w.shape[0], -1)


def mlp_forward(P, pre, x, out=None, accumulate=False, momentum=0.1, eps=1e-5):
    """Sequential(Conv1d k=1, BatchNorm1d (batch statistics), ReLU, Conv1d k=1) on token rows x [R, Cin] -> ([R, Cout], ctx)"""
    h = O.linear_fwd(x, _w2(P, pre + '0.weight'), P[pre + '0.bias'])
    a, st = O.bn_train_fwd(h, P[pre + '1.weight'], P[pre + '1.bias'], P.get(pre + '1.running_mean'), P.get(pre + '1.running_var'), eps, momentum, relu=True)
    y = O.linear_fwd(a, _w2(P, pre + '3.weight'), P[pre + '3.bias'], out=out, accumulate=accumulate)
    return y, dict(x=x, h=h, st=st, a=a)


def mlp_backward(P, pre, ctx, gy, G, need_gx=False):
    """accumulates into G when the module already has gradients there (a module run once per hand)"""
    acc = (pre + '3.weight') in G
    w3, w0 = P[pre + '3.weight'], P[pre + '0.weight']
    ga, gW3, gb3 = O.linear_bwd(gy, ctx['a'], _w2(P, pre + '3.weight'),
                                gW=G[pre + '3.weight'].view(w3.shape[0], -1) if acc else None, gb=G.get(pre + '3.bias'), accumulate=acc)
    gh, gw, gb = O.bn_train_bwd(ga, ctx['h'], P[pre + '1.weight'], ctx['st'], b=P[pre + '1.bias'], relu=True)     # BatchNorm1d + ReLU undone in one call (mask from h)
    gx, gW0, gb0 = O.linear_bwd(gh, ctx['x'], _w2(P, pre + '0.weight'),
                                gW=G[pre + '0.weight'].view(w0.shape[0], -1) if acc else None, gb=G.get(pre + '0.bias'), accumulate=acc, need_gx=need_gx)
    if acc:
        O.axpy(G[pre + '1.weight'], gw)
        O.axpy(G[pre + '1.bias'], gb)
    else:
        G[pre + '3.weight'], G[pre + '3.bias'] = gW3.view_as(w3), gb3
        G[pre + '0.weight'], G[pre + '0.bias'] = gW0.view_as(w0), gb0
        G[pre + '1.weight'], G[pre + '1.bias'] = gw, gb
    return gx


def _sub(P, pre):
    n = len(pre)
    return {k[n:]: v for k, v in P.items() if k.startswith(pre)}


# ------------------------------------------------------------------------------------------------------------------------------ forward
def stage_tokens_forward(P, mano_tables_lr, feat_nhwc, prev):
    B, S, _, C = feat_nhwc.shape
    dev = feat_nhwc.device
    c = lambda t: t.detach().float().contiguous()  # noqa: E731
    xyz = [c(prev['pd_joint_xyz_' + s]) for s in SIDES]
    uv = [c(prev['pd_joint_uv_' + s]) for s in SIDES]
    ppara = [c(prev['pd_mano_para_' + s]) for s in SIDES]
    poff = c(prev['pd_offset']).reshape(B, 3)
    pos_l, pos_r, gpos_l, gpos_r = O.stage_positions(xyz[0], xyz[1], poff)
    ctx = {'B': B, 'S': S, 'C': C, 'uv': uv, 'ppara': ppara, 'poff': poff, 'hand': []}
    cat = torch.empty(B, 2 * NJ, EMB, device=dev)
    for h, s in enumerate(SIDES):
        rows = O.grid_rows_fwd(feat_nhwc, uv[h])
        tok, c_img = mlp_forward(P, 'img2joint_%s.filters.' % s, rows)
        _, c_pos = mlp_forward(P, 'pos_emb_%s.' % s, (pos_l, pos_r)[h], out=tok, accumulate=True)               # pos + img (dir.py:100-101)
        y, c_gcn = PG.pgcn_forward(_sub(P, 'gcn_%s.' % s), tok.view(B, NJ, EMB))
        cat[:, h * NJ:(h + 1) * NJ].copy_(y)
        # + global_pos_emb (dir.py:106-110): the last Conv1d accumulates straight into this hand's rows of the concatenated token buffer
        pre = 'global_pos_emb.'
        g_in = (gpos_l, gpos_r)[h]
        hh = O.linear_fwd(g_in, _w2(P, pre + '0.weight'), P[pre + '0.bias'])
        a, st = O.bn_train_fwd(hh, P[pre + '1.weight'], P[pre + '1.bias'], P.get(pre + '1.running_mean'), P.get(pre + '1.running_var'), relu=True)
        O.gemm_strided(a, _w2(P, pre + '3.weight'), cat, NJ, EMB, EMB, EMB, EMB, EMB, tb=True, batch=B, sa=NJ * EMB, sb=0, sc=2 * NJ * EMB,
                       c_off=h * NJ * EMB, bias=P[pre + '3.bias'], accumulate=True)
        ctx['hand'].append(dict(img=c_img, pos=c_pos, gcn=c_gcn, gpos=dict(x=g_in, h=hh, st=st, a=a)))
    tok, c_ste = ST.ste_forward(_sub(P, 'interaction.'), cat)
    tok = tok.contiguous()                                                                                       # [B,42,64]
    ctx['ste'], ctx['tok'] = c_ste, tok
    # RegressorOffset (dir.py:340-352): three Linears on [tokens | previous estimate]; the concatenation is two accumulating GEMMs
    para = []
    for h, s in enumerate(SIDES):
        W, b = P['regressor.mano_%s.weight' % s], P['regressor.mano_%s.bias' % s]
        p = torch.empty(B, 64, device=dev)
        O.gemm_strided(tok, W, p, B, 64, NJ * TOK, 2 * NJ * TOK, W.shape[1], 64, tb=True, a_off=h * NJ * TOK, bias=b)
        O.gemm_strided(ppara[h], W, p, B, 64, 64, 64, W.shape[1], 64, tb=True, b_off=NJ * TOK, accumulate=True)
        para.append(p)
    W, b = P['regressor.offset.weight'], P['regressor.offset.bias']
    off = torch.empty(B, 3, device=dev)
    O.gemm_strided(tok, W, off, B, 3, 2 * NJ * TOK, 2 * NJ * TOK, W.shape[1], 3, tb=True, bias=b)
    O.gemm_strided(poff, W, off, B, 3, 3, 3, W.shape[1], 3, tb=True, b_off=2 * NJ * TOK, accumulate=True)
    mano = E.run_mano_pair(mano_tables_lr, para[0], para[1], B, mesh_uv=True)
    ctx['para'] = para
    out = {'pd_offset': off, 'joint_feat': tok}
    for h, s in enumerate(SIDES):
        out['pd_mano_para_' + s] = para[h]
        out['pd_mesh_xyz_' + s], out['pd_joint_xyz_' + s], out['pd_joint_uv_' + s], out['pd_mesh_uv_' + s] = mano[h]
    return out, ctx


# ----------------------------------------------------------------------------------------------------------------------------- backward
def stage_tokens_backward(P, mano_tables_lr, ctx, cot, g_joint_feat=None):
    B, S, C = ctx['B'], ctx['S'], ctx['C']
    dev = ctx['tok'].device
    G = {}

    def pair(key):
        ts = [cot.get(key + s) for s in SIDES]
        return None if all(t is None for t in ts) else ts
    g_para = F.mano_backward(list(mano_tables_lr), ctx['para'], g_verts=pair('pd_mesh_xyz_'), g_joints=pair('pd_joint_xyz_'),
                             g_joint_uv=pair('pd_joint_uv_'), g_mesh_uv=pair('pd_mesh_uv_'))
    for h, s in enumerate(SIDES):
        if cot.get('pd_mano_para_' + s) is not None:
            O.axpy(g_para[h], cot['pd_mano_para_' + s].float().contiguous())
    g_off = cot.get('pd_offset')
    g_off = torch.zeros(B, 3, device=dev) if g_off is None else g_off.float().contiguous()
    r = F.regress_backward(P['regressor.mano_left.weight'], P['regressor.mano_right.weight'], P['regressor.offset.weight'], ctx['tok'],
                           ctx['ppara'][0], ctx['ppara'][1], ctx['poff'], g_para[0], g_para[1], g_off)
    g_tok = r.pop('tok')
    for k, v in r.items():
        G['regressor.' + k] = v
    if g_joint_feat is not None:
        O.axpy(g_tok, g_joint_feat.float().contiguous())
    g_cat, g_ste = ST.ste_backward(_sub(P, 'interaction.'), ctx['ste'], g_tok)
    for k, v in g_ste.items():
        G['interaction.' + k] = v
    g_rows = []
    for h, s in enumerate(SIDES):
        hc = ctx['hand'][h]
        g_h = g_cat[:, h * NJ:(h + 1) * NJ].contiguous()                                      # [B,21,128]: g of (gcn out + gpos)
        mlp_backward(P, 'global_pos_emb.', hc['gpos'], g_h.view(B * NJ, EMB), G)
        g_x, g_gcn = PG.pgcn_backward(_sub(P, 'gcn_%s.' % s), hc['gcn'], g_h)
        for k, v in g_gcn.items():
            G['gcn_%s.%s' % (s, k)] = v
        g_x = g_x.contiguous().view(B * NJ, EMB)                                              # g of (pos + img)
        mlp_backward(P, 'pos_emb_%s.' % s, hc['pos'], g_x, G)
        g_rows.append(mlp_backward(P, 'img2joint_%s.filters.' % s, hc['img'], g_x, G, need_gx=True))
    g_feat = O.grid_rows_bwd(g_rows, ctx['uv'], B, S, C)
    return g_feat, G
