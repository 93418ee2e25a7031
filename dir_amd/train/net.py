"""DIR.forward in TRAINING mode and the gradient of the summed training objective w.r.t. every parameter -- what
`outs_list, loss = model(inputs, targets, meta_infos); sum(loss.values()).backward()` (train.py:66-68) computes -- composed from
libdir_hip.so kernels: dir_amd/train/{conv,blocks,spatial,stage,ste,pgcn,ops}.py, the loss gradients of dir_amd/models/loss.py and the
MANO / regressor backward.  fp32, NHWC, batch-statistics BatchNorm everywhere (running statistics in P are updated), deterministic.

    outs, ctx = forward(P, img_nchw, mano)            P: {DIR state-dict key -> fp32 cuda tensor, reference layouts}
    loss = losses(outs, target, meta_info, faces)     the 42 terms (dir_amd.models.loss.DirLoss)
    grads = backward(P, ctx, outs, target, meta_info, faces, mano)     {parameter key -> gradient of sum(loss)}

The reference's `.detach()` cuts (models/dir.py:447-453,463-469) are structural here: a stage receives the previous stage's outputs as
plain inputs.  Arithmetic (round 3): fp32 tensors; the convolutions' forward, data gradient and weight gradient in split precision on the
f16 matrix cores (dir_amd/train/conv.py: DIR_TRAIN_ARITH / DIR_TRAIN_WGRAD_ARITH = f16x3, 'f32' switches the exact kernels back), everything
else exact fp32; BatchNorm + ReLU (+ the bottleneck's residual) one launch each way.  bench.py times a step as its train_step sub-record.
"""
import os

import torch

from . import blocks as TB
from . import conv as TC
from . import hrnet as TH
from . import ops as O
from . import spatial as SP
from . import stage as TS
from .. import engine as E
from ..models import loss as L

SIDES = ('left', 'right')
LAYERS = (3, 4, 6, 3)
# bone_proj + fusion.0 in the factorised form, forward and backward (dir_bone_fusion_*, csrc/bonefuse_bwd.hip); '0': the [B,S,S,2560] map and
# the K = 23 040 convolution with its two gradient convolutions (rounds 2-3)
FACTORISED_FUSION = os.environ.get('DIR_TRAIN_FACTORISED_FUSION', '1') != '0'


def sub(P, pre):
    n = len(pre)
    return {k[n:]: v for k, v in P.items() if k.startswith(pre)}


def put(G, pre, g):
    for k, v in g.items():
        G[pre + k] = v


_MANO_KEYS = ('th_shapedirs', 'th_posedirs', 'th_v_template', 'th_J_regressor', 'th_weights', 'th_hands_mean', 'th_selected_comps')


def mano_tables(P, pre, keep, owner=None):
    """(left, right) dir_mano_tables of the two ManoLayers under `pre` ('init_regressor.' / 'decoder.projecter_4.regressor.').  The MANO tables are
    BUFFERS (manopth registers them, they are never trained): with an `owner` (the step's scale_owner) the packed forms -- two float64 products
    and a dozen copies per hand -- are made once and kept on it for as long as the buffers keep their storage and version (round 4: a step
    re-packed all six layers, ~100 torch launches)."""
    if owner is None:
        return [E.pack_mano(P, pre + 'mano_layer_' + s, s, 0, keep) for s in SIDES]
    cache = getattr(owner, '_dir_mano_cache', None)
    if cache is None:
        cache = {}
        setattr(owner, '_dir_mano_cache', cache)
    key = tuple((P[pre + 'mano_layer_' + s + '.' + k].data_ptr(), P[pre + 'mano_layer_' + s + '.' + k]._version) for s in SIDES for k in _MANO_KEYS)
    hit = cache.get(pre)
    if hit is None or hit[0] != key:
        held = []
        hit = cache[pre] = (key, [E.pack_mano(P, pre + 'mano_layer_' + s, s, 0, held) for s in SIDES], held)
    return hit[1]


# ------------------------------------------------------------------------------------------------------------------------------ pieces
def _stem_forward(P, img):
    """conv1 7x7/2 (no bias) through the engine's space-to-depth form (dir_stem_prep_s2d + a 4x4 stride-1 implicit GEMM, engine.stem_conv_op)"""
    from .. import _capi
    B = img.shape[0]
    assert tuple(img.shape[1:]) == (3, 256, 256), 'the training stem is written for 256 x 256 images (config.py: the reference trains on 256 x 256 crops); got %s' % (tuple(img.shape),)
    op = E.stem_conv_op(P['backbone.conv1.weight'], None, None, torch.float32)
    op.flags = 0                                           # raw convolution: BatchNorm (batch statistics) and ReLU follow as their own steps
    xp = torch.empty(B, 131, 132, 16, device=img.device)
    _capi.check(_capi.lib().dir_stem_prep_s2d(_capi.ptr(img), _capi.ptr(xp), B, 256, 256, 131, 132, _capi.DT_F32, _capi.stream_ptr()), 'dir_stem_prep_s2d')
    return op(xp)


def _cbr_forward(P, pre, x, k):
    """Sequential(Conv2d(k, pad k//2), BatchNorm2d, ReLU, Conv2d(1)) -- conv_final, seg, dense, attention_*, fusion (models/dir.py:57-62,227-241,404-419)"""
    st = []                                                # the BatchNorm's chunk partials from the convolution's epilogue (round 5)
    h = TC.conv_fwd(x, P[pre + '0.weight'], P.get(pre + '0.bias'), 1, k // 2, oihw=True, stats=st)
    a, s_bn = TB.bn_fwd(P, pre + '1.', h, relu=True, partials=st)
    y = TC.conv_fwd(a, P[pre + '3.weight'], P.get(pre + '3.bias'), oihw=True)
    return y, dict(x=x, bn=s_bn, a=a, k=k)


def _cbr_backward(P, pre, s, gy, G, need_gx=True):
    sp = TB.bn_bwd_spec(P, pre + '1.', s['bn'], True)
    g = TB._conv_bwd(P, pre + '3.', s['a'], gy, 1, 0, G, bn_bwd=sp)
    g = TB.bn_bwd(P, pre + '1.', s['bn'], g, G, relu=True, spec=sp)
    return TB._conv_bwd(P, pre + '0.', s['x'], g, 1, s['k'] // 2, G, need_gx=need_gx)


def _stage_image_forward(P, pre, tok, uv_l, uv_r, S, distance, want_vis=False):
    """re-embedding + bone rasterisation + fusion conv of a stage (models/dir.py:118-122)"""
    B = tok.shape[0]
    Ps = sub(P, pre)
    emb = torch.empty(B, 42, 64, device=tok.device)
    ctx_emb = []
    for h in range(2):                                     # proj_feat_emb runs once per hand (shared module, two BatchNorm calls)
        rows = tok[:, 21 * h:21 * (h + 1)].contiguous().view(B * 21, 64)
        y, c = TS.mlp_forward(Ps, 'proj_feat_emb.', rows)
        emb[:, 21 * h:21 * (h + 1)] = y.view(B, 21, 64)
        ctx_emb.append(c)
    vis = None
    if FACTORISED_FUSION:
        # bone_proj + fusion.0 as a K = 720 reduction in exact fp32 (dir_bone_fusion_*): no [B,S,S,2560] map, no 23 040-deep convolution
        fpre = pre + 'fusion.'
        h, c_bf = SP.bone_fusion_fwd(uv_l, uv_r, emb, SP.fusion_w_g(P[fpre + '0.weight']), P.get(fpre + '0.bias'), S, distance)
        a, pa, s_bn = TB.bn_relu_into_conv(P, fpre + '1.', h)          # fusion.1 / .2 applied where fusion.3 reads the map (round 5)
        img_feat = TC.conv_fwd(a, P[fpre + '3.weight'], P.get(fpre + '3.bias'), oihw=True, pre=pa)
        c_fus = dict(bf=c_bf, bn=s_bn, a=a, pa=pa)
        if want_vis:
            vis = SP.bone_proj_vis(uv_l, uv_r, emb, S, distance)
    else:
        bone = SP.bone_proj_fwd(uv_l, uv_r, emb, S, distance, want_vis=want_vis)
        if want_vis:
            bone, vis = bone
        img_feat, c_fus = _cbr_forward(P, pre + 'fusion.', bone, 3)
    return img_feat, dict(emb=emb, ctx_emb=ctx_emb, fus=c_fus, uv=(uv_l, uv_r), S=S, distance=distance, vis=vis)


def _stage_image_backward(P, pre, s, g_img_feat, G):
    """-> (g joint_feat [B,42,64], g uv_left, g uv_right)"""
    B = s['emb'].shape[0]
    if 'bf' in s['fus']:
        fpre, f = pre + 'fusion.', s['fus']
        sp = TB.bn_bwd_spec(P, fpre + '1.', f['bn'], True)
        g = TB._conv_bwd(P, fpre + '3.', f['a'], g_img_feat, 1, 0, G, pre=f.get('pa'), bn_bwd=sp)
        g = TB.bn_bwd(P, fpre + '1.', f['bn'], g, G, relu=True, spec=sp)
        g_w_g, g_emb, gul, gur = SP.bone_fusion_bwd(f['bf'], g)
        G[fpre + '0.weight'] = SP.fusion_w_g_grad_to_oihw(g_w_g)
        if (fpre + '0.bias') in P:
            G[fpre + '0.bias'] = O.colsum(g.view(-1, 256))
    else:
        g_bone = _cbr_backward(P, pre + 'fusion.', s['fus'], g_img_feat, G)
        g_emb, gul, gur = SP.bone_proj_bwd(s['uv'][0], s['uv'][1], s['emb'], g_bone, s['S'], s['distance'])
    Ps, Gs = sub(P, pre), {}
    g_tok = torch.empty(B, 42, 64, device=g_emb.device)
    for h in range(2):
        g_rows = TS.mlp_backward(Ps, 'proj_feat_emb.', s['ctx_emb'][h], g_emb[:, 21 * h:21 * (h + 1)].contiguous().view(B * 21, 64), Gs, need_gx=True)
        g_tok[:, 21 * h:21 * (h + 1)] = g_rows.view(B, 21, 64)
    put(G, pre, Gs)
    return g_tok, gul, gur


# ----------------------------------------------------------------------------------------------------------------------------- forward
def backbone_forward(P, img, ctx, pre='backbone.'):
    """ResNet.forward in training form (models/backbone/resnet.py:243-255): stem conv + bn1 + ReLU + max-pool + the 16 bottlenecks.
    -> [c1, c2, c3, c4] NHWC fp32; ctx receives what backbone_backward needs.  Used by the whole-network step below and by the stand-alone
    mirror module in .train() mode (dir_amd/models/backbone/resnet.py)."""
    if TH.is_hrnet(P, pre):                                # BASELINE config 5's backbone (no reference counterpart): dir_amd/train/hrnet.py
        return TH.hrnet_forward(P, img, ctx, pre)
    Pb = P if pre == 'backbone.' else {'backbone.' + k[len(pre):]: v for k, v in P.items() if k.startswith(pre)}
    h = _stem_forward(Pb, img)
    a, ctx['bn1'] = TB.bn_fwd(Pb, 'backbone.bn1.', h, relu=True)
    x = SP.maxpool_fwd(a)
    ctx['stem'] = (a, x)
    ctx['img'] = img
    feats, ctx['blocks'] = [], []
    for li, nb in enumerate(LAYERS):
        for bi in range(nb):
            p = 'backbone.layer%d.%d.' % (li + 1, bi)
            x, c = TB.bottleneck_forward(sub(Pb, p), x, 2 if (bi == 0 and li > 0) else 1)
            ctx['blocks'].append((p, c))
        feats.append(x)
    return feats


def backbone_backward(P, ctx, g_feats, G, flush=None, pre='backbone.'):
    """g_feats: gradients of [c1, c2, c3, c4] (None where a feature has no outside consumer).  Fills G['backbone.*'] (conv1 has no input gradient:
    the image is data)"""
    if 'hr' in ctx:
        return TH.hrnet_backward(P, ctx, g_feats, G, flush, pre)
    Pb = P if pre == 'backbone.' else {'backbone.' + k[len(pre):]: v for k, v in P.items() if k.startswith(pre)}
    if flush is None:
        flush = lambda g_: None      # noqa: E731
    g = None
    bi_end = len(ctx['blocks'])
    for li in (3, 2, 1, 0):
        if g is None:
            g = g_feats[li]
        elif g_feats[li] is not None:
            O.axpy(g, g_feats[li])
        masked, spec_mine = False, None                    # the gradient entering a layer's LAST block still needs that block's ReLU backward
        for k in range(LAYERS[li] - 1, -1, -1):
            bi_end -= 1
            p, c = ctx['blocks'][bi_end]
            # inside a layer a block's input IS the previous block's output: that block's ReLU backward is applied where this block's conv1 data
            # gradient is written (round 5); a layer's first block hands its gradient to the previous layer's tap, which the decoder's gradient joins first
            prev_y = ctx['blocks'][bi_end - 1][1]['y'] if (k > 0 and TB.FUSE_RELU_BWD) else None
            # ... and with the mask applied there, that gradient is the gradient of the previous block's bn3 output: its backward sums in the same epilogue
            pp, pc = ctx['blocks'][bi_end - 1] if prev_y is not None else (None, None)
            spec_prev = TB.bn_bwd_spec(sub(Pb, pp), 'bn3.', pc['bn3'], False) if prev_y is not None else None
            g, gb = TB.bottleneck_backward(sub(Pb, p), c, g, gy_masked=masked, mask_gx=prev_y, bn3_spec=spec_mine, prev_bn3=spec_prev)
            masked, spec_mine = prev_y is not None, spec_prev
            put(G, p, gb)
        flush(G)
    a, x_pool = ctx['stem']
    g = SP.maxpool_bwd(a, g)
    g = TB.bn_bwd(Pb, 'backbone.bn1.', ctx['bn1'], g, G, relu=True)
    img_nhwc = ctx['img'].permute(0, 2, 3, 1).contiguous()
    G['backbone.conv1.weight'] = TB._oihw(TC.conv_wgrad(img_nhwc, g, (64, 7, 7, 3), 2, 3))
    flush(G)


def forward(P, img, keep=None, scale_owner=None):
    """img NCHW fp32 [B,3,256,256] -> (outs: the three stage dicts + {'seg', 'dense'} NCHW, ctx).  scale_owner: the object that owns this
    model's cache of split-precision operand scales across steps (train_step passes the optimizer, DIR.forward the module); None = every
    convolution measures its scale on this batch (a host synchronisation per call site: tests, one-off evaluations)"""
    keep = [] if keep is None else keep
    # operand-scale cache and one-launch weight packing of THIS model (dir_amd/train/conv.py); the factorised fusion reads fusion.0.weight itself
    TC.begin_step(scale_owner, {k: v for k, v in P.items() if not k.endswith('fusion.0.weight')} if FACTORISED_FUSION else P)
    B = img.shape[0]
    dev = img.device
    ctx = {'img': img, 'keep': keep}                       # the packed MANO tables must outlive the backward pass (raw pointers in dir_mano_tables)
    # ---- backbone (models/backbone/resnet.py:243-255)
    feats = backbone_forward(P, img, ctx)
    c1, c2, c3, c4 = feats
    # ---- InitRegressor (models/dir.py:260-305)
    init, ctx['init'] = {}, {}
    pooled = []
    for h_i, s in enumerate(SIDES):
        logit, c_att = _cbr_forward(P, 'init_regressor.attention_%s.' % s, c4, 3)             # [B,8,8,1]; the Sigmoid lives in the pooling kernel
        p, attn, mean = SP.attn_pool_fwd(c4, logit.reshape(B, 64).contiguous(), want_mean=(h_i == 0))
        ctx['init'][s] = dict(att=c_att, attn=attn, pooled=p)
        pooled.append(p)
        if h_i == 0:
            ctx['init']['mean'] = mean
    init['pd_offset'] = O.linear_fwd(ctx['init']['mean'], P['init_regressor.offset.weight'], P['init_regressor.offset.bias'])
    para = [O.linear_fwd(pooled[i], P['init_regressor.mano_%s.weight' % s], P['init_regressor.mano_%s.bias' % s]) for i, s in enumerate(SIDES)]
    tabs0 = mano_tables(P, 'init_regressor.', keep, scale_owner)
    mano = E.run_mano_pair(tabs0, para[0], para[1], B, mesh_uv=True)
    for i, s in enumerate(SIDES):
        init['pd_mano_para_' + s] = para[i]
        init['pd_mesh_xyz_' + s], init['pd_joint_xyz_' + s], init['pd_joint_uv_' + s], init['pd_mesh_uv_' + s] = mano[i]
    ctx['init'].update(para=para, tabs=tabs0)
    # ---- decoder (models/dir.py:437-483)
    outs = [init]
    prev, feat_lo, skip_src = init, c4, (c3, c2)
    ctx['dec'] = []
    for si, (tag, S, dist) in enumerate((('4', 16, 1), ('3', 32, 2))):
        d = {}
        Cup = feat_lo.shape[3]
        skip, d['skip'] = TB.residual_forward(sub(P, 'decoder.skip_layer%s.' % tag), skip_src[si])
        cat = torch.empty(B, S, S, Cup + 256, device=dev)
        SP.upsample_fwd(feat_lo, out=cat, out_coff=0)
        cat[..., Cup:] = skip
        fusion_feat, d['fusion'] = TB.residual_forward(sub(P, 'decoder.fusion_layer%s.' % tag), cat)
        pre = 'decoder.projecter_%s.' % tag
        tabs = mano_tables(P, pre + 'regressor.', keep, scale_owner)
        res, d['tok'] = TS.stage_tokens_forward(sub(P, pre), tabs, fusion_feat, prev)
        img_feat, d['img'] = _stage_image_forward(P, pre, res['joint_feat'], res['pd_joint_uv_left'], res['pd_joint_uv_right'], S, dist,
                                                  want_vis=(si == 1 and 'decoder.projecter_x.0.fusion.0.weight' not in P))     # the LAST stage's map is an output (models/dir.py:481)
        enh_in = torch.cat((fusion_feat, img_feat), dim=3)
        feat_lo, d['enh'] = TB.residual_forward(sub(P, 'decoder.enhance_layer%s.' % tag), enh_in)
        d.update(tabs=tabs, Cup=Cup, S=S)
        ctx['dec'].append(d)
        outs.append(res)
        prev = res
    # ---- N more refinement stages at 32 x 32 (this build's extension, config 5: decoder.projecter_x.<i> / enhance_layer_x.<i>; the reference's
    #      classes chained the way its forward chains its two stages -- oracle/gen_golden.py::reference_with_extra_stages, goldens G7x / G21)
    ctx['extra'] = []
    while ('decoder.projecter_x.%d.fusion.0.weight' % len(ctx['extra'])) in P:
        i, d = len(ctx['extra']), {}
        pre = 'decoder.projecter_x.%d.' % i
        tabs = mano_tables(P, pre + 'regressor.', keep, scale_owner)
        x_in = feat_lo
        res, d['tok'] = TS.stage_tokens_forward(sub(P, pre), tabs, x_in, prev)
        last = ('decoder.projecter_x.%d.fusion.0.weight' % (i + 1)) not in P
        img_feat, d['img'] = _stage_image_forward(P, pre, res['joint_feat'], res['pd_joint_uv_left'], res['pd_joint_uv_right'], 32, 2, want_vis=last)
        feat_lo, d['enh'] = TB.residual_forward(sub(P, 'decoder.enhance_layer_x.%d.' % i), torch.cat((x_in, img_feat), dim=3))
        d.update(tabs=tabs)
        ctx['extra'].append(d)
        outs.append(res)
        prev = res
    feat, ctx['final'] = _cbr_forward(P, 'decoder.conv_final.', feat_lo, 3)
    seg, ctx['seg'] = _cbr_forward(P, 'decoder.seg.', feat, 3)
    dense, ctx['dense'] = _cbr_forward(P, 'decoder.dense.', feat, 3)
    outs.append({'seg': seg.permute(0, 3, 1, 2).contiguous(), 'dense': dense.permute(0, 3, 1, 2).contiguous(),
                 'proj_feat': (ctx['extra'][-1] if ctx['extra'] else ctx['dec'][1])['img']['vis']})      # the LAST stage's map (models/dir.py:481)
    ctx.update(feats=feats)
    return outs, ctx


def losses(outs, target, meta_info, faces):
    return L.DirLoss(faces[0], faces[1])(outs[:-1], outs[-1], target, meta_info)


# ---------------------------------------------------------------------------------------------------------------------------- backward
def _term_weights(grad_out, n_stages=3):
    """grad_out: None (every term weight 1: `sum(loss.values()).backward()`, train.py:68) or {loss key: 0-d tensor / float} -- the upstream
    gradient of each of the 3 + 13 n_stages terms -> (dense 3-vector or None, [n_stages 13-vectors or None])"""
    if grad_out is None:
        return None, [None] * n_stages
    dev = next(v for v in grad_out.values() if torch.is_tensor(v)).device
    f = lambda k: (grad_out[k].to(dev).float().reshape(()) if k in grad_out and grad_out[k] is not None else torch.zeros((), device=dev))  # noqa: E731
    dense = torch.stack([f('seg'), f('dense'), f('lovasz')])
    return dense, [torch.stack([f('%s_%d' % (k, i)) for k in L.STAGE_KEYS]) for i in range(n_stages)]


def backward(P, ctx, outs, target, meta_info, faces, grad_out=None, flush=None):
    """gradient of sum_k grad_out[k] * loss[k] (grad_out None: all 42 terms with weight 1, train.py:68) -> {parameter key: gradient}.
    flush: optional callable(G), called whenever a group of modules has been finished -- every key present in G at that point is FINAL
    (each parameter's gradient is written once, after all its uses), which lets the caller move gradients into the data-parallel
    bucket and start its all-reduce while the rest of the backward pass runs (dir_amd/train/step.py)."""
    G = {}
    side = TC.side_begin()                                  # convolution weight gradients on a second stream (dir_amd/train/conv.py)
    if flush is None:
        flush = lambda g: None      # noqa: E731
    elif side is not None:
        caller_flush = flush
        flush = lambda g: TC.side_run(lambda: caller_flush(g))      # noqa: E731  (the moves into the bucket follow the weight gradients on their stream)
    w_dense, w_stage = _term_weights(grad_out, len(outs) - 1)
    B = ctx['img'].shape[0]
    c1, c2, c3, c4 = ctx['feats']
    g_seg, g_dense = L.dense_loss_grads(outs[-1]['seg'], outs[-1]['dense'], target['seg'], target['dense'], grad_out=w_dense)
    g_feat = _cbr_backward(P, 'decoder.seg.', ctx['seg'], g_seg.permute(0, 2, 3, 1).contiguous(), G)
    O.axpy(g_feat, _cbr_backward(P, 'decoder.dense.', ctx['dense'], g_dense.permute(0, 2, 3, 1).contiguous(), G))
    g_lo = _cbr_backward(P, 'decoder.conv_final.', ctx['final'], g_feat, G)                      # gradient of the last enhance layer's output
    flush(G)
    for i in range(len(ctx.get('extra', ())) - 1, -1, -1):                                       # the extra stages, last first
        d, pre = ctx['extra'][i], 'decoder.projecter_x.%d.' % i
        g_cat, g = TB.residual_backward(sub(P, 'decoder.enhance_layer_x.%d.' % i), d['enh'], g_lo)
        put(G, 'decoder.enhance_layer_x.%d.' % i, g)
        g_map = g_cat[..., :256].contiguous()                                                    # the running map enters the Residual directly ...
        g_tok, gul, gur = _stage_image_backward(P, pre, d['img'], g_cat[..., 256:].contiguous(), G)
        cot = L.stage_loss_grads(outs[3 + i], target, meta_info, faces, grad_out=w_stage[3 + i])
        O.axpy(cot['pd_joint_uv_left'], gul)
        O.axpy(cot['pd_joint_uv_right'], gur)
        g_samp, g = TS.stage_tokens_backward(sub(P, pre), d['tabs'], d['tok'], cot, g_joint_feat=g_tok)
        put(G, pre, g)
        O.axpy(g_map, g_samp)                                                                    # ... and is what the stage samples its tokens from
        g_lo = g_map
        flush(G)
    g_skip_src = [None, None]
    for si in (1, 0):
        tag, d = ('4', '3')[si], ctx['dec'][si]
        pre = 'decoder.projecter_%s.' % tag
        S, Cup = d['S'], d['Cup']
        g_enh_in, g = TB.residual_backward(sub(P, 'decoder.enhance_layer%s.' % tag), d['enh'], g_lo)
        put(G, 'decoder.enhance_layer%s.' % tag, g)
        g_fusion = g_enh_in[..., :256].contiguous()
        g_tok, gul, gur = _stage_image_backward(P, pre, d['img'], g_enh_in[..., 256:].contiguous(), G)
        cot = L.stage_loss_grads(outs[si + 1], target, meta_info, faces, grad_out=w_stage[si + 1])
        O.axpy(cot['pd_joint_uv_left'], gul)                                                     # bone_proj reads the stage's own (not detached) uv
        O.axpy(cot['pd_joint_uv_right'], gur)
        g_samp, g = TS.stage_tokens_backward(sub(P, pre), d['tabs'], d['tok'], cot, g_joint_feat=g_tok)
        put(G, pre, g)
        O.axpy(g_fusion, g_samp)
        g_cat, g = TB.residual_backward(sub(P, 'decoder.fusion_layer%s.' % tag), d['fusion'], g_fusion)
        put(G, 'decoder.fusion_layer%s.' % tag, g)
        g_lo = SP.upsample_bwd(g_cat, Cup, 0)                                                    # -> enhance_layer4's output (si = 1) or c4 (si = 0)
        g_src, g = TB.residual_backward(sub(P, 'decoder.skip_layer%s.' % tag), d['skip'], g_cat[..., Cup:].contiguous())
        put(G, 'decoder.skip_layer%s.' % tag, g)
        g_skip_src[si] = g_src                                                                    # gradient into c3 (si = 0) / c2 (si = 1)
        flush(G)
    g_c4 = g_lo
    # ---- InitRegressor
    ci = ctx['init']
    cot = L.stage_loss_grads(outs[0], target, meta_info, faces, grad_out=w_stage[0])
    from .. import functional as F
    g_para = F.mano_backward(list(ci['tabs']), ci['para'], g_verts=[cot['pd_mesh_xyz_' + s] for s in SIDES], g_joints=[cot['pd_joint_xyz_' + s] for s in SIDES],
                             g_joint_uv=[cot['pd_joint_uv_' + s] for s in SIDES], g_mesh_uv=[cot['pd_mesh_uv_' + s] for s in SIDES])
    g_mean, G['init_regressor.offset.weight'], G['init_regressor.offset.bias'] = O.linear_bwd(cot['pd_offset'].contiguous(), ci['mean'], P['init_regressor.offset.weight'])
    for i, s in enumerate(SIDES):
        g_pool, G['init_regressor.mano_%s.weight' % s], G['init_regressor.mano_%s.bias' % s] = O.linear_bwd(g_para[i], ci[s]['pooled'], P['init_regressor.mano_%s.weight' % s])
        _, g_logit = SP.attn_pool_bwd(c4, ci[s]['attn'], ci[s]['pooled'], g_pool, g_mean if i == 0 else None, g_feat=g_c4)
        O.axpy(g_c4, _cbr_backward(P, 'init_regressor.attention_%s.' % s, ci[s]['att'], g_logit.view(B, 8, 8, 1), G))
    flush(G)
    # ---- backbone
    backbone_backward(P, ctx, [None, g_skip_src[1], g_skip_src[0], g_c4], G, flush)              # c1 has no consumer besides layer2
    TC.side_end()
    TC.end_step()
    return G
