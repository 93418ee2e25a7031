"""ORACLE (test infrastructure): numpy restatement of DIR.forward in eval mode
(models/dir.py:513-540): ResNet-50 pyramid (models/backbone/resnet.py:243-255, Bottleneck :120-140),
InitRegressor (models/dir.py:260-305), FusionJointInterIterDecoder (models/dir.py:437-483) with the
pre-activation Residual block (models/backbone/hourglass.py:55-70).

`sd` is a state dict {key: ndarray} with the reference's 963 keys (tests/golden/manifest_dir.json).
"""
import numpy as np

from . import nnops as N
from .tokens import mano_bufs, mano_outputs, stage_forward


def bottleneck(x, P, stride):
    out = N.relu(N.batchnorm(N.conv2d(x, P['conv1.weight']), P.sub('bn1')))
    out = N.relu(N.batchnorm(N.conv2d(out, P['conv2.weight'], None, stride, 1), P.sub('bn2')))
    out = N.batchnorm(N.conv2d(out, P['conv3.weight']), P.sub('bn3'))
    if 'downsample.0.weight' in P:
        x = N.batchnorm(N.conv2d(x, P['downsample.0.weight'], None, stride, 0), P.sub('downsample.1'))
    return N.relu(out + x)


def resnet50(x, P, taps=None):
    x = N.relu(N.batchnorm(N.conv2d(x, P['conv1.weight'], None, 2, 3), P.sub('bn1')))
    x = N.maxpool3x3s2p1(x)
    if taps is not None:
        taps['stem'] = x
    feats = []
    for li, (n, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for b in range(n):
            x = bottleneck(x, P.sub('layer%d.%d' % (li, b)), stride if b == 0 else 1)
        feats.append(x)
    return feats


def residual(x, P):
    """hourglass.Residual: BN-ReLU-1x1, BN-ReLU-3x3, BN-ReLU-1x1 (+bias each) + 1x1 skip conv."""
    skip = N.conv2d(x, P['skip_layer.conv.weight'], P['skip_layer.conv.bias']) if 'skip_layer.conv.weight' in P \
        and P['skip_layer.conv.weight'].shape[0] != P['skip_layer.conv.weight'].shape[1] else x
    out = N.relu(N.batchnorm(x, P.sub('bn1')))
    out = N.conv2d(out, P['conv1.conv.weight'], P['conv1.conv.bias'])
    out = N.relu(N.batchnorm(out, P.sub('bn2')))
    out = N.conv2d(out, P['conv2.conv.weight'], P['conv2.conv.bias'], 1, 1)
    out = N.relu(N.batchnorm(out, P.sub('bn3')))
    out = N.conv2d(out, P['conv3.conv.weight'], P['conv3.conv.bias'])
    return out + skip


def init_regressor(c4, P, root_joint=0):
    B = c4.shape[0]
    feats = {}
    for side in ('left', 'right'):
        A = P.sub('attention_' + side)
        h = N.relu(N.batchnorm(N.conv2d(c4, A['0.weight'], A['0.bias'], 1, 1), A.sub('1')))
        attn = N.sigmoid(N.conv2d(h, A['3.weight'], A['3.bias']))
        feats[side] = (c4 * attn).sum(-1).sum(-1) / (attn.sum(-1).sum(-1) + c4.dtype.type(1e-8))
    pd_offset = N.linear(c4.mean(-1).mean(-1), P['offset.weight'], P['offset.bias'])
    pl = N.linear(feats['left'], P['mano_left.weight'], P['mano_left.bias'])
    pr = N.linear(feats['right'], P['mano_right.weight'], P['mano_right.bias'])
    return mano_outputs(pl, pr, pd_offset, mano_bufs(P, 'left'), mano_bufs(P, 'right'), root_joint)


def seq_head(x, P, bias0=True):
    h = N.conv2d(x, P['0.weight'], P['0.bias'] if bias0 else None, 1, 1)
    h = N.relu(N.batchnorm(h, P.sub('1')))
    return N.conv2d(h, P['3.weight'], P['3.bias'])


def decoder(feats, init, P, root_joint=0, taps=None):
    c1, c2, c3, c4 = feats
    outs = []
    prev = init
    x = None
    for lvl, S, dist, up_src, skip_src in ((4, 16, 1, c4, c3), (3, 32, 2, None, c2)):
        up = N.upsample_bilinear2x(up_src if up_src is not None else x)
        skip = residual(skip_src, P.sub('skip_layer%d' % lvl))
        fusion = residual(np.concatenate([up, skip], 1), P.sub('fusion_layer%d' % lvl))
        res, ft = stage_forward(P.sub('projecter_%d' % lvl), S, dist, fusion,
                                prev['pd_joint_xyz_left'], prev['pd_joint_xyz_right'],
                                prev['pd_joint_uv_left'], prev['pd_joint_uv_right'],
                                prev['pd_mano_para_left'], prev['pd_mano_para_right'],
                                prev['pd_offset'][:, None, :], root_joint)
        x = residual(np.concatenate([fusion, ft['img_feat']], 1), P.sub('enhance_layer%d' % lvl))
        if taps is not None:
            taps['skip%d' % lvl], taps['fusion%d' % lvl] = skip, fusion
            taps['proj%d' % lvl], taps['enh%d' % lvl] = ft['img_feat'], x
        outs.append(dict(res, **ft))
        prev = res
    # f4 (no reference counterpart -- the reference hard-wires the two stages above, models/dir.py:395,401): N more
    # refinement iterations at the final 32x32 resolution, each with its own parameters: projecter_x.<i> (a Joint2BoneFeature like
    # projecter_3) on the running feature map, then enhance_layer_x.<i> (a Residual 512 -> 256 like enhance_layer3) on cat(map, img_feat)
    i = 0
    while ('projecter_x.%d.fusion.0.weight' % i) in P:
        res, ft = stage_forward(P.sub('projecter_x.%d' % i), 32, 2, x,
                                prev['pd_joint_xyz_left'], prev['pd_joint_xyz_right'], prev['pd_joint_uv_left'], prev['pd_joint_uv_right'],
                                prev['pd_mano_para_left'], prev['pd_mano_para_right'], prev['pd_offset'][:, None, :], root_joint)
        x = residual(np.concatenate([x, ft['img_feat']], 1), P.sub('enhance_layer_x.%d' % i))
        outs.append(dict(res, **ft))
        prev = res
        i += 1
    feat = seq_head(x, P.sub('conv_final'), bias0=False)
    if taps is not None:
        taps['final'] = feat
    return {'result_list': outs, 'seg': seq_head(feat, P.sub('seg')), 'dense': seq_head(feat, P.sub('dense')),
            'proj_feat': outs[-1]['vis_img_feat']}


OUT_KEYS = ('pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_xyz_left', 'pd_mesh_xyz_right',
            'pd_joint_xyz_left', 'pd_joint_xyz_right', 'pd_proj_left', 'pd_proj_right', 'pd_offset')


def dir_forward(sd, img, root_joint=0, dtype=np.float32, taps=None):
    """DIR.forward(eval) -> outs_list (len 4) exactly as models/dir.py:521-540 (pd_rel_joint=None)."""
    P = N.Params(sd, '', dtype)
    img = np.asarray(img).astype(dtype)
    if 'backbone.stage4.0.fuse_layers.0.1.0.weight' in P:          # f4: the HRNet-W48 backbone (oracle/hrnet.py; no reference counterpart)
        from .hrnet import hrnet_w48
        feats = hrnet_w48(img, P.sub('backbone'))
    else:
        feats = resnet50(img, P.sub('backbone'), taps)
    if taps is not None:
        taps.update(c1=feats[0], c2=feats[1], c3=feats[2], c4=feats[3])
    init = init_regressor(feats[3], P.sub('init_regressor'), root_joint)
    dec = decoder(feats, init, P.sub('decoder'), root_joint, taps)
    outs = []
    for o in [init] + dec['result_list']:
        d = {k: o[k] for k in OUT_KEYS}
        d['pd_rel_joint'] = None
        outs.append(d)
    outs.append({'dense': dec['dense'], 'seg': dec['seg'], 'proj_feat': dec['proj_feat']})
    return outs
