"""GPU parity of the small operators in ISOLATION, through the C ABI, against the numpy oracle and the reference goldens that the
end-to-end tests only cover in composition: dir_init_head_forward (models/dir.py:263-270: attention pooling with its +1e-8, three
Linears), dir_upsample2x_bilinear (nn.Upsample(scale_factor=2, mode='bilinear') = align_corners False, models/dir.py:392),
dir_maxpool3x3s2 (models/backbone/resnet.py:247), and the reference goldens G4 (ImgFeature2JointFeature, models/dir.py:197-200)
and G6 (one whole Joint2BoneFeature stage, models/dir.py:86-130) consumed directly by the HIP path."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, maxabs, relerr
from dir_amd import _capi, engine, synth
from dir_amd._capi import DT_BF16, DT_F32
from oracle import nnops as N
from oracle import tokens as OT
from oracle.golden_inputs import stage_inputs

pytestmark = pytest.mark.gpu
SEED = 1234


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize('B', [1, 3, 64])
def test_init_head_vs_oracle(B):
    """the head of InitRegressor after the two 3x3 attention convs: logits = Conv1x1(h) + b, attn = sigmoid, pooled feature =
    sum(c4 * attn) / (sum(attn) + 1e-8), offset from the plain mean, three Linears (models/dir.py:263-277)"""
    with open(os.path.join(GOLDEN, 'manifest_dir.json')) as f:
        shapes = {k[len('init_regressor.'):]: tuple(v) for k, v in json.load(f).items()
                  if k.startswith('init_regressor.') and 'mano_layer' not in k}
    sdn = synth.synth_state_dict(shapes, SEED)
    rng = np.random.RandomState(B)
    c4 = np.maximum(rng.normal(0, 1, (B, 2048, 8, 8)), 0).astype(np.float32)                     # post-ReLU like the real c4
    h = {s: np.maximum(rng.normal(0, 1, (B, 1024, 8, 8)), 0).astype(np.float32) for s in ('left', 'right')}
    for s in ('left', 'right'):                          # make the logits O(1): both sigmoid branches and a non-trivial denominator
        sdn['attention_%s.3.weight' % s] = (sdn['attention_%s.3.weight' % s] * 0 + rng.normal(0, 0.05, (1, 1024, 1, 1))).astype(np.float32)
    h['left'][0] = 0                                     # sample 0, left: logits = bias everywhere
    sdn['attention_right.3.bias'] = np.array([-30.0], np.float32)                                # right: attention ~1e-13 -> the 1e-8 matters
    P = N.Params(sdn)
    want = {}
    for s in ('left', 'right'):
        attn = N.sigmoid(N.conv2d(h[s], P['attention_%s.3.weight' % s], P['attention_%s.3.bias' % s]))
        feat = (c4 * attn).sum(-1).sum(-1) / (attn.sum(-1).sum(-1) + np.float32(1e-8))
        want[s] = N.linear(feat, P['mano_%s.weight' % s], P['mano_%s.bias' % s])
    want_off = N.linear(c4.mean(-1).mean(-1), P['offset.weight'], P['offset.bias'])
    # pack exactly as DirEngine._pack does
    H = _capi.InitHeadParams()
    t = {}
    for i, s in enumerate(('left', 'right')):
        t['aw%d' % i] = dev(sdn['attention_%s.3.weight' % s].reshape(-1))
        H.attn_w[i] = t['aw%d' % i].data_ptr()
        H.attn_b[i] = float(sdn['attention_%s.3.bias' % s][0])
        t['mb%d' % i] = dev(sdn['mano_%s.bias' % s])
        H.mano_b[i] = t['mb%d' % i].data_ptr()
    t['mwt'] = dev(np.concatenate([sdn['mano_left.weight'].T, sdn['mano_right.weight'].T], 1))
    H.mano_wt = t['mwt'].data_ptr()
    t['ow'], t['ob'] = dev(sdn['offset.weight']), dev(sdn['offset.bias'])
    H.off_w, H.off_b = t['ow'].data_ptr(), t['ob'].data_ptr()
    dc4 = dev(c4.transpose(0, 2, 3, 1))
    hh = dev(np.concatenate([h['left'], h['right']], 1).transpose(0, 2, 3, 1))                   # [B,8,8,2048] = left | right
    pl, pr, off = (torch.empty(B, 64, device='cuda'), torch.empty(B, 64, device='cuda'), torch.empty(B, 3, device='cuda'))
    _capi.check(_capi.lib().dir_init_head_forward(H, _capi.ptr(dc4), C.c_void_p(hh.data_ptr()), C.c_void_p(hh.data_ptr() + 1024 * 4),
                                                  2048, _capi.ptr(pl), _capi.ptr(pr), _capi.ptr(off), B, 64, 2048, 1024, DT_F32,
                                                  _capi.stream_ptr()), 'init_head')
    assert maxabs(pl.cpu().numpy(), want['left']) < 2e-6 * max(1.0, np.abs(want['left']).max())
    assert maxabs(pr.cpu().numpy(), want['right']) < 2e-6 * max(1.0, np.abs(want['right']).max())
    assert maxabs(off.cpu().numpy(), want_off) < 2e-6 * max(1.0, np.abs(want_off).max())
    # the 1e-8 in the denominator is visible on the right hand (attention ~1e-13 per pixel): without it the feature would be a
    # plain weighted mean, O(1); with it the feature is ~1e-3 of that
    attn_r = N.sigmoid(N.conv2d(h['right'], P['attention_right.3.weight'], P['attention_right.3.bias']))
    assert float(attn_r.sum(-1).sum(-1).max()) < 1e-8


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_upsample2x_vs_oracle(dt):
    """align_corners=False 2x bilinear, written into a channel slice of a wider NHWC buffer (the concat buffers of
    models/dir.py:444,461)"""
    for (B, H, W, Cc) in ((2, 8, 8, 64), (3, 16, 16, 256), (1, 5, 7, 32)):
        x = synth.synth_input('up.%d.%d' % (H, Cc), (B, Cc, H, W), SEED)
        if dt == torch.bfloat16:
            x = bf16_round(x)
        want = N.upsample_bilinear2x(x)
        dx = dev(x.transpose(0, 2, 3, 1)).to(dt)
        out = torch.full((B, 2 * H, 2 * W, Cc + 48), 7.0, device='cuda', dtype=dt)
        _capi.check(_capi.lib().dir_upsample2x_bilinear(_capi.ptr(dx), _capi.ptr(out), B, H, W, Cc, Cc + 48, 16,
                                                        DT_F32 if dt == torch.float32 else DT_BF16, _capi.stream_ptr()), 'upsample')
        got = out[..., 16:16 + Cc].float().permute(0, 3, 1, 2).cpu().numpy()
        if dt == torch.float32:
            assert maxabs(got, want) < 1e-6 * max(1.0, np.abs(want).max())
        else:
            assert np.array_equal(got, bf16_round(got)) and relerr(got, want) < 4e-3          # one bf16 rounding of the fp32 value
        assert float((out[..., :16].float() - 7).abs().max()) == 0 and float((out[..., 16 + Cc:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_maxpool_bit_exact(dt):
    """MaxPool2d(3, 2, 1): a selection, so bit-exact in either dtype (padding = -inf: a border window of negative values keeps its
    own maximum, never 0)"""
    for (B, H, W, Cc) in ((2, 128, 128, 64), (1, 10, 14, 8)):
        x = synth.synth_input('mp.%d' % H, (B, Cc, H, W), SEED) - np.float32(3.0)             # mostly negative
        if dt == torch.bfloat16:
            x = bf16_round(x)
        want = N.maxpool3x3s2p1(x)
        dx = dev(x.transpose(0, 2, 3, 1)).to(dt)
        out = torch.empty(B, H // 2, W // 2, Cc, device='cuda', dtype=dt)
        _capi.check(_capi.lib().dir_maxpool3x3s2(_capi.ptr(dx), _capi.ptr(out), B, H, W, Cc, DT_F32 if dt == torch.float32 else DT_BF16,
                                                 _capi.stream_ptr()), 'maxpool')
        assert np.array_equal(out.float().permute(0, 3, 1, 2).cpu().numpy(), want)


def _zero_mlp(cin, keep):
    """a token MLP whose output is identically zero (isolates the image branch of dir_grid_tokens_forward)"""
    t = dict(w1t=torch.zeros(cin, 128, device='cuda'), s1=torch.ones(128, device='cuda'), b1=torch.zeros(128, device='cuda'),
             w2t=torch.zeros(128, 128, device='cuda'), b2=torch.zeros(128, device='cuda'))
    keep.append(t)
    return _capi.TokenMlp(*(t[k].data_ptr() for k in ('w1t', 's1', 'b1', 'w2t', 'b2')))


def test_grid_tokens_vs_reference_golden_g4(golden):
    """G4 = the reference's ImgFeature2JointFeature (grid_sample bilinear / zeros / align_corners False at uv incl. out-of-range,
    exact-edge and pixel-centre values, then Conv1d-BN-ReLU-Conv1d) -- consumed by the HIP kernel directly: position embeddings
    zeroed, both "hands" given the same parameters and uv, so x0[h] is the golden's token matrix."""
    g = golden('g4_grid')
    shapes = {'filters.0.weight': (128, 256, 1), 'filters.0.bias': (128,), 'filters.1.weight': (128,),
              'filters.1.bias': (128,), 'filters.1.running_mean': (128,), 'filters.1.running_var': (128,),
              'filters.1.num_batches_tracked': (), 'filters.3.weight': (128, 128, 1), 'filters.3.bias': (128,)}
    sd = {('m.' + k): dev(v) for k, v in synth.synth_state_dict(shapes, SEED).items()}
    for S in (16, 32):
        keep = []
        mlp = engine.pack_token_mlp(sd, 'm.filters', keep)
        zero = _zero_mlp(3, keep)
        feat = synth.synth_input('grid.feat%d' % S, (2, 256, S, S), SEED)
        uv = dev(g['S%d.uv' % S])
        ref = g['S%d.y' % S].reshape(2, 128, 21).transpose(0, 2, 1)          # models/dir.py:94
        for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, None)):
            fbuf = dev(feat.transpose(0, 2, 3, 1)).to(dt)
            x0 = torch.empty(2, 2, 21, 128, device='cuda'); gp = torch.empty(2, 2, 21, 128, device='cuda')
            xyz, off = torch.zeros(2, 21, 3, device='cuda'), torch.zeros(2, 3, device='cuda')
            _capi.check(_capi.lib().dir_grid_tokens_forward(
                _capi.ptr(fbuf), DT_F32 if dt == torch.float32 else DT_BF16, S, 256, 256, 0, _capi.ptr(uv), _capi.ptr(uv), _capi.ptr(xyz),
                _capi.ptr(xyz), _capi.ptr(off), (_capi.TokenMlp * 2)(mlp, mlp), (_capi.TokenMlp * 2)(zero, zero), C.byref(zero),
                _capi.ptr(x0), _capi.ptr(gp), 2, _capi.stream_ptr()), 'grid_tokens')
            for hnd in range(2):
                if tol is not None:
                    assert maxabs(x0[hnd].cpu().numpy(), ref) < tol
                else:                                    # bf16 feature map: the sampled values carry the map's rounding only
                    want = OT.img2joint(bf16_round(feat), g['S%d.uv' % S], N.Params({k[2:]: v.cpu().numpy() for k, v in sd.items()}))
                    assert maxabs(x0[hnd].cpu().numpy(), want) < 2e-5
            assert float(gp.abs().max()) == 0.0


@pytest.mark.parametrize('S,dist', [(16, 1), (32, 2)])
def test_stage_vs_reference_golden_g6(golden, S, dist):
    """G6 = the reference's whole Joint2BoneFeature.forward (models/dir.py:86-130) on seeded inputs, consumed by the engine's stage
    (fp32 mode: grid tokens -> P-GCN -> STE -> regressor -> MANO -> bone_proj -> fusion convs) without the oracle in between."""
    g = golden('g6_stage%d' % S)
    with open(os.path.join(GOLDEN, 'manifest_stage%d.json' % S)) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {('st.' + k): dev(v) for k, v in synth.synth_state_dict(shapes, SEED).items()}
    eng = engine.DirEngine.__new__(engine.DirEngine)
    eng.dtype, eng.device, eng.sparse_fusion, eng.keep = torch.float32, torch.device('cuda'), True, []
    st = engine.StageOp(sd, 'st', S, dist, torch.float32, 0, eng.keep)
    img_feat, xyz_l, xyz_r, uv_l, uv_r, para_l, para_r, offset = stage_inputs(S)
    B = img_feat.shape[0]
    buf = torch.zeros(B, S, S, 512, device='cuda')
    buf[..., :256] = dev(img_feat.transpose(0, 2, 3, 1))
    prev = {'pd_joint_uv_left': dev(uv_l), 'pd_joint_uv_right': dev(uv_r), 'pd_joint_xyz_left': dev(xyz_l.astype(np.float32)),
            'pd_joint_xyz_right': dev(xyz_r.astype(np.float32)), 'pd_offset': dev(offset.reshape(B, 3)),
            'pd_mano_para_left': dev(para_l), 'pd_mano_para_right': dev(para_r)}
    res = eng.stage(st, buf, 512, prev, buf, 256, True)
    torch.cuda.synchronize()
    assert maxabs(res['pd_mano_para_left'].cpu().numpy(), g['pd_mano_para_left']) < 3e-5
    assert maxabs(res['pd_mano_para_right'].cpu().numpy(), g['pd_mano_para_right']) < 3e-5
    assert maxabs(res['pd_offset'].cpu().numpy(), g['pd_offset']) < 3e-5
    for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right'):
        assert maxabs(res[k].cpu().numpy(), g[k]) < 3e-6, k                  # metres
    for k in ('pd_joint_uv_left', 'pd_joint_uv_right'):
        assert maxabs(res[k].cpu().numpy(), g[k]) < 3e-5, k
    emb = res['joint_feat'].cpu().numpy()
    assert maxabs(emb[:, :21], g['joint_feat_left']) < 3e-5 and maxabs(emb[:, 21:], g['joint_feat_right']) < 3e-5
    got = buf[..., 256:].permute(0, 3, 1, 2).cpu().numpy()
    assert relerr(got, g['img_feat']) < 3e-5
    vis = res['vis_img_feat'].cpu().numpy()
    assert relerr(vis.astype(np.float64).sum((2, 3)), g['vis_sum']) < 1e-4
    assert maxabs(vis[:, 0:1280:97], g['vis_slice']) < 2e-5 * max(1.0, np.abs(g['vis_slice']).max())
