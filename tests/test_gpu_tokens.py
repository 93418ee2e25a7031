"""GPU parity for the joint-token kernels (a4, a5, a6, a7, a10, a12) through the C ABI, against the goldens the
reference produced (tests/golden) and the numpy oracle.  fp32; tolerances are those the oracle itself meets."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, maxabs, relerr
from dir_amd import _capi, engine, synth
from oracle import nnops as N
from oracle import tokens as OT
from oracle.golden_inputs import bone_uv

pytestmark = pytest.mark.gpu
SEED = 1234


def dev(a):
    """numpy -> cuda tensor.  Callers must keep the result referenced until the kernel has been enqueued: a bare
    _capi.ptr(dev(x)) would take the address of a temporary the caching allocator may hand out again."""
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pgcn_shapes(prefix=''):
    s = {}
    for i in range(4):
        p = '%sgconv_layers.%d.' % (prefix, i)
        s.update({p + 'gconv.W': (2, 21, 128, 128), p + 'gconv.e_0': (1, 21), p + 'gconv.e_1': (1, 40),
                  p + 'gconv.bias': (128,), p + 'bn.weight': (128,), p + 'bn.bias': (128,),
                  p + 'bn.running_mean': (128,), p + 'bn.running_var': (128,), p + 'bn.num_batches_tracked': ()})
    return s


def test_pgcn_stack_vs_reference(golden):
    g = golden('g2_pgcn')
    sd = {('gcn.' + k): dev(v) for k, v in synth.synth_state_dict(pgcn_shapes(), SEED).items()}
    keep = []
    layers = engine.pack_pgcn(sd, 'gcn', keep)
    x = torch.from_numpy(g['x']).cuda()
    B = x.shape[0]
    out = torch.empty(B, 21, 128, device='cuda')
    scratch = torch.empty(2, B, 21, 256, device='cuda')
    _capi.check(_capi.lib().dir_pgcn_stack_forward(layers, 4, _capi.ptr(x), None, _capi.ptr(out), 21 * 128,
                                                   _capi.ptr(scratch), B, _capi.stream_ptr()), 'pgcn')
    assert relerr(out.cpu().numpy(), g['y']) < 2e-6
    # single layer, BN+ReLU on: compare with the per-layer activation of the golden
    _capi.check(_capi.lib().dir_pgcn_stack_forward(layers, 1, _capi.ptr(x), None, _capi.ptr(out), 21 * 128,
                                                   _capi.ptr(scratch), B, _capi.stream_ptr()), 'pgcn')
    assert relerr(out.cpu().numpy(), g['layer0']) < 2e-6


def test_pgcn_batch_sizes_and_add():
    """B not a multiple of the 64-sample LDS chunk, B > 64, the `add` term and the [B,42,128] strided output."""
    sdn = synth.synth_state_dict(pgcn_shapes(), SEED)
    sd = {('gcn.' + k): torch.from_numpy(v).cuda() for k, v in sdn.items()}
    keep = []
    layers = engine.pack_pgcn(sd, 'gcn', keep)
    for B in (1, 7, 64, 150):
        x = synth.synth_input('pgcn.bs%d' % B, (B, 21, 128), SEED)
        add = synth.synth_input('pgcn.add%d' % B, (B, 21, 128), SEED)
        ref = OT.pgcn_stack(x, N.Params(sdn)) + add
        tok = torch.zeros(B, 42, 128, device='cuda')
        scratch = torch.empty(2, B, 21, 256, device='cuda')
        dx, dadd = dev(x), dev(add)
        _capi.check(_capi.lib().dir_pgcn_stack_forward(layers, 4, _capi.ptr(dx), _capi.ptr(dadd),
                                                       C.c_void_p(tok.data_ptr() + 21 * 128 * 4), 42 * 128,
                                                       _capi.ptr(scratch), B, _capi.stream_ptr()), 'pgcn')
        assert relerr(tok[:, 21:].cpu().numpy(), ref) < 3e-6
        assert float(tok[:, :21].abs().max()) == 0.0


@pytest.mark.parametrize('wdt', [torch.float32, torch.bfloat16])
def test_pgcn_fused_equals_layered(wdt):
    """Round 4: the 4-layer stack of both hands in ONE launch (tokens.hip: pgcn_fused_kernel -- persistent workgroup per (hand, node, split),
    layers separated by per-node flags) against the five launches it replaces: the same fmaf chains in the same order, so the tokens must be
    bit-identical -- at ragged and large batches, for every split count, with and without the `add` term, and over repeated launches on the
    same sync words (the flags are never reset: each launch raises them by one).  The layered path is what G2 / the oracle hold
    (test_pgcn_stack_vs_reference, test_pgcn_batch_sizes_and_add), so this pins the fused one to the reference through it; one size is also
    compared with the oracle directly."""
    L = _capi.lib()
    sdn_l, sdn_r = synth.synth_state_dict(pgcn_shapes(), SEED), synth.synth_state_dict(pgcn_shapes(), SEED + 1)
    keep = []
    lay = [engine.pack_pgcn({('gcn.' + k): dev(v) for k, v in sdn.items()}, 'gcn', keep, weight_dtype=wdt) for sdn in (sdn_l, sdn_r)]
    sync = torch.zeros(int(L.dir_pgcn_fused_sync_bytes()) // 4, dtype=torch.int32, device='cuda')
    launches = 0
    for B in (1, 7, 16, 33, 64, 150, 300):
        x = dev(synth.synth_input('pgcnf.x%d' % B, (2, B, 21, 128), SEED))
        add = dev(synth.synth_input('pgcnf.a%d' % B, (2, B, 21, 128), SEED))
        for use_add in (True, False):
            want = torch.zeros(B, 42, 128, device='cuda')
            scratch = torch.empty(4, B, 21, 256, device='cuda')
            _capi.check(L.dir_pgcn_stack_forward_pair(lay[0], lay[1], 4, _capi.ptr(x), _capi.ptr(add) if use_add else None, _capi.ptr(want),
                                                      _capi.ptr(scratch), B, _capi.stream_ptr()), 'pair')
            for splits in (0, 1, 2, 3, 4, 8):
                got = torch.full((B, 42, 128), float('nan'), device='cuda')
                sc2 = torch.full((4, B, 21, 256), float('nan'), device='cuda')
                _capi.check(L.dir_pgcn_stack_forward_fused(lay[0], lay[1], 4, _capi.ptr(x), _capi.ptr(add) if use_add else None, _capi.ptr(got),
                                                           _capi.ptr(sc2), _capi.ptr(sync), splits, B, _capi.stream_ptr()), 'fused')
                launches += 1
                assert torch.equal(got, want), (B, use_add, splits)
        if B == 33 and wdt == torch.float32:
            ref = np.concatenate([OT.pgcn_stack(x[0].cpu().numpy(), N.Params(sdn_l)), OT.pgcn_stack(x[1].cpu().numpy(), N.Params(sdn_r))], 1)
            assert relerr(got.cpu().numpy(), ref) < 3e-6
    torch.cuda.synchronize()
    assert int(sync[-4]) == 0                                     # no workgroup ever gave up waiting
    flags = sync[:-4].view(4, 2, 21, 8)
    assert int(flags[:, :, :, 0].min()) == int(flags[:, :, :, 0].max()) == launches          # split 0 exists in every launch: one raise per launch
    # fewer layers (the single-layer form the golden's per-layer activation uses)
    x = dev(synth.synth_input('pgcnf.x1', (2, 5, 21, 128), SEED))
    want, got = torch.zeros(5, 42, 128, device='cuda'), torch.zeros(5, 42, 128, device='cuda')
    scratch = torch.empty(4, 5, 21, 256, device='cuda')
    _capi.check(L.dir_pgcn_stack_forward_pair(lay[0], lay[1], 2, _capi.ptr(x), None, _capi.ptr(want), _capi.ptr(scratch), 5, _capi.stream_ptr()), 'pair')
    _capi.check(L.dir_pgcn_stack_forward_fused(lay[0], lay[1], 2, _capi.ptr(x), None, _capi.ptr(got), _capi.ptr(scratch), _capi.ptr(sync), 0, 5,
                                               _capi.stream_ptr()), 'fused')
    assert torch.equal(got, want)


def test_pgcn_fused_on_four_streams_at_once():
    """the forward-progress argument of pgcn_fused_kernel: launches of it from four streams (bench.py's four forwards in flight), each with
    its own sync words, interleaved with ordinary kernels, all finish and all give the one-at-a-time tokens"""
    L = _capi.lib()
    sdn = synth.synth_state_dict(pgcn_shapes(), SEED)
    keep = []
    lay = engine.pack_pgcn({('gcn.' + k): dev(v) for k, v in sdn.items()}, 'gcn', keep, weight_dtype=torch.bfloat16)
    B = 64
    x = dev(synth.synth_input('pgcnf.s', (2, B, 21, 128), SEED))
    want = torch.zeros(B, 42, 128, device='cuda')
    scratch = torch.empty(4, B, 21, 256, device='cuda')
    _capi.check(L.dir_pgcn_stack_forward_pair(lay, lay, 4, _capi.ptr(x), None, _capi.ptr(want), _capi.ptr(scratch), B, _capi.stream_ptr()), 'pair')
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    syncs = [torch.zeros(int(L.dir_pgcn_fused_sync_bytes()) // 4, dtype=torch.int32, device='cuda') for _ in streams]
    outs = [torch.zeros(B, 42, 128, device='cuda') for _ in streams]
    scr = [torch.empty(4, B, 21, 256, device='cuda') for _ in streams]
    filler = torch.randn(4096, 4096, device='cuda')
    torch.cuda.synchronize()
    for rnd in range(50):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                if (rnd + i) % 3 == 0:
                    filler.mul(1.0001)                            # an ordinary all-CU kernel in between
                _capi.check(L.dir_pgcn_stack_forward_fused(lay, lay, 4, _capi.ptr(x), None, _capi.ptr(outs[i]), _capi.ptr(scr[i]), _capi.ptr(syncs[i]),
                                                           0, B, _capi.stream_ptr()), 'fused')
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(outs[i], want), i
        assert int(syncs[i][-4]) == 0


def ste_shapes(prefix):
    s = {'spatial_pos_embed': (1, 42, 128), 'spatial_norm.weight': (128,), 'spatial_norm.bias': (128,),
         'head.0.weight': (128,), 'head.0.bias': (128,), 'head.1.weight': (64, 128), 'head.1.bias': (64,)}
    for i in range(4):
        p = 'STEblocks.%d.' % i
        s.update({p + 'norm1.weight': (128,), p + 'norm1.bias': (128,), p + 'norm2.weight': (128,),
                  p + 'norm2.bias': (128,), p + 'attn.qkv.weight': (384, 128), p + 'attn.qkv.bias': (384,),
                  p + 'attn.proj.weight': (128, 128), p + 'attn.proj.bias': (128,),
                  p + 'mlp.fc1.weight': (256, 128), p + 'mlp.fc1.bias': (256,),
                  p + 'mlp.fc2.weight': (128, 256), p + 'mlp.fc2.bias': (128,)})
    return s


def test_ste_vs_reference(golden):
    g = golden('g3_ste')
    sdn = synth.synth_state_dict(ste_shapes(''), SEED)
    sd = {('ste.' + k): torch.from_numpy(v).cuda() for k, v in sdn.items()}
    keep = []
    P = engine.pack_ste(sd, 'ste', keep)
    x = torch.from_numpy(g['x']).cuda()
    xpos = torch.empty_like(x)
    y = torch.empty(2, 42, 64, device='cuda')
    _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(x), _capi.ptr(xpos), _capi.ptr(y), 2, _capi.stream_ptr()), 'ste')
    assert maxabs(y.cpu().numpy(), g['y']) < 2e-5
    assert maxabs(xpos.cpu().numpy(), g['x'] + sdn['spatial_pos_embed']) < 1e-7     # the in-place `x += pos`
    # larger batch vs the oracle
    xb = synth.synth_input('ste.big', (33, 42, 128), SEED)
    yb = torch.empty(33, 42, 64, device='cuda')
    dxb = dev(xb)
    _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(dxb), None, _capi.ptr(yb), 33,
                                            _capi.stream_ptr()), 'ste')
    assert maxabs(yb.cpu().numpy(), OT.ste_forward(xb, N.Params(sdn))) < 3e-5


def test_bone_proj_vs_reference(golden):
    g = golden('g5_bone')
    for S, dist in ((16, 1), (32, 2)):
        uv = g['S%d.uv' % S]
        feat = synth.synth_input('bone.feat%d' % S, (2, 21, 64), SEED)
        ref = g['S%d.y' % S]                                                     # [2,1280,S,S] one hand
        # left hand = the golden case, right hand = the same joints mirrored (exercises the second slice)
        uv_r = uv.copy(); uv_r[..., 0] *= -1
        emb = np.concatenate([feat, feat[:, ::-1].copy()], 1)                    # [2,42,64]
        ref_r = OT.bone_proj(uv_r, emb[:, 21:], S, dist)
        duv, duvr, demb = dev(uv), dev(uv_r), dev(emb)
        for tdt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
            out = torch.empty(2, S, S, 2560, device='cuda', dtype=tdt)
            vis = torch.empty(2, 1280, S, S, device='cuda')
            _capi.check(_capi.lib().dir_bone_proj_forward(
                _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(demb), _capi.ptr(out), _capi.ptr(vis), None, 2, S, float(dist),
                0 if tdt == torch.float32 else 1, _capi.stream_ptr()), 'bone_proj')
            got = out.float().cpu().numpy().transpose(0, 3, 1, 2)
            assert np.array_equal(got[:, :1280] != 0, ref != 0), 'capsule mask differs from the reference (S=%d)' % S
            assert np.array_equal(got[:, 1280:] != 0, ref_r != 0)
            assert maxabs(got[:, :1280], ref) < tol and maxabs(got[:, 1280:], ref_r) < tol
            assert maxabs(vis.cpu().numpy(), ref + ref_r) < 2e-6                 # vis is fp32 regardless of dtype
            vis2 = torch.full((2, 1280, S, S), 3.0, device='cuda')               # proj_feat-only launch (out = NULL)
            _capi.check(_capi.lib().dir_bone_proj_forward(
                _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(demb), None, _capi.ptr(vis2), None, 2, S, float(dist),
                0 if tdt == torch.float32 else 1, _capi.stream_ptr()), 'bone_proj vis only')
            assert torch.equal(vis, vis2)


def stage_sd(S):
    with open(os.path.join(GOLDEN, 'manifest_stage%d.json' % S)) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sdn = synth.synth_state_dict(shapes, SEED)
    return sdn, {('st.' + k): torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sdn.items()}


@pytest.mark.parametrize('S,dist', [(16, 1), (32, 2)])
def test_grid_tokens_and_regress_vs_oracle(golden, S, dist):
    """dir_grid_tokens_forward and dir_regress_forward in isolation (the stage golden covers their composition)."""
    sdn, sd = stage_sd(S)
    keep = []
    st = engine.StageOp(sd, 'st', S, dist, torch.float32, 0, keep)
    P = N.Params(sdn)
    B = 3
    feat = synth.synth_input('gt.feat', (B, 256, S, S), SEED)
    uv = [bone_uv('gt.uv%d' % h, B, S) for h in range(2)]
    xyz = [synth.synth_input('gt.xyz%d' % h, (B, 21, 3), SEED) * np.float32(0.05) for h in range(2)]
    off = synth.synth_input('gt.off', (B, 3), SEED)
    # feature map inside a wider NHWC buffer at channel offset 64 (slice addressing)
    fbuf = torch.zeros(B, S, S, 384, device='cuda')
    fbuf[..., 64:320] = dev(feat.transpose(0, 2, 3, 1))
    x0 = torch.empty(2, B, 21, 128, device='cuda'); gp = torch.empty(2, B, 21, 128, device='cuda')
    duv, dxyz, doff = [dev(u) for u in uv], [dev(v) for v in xyz], dev(off)
    _capi.check(_capi.lib().dir_grid_tokens_forward(
        _capi.ptr(fbuf), 0, S, 256, 384, 64, _capi.ptr(duv[0]), _capi.ptr(duv[1]), _capi.ptr(dxyz[0]),
        _capi.ptr(dxyz[1]), _capi.ptr(doff), st.img2joint, st.pos_emb, C.byref(st.gpos), _capi.ptr(x0),
        _capi.ptr(gp), B, _capi.stream_ptr()), 'grid_tokens')
    for h, side in enumerate(('left', 'right')):
        img = OT.img2joint(feat, uv[h], P.sub('img2joint_' + side))
        pos = OT.token_mlp(xyz[h].transpose(0, 2, 1) / np.float32(0.15), P.sub('pos_emb_' + side)).transpose(0, 2, 1)
        q = xyz[h] / np.float32(0.15) + (off[:, None] / 2) * (1 if h else -1)
        gref = OT.token_mlp(q.transpose(0, 2, 1), P.sub('global_pos_emb')).transpose(0, 2, 1)
        assert maxabs(x0[h].cpu().numpy(), pos + img) < 2e-5
        assert maxabs(gp[h].cpu().numpy(), gref) < 2e-5
    # regress
    tok = synth.synth_input('rg.tok', (B, 42, 64), SEED)
    pl, pr = synth.synth_input('rg.pl', (B, 64), SEED), synth.synth_input('rg.pr', (B, 64), SEED)
    o = [torch.empty(B, 64, device='cuda'), torch.empty(B, 64, device='cuda'), torch.empty(B, 3, device='cuda'),
         torch.empty(B, 42, 64, device='cuda')]
    dtok, dpl, dpr = dev(tok), dev(pl), dev(pr)
    _capi.check(_capi.lib().dir_regress_forward(C.byref(st.reg), _capi.ptr(dtok), _capi.ptr(dpl), _capi.ptr(dpr),
                                                _capi.ptr(doff), _capi.ptr(o[0]), _capi.ptr(o[1]), _capi.ptr(o[2]),
                                                _capi.ptr(o[3]), B, _capi.stream_ptr()), 'regress')
    R = P.sub('regressor')
    fl, fr = tok[:, :21].reshape(B, -1), tok[:, 21:].reshape(B, -1)
    assert maxabs(o[0].cpu().numpy(), N.linear(np.concatenate([fl, pl], 1), R['mano_left.weight'], R['mano_left.bias'])) < 1e-5
    assert maxabs(o[1].cpu().numpy(), N.linear(np.concatenate([fr, pr], 1), R['mano_right.weight'], R['mano_right.bias'])) < 1e-5
    assert maxabs(o[2].cpu().numpy(), N.linear(np.concatenate([fl, fr, off], 1), R['offset.weight'], R['offset.bias'])) < 1e-5
    emb = OT.token_mlp(tok.transpose(0, 2, 1), P.sub('proj_feat_emb')).transpose(0, 2, 1)
    assert maxabs(o[3].cpu().numpy(), emb) < 1e-5


def test_bone_bbox_is_conservative_and_sparse_conv_is_bit_identical(golden):
    """dir_bone_proj_forward's per-(hand,bone) pixel boxes must contain every non-zero of the rasterised features
    (degenerate / off-image / coincident joints included), and dir_conv2d_sparse_forward, which skips the K-slabs
    those boxes rule out, must reproduce the dense convolution bit for bit."""
    from dir_amd import functional as F
    g = golden('g5_bone')
    for S, dist in ((16, 1), (32, 2)):
        uv = g['S%d.uv' % S].copy()
        uv_r = uv.copy(); uv_r[..., 0] *= -1
        uv_r[1, 3] = np.nan                                   # NaN joint: its two bones rasterise to nothing
        uv_r[0, 17:] = 3.0                                    # a finger completely outside the image
        B = uv.shape[0]
        emb = synth.synth_input('bbox.emb%d' % S, (B, 42, 64), SEED)
        duv, duvr, demb = dev(uv), dev(uv_r), dev(emb)
        for tdt in (torch.bfloat16, torch.float32):
            out = torch.empty(B, S, S, 2560, device='cuda', dtype=tdt)
            bbox = torch.full((B, 40, 4), -77, device='cuda', dtype=torch.int32)
            _capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(demb), _capi.ptr(out),
                                                          None, _capi.ptr(bbox), B, S, float(dist),
                                                          0 if tdt == torch.float32 else 1, _capi.stream_ptr()), 'bone')
            nz = (out.float().reshape(B, S, S, 40, 64) != 0).any(-1).cpu().numpy()          # [B,S,S,40]
            bb = bbox.cpu().numpy()
            assert (bb != -77).all()
            empty = 0
            for b in range(B):
                for gI in range(40):
                    ys, xs = np.nonzero(nz[b, :, :, gI])
                    y0, y1, x0, x1 = bb[b, gI]
                    if len(ys):
                        assert y0 <= ys.min() and ys.max() <= y1 and x0 <= xs.min() and xs.max() <= x1, (S, b, gI)
                    empty += int(y0 > y1 or x0 > x1)
            assert empty >= 2                                   # the NaN bones (and possibly off-image ones) are empty
            frac = np.mean([(max(0, bb[b, gI, 1] - bb[b, gI, 0] + 1) * max(0, bb[b, gI, 3] - bb[b, gI, 2] + 1)) / (S * S)
                            for b in range(B) for gI in range(40)])
            assert frac < 0.5                                   # the boxes are tight enough to be worth something
            w = (torch.randn(256, 3, 3, 2560, device='cuda') * 0.02).to(tdt)
            shift = torch.randn(256, device='cuda')
            dense = F.conv2d_nhwc(out, w, 1, 1, shift=shift, relu=True)
            d = _capi.ConvDesc(B, S, S, 2560, 2560, 0, 256, 256, 0, 0, 0, 3, 3, 1, 1, F._dt(out), F._dt(out), 1, 0, 0)
            sparse = torch.empty_like(dense)
            _capi.check(_capi.lib().dir_conv2d_sparse_forward(d, _capi.ptr(out), _capi.ptr(w), None, _capi.ptr(shift), None,
                                                              _capi.ptr(sparse), _capi.ptr(bbox), _capi.stream_ptr()), 'sparse')
            assert torch.equal(dense, sparse), 'sparse-K conv differs from dense (S=%d, %s)' % (S, tdt)


@pytest.mark.parametrize('S,dist,B', [(16, 1, 3), (32, 2, 2), (32, 2, 5)])
def test_bone_fusion_factorised_vs_oracle(golden, S, dist, B):
    """dir_bone_fusion_forward (bone_proj + fusion conv + BN + ReLU as a K = 720 reduction) against the oracle's
    bone_proj -> conv2d 3x3 -> scale/shift -> ReLU in float64, against the reference's own bone map (g5_bone golden) pushed
    through the same float64 conv, and against the materialised GPU path (dir_bone_proj_forward + dir_conv2d_forward).
    Tolerance: bf16 operands (weights, G, pixel weights: 2^-9 each) and a bf16 output -> 1.5e-2 of the output scale."""
    g = golden('g5_bone')
    uv0 = g['S%d.uv' % S]                                                        # [2,21,2] from the reference fixture
    rng = np.random.default_rng(7 + S + B)
    uv_l = np.concatenate([uv0, uv0[::-1]], 0)[:B] if B <= 4 else np.concatenate([uv0, uv0[::-1], uv0[:1] * 0.7], 0)
    uv_l = np.ascontiguousarray(uv_l, np.float32)
    uv_r = uv_l.copy(); uv_r[..., 0] *= -0.9
    emb = synth.synth_input('bonefuse.emb%d' % S, (B, 42, 64), SEED)
    w = (rng.standard_normal((256, 2560, 3, 3)) * 0.02).astype(np.float32)
    w = torch.from_numpy(w).to(torch.bfloat16).float().numpy()                   # the bf16 mode's weights
    scale = (1 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    shift = (0.1 * rng.standard_normal(256)).astype(np.float32)
    # oracle, float64
    img = np.concatenate([OT.bone_proj(uv_l, emb[:, :21], S, dist), OT.bone_proj(uv_r, emb[:, 21:], S, dist)], 1)   # [B,2560,S,S]
    if B == 2:                                                # left-hand joints == the reference fixture's: same capsule mask
        assert np.array_equal((img[:, :1280] != 0).reshape(2, 20, 64, S, S).any(2), (g['S%d.y' % S] != 0).reshape(2, 20, 64, S, S).any(2))
    ref = N.conv2d(img.astype(np.float64), w.astype(np.float64), None, 1, 1)
    ref = np.maximum(ref * scale[None, :, None, None] + shift[None, :, None, None], 0)
    # GPU: factorised
    L = _capi.lib()
    dw = torch.from_numpy(w).cuda()
    wg = dw.reshape(256, 40, 64, 9).permute(3, 1, 2, 0).contiguous()
    dsc, dsh = dev(scale), dev(shift)
    P = _capi.BoneFusionParams(wg.data_ptr(), dsc.data_ptr(), dsh.data_ptr())
    duv, duvr, demb = dev(uv_l), dev(uv_r), dev(emb)
    scratch = torch.empty(L.dir_bone_fusion_scratch_bytes(B), device='cuda', dtype=torch.uint8)
    y = torch.full((B, S, S, 320), 7.0, device='cuda', dtype=torch.bfloat16)      # written into channels [32, 288)
    _capi.check(L.dir_bone_fusion_prepare(P, _capi.ptr(demb), _capi.ptr(scratch), B, _capi.stream_ptr()), 'bone_fusion_prepare')
    _capi.check(L.dir_bone_fusion_forward(P, _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(scratch), _capi.ptr(y),
                                          B, S, float(dist), 320, 32, 1, _capi.stream_ptr()), 'bone_fusion')
    torch.cuda.synchronize()
    got = y[..., 32:288].float().cpu().numpy().transpose(0, 3, 1, 2)
    sc = np.abs(ref).max()
    assert maxabs(got, ref) <= 1.5e-2 * sc, (maxabs(got, ref), sc)
    assert float(y[..., :32].float().min()) == 7.0 and float(y[..., 288:].float().max()) == 7.0     # slice untouched outside
    # GPU: materialised bone map + implicit-GEMM conv (the fp32-mode path, here in bf16)
    bone = torch.empty(B, S, S, 2560, device='cuda', dtype=torch.bfloat16)
    _capi.check(L.dir_bone_proj_forward(_capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(demb), _capi.ptr(bone), None, None, B, S,
                                        float(dist), 1, _capi.stream_ptr()), 'bone_proj')
    from dir_amd import functional as Fn
    wp = dw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    y2 = Fn.conv2d_nhwc(bone, wp, 1, 1, scale=dsc, shift=dsh, relu=True).float().cpu().numpy().transpose(0, 3, 1, 2)
    assert maxabs(got, y2) <= 1.5e-2 * sc
    # and the factorised result is at least as close to the float64 value as the materialised bf16 path
    assert maxabs(got, ref) <= 1.25 * maxabs(y2, ref) + 1e-3 * sc

@pytest.mark.parametrize('S,dist,B', [(16, 1, 2), (32, 2, 2), (32, 2, 5)])
def test_bone_fusion_exact_fp32_vs_oracle(golden, S, dist, B):
    """dir_bone_fusion_* with exact_f32 = 1 (the f16x3 parity mode's fusion path): unrounded weights, fp32 G and pixel weights, exact fp32
    matrix cores.  Against the float64 composition bone_proj -> conv3x3 -> scale / shift -> ReLU: 5e-6 of the output scale (the
    materialised exact-fp32 convolution of the same operands, K = 23040, sits at the same distance: both are fp32 summations of the same
    products in different orders)."""
    g = golden('g5_bone')
    uv0 = g['S%d.uv' % S]
    rng = np.random.default_rng(17 + S + B)
    uv_l = np.concatenate([uv0, uv0[::-1]], 0)[:B] if B <= 4 else np.concatenate([uv0, uv0[::-1], uv0[:1] * 0.7], 0)
    uv_l = np.ascontiguousarray(uv_l, np.float32)
    uv_r = uv_l.copy(); uv_r[..., 0] *= -0.9
    emb = synth.synth_input('bonefuse.emb%d' % S, (B, 42, 64), SEED)
    w = (rng.standard_normal((256, 2560, 3, 3)) * 0.02).astype(np.float32)
    scale = (1 + 0.1 * rng.standard_normal(256)).astype(np.float32)
    shift = (0.1 * rng.standard_normal(256)).astype(np.float32)
    img = np.concatenate([OT.bone_proj(uv_l, emb[:, :21], S, dist), OT.bone_proj(uv_r, emb[:, 21:], S, dist)], 1)   # [B,2560,S,S]
    ref = N.conv2d(img.astype(np.float64), w.astype(np.float64), None, 1, 1)
    ref = np.maximum(ref * scale[None, :, None, None] + shift[None, :, None, None], 0)
    L = _capi.lib()
    dw = torch.from_numpy(w).cuda()
    wg = dw.reshape(256, 40, 64, 9).permute(3, 1, 2, 0).contiguous()
    dsc, dsh = dev(scale), dev(shift)
    P = _capi.BoneFusionParams(wg.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), 1)
    duv, duvr, demb = dev(uv_l), dev(uv_r), dev(emb)
    scratch = torch.empty(L.dir_bone_fusion_scratch_bytes(B), device='cuda', dtype=torch.uint8)
    y = torch.full((B, S, S, 320), 7.0, device='cuda')                            # written into channels [32, 288)
    _capi.check(L.dir_bone_fusion_prepare(P, _capi.ptr(demb), _capi.ptr(scratch), B, _capi.stream_ptr()), 'bone_fusion_prepare')
    _capi.check(L.dir_bone_fusion_forward(P, _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(scratch), _capi.ptr(y),
                                          B, S, float(dist), 320, 32, 1, _capi.stream_ptr()), 'bone_fusion')
    torch.cuda.synchronize()
    got = y[..., 32:288].cpu().numpy().transpose(0, 3, 1, 2)
    sc = np.abs(ref).max()
    print('exact-fp32 factorised fusion S=%d B=%d: %.2e of the output scale' % (S, B, maxabs(got, ref) / sc))
    assert maxabs(got, ref) <= 5e-6 * sc, (maxabs(got, ref), sc)
    assert float(y[..., :32].min()) == 7.0 and float(y[..., 288:].max()) == 7.0     # slice untouched outside
    bone = torch.empty(B, S, S, 2560, device='cuda')
    _capi.check(L.dir_bone_proj_forward(_capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(demb), _capi.ptr(bone), None, None, B, S,
                                        float(dist), 0, _capi.stream_ptr()), 'bone_proj')
    from dir_amd import functional as Fn
    y2 = Fn.conv2d_nhwc(bone, dw.permute(0, 2, 3, 1).contiguous(), 1, 1, scale=dsc, shift=dsh, relu=True).cpu().numpy().transpose(0, 3, 1, 2)
    print('   materialised exact-fp32 conv: %.2e' % (maxabs(y2, ref) / sc))
    assert maxabs(got, ref) <= 2.0 * maxabs(y2, ref) + 1e-6 * sc
    # split precision on the f16 matrix cores (g_scale > 0): the same distance from float64 as the exact kernel
    import math
    amax = float(scratch.view(torch.float32)[:B * 9 * 40 * 256 * 2].abs().max())
    P.g_scale = 2.0 ** (10 - math.frexp(amax)[1])
    y3 = torch.full((B, S, S, 320), 7.0, device='cuda')
    _capi.check(L.dir_bone_fusion_forward(P, _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(scratch), _capi.ptr(y3),
                                          B, S, float(dist), 320, 32, 1, _capi.stream_ptr()), 'bone_fusion')
    torch.cuda.synchronize()
    got3 = y3[..., 32:288].cpu().numpy().transpose(0, 3, 1, 2)
    print('   split-precision factorised fusion (g_scale 2^%d): %.2e' % (math.frexp(P.g_scale)[1] - 1, maxabs(got3, ref) / sc))
    assert maxabs(got3, ref) <= 5e-6 * sc, (maxabs(got3, ref), sc)
    assert float(y3[..., :32].min()) == 7.0 and float(y3[..., 288:].max()) == 7.0
    P.g_scale = 3.0                                                                # not a power of two
    assert L.dir_bone_fusion_forward(P, _capi.ptr(duv), _capi.ptr(duvr), _capi.ptr(scratch), _capi.ptr(y3), B, S, float(dist), 320, 32, 1,
                                     _capi.stream_ptr()) != 0


def test_bone_fusion_rejects_bad_arguments():
    L = _capi.lib()
    z = torch.zeros(16, device='cuda')
    P = _capi.BoneFusionParams(z.data_ptr(), z.data_ptr(), z.data_ptr())
    for S, cs, co in ((24, 256, 0), (16, 250, 0), (16, 256, 8)):
        rc = L.dir_bone_fusion_forward(P, _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), 1, S, 1.0, cs, co, 1, _capi.stream_ptr())
        assert rc != 0 and b'dir_bone_fusion_forward' in L.dir_last_error()
    assert L.dir_bone_fusion_forward(P, _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), 0, 16, 1.0, 256, 0, 1,
                                     _capi.stream_ptr()) == 0                       # empty batch is a no-op
    assert L.dir_bone_fusion_prepare(P, _capi.ptr(z), _capi.ptr(z), 0, _capi.stream_ptr()) == 0
    assert L.dir_bone_fusion_scratch_bytes(3) == 3 * 9 * 40 * 256 * 8      # float2 per (tap, hand-bone, channel): sized for exact_f32


def test_ste_bf16_linears_autocast_semantics(golden):
    """weight_dtype = bf16: nn.Linear on the bf16 matrix cores (fp32 accumulate), everything else fp32 -- what
    torch.autocast(bfloat16) does to transformer/mixSTE.py.  Checked against (a) the reference golden within the bf16
    envelope and (b) an oracle run whose Linear operands are rounded to bf16 the same way (tight)."""
    g = golden('g3_ste')
    sdn = synth.synth_state_dict(ste_shapes(''), SEED)
    sd = {('ste.' + k): torch.from_numpy(v).cuda() for k, v in sdn.items()}
    keep = []
    P = engine.pack_ste(sd, 'ste', keep, weight_dtype=torch.bfloat16)
    x = torch.from_numpy(g['x']).cuda()
    y = torch.empty(2, 42, 64, device='cuda')
    _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(x), None, _capi.ptr(y), 2, _capi.stream_ptr()), 'ste bf16')
    got = y.cpu().numpy()
    sc = np.abs(g['y']).max()
    assert maxabs(got, g['y']) < 2e-2 * sc, (maxabs(got, g['y']), sc)

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()

    orig = N.linear                                      # oracle with bf16-rounded Linear operands, fp32 accumulation
    try:
        N.linear = lambda xx, w, b=None: orig(bf(xx), bf(w), b)
        want = OT.ste_forward(g['x'].copy(), N.Params(sdn))
        # a ragged batch through the 16-wave kernel of round 4 (ste16_kernel: attention probabilities in registers, spatial_norm + LayerNorm fused)
        xb = synth.synth_input('ste.bf16.big', (37, 42, 128), SEED)
        want_b = OT.ste_forward(xb.copy(), N.Params(sdn))
    finally:
        N.linear = orig
    assert maxabs(got, want) < 2e-3 * sc, (maxabs(got, want), sc)
    dxb = dev(xb)
    xpos = torch.empty(37, 42, 128, device='cuda')
    yb = torch.empty(37, 42, 64, device='cuda')
    _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(dxb), _capi.ptr(xpos), _capi.ptr(yb), 37, _capi.stream_ptr()), 'ste bf16 batch')
    assert maxabs(yb.cpu().numpy(), want_b) < 2e-3 * np.abs(want_b).max()
    assert maxabs(xpos.cpu().numpy(), xb + sdn['spatial_pos_embed'].reshape(1, 42, 128)) < 1e-6          # the in-place `x += pos_embed` the reference performs


def test_pgcn_bf16_matmuls_autocast_semantics(golden):
    """w_dtype = bf16: the two matmuls of PGraphConv (SemGCN/p_graph_conv.py:47-48) on the bf16 matrix cores with fp32
    accumulation, everything else (edge softmax, neighbour mix, bias, BatchNorm, ReLU) fp32 -- torch.autocast semantics.
    Checked against the reference golden within the bf16 envelope and, tightly, against the oracle with the matmul operands
    rounded to bf16 the same way (per layer input and weights)."""
    g = golden('g2_pgcn')
    sdn = synth.synth_state_dict(pgcn_shapes(), SEED)
    sd = {('gcn.' + k): dev(v) for k, v in sdn.items()}
    keep = []
    layers = engine.pack_pgcn(sd, 'gcn', keep, weight_dtype=torch.bfloat16)
    x = torch.from_numpy(g['x']).cuda()
    B = x.shape[0]
    out = torch.empty(B, 21, 128, device='cuda')
    scratch = torch.empty(2, B, 21, 256, device='cuda')
    _capi.check(_capi.lib().dir_pgcn_stack_forward(layers, 4, _capi.ptr(x), None, _capi.ptr(out), 21 * 128,
                                                   _capi.ptr(scratch), B, _capi.stream_ptr()), 'pgcn bf16')
    got = out.cpu().numpy()
    sc = np.abs(g['y']).max()
    assert maxabs(got, g['y']) < 2e-2 * sc, (maxabs(got, g['y']), sc)

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()

    orig = OT.pgraphconv

    def rounded(xx, P):
        class Q(dict):
            pass
        P2 = {'W': bf(P['W']), 'e_0': P['e_0'], 'e_1': P['e_1'], 'bias': P['bias']}
        return orig(bf(xx), P2)
    try:
        OT.pgraphconv = rounded
        want = OT.pgcn_stack(g['x'].copy(), N.Params(sdn))
    finally:
        OT.pgraphconv = orig
    assert maxabs(got, want) < 2e-3 * sc, (maxabs(got, want), sc)
