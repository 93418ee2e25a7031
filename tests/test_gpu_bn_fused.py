"""Round 5: a training-mode BatchNorm + ReLU whose only consumer is a convolution is applied where that convolution READS the map
(dir_bn_train_stats -> pre_scale / pre_shift; dir_conv2d_forward's pre-activation / dir_split_f16_forward's; dir_conv2d_wgrad_f16x3_pre)
instead of in a pass of its own (models/backbone/resnet.py:125-131 bn1 / bn2, models/backbone/hourglass.py:60-67).  Held to the unfused
step (same statistics bits, outputs and gradients to fp32 rounding) and -- through tests/test_gpu_blocks_bwd.py, test_gpu_full_bwd.py -- to the
reference's autograd."""
import numpy as np
import pytest
import torch

from dir_amd import _capi
from dir_amd.train import blocks as TB
from dir_amd.train import conv as TC
from dir_amd.train import ops as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_bn_train_stats_gives_the_statistics_of_bn_train_forward():
    g = torch.Generator(device='cuda').manual_seed(11)
    for R, C in ((3000, 64), (8192, 256), (700, 32)):
        x = torch.randn(R, C, device='cuda', generator=g) * 3 + 0.7
        w, b = torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g)
        rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        rm2, rv2 = rm.clone(), rv.clone()
        y, (sm, sr) = O.bn_train_fwd(x, w, b, rm, rv, relu=True)
        (sm2, sr2), (ps, pb) = O.bn_train_stats(x, w, b, rm2, rv2)
        assert torch.equal(sm, sm2) and torch.equal(sr, sr2) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
        y2 = torch.relu(x * ps + pb)
        assert float((y - y2).abs().max()) < 1e-5 * float(y.abs().max())


@pytest.mark.parametrize('shape', [(4, 32, 32, 64, 64, 3, 1), (2, 64, 64, 128, 256, 1, 1), (3, 17, 20, 32, 96, 3, 2), (2, 16, 16, 256, 128, 1, 1)])
def test_wgrad_with_pre_activation_equals_wgrad_of_the_materialised_operand(shape):
    B, H, W, Cin, Cout, k, stride = shape
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    ps, pb = torch.rand(Cin, device='cuda', generator=g) + 0.5, torch.randn(Cin, device='cuda', generator=g) * 0.5
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randn(B, Ho, Wo, Cout, device='cuda', generator=g)
    a = torch.relu(torch.addcmul(pb, x, ps))
    TC.end_step()
    want = TC.conv_wgrad(a, gy, (Cout, k, k, Cin), stride, pad)
    got = TC.conv_wgrad(x, gy, (Cout, k, k, Cin), stride, pad, pre=(ps, pb))
    assert rel(got, want) < 2e-6, rel(got, want)          # (the operand differs by the rounding of fmaf vs multiply-add only)


@pytest.mark.parametrize('kind', ['bottleneck', 'bottleneck_s2', 'residual', 'residual_skip'])
def test_blocks_with_the_batchnorm_applied_in_the_consumer_equal_the_unfused_blocks(kind):
    g = torch.Generator(device='cuda').manual_seed(3)

    def rnd(*s, scale=1.0):
        return torch.randn(*s, device='cuda', generator=g) * scale
    B, S = 4, 32
    if kind.startswith('bottleneck'):
        stride = 2 if kind.endswith('s2') else 1
        cin, pl = 256, 64
        P = {'conv1.weight': rnd(pl, cin, 1, 1, scale=0.06), 'conv2.weight': rnd(pl, pl, 3, 3, scale=0.04), 'conv3.weight': rnd(4 * pl, pl, 1, 1, scale=0.1)}
        for n, c in (('bn1.', pl), ('bn2.', pl), ('bn3.', 4 * pl)):
            P.update({n + 'weight': torch.rand(c, device='cuda', generator=g) + 0.5, n + 'bias': rnd(c, scale=0.3), n + 'running_mean': torch.zeros(c, device='cuda'),
                      n + 'running_var': torch.ones(c, device='cuda')})
        if stride == 2:
            P['downsample.0.weight'] = rnd(4 * pl, cin, 1, 1, scale=0.06)
            P.update({'downsample.1.weight': torch.rand(4 * pl, device='cuda', generator=g) + 0.5, 'downsample.1.bias': rnd(4 * pl, scale=0.3),
                      'downsample.1.running_mean': torch.zeros(4 * pl, device='cuda'), 'downsample.1.running_var': torch.ones(4 * pl, device='cuda')})
        fwd = lambda P_, x_: TB.bottleneck_forward(P_, x_, stride)          # noqa: E731
        bwd = TB.bottleneck_backward
        x = rnd(B, S, S, cin)
    else:
        cin, cout = (128, 256) if kind.endswith('skip') else (256, 256)
        mid = cout // 2
        P = {'conv1.conv.weight': rnd(mid, cin, 1, 1, scale=0.08), 'conv1.conv.bias': rnd(mid, scale=0.1), 'conv2.conv.weight': rnd(mid, mid, 3, 3, scale=0.03),
             'conv2.conv.bias': rnd(mid, scale=0.1), 'conv3.conv.weight': rnd(cout, mid, 1, 1, scale=0.08), 'conv3.conv.bias': rnd(cout, scale=0.1),
             'skip_layer.conv.weight': rnd(cout, cin, 1, 1, scale=0.08), 'skip_layer.conv.bias': rnd(cout, scale=0.1)}
        for n, c in (('bn1.', cin), ('bn2.', mid), ('bn3.', mid)):
            P.update({n + 'weight': torch.rand(c, device='cuda', generator=g) + 0.5, n + 'bias': rnd(c, scale=0.3), n + 'running_mean': torch.zeros(c, device='cuda'),
                      n + 'running_var': torch.ones(c, device='cuda')})
        fwd, bwd = TB.residual_forward, TB.residual_backward
        x = rnd(B, S, S, cin)
    res = {}
    for fused in (False, True):
        O.FUSE_BN = fused
        try:
            Pc = {k: v.clone() for k, v in P.items()}
            TC.end_step()
            y, ctx = fwd(Pc, x)
            assert ('p1' in ctx and ctx['p1'] is not None) == fused
            gy = torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(9))
            gx, G = bwd(Pc, ctx, gy)
            res[fused] = (y, gx, G, Pc)
        finally:
            O.FUSE_BN = True
    (y0, gx0, G0, P0), (y1, gx1, G1, P1) = res[False], res[True]
    assert rel(y1, y0) < 2e-6 and rel(gx1, gx0) < 2e-5, (rel(y1, y0), rel(gx1, gx0))
    assert set(G0) == set(G1)
    gmax = max(float(v.abs().max()) for v in G0.values())
    for k in G0:
        if k in ('conv1.conv.bias', 'conv2.conv.bias'):          # a bias in front of a BatchNorm: its gradient is rounding noise around zero both ways
            assert float(G1[k].abs().max()) < 1e-4 * gmax and float(G0[k].abs().max()) < 1e-4 * gmax, k
            continue
        assert rel(G1[k], G0[k]) < 2e-5, (k, rel(G1[k], G0[k]))
    for k in P0:
        if 'running' in k:
            assert rel(P1[k], P0[k]) < 1e-6, k


@pytest.mark.parametrize('shape', [(4, 32, 32, 64, 64, 3, 1), (2, 64, 64, 64, 256, 1, 1), (32, 16, 16, 256, 256, 3, 1), (3, 17, 20, 32, 96, 3, 2),
                                   (8, 8, 8, 512, 2048, 1, 1), (2, 64, 64, 256, 64, 1, 1)])
def test_convolution_epilogue_forms_the_chunk_partials_of_the_following_batchnorm(shape):
    """dir_conv2d_forward_stats: p1 = per M tile and channel the sum of the output's valid rows, p2 = the sum of squared deviations from the tile mean;
    the BatchNorm statistics combined from them equal those formed from the stored map (dir_bn_train_forward)"""
    B, H, W, Cin, Cout, k, stride = shape
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    w = torch.randn(Cout, Cin, k, k, device='cuda', generator=g) * (1.0 / (Cin * k * k) ** 0.5)
    bias = torch.randn(Cout, device='cuda', generator=g)
    TC.end_step()
    st = []
    y = TC.conv_fwd(x, w, bias, stride, k // 2, oihw=True, stats=st)
    y0 = TC.conv_fwd(x, w, bias, stride, k // 2, oihw=True)
    assert torch.equal(y, y0)
    assert len(st) == 1 and st[0][2] in (64, 128, 256), st and st[0][2]
    p1, p2, rows = st[0]
    y2 = y.reshape(-1, Cout).double()
    R = y2.shape[0]
    nch = (R + rows - 1) // rows
    for c in range(nch):
        blk = y2[c * rows:(c + 1) * rows]
        s1 = blk.sum(0)
        m2 = ((blk - blk.mean(0)) ** 2).sum(0)
        assert float((p1[c].double() - s1).abs().max()) < 1e-5 * float(s1.abs().max() + blk.abs().max()), (c, 'sum')
        assert float((p2[c].double() - m2).abs().max()) < 1e-5 * float(m2.abs().max()), (c, 'M2')
    if R > 512:
        wbn, bbn = torch.rand(Cout, device='cuda', generator=g) + 0.5, torch.randn(Cout, device='cuda', generator=g)
        rm, rv = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
        rm2, rv2 = rm.clone(), rv.clone()
        yb, (sm, sr) = O.bn_train_fwd(y.view(-1, Cout), wbn, bbn, rm, rv, relu=True)
        (sm2, sr2), (ps, pb) = O.bn_train_stats_from_partials(st[0], R, wbn, bbn, rm2, rv2)
        assert rel(sm2, sm) < 1e-5 and rel(sr2, sr) < 1e-5 and rel(rm2, rm) < 1e-5 and rel(rv2, rv) < 1e-5
        ya = O.bn_train_apply(y.view(-1, Cout), wbn, bbn, (sm2, sr2), relu=True)
        assert float((ya - yb).abs().max()) < 1e-5 * float(yb.abs().max())


def test_data_gradient_with_the_relu_mask_in_its_epilogue():
    """dir_conv2d_forward_masked through conv_dgrad(mask=...): gx = (dgrad + add) where mask > 0, else 0 -- bit for bit the separate relu_bwd pass"""
    g = torch.Generator(device='cuda').manual_seed(13)
    for (B, H, Cin, Cout, k) in ((4, 32, 256, 64, 1), (2, 64, 64, 64, 3), (8, 16, 1024, 256, 1)):
        w = torch.randn(Cout, Cin, k, k, device='cuda', generator=g) * 0.05
        gy = torch.randn(B, H, H, Cout, device='cuda', generator=g)
        add = torch.randn(B, H, H, Cin, device='cuda', generator=g)
        y_prev = torch.relu(torch.randn(B, H, H, Cin, device='cuda', generator=g))
        TC.end_step()
        want = O.relu_bwd(TC.conv_dgrad(w, gy, 1, k // 2, H, H, oihw=True, add=add), y_prev)
        got = TC.conv_dgrad(w, gy, 1, k // 2, H, H, oihw=True, add=add, mask=y_prev)
        assert torch.equal(got, want), (B, H, Cin, Cout, k)


def test_backbone_backward_with_and_without_the_fused_relu_backward():
    """the 12 in-layer block boundaries of the ResNet take their ReLU backward in the next block's conv1 data gradient: same gradients bit for bit"""
    import json
    import os
    from dir_amd import synth
    from dir_amd.train import net as TN
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items() if k.startswith('backbone.')}
    sd = synth.synth_state_dict(shapes, 1234)
    img = torch.from_numpy(synth.synth_input('relu_bwd.img', (2, 3, 256, 256), 1234)).cuda()          # (the training stem is written for 256 x 256 inputs, like the reference's data)
    res = {}
    TC.BN_BWD_IN_EPILOGUE = False          # (bit equality is about the mask alone: with it, bn3's backward sums move into the same epilogue -- another summation order)
    for fused in (True, False):
        TB.FUSE_RELU_BWD = fused
        try:
            P = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items() if 'num_batches' not in k}
            TC.end_step()
            ctx = {}
            feats = TN.backbone_forward(P, img, ctx)
            gen = torch.Generator(device='cuda').manual_seed(4)
            gf = [torch.randn(f.shape, device='cuda', generator=gen) for f in feats]
            G = {}
            TN.backbone_backward(P, ctx, gf, G)
            res[fused] = G
        finally:
            TB.FUSE_RELU_BWD = True
    TC.BN_BWD_IN_EPILOGUE = True
    assert set(res[True]) == set(res[False])
    for k in res[True]:
        assert torch.equal(res[True][k], res[False][k]), k


def test_data_gradient_epilogue_forms_the_batchnorm_backward_sums():
    """dir_conv2d_forward_ex through conv_dgrad(bn_bwd=spec): the chunk partials of sum(g m) and sum(g m xhat) (m: the ReLU mask BatchNorm(z) > 0, or the
    output mask already applied) pooled over the chunks equal the sums formed from the stored gradient; dir_bn_train_backward_from_partials then gives
    dir_bn_train_backward's gradients"""
    g = torch.Generator(device='cuda').manual_seed(17)
    for (B, H, Cin, Cout, k, relu, masked) in ((4, 32, 256, 64, 1, True, False), (2, 64, 64, 64, 3, True, False), (8, 16, 1024, 256, 1, False, True), (3, 20, 96, 32, 3, True, False)):
        w = torch.randn(Cout, Cin, k, k, device='cuda', generator=g) * 0.05
        gy = torch.randn(B, H, H, Cout, device='cuda', generator=g)
        z = torch.randn(B, H, H, Cin, device='cuda', generator=g) * 2 + 0.3
        wbn, bbn = torch.rand(Cin, device='cuda', generator=g) + 0.5, torch.randn(Cin, device='cuda', generator=g) * 0.3
        _, st = O.bn_train_fwd(z.view(-1, Cin), wbn, bbn, relu=relu)
        add = torch.randn(B, H, H, Cin, device='cuda', generator=g) if masked else None
        y_prev = torch.relu(torch.randn(B, H, H, Cin, device='cuda', generator=g)) if masked else None
        TC.end_step()
        want_g = TC.conv_dgrad(w, gy, 1, k // 2, H, H, oihw=True, add=add, mask=y_prev)
        spec = O.bn_bwd_spec(z.view(-1, Cin), wbn, bbn, st, relu)
        assert spec is not None
        got_g = TC.conv_dgrad(w, gy, 1, k // 2, H, H, oihw=True, add=add, mask=y_prev, bn_bwd=spec)
        assert torch.equal(got_g, want_g)
        assert len(spec['out']) == 1, 'the epilogue did not form the sums'
        p1, p2, chunks = spec['out'][0]
        xh = ((z.view(-1, Cin) - st[0]) * st[1]).double()
        gm = want_g.view(-1, Cin).double()
        if relu:
            gm = gm * ((xh.float() * wbn + bbn) > 0)
        t1, t2 = gm.sum(0), (gm * xh).sum(0)
        assert float((p1[:chunks].double().sum(0) - t1).abs().max()) < 2e-5 * float(gm.abs().sum(0).max())
        assert float((p2[:chunks].double().sum(0) - t2).abs().max()) < 2e-5 * float((gm * xh).abs().sum(0).max())
        a = O.bn_train_bwd(want_g.view(-1, Cin), z.view(-1, Cin), wbn, st, b=bbn, relu=relu)
        b = O.bn_train_bwd(want_g.view(-1, Cin), z.view(-1, Cin), wbn, st, b=bbn, relu=relu, partials=spec['out'][0])
        for u, v in zip(a, b):
            assert rel(v, u) < 2e-5, rel(v, u)


@pytest.mark.parametrize('kind', ['bottleneck', 'residual_skip'])
def test_blocks_with_the_backward_sums_from_the_epilogue_equal_the_separate_pass(kind):
    g = torch.Generator(device='cuda').manual_seed(5)

    def rnd(*s, scale=1.0):
        return torch.randn(*s, device='cuda', generator=g) * scale
    B, S = 4, 32
    if kind == 'bottleneck':
        cin, pl = 256, 64
        P = {'conv1.weight': rnd(pl, cin, 1, 1, scale=0.06), 'conv2.weight': rnd(pl, pl, 3, 3, scale=0.04), 'conv3.weight': rnd(4 * pl, pl, 1, 1, scale=0.1)}
        bns = (('bn1.', pl), ('bn2.', pl), ('bn3.', 4 * pl))
        fwd, bwd = (lambda P_, x_: TB.bottleneck_forward(P_, x_, 1)), TB.bottleneck_backward
    else:
        cin, cout = 128, 256
        mid = cout // 2
        P = {'conv1.conv.weight': rnd(mid, cin, 1, 1, scale=0.08), 'conv1.conv.bias': rnd(mid, scale=0.1), 'conv2.conv.weight': rnd(mid, mid, 3, 3, scale=0.03),
             'conv2.conv.bias': rnd(mid, scale=0.1), 'conv3.conv.weight': rnd(cout, mid, 1, 1, scale=0.08), 'conv3.conv.bias': rnd(cout, scale=0.1),
             'skip_layer.conv.weight': rnd(cout, cin, 1, 1, scale=0.08), 'skip_layer.conv.bias': rnd(cout, scale=0.1)}
        bns = (('bn1.', cin), ('bn2.', mid), ('bn3.', mid))
        fwd, bwd = TB.residual_forward, TB.residual_backward
    for n, c in bns:
        P.update({n + 'weight': torch.rand(c, device='cuda', generator=g) + 0.5, n + 'bias': rnd(c, scale=0.3), n + 'running_mean': torch.zeros(c, device='cuda'),
                  n + 'running_var': torch.ones(c, device='cuda')})
    x = rnd(B, S, S, cin)
    res = {}
    for on in (False, True):
        TC.BN_BWD_IN_EPILOGUE = on
        try:
            Pc = {k: v.clone() for k, v in P.items()}
            TC.end_step()
            y, ctx = fwd(Pc, x)
            gy = torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(9))
            res[on] = bwd(Pc, ctx, gy)
        finally:
            TC.BN_BWD_IN_EPILOGUE = True
    (gx0, G0), (gx1, G1) = res[False], res[True]
    assert rel(gx1, gx0) < 2e-5, rel(gx1, gx0)
    gmax = max(float(v.abs().max()) for v in G0.values())
    for k in G0:
        if k in ('conv1.conv.bias', 'conv2.conv.bias'):
            assert float(G1[k].abs().max()) < 1e-4 * gmax
            continue
        assert rel(G1[k], G0[k]) < 2e-5, (k, rel(G1[k], G0[k]))
