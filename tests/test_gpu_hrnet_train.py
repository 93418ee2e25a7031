"""VERDICT r4 missing 4 / SURVEY.md 8f rank 4: the HRNet-W48 backbone TRAINS (dir_amd/train/hrnet.py) -- stand-alone through the mirror module's
`.train()` mode, and as the backbone of the whole network (`DIR(backbone='hrnet_w48').train()`, BASELINE config 5's per-GPU work).  The reference
has no HRNet, so the yardstick is torch itself: the mirror module is a container of real nn.Conv2d / nn.BatchNorm2d parameters, and the test
states the architecture's forward on those modules with torch ops (float64, CPU, training-mode BatchNorm) and lets autograd differentiate it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dir_amd import synth

pytestmark = pytest.mark.gpu
SEED = 1234


def _torch_forward(m, x):
    """HRNetW48's forward as its docstring words it, on the module's own layers (training-mode BatchNorm when m.training)"""
    def cb(seq, t, relu):
        t = seq[1](seq[0](t))
        return F.relu(t) if relu else t

    def bottleneck(b, t):
        idn = t
        o = F.relu(b.bn1(b.conv1(t)))
        o = F.relu(b.bn2(b.conv2(o)))
        o = b.bn3(b.conv3(o))
        if b.downsample is not None:
            idn = b.downsample(t)
        return F.relu(o + idn)

    def basic(b, t):
        o = F.relu(b.bn1(b.conv1(t)))
        o = b.bn2(b.conv2(o))
        return F.relu(o + t)

    def module(mod, xs):
        nb = len(xs)
        ys = []
        for b in range(nb):
            y = xs[b]
            for blk in mod.branches[b]:
                y = basic(blk, y)
            ys.append(y)
        outs = []
        for i in range(nb):
            acc = None
            for j in range(nb):
                if j == i:
                    t = ys[j]
                elif j > i:
                    t = cb(mod.fuse_layers[i][j], ys[j], False)
                    t = t.repeat_interleave(2 ** (j - i), 2).repeat_interleave(2 ** (j - i), 3)
                else:
                    t = ys[j]
                    for s, seq in enumerate(mod.fuse_layers[i][j]):
                        t = cb(seq, t, s < i - j - 1)
                acc = t if acc is None else acc + t
            outs.append(F.relu(acc))
        return outs
    x = F.relu(m.bn1(m.conv1(x)))
    x = F.relu(m.bn2(m.conv2(x)))
    for b in m.layer1:
        x = bottleneck(b, x)
    xs = [cb(m.transition1[0], x, True), cb(m.transition1[1], x, True)]
    for st in (2, 3, 4):
        if st > 2:
            xs = xs + [cb(getattr(m, 'transition%d' % (st - 1)), xs[-1], True)]
        for mod in getattr(m, 'stage%d' % st):
            xs = module(mod, xs)
    return [cb(m.incre[b], xs[b], True) for b in range(4)]


def _load(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed, cond=True)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)


def test_upsample_nearest_add_and_backward():
    from dir_amd import _capi
    L = _capi.lib()
    g = torch.Generator(device='cuda').manual_seed(3)
    for f in (1, 2, 4, 8):
        src = torch.randn(3, 5, 6, 48, device='cuda', generator=g)
        dst = torch.randn(3, 5 * f, 6 * f, 48, device='cuda', generator=g)
        want = dst + src.repeat_interleave(f, 1).repeat_interleave(f, 2)
        _capi.check(L.dir_upsample_nearest_add_f32(_capi.ptr(src), _capi.ptr(dst), 3, 5, 6, 48, f, _capi.stream_ptr()), 'up')
        assert torch.equal(dst, want), f
        gy = torch.randn(3, 5 * f, 6 * f, 48, device='cuda', generator=g)
        gx = torch.empty(3, 5, 6, 48, device='cuda')
        _capi.check(L.dir_upsample_nearest_backward_f32(_capi.ptr(gy), _capi.ptr(gx), 3, 5, 6, 48, f, _capi.stream_ptr()), 'upb')
        ref = gy.double().view(3, 5, f, 6, f, 48).sum((2, 4))
        assert float((gx.double() - ref).abs().max()) < 1e-5 * max(1.0, f), f


@pytest.mark.parametrize('frozen,arith', [(True, 'f32'), (False, 'f32'), (True, 'f16x3'), (False, 'f16x3')])
def test_hrnet_module_trains_like_torch_autograd(frozen, arith, monkeypatch):
    """forward, running statistics and every parameter's gradient of the stand-alone HRNetW48 module in .train() mode against torch autograd in
    float64 over the same parameters (6 images of 64x64).
    frozen = True: BatchNorm normalises with its running statistics (dir_amd.train.ops.frozen_batchnorm(); torch: the BatchNorm modules in
    .eval()); frozen = False: batch statistics.  A float32 gradient of ~300 stacked convolution / BatchNorm / ReLU layers is only reproducible
    to what its own rounding allows (ReLU masks flip on values within rounding of zero; the training-mode BatchNorm backward amplifies rounding
    by 1e4..1e5: G20c / G20e of the ResNet network), so the gate is torch itself: the same graph run by torch in float32 against its float64 run
    gives the noise floor, and this implementation must sit within 6x of it at the median, the 90th percentile and the maximum over the 927
    tensors.  The forward and the running statistics are held to 2e-5.
    arith = 'f32': exact fp32 convolutions (DIR_TRAIN_ARITH=f32) -- THE PIN of the composition (which tensor feeds which, every fuse path, every
    accumulation).  arith = 'f16x3': the training step's default split-precision convolutions.  Same forward (1e-6), same weight-gradient
    kernels; the DATA-gradient convolutions split the incoming gradient into two f16 parts under ONE power-of-two scale per tensor, and on
    this network's heavy-tailed gradients (frozen synthetic statistics, 2x2 maps) the elements 2^-18 below the tensor's maximum keep ~16 bits:
    measured median 1.4e-4 / max 2e-2 with frozen statistics -- gated as a band (the ResNet network sits at 1e-6 under the same kernels, G20e)."""
    from dir_amd.models.backbone.hrnet import HRNetW48
    from dir_amd.train import ops as O, conv as TC
    import contextlib
    monkeypatch.setattr(TC, 'ARITH', arith)
    monkeypatch.setattr(TC, 'WGRAD_ARITH', arith)
    torch.manual_seed(0)
    net = HRNetW48()
    _load(net, SEED)
    ref = HRNetW48()
    ref.load_state_dict(net.state_dict())
    ref = ref.double().train()
    if frozen:
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}          # before the step (training-mode BatchNorm moves the running statistics)
    net = net.cuda().train()
    img = synth.synth_input('hrtrain.img', (6, 3, 64, 64), SEED)
    gys = [synth.synth_input('hrtrain.g%d' % b, (6, c, 16 >> b, 16 >> b), SEED) for b, c in enumerate((256, 512, 1024, 2048))]
    x = torch.from_numpy(img).cuda()
    with (O.frozen_batchnorm() if frozen else contextlib.nullcontext()):
        feats = net(x)
        sum((f * torch.from_numpy(g).cuda()).sum() for f, g in zip(feats, gys)).backward()
    want = _torch_forward(ref, torch.from_numpy(img).double())
    sum((f * torch.from_numpy(g).double()).sum() for f, g in zip(want, gys)).backward()
    # the yardstick's own fp32 noise: the same torch graph in float32 on the CPU against its float64 self
    ref32 = HRNetW48()
    ref32.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()} if frozen else {k: v.float() for k, v in sd0.items()})
    ref32 = ref32.float().train()
    if frozen:
        for m in ref32.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    w32 = _torch_forward(ref32, torch.from_numpy(img))
    sum((f * torch.from_numpy(g)).sum() for f, g in zip(w32, gys)).backward()
    fe, fe32 = [], []
    for b in range(4):
        fe.append(float((feats[b].detach().double().cpu() - want[b].detach()).abs().max() / want[b].detach().abs().max()))
        fe32.append(float((w32[b].detach().double() - want[b].detach()).abs().max() / want[b].detach().abs().max()))
    print('forward c1..c4 vs float64: %s (torch float32: %s)' % (np.array2string(np.array(fe), precision=2), np.array2string(np.array(fe32), precision=2)))
    assert max(fe) < 2e-5, fe
    sd_n, sd_r = net.state_dict(), ref.state_dict()
    for k in sd_r:
        if 'running_' in k:
            e = float((sd_n[k].double().cpu() - sd_r[k]).abs().max() / (sd_r[k].abs().max() + 1e-12))
            assert e < 2e-5, (k, e)
    errs, e32 = {}, {}
    gp, gp32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    for k, p in net.named_parameters():
        assert p.grad is not None and gp[k].grad is not None, k
        den = gp[k].grad.abs().max() + 1e-30
        errs[k] = float((p.grad.double().cpu() - gp[k].grad).abs().max() / den)
        e32[k] = float((gp32[k].grad.double() - gp[k].grad).abs().max() / den)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    q = lambda d: np.percentile(list(d.values()), (50, 90, 100))  # noqa: E731
    print('HRNet-W48 training (%s statistics, %s convolutions): %d parameter gradients vs torch autograd in float64, relative to each tensor\'s maximum: '
          'median / 90 %% / max  %s   (torch float32 on the CPU against the same: %s)   worst %s'
          % ('frozen' if frozen else 'batch', arith, len(errs), np.array2string(q(errs), precision=2), np.array2string(q(e32), precision=2), worst))
    if arith == 'f32' or not frozen:
        # gate: no further from float64 than six times what torch's own float32 run of the same graph is.  The MAXIMUM of the batch-statistics /
        # f16x3 case gets ten times: it is one tensor of 927 in a chaotic regime (BatchNorms over 24 rows of 2 x 2 maps amplify a 1e-7 change of
        # an upstream statistic to tens of percent of ONE bias gradient) -- round 5 measured 0.10 / 0.10 / 0.32 for three equally exact orders of
        # forming the batch statistics (stored map in 256-row chunks; the same + consumer-side apply; convolution-epilogue tiles), with the same
        # forward (1e-6), running statistics (2e-5), median and 90th percentile every time, and 0.10 in exact fp32 under all three.
        lim = (6.0, 6.0, 10.0 if (arith == 'f16x3' and not frozen) else 6.0)
        assert all(a_ < l_ * b_ + 1e-6 for a_, b_, l_ in zip(q(errs), q(e32), lim)), (q(errs), q(e32))
    else:
        assert q(errs)[0] < 1e-3 and q(errs)[2] < 0.1, q(errs)


def test_config5_network_takes_training_steps():
    """DIR(backbone='hrnet_w48').train(): the whole-network step of BASELINE config 5 (HRNet-W48 + init + stages) -- the 42-term objective falls under
    AdamW and every HRNet parameter receives a gradient"""
    from conftest import loss_case
    import os
    from dir_amd.models.dir import DIR
    HERE = os.path.dirname(os.path.abspath(__file__))
    g8 = dict(np.load(os.path.join(HERE, 'golden', 'g8_loss.npz')))
    net = DIR(21, 'unused', 0, backbone='hrnet_w48')
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, SEED, cond=True)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    net = net.cuda().train()
    img = torch.from_numpy(synth.synth_input('loss.img', (2, 3, 256, 256), SEED)).cuda()
    preds, gt, faces, _, _, gt_seg, gt_dense = loss_case(g8)
    for m, f in zip((net.init_regressor.mano_layer_left, net.init_regressor.mano_layer_right), faces):     # non-degenerate triangles, as in G8
        m.th_faces.copy_(torch.from_numpy(f.astype(np.int64)).to(m.th_faces.dtype))
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    target = {k: dv(v) for k, v in gt.items() if 'center' not in k}
    target.update(seg=dv(gt_seg), dense=dv(gt_dense))
    meta = {k: dv(v) for k, v in gt.items() if 'center' in k}
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=2e-5)
    totals = []
    for _ in range(3):
        opt.zero_grad()
        outs_list, loss = net({'img': img}, target, meta)
        total = sum(loss[k] for k in loss)
        total.backward()
        opt.step()
        totals.append(float(total.detach()))
    got = {k for k, p in net.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0}
    dead = ('backbone.incre.0.', 'backbone.stage4.2.fuse_layers.0.')          # feed c1 only, which DIR does not read: no gradient, like under torch
    hr = [k for k, _ in net.named_parameters() if k.startswith('backbone.') and not k.startswith(dead)]
    missing = [k for k in hr if k not in got]
    assert len(hr) > 900 and not missing, missing[:8]
    assert all(p.grad is None for k, p in net.named_parameters() if k.startswith(dead))
    assert all(np.isfinite(totals)) and totals[2] < totals[0], totals
