"""HRNet-W48 backbone in TRAINING form (batch-statistics BatchNorm2d), forward + backward from libdir_hip.so -- VERDICT r4 missing 4 / SURVEY.md 8f
rank 4 (BASELINE config 5 trains "whatever DIR it builds", train.py:58-91).  NO REFERENCE COUNTERPART: /root/reference has no HRNet; the
architecture is the one dir_amd/models/backbone/hrnet.py states (its docstring) and oracle/hrnet.py restates for inference:

    stem conv3x3/2 bn relu x 2 -> layer1 (4 Bottlenecks) -> transition1 -> stage2 (1 module, 2 branches) -> transition2 -> stage3 (4 modules,
    3 branches) -> transition3 -> stage4 (3 modules, 4 branches) -> incre (conv1x1 bn relu to 256 / 512 / 1024 / 2048)
    module: 4 BasicBlocks per branch, then y_i = relu(sum_j f_ij(x_j)); f_ii = identity; j > i: conv1x1 bn, nearest upsample x 2^(j-i);
            j < i: (i - j) conv3x3/2 bn, ReLU after all but the last

P: {state-dict key -> fp32 cuda tensor, reference layouts}; activations NHWC fp32; running statistics updated like nn.BatchNorm2d.  Every
arithmetic step is a library call: the convolutions / BatchNorms / ReLUs of dir_amd/train/{conv,blocks,ops}.py (the widths 48 are padded to the
split-precision kernels' 32-channel slabs on the fly by conv_fwd / conv_dgrad), dir_upsample_nearest_add_f32 / _backward_f32, dir_axpy_f32.
Gradients: tests/test_gpu_hrnet_train.py against torch autograd (float64) over the mirror module's own parameters.
"""
import torch

from .. import _capi
from . import blocks as TB
from . import conv as TC
from . import ops as O

WIDTHS = (48, 96, 192, 384)
MODULES = ((2, 1), (3, 4), (4, 3))


def is_hrnet(P, pre='backbone.'):
    return (pre + 'stage4.0.fuse_layers.0.1.0.weight') in P


def _sub(P, pre):
    n = len(pre)
    return {k[n:]: v for k, v in P.items() if k.startswith(pre)}


def _put(G, pre, g):
    for k, v in g.items():
        G[pre + k] = v


# ------------------------------------------------------------------------------------------------------------------------------ pieces
def _up_add(src, dst, f):
    B, h, w, C = src.shape
    _capi.check(_capi.lib().dir_upsample_nearest_add_f32(_capi.ptr(src), _capi.ptr(dst), B, h, w, C, f, _capi.stream_ptr()), 'dir_upsample_nearest_add_f32')


def _up_bwd(gy, f):
    B, H, W, C = gy.shape
    gx = torch.empty(B, H // f, W // f, C, device=gy.device)
    _capi.check(_capi.lib().dir_upsample_nearest_backward_f32(_capi.ptr(gy), _capi.ptr(gx), B, H // f, W // f, C, f, _capi.stream_ptr()),
                'dir_upsample_nearest_backward_f32')
    return gx


def cb_forward(P, pre, x, stride, pad, relu, residual=None):
    """Sequential(Conv2d(bias=False), BatchNorm2d) (+ ReLU) (+ residual before it): keys pre + '0.weight', pre + '1.*'"""
    st = []                                                # the BatchNorm's chunk partials from the convolution's epilogue (round 5)
    h = TC.conv_fwd(x, P[pre + '0.weight'], None, stride, pad, oihw=True, stats=st)
    y, s_bn = TB.bn_fwd(P, pre + '1.', h, relu=relu, residual=residual, partials=st)
    return y, dict(x=x, bn=s_bn, stride=stride, pad=pad, relu=relu)


def cb_backward(P, pre, s, gy, G, need_gx=True, add_gx=None):
    """gy: the gradient of the BatchNorm's output (after the ReLU's mask when a residual was added: the caller applies relu_bwd there)"""
    g = TB.bn_bwd(P, pre + '1.', s['bn'], gy, G, relu=s['relu'])
    return TB._conv_bwd(P, pre + '0.', s['x'], g, s['stride'], s['pad'], G, need_gx=need_gx, add_gx=add_gx)


def basic_forward(P, x):
    """BasicBlock: conv3x3 bn relu conv3x3 bn, + x, relu"""
    st1, st2 = [], []
    h = TC.conv_fwd(x, P['conv1.weight'], None, 1, 1, oihw=True, stats=st1)
    a1, p1, s1 = TB.bn_relu_into_conv(P, 'bn1.', h, partials=st1)          # widths that are whole 32-channel slabs: applied where conv2 reads the map (round 5)
    h = TC.conv_fwd(a1, P['conv2.weight'], None, 1, 1, oihw=True, pre=p1, stats=st2)
    y, s2 = TB.bn_fwd(P, 'bn2.', h, relu=True, residual=x, partials=st2)
    return y, dict(x=x, a1=a1, p1=p1, bn1=s1, bn2=s2, y=y)


def basic_backward(P, s, gy, gy_masked=False, mask_gx=None):
    """gy_masked / mask_gx: the block's final ReLU backward applied by the NEXT block's conv1 data gradient / the previous block's applied here
    (dir_amd.train.blocks.bottleneck_backward; inside a branch a block's input is the previous block's output)"""
    G = {}
    g = gy.contiguous() if gy_masked else O.relu_bwd(gy.contiguous(), s['y'])                  # gradient of (bn2 out + x)
    g2 = TB.bn_bwd(P, 'bn2.', s['bn2'], g, G)
    sp = TB.bn_bwd_spec(P, 'bn1.', s['bn1'], True)
    g1 = TB._conv_bwd(P, 'conv2.', s['a1'], g2, 1, 1, G, pre=s.get('p1'), bn_bwd=sp)
    g1 = TB.bn_bwd(P, 'bn1.', s['bn1'], g1, G, relu=True, spec=sp)
    gx = TB._conv_bwd(P, 'conv1.', s['x'], g1, 1, 1, G, add_gx=g, mask_gx=mask_gx)
    return gx, G


def module_forward(P, xs):
    """HRModule: xs -> ys (one map per branch)"""
    nb = len(xs)
    ctx = {'blocks': [], 'fuse': []}
    ys = []
    for b in range(nb):
        y = xs[b]
        for k in range(4):
            p = 'branches.%d.%d.' % (b, k)
            y, c = basic_forward(_sub(P, p), y)
            ctx['blocks'].append((p, c))
        ys.append(y)
    outs = []
    for i in range(nb):
        # same-resolution terms first (identity, then the strided chains, each added in its last BatchNorm's launch), then the upsampled ones
        acc, rec = ys[i], {'down': [], 'up': []}
        for j in range(i):
            t, chain = ys[j], []
            for s_ in range(i - j):
                last = s_ == i - j - 1
                p = 'fuse_layers.%d.%d.%d.' % (i, j, s_)
                t, c = cb_forward(P, p, t, 2, 1, relu=not last, residual=acc if last else None)
                chain.append((p, c))
            acc = t
            rec['down'].append((j, chain))
        if acc is ys[i]:
            acc = ys[i].clone()                              # the identity term must not be overwritten by the additions below
        for j in range(i + 1, nb):
            p = 'fuse_layers.%d.%d.' % (i, j)
            t, c = cb_forward(P, p, ys[j], 1, 0, relu=False)
            _up_add(t, acc, 2 ** (j - i))
            rec['up'].append((j, p, c))
        y = O.relu_fwd(acc)
        rec['y'] = y
        outs.append(y)
        ctx['fuse'].append(rec)
    return outs, ctx


def module_backward(P, ctx, gys, G):
    """gys: gradients of the module's outputs (one per branch) -> gradients of its inputs"""
    nb = len(gys)
    g_ys = [None] * nb                                       # gradients of the branches' outputs (before the fuse layers)

    def acc_(j, g):
        if g_ys[j] is None:
            g_ys[j] = g
        else:
            O.axpy(g_ys[j], g)
    for i in range(nb):
        rec = ctx['fuse'][i]
        if gys[i] is None:                                   # an output nobody reads (c1's path under DIR): its fuse layers get no gradient
            continue
        g = O.relu_bwd(gys[i].contiguous(), rec['y'])        # gradient of the sum: every term receives it
        acc_(i, g)                                           # (may alias g_ys[i]: it is only added to by LATER outputs' terms, when g is done with)
        for j, p, c in rec['up']:
            gs = _up_bwd(g, 2 ** (j - i))
            acc_(j, cb_backward(P, p, c, gs, G))
        for j, chain in rec['down']:
            t = g
            for p, c in reversed(chain):
                t = cb_backward(P, p, c, t, G)
            acc_(j, t)
    gxs = []
    bi = len(ctx['blocks'])
    for b in reversed(range(nb)):
        g = g_ys[b]
        masked = False
        for k in range(3, -1, -1):
            bi -= 1
            p, c = ctx['blocks'][bi]
            prev_y = ctx['blocks'][bi - 1][1]['y'] if (k > 0 and TB.FUSE_RELU_BWD and c['x'].shape[-1] % 32 == 0) else None
            g, gb = basic_backward(_sub(P, p), c, g, gy_masked=masked, mask_gx=prev_y)
            masked = prev_y is not None
            _put(G, p, gb)
        gxs.append(g)
    return gxs[::-1]


# ----------------------------------------------------------------------------------------------------------------------------- backbone
def hrnet_forward(P, img, ctx, pre='backbone.'):
    """HRNetW48.forward in training form.  img NCHW fp32 [B,3,H,W] (H, W multiples of 32) -> [c1, c2, c3, c4] NHWC fp32; ctx receives what
    hrnet_backward needs"""
    Pb = _sub(P, pre)
    x = img.permute(0, 2, 3, 1).contiguous()
    ctx['hr'] = c = {'img': x}
    x, c['stem1'] = _stem(Pb, 'conv1.weight', 'bn1.', x)
    x, c['stem2'] = _stem(Pb, 'conv2.weight', 'bn2.', x)
    c['layer1'] = []
    for b in range(4):
        p = 'layer1.%d.' % b
        x, cb = TB.bottleneck_forward(_sub(Pb, p), x, 1)
        c['layer1'].append((p, cb))
    c['t1_in'] = x
    x0, c['t1_0'] = cb_forward(Pb, 'transition1.0.', x, 1, 1, relu=True)
    x1, c['t1_1'] = cb_forward(Pb, 'transition1.1.', x, 2, 1, relu=True)
    xs = [x0, x1]
    c['stages'] = []
    for st, n in MODULES:
        tr = None
        if st > 2:
            t, tr = cb_forward(Pb, 'transition%d.' % (st - 1), xs[-1], 2, 1, relu=True)
            xs = xs + [t]
        mods = []
        for m in range(n):
            p = 'stage%d.%d.' % (st, m)
            xs, cm = module_forward(_sub(Pb, p), xs)
            mods.append((p, cm))
        c['stages'].append((st, tr, mods))
    feats, c['incre'] = [], []
    for b in range(4):
        y, ci = cb_forward(Pb, 'incre.%d.' % b, xs[b], 1, 0, relu=True)
        feats.append(y)
        c['incre'].append(ci)
    return feats


def _stem(Pb, wkey, bnpre, x):
    st = []
    h = TC.conv_fwd(x, Pb[wkey], None, 2, 1, oihw=True, stats=st)
    y, s_bn = TB.bn_fwd(Pb, bnpre, h, relu=True, partials=st)
    return y, dict(x=x, bn=s_bn)


def hrnet_backward(P, ctx, g_feats, G, flush=None, pre='backbone.'):
    """g_feats: gradients of [c1, c2, c3, c4]; None = the feature has no consumer (DIR reads c2, c3, c4: c1's `incre.0` and the last module's
    fuse layers into branch 0 then receive no gradient, like under torch).  Fills G[pre + *]"""
    Pb, c = _sub(P, pre), ctx['hr']
    Gb = {}
    if flush is None:
        flush = lambda g_: None      # noqa: E731
    gxs = []
    for b in range(4):
        g = g_feats[b]
        gxs.append(None if g is None else cb_backward(Pb, 'incre.%d.' % b, c['incre'][b], g.contiguous(), Gb))
    for st, tr, mods in reversed(c['stages']):
        for p, cm in reversed(mods):
            gm = {}
            gxs = module_backward(_sub(Pb, p), cm, gxs, gm)
            _put(Gb, p, gm)
        if tr is not None:                                   # the stage's new branch came from the previous stage's last branch
            g_new = gxs.pop()
            O.axpy(gxs[-1], cb_backward(Pb, 'transition%d.' % (st - 1), tr, g_new, Gb))
        _put(G, pre, Gb)
        Gb = {}
        flush(G)
    g = cb_backward(Pb, 'transition1.0.', c['t1_0'], gxs[0], Gb)
    g = cb_backward(Pb, 'transition1.1.', c['t1_1'], gxs[1], Gb, add_gx=g)
    for p, cb in reversed(c['layer1']):
        g, gb = TB.bottleneck_backward(_sub(Pb, p), cb, g)
        _put(Gb, p, gb)
    g = TB.bn_bwd(Pb, 'bn2.', c['stem2']['bn'], g, Gb, relu=True)
    g = TB._conv_bwd(Pb, 'conv2.', c['stem2']['x'], g, 2, 1, Gb)
    g = TB.bn_bwd(Pb, 'bn1.', c['stem1']['bn'], g, Gb, relu=True)
    TB._conv_bwd(Pb, 'conv1.', c['stem1']['x'], g, 2, 1, Gb, need_gx=False)          # the image is data
    _put(G, pre, Gb)
    flush(G)
