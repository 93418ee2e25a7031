"""The two residual blocks of the image half in TRAINING form (batch-statistics BatchNorm2d), forward + backward from libdir_hip.so:

    Bottleneck   models/backbone/resnet.py:86-142    conv1x1 - bn - relu - conv3x3(stride) - bn - relu - conv1x1 - bn (+ downsample(x)) - relu
    Residual     models/backbone/hourglass.py:33-70  bn - relu - conv1x1 - bn - relu - conv3x3 - bn - relu - conv1x1 (+ skip_layer(x) | x)

P: {state-dict key relative to the block -> fp32 cuda tensor, reference layouts (conv weights OIHW)}; running statistics in P are
updated like nn.BatchNorm2d does.  Activations NHWC fp32.  y, ctx = *_forward(P, x);  gx, grads = *_backward(P, ctx, gy) with grads in the
reference layouts.  Every arithmetic step is a library call (dir_conv2d_forward, dir_conv2d_wgrad_f32, dir_bn_train_*, dir_relu_*,
dir_axpy_f32, dir_colsum_f32).
"""
from . import conv as TC
from . import ops as O


import os

FUSE_RELU_BWD = os.environ.get('DIR_TRAIN_FUSE_RELU_BWD', '1') == '1'      # round 5: a block's final ReLU backward inside the next block's conv1 data gradient


def _ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def _oihw(g):
    return g.permute(0, 3, 1, 2).contiguous()


def bn_fwd(P, pre, x, momentum=0.1, eps=1e-5, relu=False, residual=None, partials=None):
    """BatchNorm2d in training mode; relu=True: the nn.ReLU that follows it in the same launch (bn_bwd(..., relu=True) undoes both);
    residual: added before that ReLU (a bottleneck's tail; its backward masks with the saved output: relu_bwd, then bn_bwd(relu=False));
    partials: the chunk partials the producing convolution's epilogue formed (conv_fwd(stats=[...])) -- the statistics pass over x is then skipped"""
    C = x.shape[-1]
    if partials:
        x2, w, b = x.view(-1, C), P.get(pre + 'weight'), P.get(pre + 'bias')
        r2 = None if residual is None else residual.contiguous().view(-1, C)
        if O.bn_partials_usable(x2, w, b, partials, r2):
            y, st = O.bn_train_fwd_from_partials(x2, partials[-1], w, b, P.get(pre + 'running_mean'), P.get(pre + 'running_var'), eps, momentum, relu=relu, residual=r2)
            return y.view(x.shape), (x, st)
    y, st = O.bn_train_fwd(x.view(-1, C), P[pre + 'weight'], P[pre + 'bias'], P.get(pre + 'running_mean'), P.get(pre + 'running_var'), eps, momentum,
                           relu=relu, residual=None if residual is None else residual.contiguous().view(-1, C))
    return y.view(x.shape), (x, st)


def bn_relu_into_conv(P, pre, x, momentum=0.1, eps=1e-5, partials=None):
    """BatchNorm2d (training mode) + ReLU whose only consumer is ONE convolution.  -> (operand, pre_act, saved): with the fused path (ops.FUSE_BN,
    maps of more than 512 rows) the operand is x itself and pre_act = (pre_scale, pre_shift) goes to conv_fwd / conv_bwd(pre=...) -- the
    normalised map is never written (round 5); otherwise the operand is the normalised map and pre_act is None.  saved: for bn_bwd(relu=True)"""
    C = x.shape[-1]
    x2 = x.view(-1, C)
    w, b = P.get(pre + 'weight'), P.get(pre + 'bias')
    if O.bn_can_fuse(x2, w, b):
        if O.bn_partials_usable(x2, w, b, partials):       # the producing convolution formed the chunk partials: the map is not read for its statistics
            st, pa = O.bn_train_stats_from_partials(partials[-1], x2.shape[0], w, b, P.get(pre + 'running_mean'), P.get(pre + 'running_var'), eps, momentum)
        else:
            st, pa = O.bn_train_stats(x2, w, b, P.get(pre + 'running_mean'), P.get(pre + 'running_var'), eps, momentum)
        return x, pa, (x, st)
    a, saved = bn_fwd(P, pre, x, momentum, eps, relu=True, partials=partials)
    return a, None, saved


def bn_bwd(P, pre, saved, gy, G, relu=False, spec=None):
    """spec: what bn_bwd_spec returned for this BatchNorm and was handed to the data-gradient convolution that wrote gy (conv_bwd(bn_bwd=spec)): when that
    convolution's epilogue formed the backward sums (spec['out']), the pass over (gy, x) for them is skipped"""
    x, st = saved
    C = x.shape[-1]
    pt = spec['out'][-1] if (spec is not None and spec['out']) else None
    gx, G[pre + 'weight'], G[pre + 'bias'] = O.bn_train_bwd(gy.contiguous().view(-1, C), x.view(-1, C), P[pre + 'weight'], st, b=P[pre + 'bias'], relu=relu,
                                                            partials=pt)
    return gx.view(x.shape)


def bn_bwd_spec(P, pre, saved, relu):
    """the backward sums of BatchNorm `pre` out of the data-gradient convolution that writes the gradient of its output (round 5): pass the result to
    _conv_bwd(bn_bwd=...) and then to bn_bwd(spec=...); None = the separate pass"""
    x, st = saved
    C = x.shape[-1]
    return O.bn_bwd_spec(x.view(-1, C), P.get(pre + 'weight'), P.get(pre + 'bias'), st, relu)


def _conv_bwd(P, key, x, gy, stride, pad, G, need_gx=True, add_gx=None, pre=None, mask_gx=None, bn_bwd=None):
    """add_gx: another gradient of x, summed into gx in the data-gradient convolution's epilogue; pre: the pre-activation the forward convolution
    applied to x (bn_relu_into_conv) -- gx is then the gradient of the ACTIVATED operand, as before"""
    has_bias = (key + 'bias') in P
    gx, G[key + 'weight'], gb = TC.conv_bwd(x, P[key + 'weight'], gy, stride, pad, need_gx=need_gx, has_bias=has_bias, oihw=True, add_gx=add_gx, gw_oihw=True,
                                            pre=pre, mask_gx=mask_gx, bn_bwd=bn_bwd)
    if has_bias:
        G[key + 'bias'] = gb
    return gx


# ------------------------------------------------------------------------------------------------------------------------ Bottleneck
def bottleneck_forward(P, x, stride=1):
    ctx = {'x': x, 'stride': stride}
    s1, s2, s3, sd = [], [], [], []                                # chunk partials of bn1 / bn2 / bn3 / downsample.1 from the convolutions' epilogues
    h = TC.conv_fwd(x, P['conv1.weight'], oihw=True, stats=s1)
    a1, p1, ctx['bn1'] = bn_relu_into_conv(P, 'bn1.', h, partials=s1)             # (a1 = conv1's raw output + the affine conv2 applies, or the normalised map)
    h = TC.conv_fwd(a1, P['conv2.weight'], None, stride, 1, oihw=True, pre=p1, stats=s2)
    a2, p2, ctx['bn2'] = bn_relu_into_conv(P, 'bn2.', h, partials=s2)
    h = TC.conv_fwd(a2, P['conv3.weight'], oihw=True, pre=p2, stats=s3)
    if 'downsample.0.weight' in P:
        idn = TC.conv_fwd(x, P['downsample.0.weight'], None, stride, 0, oihw=True, stats=sd)
        idn, ctx['bnd'] = bn_fwd(P, 'downsample.1.', idn, partials=sd)
    else:
        idn = x
    y, ctx['bn3'] = bn_fwd(P, 'bn3.', h, relu=True, residual=idn, partials=s3)      # relu(bn3(.) + identity) (resnet.py:136-140), one launch
    ctx.update(a1=a1, a2=a2, p1=p1, p2=p2, y=y)
    return y, ctx


def bottleneck_backward(P, ctx, gy, need_gx=True, gy_masked=False, mask_gx=None, bn3_spec=None, prev_bn3=None):
    """gy_masked: gy already went through this block's final ReLU backward (the NEXT block's conv1 data gradient applied it: mask_gx there);
    mask_gx: the stored output of the block that produced x -- its ReLU backward is applied to the returned gx in conv1's data-gradient epilogue
    (round 5: one pass over (gradient, output) per block boundary less; only where nothing else is added to that gradient first).
    prev_bn3: bn_bwd_spec of the PREVIOUS block's bn3 (with mask_gx: the masked gx is exactly the gradient of that BatchNorm's output) -- its
    backward sums are then formed in the same epilogue; the previous block receives the spec back as bn3_spec"""
    G = {}
    x, stride = ctx['x'], ctx['stride']
    g = gy.contiguous() if gy_masked else O.relu_bwd(gy.contiguous(), ctx['y'])               # gradient of (bn3 out + identity)
    g3 = bn_bwd(P, 'bn3.', ctx['bn3'], g, G, spec=bn3_spec)
    s2 = bn_bwd_spec(P, 'bn2.', ctx['bn2'], True)            # bn2's / bn1's backward sums out of the data-gradient convolution that writes their gradient
    g2 = _conv_bwd(P, 'conv3.', ctx['a2'], g3, 1, 0, G, pre=ctx.get('p2'), bn_bwd=s2)
    g2 = bn_bwd(P, 'bn2.', ctx['bn2'], g2, G, relu=True, spec=s2)
    s1 = bn_bwd_spec(P, 'bn1.', ctx['bn1'], True)
    g1 = _conv_bwd(P, 'conv2.', ctx['a1'], g2, stride, 1, G, pre=ctx.get('p1'), bn_bwd=s1)
    g1 = bn_bwd(P, 'bn1.', ctx['bn1'], g1, G, relu=True, spec=s1)
    # the identity / projection path's gradient joins conv1's data gradient in that convolution's epilogue (no separate dir_axpy_f32)
    if 'downsample.0.weight' in P:
        gd = bn_bwd(P, 'downsample.1.', ctx['bnd'], g, G)
        other = _conv_bwd(P, 'downsample.0.', x, gd, stride, 0, G, need_gx=need_gx)
    else:
        other = g
    gx = _conv_bwd(P, 'conv1.', x, g1, 1, 0, G, need_gx=need_gx, add_gx=other if need_gx else None, mask_gx=mask_gx if need_gx else None,
                   bn_bwd=prev_bn3 if (need_gx and mask_gx is not None) else None)
    return gx, G


# -------------------------------------------------------------------------------------------------------------------------- Residual
def residual_forward(P, x):
    ctx = {'x': x}
    a0, p0, ctx['bn1'] = bn_relu_into_conv(P, 'bn1.', x)           # pre-activation block: every BatchNorm + ReLU feeds exactly one convolution
    s2, s3 = [], []
    h = TC.conv_fwd(a0, P['conv1.conv.weight'], P['conv1.conv.bias'], oihw=True, pre=p0, stats=s2)
    a1, p1, ctx['bn2'] = bn_relu_into_conv(P, 'bn2.', h, partials=s2)
    h = TC.conv_fwd(a1, P['conv2.conv.weight'], P['conv2.conv.bias'], 1, 1, oihw=True, pre=p1, stats=s3)
    a2, p2, ctx['bn3'] = bn_relu_into_conv(P, 'bn3.', h, partials=s3)
    need_skip = P['skip_layer.conv.weight'].shape[0] != P['skip_layer.conv.weight'].shape[1]          # hourglass.py:49-52
    if need_skip:
        y = TC.conv_fwd(a2, P['conv3.conv.weight'], P['conv3.conv.bias'], oihw=True, pre=p2)
        y = TC.conv_fwd(x, P['skip_layer.conv.weight'], P['skip_layer.conv.bias'], oihw=True, residual=y)       # + skip_layer(x), same launch
    else:
        y = TC.conv_fwd(a2, P['conv3.conv.weight'], P['conv3.conv.bias'], oihw=True, residual=x, pre=p2)
    ctx.update(a0=a0, a1=a1, a2=a2, p0=p0, p1=p1, p2=p2, need_skip=need_skip)
    return y, ctx


def residual_backward(P, ctx, gy, need_gx=True):
    G = {}
    x = ctx['x']
    gy = gy.contiguous()
    s3 = bn_bwd_spec(P, 'bn3.', ctx['bn3'], True)            # each BatchNorm's backward sums out of the data-gradient convolution that writes its output's gradient
    g = _conv_bwd(P, 'conv3.conv.', ctx['a2'], gy, 1, 0, G, pre=ctx.get('p2'), bn_bwd=s3)
    g = bn_bwd(P, 'bn3.', ctx['bn3'], g, G, relu=True, spec=s3)
    s2 = bn_bwd_spec(P, 'bn2.', ctx['bn2'], True)
    g = _conv_bwd(P, 'conv2.conv.', ctx['a1'], g, 1, 1, G, pre=ctx.get('p1'), bn_bwd=s2)
    g = bn_bwd(P, 'bn2.', ctx['bn2'], g, G, relu=True, spec=s2)
    s1 = bn_bwd_spec(P, 'bn1.', ctx['bn1'], True)
    g = _conv_bwd(P, 'conv1.conv.', ctx['a0'], g, 1, 0, G, pre=ctx.get('p0'), bn_bwd=s1)
    gx = bn_bwd(P, 'bn1.', ctx['bn1'], g, G, relu=True, spec=s1)
    if ctx['need_skip']:
        gs = _conv_bwd(P, 'skip_layer.conv.', x, gy, 1, 0, G, need_gx=need_gx, add_gx=gx if need_gx else None)      # + gx, same launch
        gx = gs if need_gx else gx
    else:
        # an unused skip_layer keeps its parameters without gradient, like torch (hourglass.py:56-59)
        O.axpy(gx, gy)
    return gx, G
