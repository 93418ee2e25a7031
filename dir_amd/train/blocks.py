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


def _ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def _oihw(g):
    return g.permute(0, 3, 1, 2).contiguous()


def bn_fwd(P, pre, x, momentum=0.1, eps=1e-5, relu=False, residual=None):
    """BatchNorm2d in training mode; relu=True: the nn.ReLU that follows it in the same launch (bn_bwd(..., relu=True) undoes both);
    residual: added before that ReLU (a bottleneck's tail; its backward masks with the saved output: relu_bwd, then bn_bwd(relu=False))"""
    C = x.shape[-1]
    y, st = O.bn_train_fwd(x.view(-1, C), P[pre + 'weight'], P[pre + 'bias'], P.get(pre + 'running_mean'), P.get(pre + 'running_var'), eps, momentum,
                           relu=relu, residual=None if residual is None else residual.contiguous().view(-1, C))
    return y.view(x.shape), (x, st)


def bn_bwd(P, pre, saved, gy, G, relu=False):
    x, st = saved
    C = x.shape[-1]
    gx, G[pre + 'weight'], G[pre + 'bias'] = O.bn_train_bwd(gy.contiguous().view(-1, C), x.view(-1, C), P[pre + 'weight'], st, b=P[pre + 'bias'], relu=relu)
    return gx.view(x.shape)


def _conv_bwd(P, key, x, gy, stride, pad, G, need_gx=True, add_gx=None):
    """add_gx: another gradient of x, summed into gx in the data-gradient convolution's epilogue"""
    has_bias = (key + 'bias') in P
    gx, G[key + 'weight'], gb = TC.conv_bwd(x, P[key + 'weight'], gy, stride, pad, need_gx=need_gx, has_bias=has_bias, oihw=True, add_gx=add_gx, gw_oihw=True)
    if has_bias:
        G[key + 'bias'] = gb
    return gx


# ------------------------------------------------------------------------------------------------------------------------ Bottleneck
def bottleneck_forward(P, x, stride=1):
    ctx = {'x': x, 'stride': stride}
    h = TC.conv_fwd(x, P['conv1.weight'], oihw=True)
    a1, ctx['bn1'] = bn_fwd(P, 'bn1.', h, relu=True)
    h = TC.conv_fwd(a1, P['conv2.weight'], None, stride, 1, oihw=True)
    a2, ctx['bn2'] = bn_fwd(P, 'bn2.', h, relu=True)
    h = TC.conv_fwd(a2, P['conv3.weight'], oihw=True)
    if 'downsample.0.weight' in P:
        idn = TC.conv_fwd(x, P['downsample.0.weight'], None, stride, 0, oihw=True)
        idn, ctx['bnd'] = bn_fwd(P, 'downsample.1.', idn)
    else:
        idn = x
    y, ctx['bn3'] = bn_fwd(P, 'bn3.', h, relu=True, residual=idn)                   # relu(bn3(.) + identity) (resnet.py:136-140), one launch
    ctx.update(a1=a1, a2=a2, y=y)
    return y, ctx


def bottleneck_backward(P, ctx, gy, need_gx=True):
    G = {}
    x, stride = ctx['x'], ctx['stride']
    g = O.relu_bwd(gy.contiguous(), ctx['y'])               # gradient of (bn3 out + identity)
    g3 = bn_bwd(P, 'bn3.', ctx['bn3'], g, G)
    g2 = _conv_bwd(P, 'conv3.', ctx['a2'], g3, 1, 0, G)
    g2 = bn_bwd(P, 'bn2.', ctx['bn2'], g2, G, relu=True)
    g1 = _conv_bwd(P, 'conv2.', ctx['a1'], g2, stride, 1, G)
    g1 = bn_bwd(P, 'bn1.', ctx['bn1'], g1, G, relu=True)
    # the identity / projection path's gradient joins conv1's data gradient in that convolution's epilogue (no separate dir_axpy_f32)
    if 'downsample.0.weight' in P:
        gd = bn_bwd(P, 'downsample.1.', ctx['bnd'], g, G)
        other = _conv_bwd(P, 'downsample.0.', x, gd, stride, 0, G, need_gx=need_gx)
    else:
        other = g
    gx = _conv_bwd(P, 'conv1.', x, g1, 1, 0, G, need_gx=need_gx, add_gx=other if need_gx else None)
    return gx, G


# -------------------------------------------------------------------------------------------------------------------------- Residual
def residual_forward(P, x):
    ctx = {'x': x}
    a0, ctx['bn1'] = bn_fwd(P, 'bn1.', x, relu=True)
    h = TC.conv_fwd(a0, P['conv1.conv.weight'], P['conv1.conv.bias'], oihw=True)
    a1, ctx['bn2'] = bn_fwd(P, 'bn2.', h, relu=True)
    h = TC.conv_fwd(a1, P['conv2.conv.weight'], P['conv2.conv.bias'], 1, 1, oihw=True)
    a2, ctx['bn3'] = bn_fwd(P, 'bn3.', h, relu=True)
    need_skip = P['skip_layer.conv.weight'].shape[0] != P['skip_layer.conv.weight'].shape[1]          # hourglass.py:49-52
    if need_skip:
        y = TC.conv_fwd(a2, P['conv3.conv.weight'], P['conv3.conv.bias'], oihw=True)
        y = TC.conv_fwd(x, P['skip_layer.conv.weight'], P['skip_layer.conv.bias'], oihw=True, residual=y)       # + skip_layer(x), same launch
    else:
        y = TC.conv_fwd(a2, P['conv3.conv.weight'], P['conv3.conv.bias'], oihw=True, residual=x)
    ctx.update(a0=a0, a1=a1, a2=a2, need_skip=need_skip)
    return y, ctx


def residual_backward(P, ctx, gy, need_gx=True):
    G = {}
    x = ctx['x']
    gy = gy.contiguous()
    g = _conv_bwd(P, 'conv3.conv.', ctx['a2'], gy, 1, 0, G)
    g = bn_bwd(P, 'bn3.', ctx['bn3'], g, G, relu=True)
    g = _conv_bwd(P, 'conv2.conv.', ctx['a1'], g, 1, 1, G)
    g = bn_bwd(P, 'bn2.', ctx['bn2'], g, G, relu=True)
    g = _conv_bwd(P, 'conv1.conv.', ctx['a0'], g, 1, 0, G)
    gx = bn_bwd(P, 'bn1.', ctx['bn1'], g, G, relu=True)
    if ctx['need_skip']:
        gs = _conv_bwd(P, 'skip_layer.conv.', x, gy, 1, 0, G, need_gx=need_gx, add_gx=gx if need_gx else None)      # + gx, same launch
        gx = gs if need_gx else gx
    else:
        # an unused skip_layer keeps its parameters without gradient, like torch (hourglass.py:56-59)
        O.axpy(gx, gy)
    return gx, G
