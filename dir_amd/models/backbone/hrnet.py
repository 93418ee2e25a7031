"""HRNet-W48 feature backbone for DIR (BASELINE config 5).  NO REFERENCE COUNTERPART: /root/reference has no HRNet
(models/backbone/ holds resnet.py and hourglass.py only), so this is the build's own statement of the published architecture
(Sun et al., "Deep High-Resolution Representation Learning", CVPR 2019; W48 = branch widths 48 / 96 / 192 / 384):

    stem        conv3x3/2 (3 -> 64) bn relu, conv3x3/2 (64 -> 64) bn relu                                   256x256 -> 64x64
    layer1      4 Bottlenecks (planes 64 -> 256 channels; the first with a 1x1 projection shortcut)
    transition1 branch 0: conv3x3 (256 -> 48) bn relu; branch 1: conv3x3/2 (256 -> 96) bn relu
    stage2      1 module of 2 branches, stage3: 4 modules of 3 branches (new branch: conv3x3/2 of the last one, 96 -> 192), stage4: 3 modules of
                4 branches (192 -> 384).  A module = 4 BasicBlocks per branch (conv3x3 bn relu conv3x3 bn, + x, relu) + fuse layers:
                y_i = relu(sum_j f_ij(x_j)), f_ii = identity, j > i: conv1x1 (C_j -> C_i) bn, nearest upsample x 2^(j-i);
                j < i: (i - j) conv3x3/2, the last one to C_i with bn only, the others keeping C_j with bn relu
    incre       the four branch outputs (48@64^2, 96@32^2, 192@16^2, 384@8^2) -> conv1x1 bn relu to DIR's pyramid widths (256, 512, 1024, 2048)

forward(x) -> [c1, c2, c3, c4] like models/backbone/resnet.py:243-255, so DIR's InitRegressor / decoder are unchanged.  The module is a
parameter container (same convention as resnet.py here); the arithmetic is dir_amd.engine.HRNetOp on libdir_hip.so in .eval() mode and
dir_amd/train/hrnet.py (training kernels behind one autograd node) in .train() mode."""
import torch
import torch.nn as nn

from .resnet import Bottleneck

WIDTHS = (48, 96, 192, 384)
MODULES = {2: 1, 3: 4, 4: 3}          # stage -> number of modules (branches = stage)
OUT = (256, 512, 1024, 2048)


class BasicBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)


def _cb(cin, cout, k, s):
    return nn.Sequential(nn.Conv2d(cin, cout, k, s, k // 2, bias=False), nn.BatchNorm2d(cout))


class HRModule(nn.Module):
    def __init__(self, nb):
        super().__init__()
        self.branches = nn.ModuleList(nn.Sequential(*[BasicBlock(WIDTHS[b]) for _ in range(4)]) for b in range(nb))
        fuse = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j == i:
                    row.append(nn.Identity())
                elif j > i:
                    row.append(_cb(WIDTHS[j], WIDTHS[i], 1, 1))
                else:                                                        # i - j strided 3x3 convs; the last one changes the width
                    row.append(nn.Sequential(*[_cb(WIDTHS[j], WIDTHS[i] if t == i - j - 1 else WIDTHS[j], 3, 2) for t in range(i - j)]))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)


class HRNetW48(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = OUT[3]                                                  # read by DIR (models/dir.py:501)
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False), nn.BatchNorm2d(64)
        self.conv2, self.bn2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False), nn.BatchNorm2d(64)
        ds = nn.Sequential(nn.Conv2d(64, 256, 1, 1, bias=False), nn.BatchNorm2d(256))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1, ds), *[Bottleneck(256, 64) for _ in range(3)])
        self.transition1 = nn.ModuleList([_cb(256, WIDTHS[0], 3, 1), _cb(256, WIDTHS[1], 3, 2)])
        self.transition2 = _cb(WIDTHS[1], WIDTHS[2], 3, 2)
        self.transition3 = _cb(WIDTHS[2], WIDTHS[3], 3, 2)
        for st, n in MODULES.items():
            setattr(self, 'stage%d' % st, nn.ModuleList(HRModule(st) for _ in range(n)))
        self.incre = nn.ModuleList(_cb(WIDTHS[b], OUT[b], 1, 1) for b in range(4))

    def forward(self, x, compute_dtype=torch.float32):
        from ...engine import hrnet_standalone
        if self.training and torch.is_grad_enabled():
            return self._train_forward(x)
        return hrnet_standalone(self, x, compute_dtype)

    def _train_forward(self, x):
        """.train(): batch-statistics BatchNorm and autograd -- dir_amd/train/hrnet.py (hrnet_forward / hrnet_backward, the backbone part of the
        whole-network training step of a DIR built on this backbone) behind ONE autograd node; returns [c1, c2, c3, c4] NCHW."""
        from ... import _capi
        from ...train import autograd as AG
        from ...train import conv as TC
        from ...train import hrnet as TH
        _capi.require_cuda(x)
        params = {'backbone.' + k: p for k, p in self.named_parameters()}
        buffers = {'backbone.' + k: b for k, b in self.named_buffers() if 'num_batches_tracked' not in k}
        for m in self.modules():
            if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None:
                m.num_batches_tracked += 1

        def fwd(P, img):
            TC.begin_step(None)
            ctx = {}
            feats = TH.hrnet_forward(P, _capi.f32c(img), ctx)
            return tuple(f.permute(0, 3, 1, 2) for f in feats), ctx

        def bwd(P, ctx, *g_feats):
            G = {}
            TH.hrnet_backward(P, ctx, [None if g is None else g.permute(0, 2, 3, 1).contiguous() for g in g_feats], G)
            TC.end_step()
            return (None,), G              # the image is data
        with torch.cuda.device(x.device):
            return list(AG.run(fwd, bwd, [x], params, buffers))


def hrnet_w48():
    return HRNetW48()
