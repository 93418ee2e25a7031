"""Drop-in for models/backbone/resnet.py:85-324 (Bottleneck, ResNet, resnet50): identical parameter tree (incl. the
unused fc), forward(x) -> [c1, c2, c3, c4] (NCHW) computed by the MFMA implicit-GEMM kernels."""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), img_dim=3, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(img_dim, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)            # unused by the network (models/backbone/resnet.py:255)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x, compute_dtype=torch.float32):
        from ...engine import backbone_standalone
        if self.training and torch.is_grad_enabled():
            return self._train_forward(x)
        return backbone_standalone(self, x, compute_dtype)

    def _train_forward(self, x):
        """.train(): models/backbone/resnet.py:243-255 with batch-statistics BatchNorm and autograd -- the backbone part of the whole-network
        training step (dir_amd/train/net.py: backbone_forward / backbone_backward) behind ONE autograd node; returns [c1, c2, c3, c4] NCHW.
        The classifier `fc` is never called (resnet.py:243-255 returns the pyramid): no gradient, like under torch."""
        from ... import _capi
        from ...train import autograd as AG
        from ...train import conv as TC
        from ...train import net as TN
        _capi.require_cuda(x)
        params = {'backbone.' + k: p for k, p in self.named_parameters() if not k.startswith('fc.')}
        buffers = {'backbone.' + k: b for k, b in self.named_buffers() if 'num_batches_tracked' not in k}
        for m in self.modules():
            if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None:
                m.num_batches_tracked += 1

        def fwd(P, img):
            TC.begin_step(None)
            ctx = {}
            feats = TN.backbone_forward(P, _capi.f32c(img), ctx)
            return tuple(f.permute(0, 3, 1, 2) for f in feats), ctx

        def bwd(P, ctx, *g_feats):
            G = {}
            TN.backbone_backward(P, ctx, [None if g is None else g.permute(0, 2, 3, 1).contiguous() for g in g_feats], G)
            TC.end_step()
            return (None,), G              # the image is data (conv1's input gradient is never formed, as in the training step)
        with torch.cuda.device(x.device):
            return list(AG.run(fwd, bwd, [x], params, buffers))


def resnet50(**kwargs):
    return ResNet((3, 4, 6, 3), **kwargs)
