"""Drop-in for models/backbone/hourglass.py:10-70 (Conv, Residual): identical parameter tree; Residual.forward (NCHW)
runs as four MFMA convs with the pre-activation BN+ReLU applied in the conv's register-staging prologue."""
import torch
import torch.nn as nn


class Conv(nn.Module):
    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=False, relu=True):
        super().__init__()
        self.inp_dim = inp_dim
        self.conv = nn.Conv2d(inp_dim, out_dim, kernel_size, stride, padding=(kernel_size - 1) // 2, bias=True)
        self.relu = nn.ReLU() if relu else None
        self.bn = nn.BatchNorm2d(out_dim) if bn else None


class Residual(nn.Module):
    def __init__(self, inp_dim, out_dim):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inp_dim)
        self.relu1 = nn.ReLU()
        self.conv1 = Conv(inp_dim, int(out_dim / 2), 1, relu=False)
        self.bn2 = nn.BatchNorm2d(int(out_dim / 2))
        self.relu2 = nn.ReLU()
        self.conv2 = Conv(int(out_dim / 2), int(out_dim / 2), 3, relu=False)
        self.bn3 = nn.BatchNorm2d(int(out_dim / 2))
        self.relu3 = nn.ReLU()
        self.conv3 = Conv(int(out_dim / 2), out_dim, 1, relu=False)
        self.skip_layer = Conv(inp_dim, out_dim, 1, relu=False)
        self.need_skip = inp_dim != out_dim

    def _train_forward(self, x):
        """.train(): models/backbone/hourglass.py:55-70 with batch-statistics BatchNorm and autograd (dir_amd/train/blocks.py behind one node)"""
        from ... import _capi
        from ...train import autograd as AG
        from ...train import blocks as TB
        from ...train import conv as TC
        params = dict(self.named_parameters())
        buffers = {k: b for k, b in self.named_buffers() if 'num_batches_tracked' not in k}
        for m in (self.bn1, self.bn2, self.bn3):
            if m.num_batches_tracked is not None:
                m.num_batches_tracked += 1

        def fwd(P, xx):
            TC.begin_step(None)
            y, ctx = TB.residual_forward(P, _capi.f32c(xx.permute(0, 2, 3, 1)))
            return (y.permute(0, 3, 1, 2),), ctx

        def bwd(P, ctx, gy):
            gx, G = TB.residual_backward(P, ctx, gy.permute(0, 2, 3, 1).contiguous())
            TC.end_step()
            return (gx.permute(0, 3, 1, 2),), G
        with torch.cuda.device(x.device):
            return AG.run(fwd, bwd, [x], params, buffers)[0]

    def forward(self, x, compute_dtype=torch.float32):
        from ...engine import ResidualOp
        from ... import _capi
        _capi.require_cuda(x)
        if self.training and torch.is_grad_enabled():
            return self._train_forward(x)
        sd = {'r.' + k: v.detach() for k, v in self.state_dict().items()}
        op = ResidualOp(sd, 'r', compute_dtype)
        y = op(x.detach().permute(0, 2, 3, 1).contiguous().to(compute_dtype))
        return y.permute(0, 3, 1, 2).float()
