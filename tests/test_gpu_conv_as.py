"""GPU parity of the activation-stationary convolution kernel (conv_as.hip, dir_conv2d_as_forward, DIR_CONV_VARIANT 25 .. 28) through the engine's
ConvOp: every (A, PB) shape against the numpy oracle conv2d on the 16-bit-rounded operands, and BIT FOR BIT against the tiled kernel the library
picks itself (same K order and k-slot assignment: what lets the engine's autotune choose the kernel per layer).  Shapes: the small maps of the path --
ResNet layer3 / layer4 (models/backbone/resnet.py:120-140), the decoder's 16x16 / 32x32 Residual blocks (models/backbone/hourglass.py:55-70) --
cut down in batch; residual + ReLU, channel-slice input / output (the concat buffers), both storage kinds."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import relerr
from dir_amd import _capi, synth
from dir_amd import engine as E
from oracle import nnops as N

pytestmark = pytest.mark.gpu
SEED = 1234

CASES = [
    # B, H, W, Cin, Cout, k, residual, relu
    (3, 16, 16, 256, 256, 3, False, True),      # layer3 conv2
    (2, 8, 8, 512, 512, 3, False, True),        # layer4 conv2
    (2, 16, 16, 1024, 256, 1, False, True),     # layer3 conv1
    (2, 16, 16, 256, 1024, 1, True, True),      # layer3 conv3 + identity
    (2, 32, 32, 128, 128, 3, False, False),     # decoder Residual conv2 @32x32
    (1, 8, 8, 2048, 512, 1, False, True),       # layer4 conv1 (patch 263 KB: served by (4, 2) on a ring of two 512-channel chunks, the others fall back)
    (2, 8, 8, 2048, 512, 3, False, True),       # the attention convolution's geometry (models/dir.py:227-241; Cout cut down): 8 chunks of 256 channels
    (2, 16, 16, 128, 256, 3, True, False),
]


def _round(a, tdt):
    return torch.from_numpy(a).to(tdt).float().numpy()


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_as_variants_match_oracle_and_the_tiled_kernel_bit_for_bit(case, dt):
    B, H, W, Ci, Co, k, with_res, relu = case
    tag = 'convas.%s' % '_'.join(map(str, case[:6]))
    x = _round(synth.synth_input(tag + '.x', (B, Ci, H, W), SEED), dt)
    w = _round(synth.synth_input(tag + '.w', (Co, Ci, k, k), SEED) * np.float32(np.sqrt(2.0 / (k * k * Ci))), dt)
    scale = synth.synth_input(tag + '.s', (Co,), SEED, kind='uniform', lo=0.5, hi=1.5)
    shift = synth.synth_input(tag + '.b', (Co,), SEED) * np.float32(0.3)
    res = _round(synth.synth_input(tag + '.r', (B, Co, H, W), SEED), dt) if with_res else None
    ref = N.conv2d(x.astype(np.float64), w.astype(np.float64), None, 1, k // 2) * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)
    if with_res:
        ref = ref + res
    if relu:
        ref = np.maximum(ref, 0)
    op = E.ConvOp(torch.from_numpy(w).cuda(), dt, stride=1, pad=k // 2, scale=torch.from_numpy(scale).cuda(), shift=torch.from_numpy(shift).cuda(), relu=relu)
    # the input is a channel slice [32, 32 + Ci) of a wider buffer, the output a slice [64, 64 + Co) of another (concat buffers of the decoder)
    xbuf = torch.zeros(B, H, W, Ci + 96, device='cuda', dtype=dt)
    xbuf[..., 32:32 + Ci] = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda().to(dt)
    xbuf[..., :32] = 7.0
    xbuf[..., 32 + Ci:] = -5.0
    rbuf = None if res is None else torch.from_numpy(np.ascontiguousarray(res.transpose(0, 2, 3, 1))).cuda().to(dt).contiguous()

    def run(variant):
        E._TLS.variant = variant
        try:
            out = torch.full((B, H, W, Co + 128), 3.0, device='cuda', dtype=dt)
            op(xbuf, out=out, out_coff=64, in_coff=32, residual=rbuf)
            torch.cuda.synchronize()
            return out
        finally:
            E._TLS.variant = None
    base = run(0)
    got = base[..., 64:64 + Co].float().cpu().numpy().transpose(0, 3, 1, 2)
    assert relerr(got, ref) < (1e-2 if dt == torch.bfloat16 else 2e-3)
    d = _capi.ConvDesc(B, H, W, Ci, Ci + 96, 32, Co, Co + 128, 64, Co if with_res else 0, 0, k, k, 1, k // 2, E._dt(dt), E._dt(dt), 0, 0, 0, 1.0)
    ran = 0
    for v, (A, PB) in E.AS_VARIANTS.items():
        supported = bool(_capi.lib().dir_conv2d_as_supported(d, A, PB))
        _capi.lib().dir_launch_log_reset()
        out = run(v)
        buf = C.create_string_buffer(256)
        _capi.lib().dir_launch_log_get(buf, 256)
        names = buf.value.decode()
        assert ('conv_as_kernel' in names) == supported, (v, names, supported)
        ran += supported
        assert torch.equal(out, base), 'variant %d (A=%d, PB=%d) differs from the tiled kernel' % (v, A, PB)          # incl. the untouched slices of the buffer
    assert ran >= (1 if Ci == 2048 or Co % 256 else 2), 'the kernel should serve this layer'


def test_first_use_inside_a_graph_capture():
    """The kernel opts into > 64 KB of dynamic LDS (hipFuncSetAttribute) the first time an instantiation is launched.  A caller that loads a kernel
    table and captures its first forward straight away (engine.ForwardPipeline after load_tuning_table) makes that first launch INSIDE a stream
    capture: it must work there too.  Own process, so that no earlier test has launched the instantiation."""
    import os
    import subprocess
    import sys
    code = r'''
import torch
from dir_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(3)
w = torch.randn(256, 256, 3, 3, device='cuda', generator=g) * 0.02
op = E.ConvOp(w, torch.float16, stride=1, pad=1, scale=torch.ones(256, device='cuda'), shift=torch.zeros(256, device='cuda'), relu=True)
x = torch.randn(4, 16, 16, 256, device='cuda', generator=g).half()
ref = op(x).clone()                                   # the library's own choice (a tiled kernel), eager
torch.cuda.synchronize()
out = torch.empty_like(ref)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    E._TLS.variant = 25
    with torch.cuda.graph(gr, stream=s):
        op(x, out=out)                                # first launch of this conv_as_kernel instantiation: inside the capture
    E._TLS.variant = None
    gr.replay()
torch.cuda.synchronize()
assert torch.equal(out, ref), 'captured first use differs'
print('OK')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
