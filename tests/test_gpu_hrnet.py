"""SURVEY.md 8f rank 4 / BASELINE config 5: the HRNet-W48 backbone (dir_amd/models/backbone/hrnet.py, dir_amd.engine.HRNetOp) and the whole
config-5 network -- HRNet-W48 + init regression + 4 refinement stages ("5 refinement iters") -- against the numpy oracle (oracle/hrnet.py,
oracle/dir_forward.py).  The reference has neither an HRNet nor more than two stages: parity is pinned to the oracle ONLY, whose convolution /
BatchNorm / stage functions are the ones the reference goldens hold; the trained-like flavour of the synthetic parameters keeps activations
O(10)."""
import numpy as np
import pytest
import torch

from conftest import maxabs, relerr
from dir_amd import synth

pytestmark = pytest.mark.gpu
SEED = 1234


def test_dir_add_upsampled():
    from dir_amd import _capi
    g = torch.Generator(device='cuda').manual_seed(2)
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        for f in (1, 2, 8):
            acc = torch.randn(3, 16, 24, 64, device='cuda', generator=g).to(dt)
            src = torch.randn(3, 16 // f, 24 // f, 64, device='cuda', generator=g).to(dt)
            want = acc.float() + src.float().repeat_interleave(f, 1).repeat_interleave(f, 2)
            for relu in (0, 1):
                a = acc.clone()
                _capi.check(_capi.lib().dir_add_upsampled(_capi.ptr(a), _capi.ptr(src), 3, 16, 24, 64, f, relu, code, _capi.stream_ptr()), 'add')
                w = (torch.relu(want) if relu else want).to(dt)
                assert torch.equal(a, w), (dt, f, relu)


def test_dir_fuse_sum_is_the_sum_rounded_once():
    """out = act(base + sum_t up(src_t, f_t)) in fp32, one rounding (dir_fuse_sum); in place and into a new map; 0 .. 4 sources"""
    import ctypes as C
    from dir_amd import _capi
    g = torch.Generator(device='cuda').manual_seed(3)
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1), (torch.float16, _capi.DT_F16)):
        base = torch.randn(3, 16, 24, 64, device='cuda', generator=g).to(dt)
        for facs in ((), (1,), (2, 4, 8), (1, 2, 4, 8)):
            srcs = [torch.randn(3, 16 // f, 24 // f, 64, device='cuda', generator=g).to(dt) for f in facs]
            want = base.float()
            for f, t in zip(facs, srcs):
                want = want + t.float().repeat_interleave(f, 1).repeat_interleave(f, 2)
            n = len(facs)
            ps = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in srcs])
            fs = (C.c_int * max(n, 1))(*facs)
            for relu in (0, 1):
                w = (torch.relu(want) if relu else want).to(dt)
                out = torch.empty_like(base)
                _capi.check(_capi.lib().dir_fuse_sum(_capi.ptr(out), _capi.ptr(base), ps, fs, n, 3, 16, 24, 64, relu, code, _capi.stream_ptr()), 'fuse_sum')
                assert torch.equal(out, w), (dt, facs, relu)
                a = base.clone()
                _capi.check(_capi.lib().dir_fuse_sum(_capi.ptr(a), _capi.ptr(a), ps, fs, n, 3, 16, 24, 64, relu, code, _capi.stream_ptr()), 'fuse_sum')
                assert torch.equal(a, w), (dt, facs, relu, 'in place')


@pytest.mark.parametrize('mode', ['f32', 'f16x3', 'f16', 'bf16', 'f16s'])
def test_hrnet_w48_backbone_vs_oracle(mode):
    from dir_amd.engine import HRNetOp
    from dir_amd.models.backbone.hrnet import HRNetW48
    from oracle import nnops as N
    from oracle.hrnet import hrnet_w48
    m = HRNetW48()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd_np = synth.synth_state_dict(shapes, SEED, cond=True)
    img = synth.synth_input('hr.img', (2, 3, 256, 256), SEED)
    ref = hrnet_w48(img.astype(np.float64), N.Params(sd_np, '', np.float64))
    sd = {'b.' + k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd_np.items()}
    dt = {'bf16': torch.bfloat16, 'f16s': torch.float16}.get(mode, torch.float32)        # f16s: f16 STORAGE (round 5), the bf16 data path on IEEE f16
    from dir_amd import engine as E
    E._TLS.arith = mode if mode in ('f16x3', 'f16') else None
    try:
        op = HRNetOp(sd, 'b', dt, torch.device('cuda'))
    finally:
        E._TLS.arith = None
    x = torch.from_numpy(img).cuda()
    if mode in ('f16x3', 'f16'):
        E._TLS.calibrating = True
        try:
            op(x)
        finally:
            E._TLS.calibrating = False
    feats = op(x)
    torch.cuda.synchronize()
    errs = []
    for f, r, c in zip(feats, ref, (256, 512, 1024, 2048)):
        assert f.shape[3] == c and f.shape[0] == 2
        errs.append(relerr(f.float().permute(0, 3, 1, 2).cpu().numpy(), r))
    print('HRNet-W48 %s: c1..c4 relative to each map\'s maximum vs the float64 oracle: %s' % (mode, np.array2string(np.array(errs), precision=2)))
    # ~300 convolutions deep: bf16 accumulates 2^-9 per layer, f16 (config 5's named arithmetic: one f16 MFMA per product) 2^-12
    assert max(errs) < {'bf16': 6e-2, 'f16': 8e-3, 'f16s': 8e-3}.get(mode, 2e-5), errs


@pytest.mark.parametrize('mode', ['f32', 'f16x3', 'f16', 'bf16', 'f16s'])
def test_config5_network_vs_oracle(mode):
    """HRNet-W48 + init regression + 4 refinement stages through the drop-in module (DIR(backbone='hrnet_w48', extra_stages=2))"""
    from dir_amd.models.dir import DIR
    from oracle.dir_forward import dir_forward
    net = DIR(21, './misc/mano', 0, backbone='hrnet_w48', extra_stages=2, compute_dtype={'bf16': torch.bfloat16, 'f16s': torch.float16}.get(mode, torch.float32),
              arith=mode if mode in ('f16x3', 'f16') else None)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd_np = synth.synth_state_dict(shapes, SEED, cond=True)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    img = synth.synth_input('dir.img', (2, 3, 256, 256), SEED)
    ref = dir_forward(sd_np, img, dtype=np.float64)           # float64: ~300 convolutions deep, an fp32 oracle carries as much noise as the kernels
    outs, loss = net({'img': torch.from_numpy(img)}, None, None)
    assert loss == {} and len(outs) == 6
    worst, mpjpe = 0.0, []
    for i in range(5):
        for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right'):
            worst = max(worst, maxabs(outs[i][k].cpu().numpy(), ref[i][k]))
        d = outs[i]['pd_joint_xyz_left'].cpu().numpy() - ref[i]['pd_joint_xyz_left']
        mpjpe.append(float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3)
    print('config 5 (%s): worst |xyz - oracle| %.3e m; mean per-joint error per stage (mm) %s' % (mode, worst, np.round(mpjpe, 5)))
    if mode == 'bf16':
        assert max(mpjpe[1:]) < 0.02 and mpjpe[0] < 0.2, mpjpe
    elif mode in ('f16', 'f16s'):      # BASELINE configs[4]'s named arithmetic (f16s: f16 storage at bf16 speed): every stage inside the 0.01 mm MPJPE budget
        assert max(mpjpe) < 0.01, mpjpe
        assert relerr(outs[5]['seg'].cpu().numpy(), ref[5]['seg']) < 2e-2
    else:
        assert worst < 5e-7, worst          # vs float64; the ResNet network sits at 9e-8 m, this backbone is six times deeper
        assert relerr(outs[5]['seg'].cpu().numpy(), ref[5]['seg']) < 1e-3


@pytest.mark.parametrize('mode', ['f16', 'bf16'])
def test_config5_batch_32_rows_equal_the_oracle_pinned_small_batch(mode):
    """BASELINE configs[4] at its per-GPU size (batch 256 on 8 GPUs = 32 per GPU), in the arithmetic it names (fp16 MFMA path) and in bf16:
    the two oracle-pinned images sit at rows 3 and 31 of a batch of 30 others.  Samples are independent (eval-mode BN), so those rows must
    reproduce the B = 2 run (which test_config5_network_vs_oracle holds to the float64 oracle) whatever tiles the larger batch selects:
    bit for bit in the f16 mode (fp32 tensors, every kernel variant accumulates in the same order), within the bf16 rounding envelope in bf16."""
    from dir_amd.models.dir import DIR
    from oracle.dir_forward import dir_forward
    net = DIR(21, './misc/mano', 0, backbone='hrnet_w48', extra_stages=2, compute_dtype={'bf16': torch.bfloat16, 'f16s': torch.float16}.get(mode, torch.float32),
              arith='f16' if mode == 'f16' else None)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd_np = synth.synth_state_dict(shapes, SEED, cond=True)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}, strict=True)
    net = net.cuda().eval()
    net.autotune = False
    img = synth.synth_input('dir.img', (2, 3, 256, 256), SEED)
    small, _ = net({'img': torch.from_numpy(img)}, None, None)           # (calibrates the f16 operand scales on this batch)
    keys = ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left', 'pd_joint_xyz_right', 'pd_joint_uv_left')
    want = [{k: small[s][k].clone() for k in keys} for s in range(5)]
    big = torch.randn(32, 3, 256, 256, generator=torch.Generator().manual_seed(32))
    big[3], big[31] = torch.from_numpy(img[0]), torch.from_numpy(img[1])
    outs, _ = net({'img': big}, None, None)
    torch.cuda.synchronize()
    worst = 0.0
    for s in range(5):
        for k in keys:
            got = outs[s][k][[3, 31]]
            if mode == 'f16':
                assert torch.equal(got, want[s][k]), (s, k)
            else:
                worst = max(worst, float((got - want[s][k]).abs().max()))
    ref = dir_forward(sd_np, img, dtype=np.float64)
    d = outs[4]['pd_joint_xyz_left'][[3, 31]].cpu().numpy() - ref[4]['pd_joint_xyz_left']
    mp = float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3
    print('config 5 at 32 per GPU (%s): last-stage mean per-joint error vs the float64 oracle %.5f mm; worst row difference vs the B=2 run %.3e' % (mode, mp, worst))
    assert mp < (0.02 if mode == 'bf16' else 0.01)
    assert worst < 2e-3
