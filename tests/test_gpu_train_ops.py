"""GPU parity of the training building blocks (dir_gemm_f32, dir_layernorm_*, dir_gelu_*, dir_attention_*, dir_bn_train_*) against numpy in
float64, and of the composed STE forward + backward (dir_amd/train/ste.py) against G15 = torch autograd through the reference's STE
(transformer/mixSTE.py:159-205) and the analytic oracle (oracle/ste_grad.py).  fp32 kernels: 1e-5 of each tensor's maximum."""
import numpy as np
import pytest
import torch

from conftest import check_compact_grads, maxabs
from dir_amd import synth
from dir_amd.train import ops as O
from dir_amd.train import ste as STE

pytestmark = pytest.mark.gpu
SEED = 1234


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(a.detach().cpu().numpy().astype(np.float64).reshape(b.shape) - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('M,N,K,ta,tb', [(126, 384, 128, 0, 1), (384, 128, 126, 1, 0), (126, 128, 384, 0, 0), (1, 5376, 3, 0, 0), (70, 65, 17, 1, 1),
                                        (2688, 256, 128, 0, 1), (128, 128, 672, 1, 0), (130, 3, 1344, 1, 0), (384, 128, 1344, 1, 0), (32, 64, 2050, 0, 1)])
def test_gemm(M, N, K, ta, tb):
    rng = np.random.RandomState(M + N)
    A = rng.normal(0, 1, (K, M) if ta else (M, K)).astype(np.float32)
    B = rng.normal(0, 1, (N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.normal(0, 1, N).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    got = O.gemm(dev(A), dev(B), bool(ta), bool(tb), bias=dev(bias))
    assert rel(got, ref + bias) < 2e-6
    C0 = rng.normal(0, 1, (M, N)).astype(np.float32)
    out = dev(C0)
    O.gemm(dev(A), dev(B), bool(ta), bool(tb), out=out, accumulate=True)
    assert rel(out, ref + C0) < 2e-6
    assert torch.equal(O.gemm(dev(A), dev(B), bool(ta), bool(tb)), O.gemm(dev(A), dev(B), bool(ta), bool(tb)))          # deterministic


def test_gemm_batched():
    rng = np.random.RandomState(3)
    A, B = rng.normal(0, 1, (21, 48, 128)).astype(np.float32), rng.normal(0, 1, (21, 128, 128)).astype(np.float32)
    got = O.gemm(dev(A), dev(B))
    assert rel(got, A.astype(np.float64) @ B.astype(np.float64)) < 2e-6
    got = O.gemm(dev(A), dev(B[0]), tb=True)                        # shared second operand
    assert rel(got, A.astype(np.float64) @ B[0].astype(np.float64).T) < 2e-6


def test_layernorm_gelu_bn():
    rng = np.random.RandomState(7)
    for R, C, eps in ((126, 128, 1e-6), (5, 64, 1e-5), (300, 256, 1e-5)):
        x, w, b, gy = (rng.normal(0, 1, s).astype(np.float32) for s in ((R, C), (C,), (C,), (R, C)))
        x64 = x.astype(np.float64)
        mu, var = x64.mean(1, keepdims=True), x64.var(1, keepdims=True)
        xh = (x64 - mu) / np.sqrt(var + eps)
        y, st = O.layernorm_fwd(dev(x), dev(w), dev(b), eps)
        assert rel(y, xh * w + b) < 2e-6
        gh = gy.astype(np.float64) * w
        gx_ref = (gh - gh.mean(1, keepdims=True) - xh * (gh * xh).mean(1, keepdims=True)) / np.sqrt(var + eps)
        gx, gw, gb = O.layernorm_bwd(dev(gy), dev(x), dev(w), st)
        assert rel(gx, gx_ref) < 5e-6 and rel(gw, (gy * xh).sum(0)) < 5e-6 and rel(gb, gy.astype(np.float64).sum(0)) < 5e-6
        base = dev(rng.normal(0, 1, (R, C)))
        keep = base.clone()
        O.layernorm_bwd(dev(gy), dev(x), dev(w), st, gx=base, accumulate_x=True)
        assert rel(base, gx_ref + keep.cpu().numpy()) < 5e-6
        # BatchNorm, training mode, against torch's own module on the CPU (statistics, running update, gradients)
        bn = torch.nn.BatchNorm1d(C).train()
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(w)); bn.bias.copy_(torch.from_numpy(b))
        xt = torch.from_numpy(x).requires_grad_(True)
        yt = bn(xt)
        yt.backward(torch.from_numpy(gy))
        rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        y, st = O.bn_train_fwd(dev(x), dev(w), dev(b), rm, rv)
        assert rel(y, yt.detach().numpy()) < 5e-6 and rel(rm, bn.running_mean.numpy()) < 5e-6 and rel(rv, bn.running_var.numpy()) < 5e-6
        gx, gw, gb = O.bn_train_bwd(dev(gy), dev(x), dev(w), st)
        assert rel(gx, xt.grad.numpy()) < 2e-5 and rel(gw, bn.weight.grad.numpy()) < 5e-6 and rel(gb, bn.bias.grad.numpy()) < 5e-6
    from scipy.special import erf
    h = rng.normal(0, 2, (77, 256)).astype(np.float32)
    g = rng.normal(0, 1, (77, 256)).astype(np.float32)
    h64 = h.astype(np.float64)
    assert rel(O.gelu_fwd(dev(h)), 0.5 * h64 * (1 + erf(h64 / np.sqrt(2)))) < 2e-6
    assert rel(O.gelu_bwd(dev(g), dev(h)), g * (0.5 * (1 + erf(h64 / np.sqrt(2))) + h64 * np.exp(-0.5 * h64 ** 2) / np.sqrt(2 * np.pi))) < 2e-6


def test_attention():
    rng = np.random.RandomState(9)
    B, T, H, D = 3, 42, 4, 32
    qkv = rng.normal(0, 1, (B, T, 3, H, D)).astype(np.float32)
    go = rng.normal(0, 1, (B, T, H * D)).astype(np.float32)
    q, k, v = (qkv.astype(np.float64).transpose(2, 0, 3, 1, 4)[j] for j in range(3))
    S = q @ k.transpose(0, 1, 3, 2) * D ** -0.5
    P = np.exp(S - S.max(-1, keepdims=True)); P /= P.sum(-1, keepdims=True)
    o = (P @ v).transpose(0, 2, 1, 3).reshape(B, T, H * D)
    out, probs = O.attention_fwd(dev(qkv.reshape(B * T, -1)), B, T, H, D ** -0.5)
    assert rel(out, o) < 3e-6 and rel(probs, P) < 3e-6
    g = go.astype(np.float64).reshape(B, T, H, D).transpose(0, 2, 1, 3)
    gv, gP = P.transpose(0, 1, 3, 2) @ g, g @ v.transpose(0, 1, 3, 2)
    gS = P * (gP - (gP * P).sum(-1, keepdims=True)) * D ** -0.5
    ref = np.stack([gS @ k, gS.transpose(0, 1, 3, 2) @ q, gv]).transpose(1, 3, 0, 2, 4).reshape(B * T, 3 * H * D)
    assert rel(O.attention_bwd(dev(qkv.reshape(B * T, -1)), probs, dev(go.reshape(B * T, -1)), B, T, H, D ** -0.5), ref) < 5e-6


def ste_params():
    import json
    import os
    shapes = {'spatial_pos_embed': (1, 42, 128), 'spatial_norm.weight': (128,), 'spatial_norm.bias': (128,),
              'head.0.weight': (128,), 'head.0.bias': (128,), 'head.1.weight': (64, 128), 'head.1.bias': (64,)}
    for i in range(4):
        p = 'STEblocks.%d.' % i
        shapes.update({p + 'norm1.weight': (128,), p + 'norm1.bias': (128,), p + 'norm2.weight': (128,), p + 'norm2.bias': (128,),
                       p + 'attn.qkv.weight': (384, 128), p + 'attn.qkv.bias': (384,), p + 'attn.proj.weight': (128, 128), p + 'attn.proj.bias': (128,),
                       p + 'mlp.fc1.weight': (256, 128), p + 'mlp.fc1.bias': (256,), p + 'mlp.fc2.weight': (128, 256), p + 'mlp.fc2.bias': (128,)})
    return synth.synth_state_dict(shapes, SEED)


def test_ste_forward_backward_vs_reference_autograd(golden):
    g = golden('g15_ste_grad')
    sdn = ste_params()
    P = {k: dev(v) for k, v in sdn.items()}
    x = dev(synth.synth_input('stegrad.x', (3, 42, 128), SEED))
    gy = dev(synth.synth_input('stegrad.gy', (3, 42, 64), SEED))
    y, ctx = STE.ste_forward(P, x)
    assert maxabs(y.cpu().numpy(), g['y']) < 3e-5
    gx, G = STE.ste_backward(P, ctx, gy)
    assert rel(gx, g['grad.x']) < 1e-5
    assert not any(k.startswith('STEblocks.0.') for k in G) and len(G) == 43
    worst = check_compact_grads({k: v.cpu().numpy() for k, v in G.items()}, g, 1e-5)
    print('STE backward vs torch autograd through the reference: worst %.2e' % worst)
    # and the analytic float64 oracle at another batch size
    from oracle.ste_grad import ste_forward_backward
    rng = np.random.RandomState(4)
    x2, gy2 = rng.normal(0, 1, (7, 42, 128)).astype(np.float32), rng.normal(0, 1, (7, 42, 64)).astype(np.float32)
    yr, gxr, Gr = ste_forward_backward(sdn, x2, gy2)
    y2, ctx2 = STE.ste_forward(P, dev(x2))
    gx2, G2 = STE.ste_backward(P, ctx2, dev(gy2))
    assert rel(y2, yr) < 1e-5 and rel(gx2, gxr) < 1e-5
    for k in Gr:
        assert rel(G2[k], Gr[k]) < 1e-5, k


def pgcn_params():
    s = {}
    for i in range(4):
        p = 'gconv_layers.%d.' % i
        s.update({p + 'gconv.W': (2, 21, 128, 128), p + 'gconv.e_0': (1, 21), p + 'gconv.e_1': (1, 40), p + 'gconv.bias': (128,), p + 'bn.weight': (128,),
                  p + 'bn.bias': (128,), p + 'bn.running_mean': (128,), p + 'bn.running_var': (128,), p + 'bn.num_batches_tracked': ()})
    return synth.synth_state_dict(s, SEED)


def test_pgcn_train_forward_backward_vs_reference_autograd(golden):
    """ResSimplePGCN in training mode (batch-statistics BatchNorm1d): G16 = torch autograd through the reference's own modules"""
    from dir_amd.train import pgcn as PG
    from oracle.pgcn_grad import pgcn_train_forward_backward
    g = golden('g16_pgcn_grad')
    sdn = pgcn_params()
    P = {k: dev(v) for k, v in sdn.items() if np.asarray(v).dtype.kind == 'f'}
    x = dev(synth.synth_input('pgcngrad.x', (5, 21, 128), SEED))
    gy = dev(synth.synth_input('pgcngrad.gy', (5, 21, 128), SEED))
    y, ctx = PG.pgcn_forward(P, x)
    assert rel(y, g['y']) < 1e-5
    for k in g:
        if k.startswith('after.') and 'running' in k:
            assert rel(P[k[6:]], g[k]) < 1e-5, k                               # running statistics updated like torch (momentum 0.1, unbiased var)
    gx, G = PG.pgcn_backward(P, ctx, gy)
    assert rel(gx, g['grad.x']) < 2e-5
    Gn = {k: (v.cpu().numpy().reshape(2 * 21 * 128, 128) if k.endswith('gconv.W') else v.cpu().numpy()) for k, v in G.items()}
    worst = check_compact_grads(Gn, g, 2e-5, zero_suffixes=('gconv.bias', 'gconv.e_0'))
    print('P-GCN (training mode) backward vs torch autograd through the reference: worst %.2e' % worst)
    # the float64 oracle at a batch size that is not a multiple of anything
    rng = np.random.RandomState(2)
    x2, gy2 = rng.normal(0, 1, (13, 21, 128)).astype(np.float32), rng.normal(0, 1, (13, 21, 128)).astype(np.float32)
    yr, gxr, Gr, _ = pgcn_train_forward_backward(sdn, x2, gy2)
    P2 = {k: dev(v) for k, v in sdn.items() if np.asarray(v).dtype.kind == 'f'}
    y2, ctx2 = PG.pgcn_forward(P2, dev(x2))
    gx2, G2 = PG.pgcn_backward(P2, ctx2, dev(gy2))
    assert rel(y2, yr) < 1e-5 and rel(gx2, gxr) < 2e-5
    for k in Gr:
        if not k.endswith(('gconv.bias', 'gconv.e_0')):
            assert rel(G2[k], Gr[k]) < 2e-5, k


@pytest.mark.parametrize('R,C', [(300, 128), (5000, 64), (70000, 256), (4097, 6), (1537, 36)])
@pytest.mark.parametrize('relu', [False, True])
def test_batchnorm_relu_fused(R, C, relu):
    """dir_bn_train_forward / _backward (one-thread-per-channel, scalar chunked and 16-byte chunked paths), plain and with the fused ReLU,
    against torch's BatchNorm1d (+ ReLU) in float64 on the CPU"""
    rng = np.random.RandomState(R + C)
    x = (rng.normal(0, 1, (R, C)) * rng.uniform(0.5, 3, C) + rng.normal(0, 2, C)).astype(np.float32)
    w, b, gy = rng.normal(0, 1, C).astype(np.float32), rng.normal(0, 0.5, C).astype(np.float32), rng.normal(0, 1, (R, C)).astype(np.float32)
    bn = torch.nn.BatchNorm1d(C).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(w)); bn.bias.copy_(torch.from_numpy(b))
    xt = torch.from_numpy(x).double().requires_grad_(True)
    yt = bn(xt)
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y, st = O.bn_train_fwd(dev(x), dev(w), dev(b), rm, rv, relu=relu)
    yref = torch.relu(yt) if relu else yt
    assert rel(y, yref.detach().numpy()) < 5e-6 and rel(rm, bn.running_mean.numpy()) < 5e-6 and rel(rv, bn.running_var.numpy()) < 5e-6
    # the ReLU's gradient mask is the fp32 forward's (an element within rounding of zero may sit on the other side in float64): the reference
    # differentiates BatchNorm in float64 under THAT mask, which is what autograd does with the fp32 forward's saved output
    mask = (y > 0).double().cpu() if relu else torch.ones(R, C, dtype=torch.float64)
    if relu:
        assert float((mask - (yt.detach() > 0).double()).abs().mean()) < 1e-5
    yt.backward(torch.from_numpy(gy).double() * mask)
    gx, gw, gb = O.bn_train_bwd(dev(gy), dev(x), dev(w), st, b=dev(b), relu=relu)
    assert rel(gx, xt.grad.numpy()) < 3e-5 and rel(gw, bn.weight.grad.numpy()) < 1e-5 and rel(gb, bn.bias.grad.numpy()) < 1e-5
    y2, _ = O.bn_train_fwd(dev(x), dev(w), dev(b), relu=relu)
    assert torch.equal(y, y2)
    # residual added before the ReLU (a bottleneck's tail): relu(bn(x) + res)
    res = rng.normal(0, 1, (R, C)).astype(np.float32)
    y3, _ = O.bn_train_fwd(dev(x), dev(w), dev(b), relu=relu, residual=dev(res))
    want = yt.detach() + torch.from_numpy(res).double()
    assert rel(y3, (torch.relu(want) if relu else want).numpy()) < 5e-6


@pytest.mark.parametrize('R,C', [(5000, 64), (70000, 256), (131072, 64), (2048, 2048), (8192, 1024), (1537, 36), (33000, 48)])
def test_batchnorm_one_launch_gives_the_bits_of_the_three_launches(R, C):
    """BatchNorm over feature maps in ONE launch each way (bn_one_fwd_kernel / bn_one_bwd_kernel: persistent grid, per 64-channel
    group the last chunk to arrive combines, the others wait on the group's flag) against the three dependent launches it replaces
    (dir_bn_one_launch_enable(0)): the same chunk partials combined in the same order -> y, running statistics, saved statistics, g x, g w, g b
    equal bit for bit; with ReLU, with a residual, without g x; again on a second stream and from a replayed HIP graph (the sync words are left
    clean by every launch); and no workgroup ever gave up waiting (dir_bn_one_launch_status)."""
    from dir_amd import _capi
    L = _capi.lib()
    rng = np.random.RandomState(R + C)
    x = dev((rng.normal(0, 1, (R, C)) * rng.uniform(0.5, 3, C) + rng.normal(0, 2, C)).astype(np.float32))
    w, b, gy = dev(rng.normal(0, 1, C).astype(np.float32)), dev(rng.normal(0, 0.5, C).astype(np.float32)), dev(rng.normal(0, 1, (R, C)).astype(np.float32))
    res = dev(rng.normal(0, 1, (R, C)).astype(np.float32))

    def run(relu, residual, need_gx=True):
        rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
        y, st = O.bn_train_fwd(x, w, b, rm, rv, relu=relu, residual=residual)
        gx, gw, gb = O.bn_train_bwd(gy, x, w, st, need_gx=need_gx, b=b, relu=relu and residual is None)
        return [y, st[0], st[1], rm, rv, gw, gb] + ([gx] if need_gx else [])
    cases = [(False, None, True), (True, None, True), (True, res, True), (False, None, False)]
    was = L.dir_bn_one_launch_enable(0)
    try:
        want = [run(*c) for c in cases]
        L.dir_bn_one_launch_enable(1)
        got = [run(*c) for c in cases]
        again = [run(*c) for c in cases]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            on_side = [run(*c) for c in cases]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            captured = [run(*c) for c in cases]
        for t in (t for c in captured for t in c):
            t.zero_()
        g.replay()
        g.replay()
        torch.cuda.synchronize()
    finally:
        L.dir_bn_one_launch_enable(was)
    for name, other in (('one launch', got), ('second call', again), ('side stream', on_side), ('graph replay', captured)):
        for c, ws, gs in zip(cases, want, other):
            for i, (a_, b_) in enumerate(zip(ws, gs)):
                assert torch.equal(a_, b_), (name, c[0], c[1] is not None, c[2], i, float((a_ - b_).abs().max()))
    assert L.dir_bn_one_launch_status() == 0


def test_axpy_multi_matches_per_tensor_axpy():
    """dir_axpy_multi_f32 (work split by elements: one huge tensor beside hundreds of tiny ones, odd lengths, unaligned views)"""
    g = torch.Generator(device='cuda').manual_seed(5)
    sizes = [37748736 // 8, 3, 64, 1, 8192, 8193, 100003] + [64 + 7 * i for i in range(90)]
    pool = torch.randn(sum(sizes) + len(sizes), device='cuda', generator=g)
    dsts, srcs, o = [], [], 0
    for i, n in enumerate(sizes):
        dsts.append(pool[o + (i & 1):o + (i & 1) + n])                     # every other view starts off a 16-byte boundary
        srcs.append(torch.randn(n, device='cuda', generator=g))
        o += n + 1
    want = [d + 0.25 * s_ for d, s_ in zip(dsts, srcs)]
    O.axpy_multi(dsts, srcs, 0.25)
    for d, w_ in zip(dsts, want):
        assert float((d - w_).abs().max()) <= 2.4e-7 * float(w_.abs().max())            # (the kernel's multiply-add is fused)


@pytest.mark.parametrize('M,N,K,ta,tb', [(80, 150, 70, True, False), (80, 130, 64, False, True), (100, 70, 45, False, True), (37, 64, 33, True, False)])
def test_gemm_grouped_equals_one_call_per_group(M, N, K, ta, tb):
    """dir_gemm_f32_grouped (taps as pointer shifts, csrc/bonefuse_bwd.hip): independent groups = the same bits as one dir_gemm_f32 call per group
    (the same k order per element on the 64 x 64 and on the 80-row tiles); reduced groups = their sum (another association: 5e-6 of the largest sum)."""
    torch.manual_seed(M + N + K)
    batch, ny, nx = 3, 3, 3
    ra, ca = (K, M) if ta else (M, K)
    rb, cb = (N, K) if tb else (K, N)
    pa, pb = 5, 7                                                  # row displacement per gy step; gx steps one row
    A = torch.randn(batch, ra + pa * ny + nx, ca, device='cuda')
    Bm = torch.randn(batch, rb + pb * ny + nx, cb, device='cuda')
    sa, sb = A[0].numel(), Bm[0].numel()
    ref = torch.empty(batch, ny * nx, M, N, device='cuda')
    for g in range(ny * nx):
        gy, gx = divmod(g, nx)
        O.gemm_strided(A, Bm, ref, M, N, K, ca, cb, N, ta=ta, tb=tb, batch=batch, sa=sa, sb=sb, sc=ny * nx * M * N,
                       a_off=(gy * pa + gx) * ca, b_off=(gy * pb + gx) * cb, c_off=g * M * N)
    out = torch.full_like(ref, float('nan'))
    O.gemm_grouped(A, Bm, out, M, N, K, ca, cb, N, (ny, nx), (pa * ca, ca), (pb * cb, cb), (nx * M * N, M * N), ta=ta, tb=tb, batch=batch, sa=sa, sb=sb,
                   sc=ny * nx * M * N)
    assert torch.equal(out, ref)
    red = torch.full((batch, M, N), float('nan'), device='cuda')
    O.gemm_grouped(A, Bm, red, M, N, K, ca, cb, N, (ny, nx), (pa * ca, ca), (pb * cb, cb), reduce=True, ta=ta, tb=tb, batch=batch, sa=sa, sb=sb, sc=M * N)
    want = ref.double().sum(1)
    assert float((red.double() - want).abs().max() / want.abs().max()) < 5e-6


@pytest.mark.parametrize('M,N,K,batch,ta,tb', [(32, 128, 128, 21, False, True), (150, 130, 384, 1, False, False), (64, 128, 100, 3, True, False),
                                               (70, 64, 257, 2, True, True)])
def test_gemm_grouped_with_one_group_gives_the_bits_of_gemm(M, N, K, batch, ta, tb):
    """include/dir_hip.h: "one group: the same bits as dir_gemm_f32" (the same tiles, the same k order per element)"""
    torch.manual_seed(K)
    A = torch.randn(batch, *((K, M) if ta else (M, K)), device='cuda')
    Bm = torch.randn(batch, *((N, K) if tb else (K, N)), device='cuda')
    c1 = torch.full((batch, M, N), float('nan'), device='cuda')
    c2 = torch.full((batch, M, N), float('nan'), device='cuda')
    O.gemm_strided(A, Bm, c1, M, N, K, A.shape[2], Bm.shape[2], N, ta=ta, tb=tb, batch=batch, sa=A[0].numel(), sb=Bm[0].numel(), sc=M * N)
    O.gemm_grouped(A, Bm, c2, M, N, K, A.shape[2], Bm.shape[2], N, (1, 1), (0, 0), (0, 0), ta=ta, tb=tb, batch=batch, sa=A[0].numel(), sb=Bm[0].numel(), sc=M * N)
    assert torch.equal(c1, c2)
    ref = (A.double().transpose(1, 2) if ta else A.double()) @ (Bm.double().transpose(1, 2) if tb else Bm.double())
    assert float((c1.double() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize('R,N', [(1344, 200), (672, 64), (130, 4), (4096, 388), (5000, 128), (100, 36)])
def test_colsum_every_path_vs_float64(R, N):
    """dir_colsum_f32: the one-launch form of 128 .. 4096 rows (colsum_mid_kernel), the serial form below and the chunked form above it"""
    torch.manual_seed(R + N)
    x = torch.randn(R, N, device='cuda')
    ref = x.double().sum(0)
    got = O.colsum(x)
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    acc = torch.randn(N, device='cuda')
    want = acc.double() + ref
    O.colsum(x, out=acc, accumulate=True)
    assert float((acc.double() - want).abs().max() / want.abs().max()) < 2e-6
    assert torch.equal(O.colsum(x), got)
