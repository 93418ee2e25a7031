"""The training step's bone_proj + fusion.0 (dir_bone_fusion_prepare / _forward exact fp32, dir_bone_fusion_backward) alone, next to the materialised
form it replaced (dir_bone_proj_forward + the K = 23 040 convolution and its two gradients + dir_bone_proj_backward).
usage: bench_bone_fusion_train.py [B] [S] [reps]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd.train import conv as TC  # noqa: E402
from dir_amd.train import spatial as SP  # noqa: E402
from oracle.golden_inputs import bone_grad_inputs  # noqa: E402  (seeded joint positions only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dist = 2 if S == 32 else 1
rng = np.random.default_rng(0)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
uv_l = dv(bone_grad_inputs(S, B)[0])
uv_r = dv(bone_grad_inputs(S, B)[0][::-1].copy())
emb = dv(rng.standard_normal((B, 42, 64)).astype(np.float32))
W = dv((rng.standard_normal((256, 2560, 3, 3)) * 0.02).astype(np.float32))
bias = dv(rng.standard_normal(256).astype(np.float32))
gy = dv(rng.standard_normal((B, S, S, 256)).astype(np.float32))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


w_g = SP.fusion_w_g(W)
y, ctx = SP.bone_fusion_fwd(uv_l, uv_r, emb, w_g, bias, S, dist)
t_pack = timed(lambda: SP.fusion_w_g(W))
t_f = timed(lambda: SP.bone_fusion_fwd(uv_l, uv_r, emb, w_g, bias, S, dist))
t_b = timed(lambda: SP.bone_fusion_bwd(ctx, gy))
t_unpack = timed(lambda: SP.fusion_w_g_grad_to_oihw(w_g))
print('factorised  B=%d S=%d: weight permute %.3f ms, forward %.3f ms, backward %.3f ms, gradient permute %.3f ms' % (B, S, t_pack, t_f, t_b, t_unpack))
if os.environ.get('MATERIALISED', '1') != '0':
    bone = SP.bone_proj_fwd(uv_l, uv_r, emb, S, dist)
    t_bp = timed(lambda: SP.bone_proj_fwd(uv_l, uv_r, emb, S, dist))
    t_cf = timed(lambda: TC.conv_fwd(bone, W, bias, 1, 1, oihw=True))
    t_cb = timed(lambda: TC.conv_bwd(bone, W, gy, 1, 1, need_gx=True, has_bias=True, oihw=True))
    gx = TC.conv_bwd(bone, W, gy, 1, 1, need_gx=True, has_bias=True, oihw=True)[0]
    t_bb = timed(lambda: SP.bone_proj_bwd(uv_l, uv_r, emb, gx, S, dist))
    print('materialised B=%d S=%d: bone_proj %.3f ms, convolution %.3f ms, its two gradients %.3f ms, bone_proj backward %.3f ms' % (B, S, t_bp, t_cf, t_cb, t_bb))
