"""BASELINE config 5 on one GPU: HRNet-W48 + init regression + 4 refinement stages ("5 refinement iters"), 32 images per GPU (batch 256 over 8)
and 64; one captured forward replayed, bf16 and the fp16 MFMA path.  python tools/bench_config5.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import synth
from dir_amd.engine import DirEngine
from dir_amd.models.dir import DIR

net = DIR(21, 'x', 0, backbone='hrnet_w48', extra_stages=2)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=True).items()}
for dt, arith, tag in ((torch.bfloat16, None, 'bf16'), (torch.float32, 'f16', 'fp16 MFMA path (fp32 feature maps)')):
    eng = DirEngine(sd, dtype=dt, arith=arith)
    for B in (32, 64):
        img = torch.randn(B, 3, 256, 256, device='cuda')
        eng.calibrate(img)
        eng.forward(img); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.forward(img)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / 10
        print('config 5, %s, B=%d: %.2f ms per forward = %.0f images/s (one forward in flight, heuristic kernel choice)' % (tag, B, dt_ * 1e3, B / dt_))
