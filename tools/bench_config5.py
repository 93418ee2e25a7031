"""BASELINE config 5 on one GPU: HRNet-W48 + init regression + 4 refinement stages ("5 refinement iters"), 32 images per GPU (batch 256 over 8):
captured forwards, live autotune, 1 / 2 / 4 / 6 / 8 forwards in flight, bf16 and the fp16 MFMA path.  python tools/bench_config5.py [B]"""
import os, sys, time, statistics
os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import synth, power as P
from dir_amd.engine import DirEngine, ForwardPipeline
from dir_amd.models.dir import DIR

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = DIR(21, 'x', 0, backbone='hrnet_w48', extra_stages=2)
shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
del net
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234, cond=True).items()}
g = torch.Generator(device='cuda').manual_seed(5)
imgs = [torch.randn(B, 3, 256, 256, device='cuda', generator=g) for _ in range(8)]
for dt, arith, tag in ((torch.bfloat16, None, 'bf16'), (torch.float32, 'f16', 'fp16 MFMA path (fp32 feature maps)')):
    eng = DirEngine(sd, dtype=dt, arith=arith)
    eng.calibrate(imgs[0])
    eng.forward(imgs[0]); torch.cuda.synchronize()
    eng.autotune(imgs[0], reps=1)
    streams = [torch.cuda.Stream() for _ in range(8)]
    row = 'config 5, %s, B=%d:' % (tag, B)
    for nfl in (1, 2, 4, 6, 8):
        pipe = ForwardPipeline(eng, imgs[:nfl], streams=streams[:nfl])
        k = [0]

        def step():
            pipe.launch(k[0] % nfl); k[0] += 1
        for _ in range(2 * nfl):
            step()
        torch.cuda.synchronize()
        e0 = P.energy_joules(); t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            for _ in range(16):
                step()
            torch.cuda.synchronize(); n += 16
        dt_ = (time.perf_counter() - t0) / n
        e1 = P.energy_joules()
        row += '   %d in flight %.2f ms = %.0f img/s (%.2f J)' % (nfl, dt_ * 1e3, B / dt_, (e1[0] - e0[0]) / n if e0 and e1 else float('nan'))
        del pipe
    print(row, flush=True)
    del eng
