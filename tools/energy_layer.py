"""Joules per launch of one convolution layer per kernel variant (socket energy counter over 0.6 s of back-to-back launches, idle power subtracted):
python tools/energy_layer.py B H W Cin Cout k [variants...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import engine as E, power
B, H, W, Ci, Co, k = (int(v) for v in sys.argv[1:7])
variants = [int(v) for v in sys.argv[7:]] or [0, 8, 9, 11, 12, 25, 26, 27, 28]
dt = torch.float16
g = torch.Generator(device='cuda').manual_seed(1)
w = torch.randn(Co, Ci, k, k, device='cuda', generator=g) * (2.0 / (k * k * Ci)) ** 0.5
op = E.ConvOp(w, dt, stride=1, pad=k // 2, scale=torch.ones(Co, device='cuda'), shift=torch.zeros(Co, device='cuda'), relu=True)
x = torch.randn(B, H, W, Ci, device='cuda', generator=g).to(dt)
out = torch.empty(B, H, W, Co, device='cuda', dtype=dt)
flops = 2.0 * B * H * W * Co * k * k * Ci
for rnd in range(2):
    for v in variants:
        E._TLS.variant = v
        for _ in range(5): op(x, out=out)
        torch.cuda.synchronize()
        e0 = power.energy_joules(); t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 0.6:
            for _ in range(50): op(x, out=out)
            torch.cuda.synchronize(); n += 50
        dtm = time.perf_counter() - t0; e1 = power.energy_joules()
        wsock = (e1[0] - e0[0]) / dtm
        us = dtm / n * 1e6
        print('round %d variant %2d: %7.1f us  %5.0f W  %7.1f mJ above %d W idle  (%.0f TF/s, %.2f pJ/FLOP)' % (rnd, v, us, wsock, us * (wsock - power.IDLE_W) / 1e3, power.IDLE_W, flops / us / 1e6, us * (wsock - power.IDLE_W) * 1e-6 / flops * 1e12), flush=True)
E._TLS.variant = None
