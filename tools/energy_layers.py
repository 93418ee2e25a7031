"""Scratch: socket power while ONE conv layer (B = 64, bf16, rotating buffers) runs back to back with a forced DIR_CONV_VARIANT, for ~1 s per
(layer, variant): time per launch, median rocm-smi power over the second half of the run, joules per launch.  Under the package power cap
(four forwards in flight: DESIGN.md 9) the cheaper variant in joules, not in microseconds, is the one that raises throughput."""
import os, sys, time, threading, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import functional as F
from dir_amd.power import smi_sample

L = [  # name, H, Cin, Cout, k, stride, residual
    ('l2.c2 3x3 128->128 @32', 32, 128, 128, 3, 1, 0),
    ('l3.c2 3x3 256->256 @16', 16, 256, 256, 3, 1, 0),
    ('l3.c1 1x1 1024->256 @16', 16, 1024, 256, 1, 1, 0),
    ('l4.c2 3x3 512->512 @8', 8, 512, 512, 3, 1, 0),
    ('l4.c1 1x1 2048->512 @8', 8, 2048, 512, 1, 1, 0),
    ('attn 3x3 2048->2048 @8', 8, 2048, 2048, 3, 1, 0),
    ('dec 1x1 512->128 @32', 32, 512, 128, 1, 1, 0),
    ('dec 3x3 256->256 @32', 32, 256, 256, 3, 1, 0),
    ('dec 3x3 128->128 @16', 16, 128, 128, 3, 1, 0),
]
VARS = [int(v) for v in os.environ.get('VARS', '1,17,2,18,20,8,9,10,11,12,13,14,15').split(',')]
names = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x128', 4: '64x64', 17: '128x128r', 18: '128x64r', 20: '64x64r', 8: 'P256x128', 9: 'P128x128',
         10: 'P256x64', 11: 'B256x256', 12: 'H256x128', 13: 'H128x128', 14: 'H256x64', 15: 'P128x64'}
B, dt, NROT, SECS = 64, torch.bfloat16, 3, float(os.environ.get('SECS', 1.0))
only = os.environ.get('ONLY')
for name, H, Ci, Co, k, s, res in L:
    if only and only not in name:
        continue
    Ho = H // s
    xs = [torch.randn(B, H, H, Ci, device='cuda').to(dt) for _ in range(NROT)]
    ys = [torch.empty(B, Ho, Ho, Co, device='cuda', dtype=dt) for _ in range(NROT)]
    w = (torch.randn(Co, k, k, Ci, device='cuda') * 0.02).to(dt)
    sc, sh = torch.ones(Co, device='cuda'), torch.zeros(Co, device='cuda')
    rows = []
    for v in VARS:
        def burst(n):
            for i in range(n):
                F.conv2d_nhwc(xs[i % NROT], w, s, k // 2, scale=sc, shift=sh, relu=True, out=ys[i % NROT], variant=v)
        burst(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); burst(20); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        samples, stop = [], threading.Event()

        def sampler():
            t0 = time.perf_counter()
            while not stop.is_set():
                r = smi_sample()
                if r and time.perf_counter() - t0 > SECS * 0.5:
                    samples.append(r)
                stop.wait(0.05)
        th = threading.Thread(target=sampler, daemon=True); th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < SECS:
            burst(200); torch.cuda.synchronize(); n += 200
        dtw = time.perf_counter() - t0
        stop.set(); th.join()
        if not samples:
            print('no rocm-smi samples'); sys.exit(1)
        pw = statistics.median(x['w'] for x in samples)
        clk = statistics.median(x['sclk'] for x in samples)
        rows.append((v, us, dtw / n * 1e6, pw, clk))
        time.sleep(0.3)
    print(name)
    for v, us, usw, pw, clk in rows:
        print('   %-9s %7.1f us alone  %7.1f us in the loop  %6.0f W  %5.0f MHz  %7.4f J/launch  (dynamic %7.4f J at 243 W idle)' %
              (names.get(v, str(v)), us, usw, pw, clk, pw * usw * 1e-6, (pw - 243.0) * usw * 1e-6), flush=True)
