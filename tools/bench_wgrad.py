"""dir_conv2d_wgrad_f16x3 alone on the training step's heaviest shapes (HIP-event time, TFLOP/s) -- the target of tools/pmc_wgrad.sh.
python tools/bench_wgrad.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dir_amd.train import conv as TC
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = os.environ.get('ONLY')
CASES = [('fusion 32^2', 32, 32, 2560, 256, 3, 1), ('layer2 3x3', 32, 32, 128, 128, 3, 1), ('layer1 3x3', 32, 64, 64, 64, 3, 1), ('layer1 1x1', 32, 64, 64, 256, 1, 1),
         ('layer3 3x3', 32, 16, 256, 256, 3, 1), ('attention 8^2', 32, 8, 2048, 1024, 3, 1)]
for name, B, H, Cin, Cout, k, s in CASES:
    if only and only not in name:
        continue
    x = torch.randn(B, H, H, Cin, device='cuda')
    gy = torch.randn(B, H, H, Cout, device='cuda')
    f = lambda: TC.conv_wgrad(x, gy, (Cout, k, k, Cin), s, k // 2)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * H * H * Cout * Cin * k * k
    print('%-14s M=%6d Cout=%4d Cin=%4d k%d  %8.3f ms  %6.1f TFLOP/s (x3 peak 833)' % (name, B * H * H, Cout, Cin, k, t, fl / t / 1e9))
