import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from dir_amd import _capi, engine, synth
from test_gpu_tokens import ste_shapes
sdn = synth.synth_state_dict(ste_shapes(''), 1234)
sd = {('ste.' + k): torch.from_numpy(v).cuda() for k, v in sdn.items()}
keep = []; P = engine.pack_ste(sd, 'ste', keep)
B = 64
x = torch.randn(B, 42, 128, device='cuda'); y = torch.empty(B, 42, 64, device='cuda')
L = C.CDLL(_capi.LIB_PATH)
buf = (C.c_longlong * 32)()
for it in range(3):
    _capi.check(_capi.lib().dir_ste_forward(C.byref(P), _capi.ptr(x), None, _capi.ptr(y), B, _capi.stream_ptr()), 'ste')
    torch.cuda.synchronize()
    L.dir_debug_ste_timing(buf)
names = ['(pre/loopback)', 'LN1', 'qkv linear', 'QK^T mfma', 'softmax', 'PV mfma', 'proj linear', 'LN2', 'fc1 linear', 'fc2 linear', 'snorm+copy(next iter start)', 'head']
tot = sum(buf[:12])
for i, n in enumerate(names):
    print('%-28s %9d cycles  %5.1f%%' % (n, buf[i], 100.0 * buf[i] / tot))
print('total', tot, 'cycles (block 0, thread 0; counter at 100 MHz?)')
