"""Scratch: do two independent 64-workgroup kernels on two streams overlap -- eagerly and inside a captured HIP graph?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctypes as C
from dir_amd import _capi, engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in synth.synth_state_dict(shapes, 1234).items()}
keep = []
st = E.StageOp(sd, 'decoder.projecter_3', 32, 2, torch.bfloat16, 0, keep)
L = _capi.lib()
B = 64
xs = [torch.randn(B, 42, 128, device='cuda') for _ in range(2)]
ys = [torch.empty(B, 42, 64, device='cuda') for _ in range(2)]
side = torch.cuda.Stream()


def ste(i):
    L.dir_ste_forward(C.byref(st.ste), _capi.ptr(xs[i]), None, _capi.ptr(ys[i]), B, _capi.stream_ptr())


def serial():
    ste(0); ste(1)


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    ste(0)
    with torch.cuda.stream(side):
        ste(1)
    main.wait_stream(side)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('eager serial : %.1f us' % timeit(serial))
print('eager forked : %.1f us' % timeit(forked))
for name, fn in (('serial', serial), ('forked', forked)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    print('graph %s : %.1f us' % (name, timeit(g.replay)))
