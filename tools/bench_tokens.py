"""Scratch: token-path kernels (P-GCN stack, STE, MANO pair) alone, over batch sizes: time per call and algorithmic GB/s.
P-GCN bytes per stack call (both hands, 4 layers): weights 4 x 5.5 MB (once) + per sample the layer inputs / outputs."""
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


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


for B in (64, 256, 1024, 4096):
    x0 = torch.randn(2, B, 21, 128, device='cuda'); gp = torch.randn(2, B, 21, 128, device='cuda')
    tok = torch.empty(B, 42, 128, device='cuda'); scratch = torch.empty(4, B, 21, 256, device='cuda')
    sp = _capi.stream_ptr()
    t = timeit(lambda: _capi.check(L.dir_pgcn_stack_forward_pair(st.gcn[0], st.gcn[1], 4, _capi.ptr(x0), _capi.ptr(gp), _capi.ptr(tok),
                                                                 _capi.ptr(scratch), B, sp), 'pgcn'))
    w_bytes = 4 * 2 * 2 * 21 * 128 * 128 * 4
    act = B * 2 * 21 * 4 * (128 + 256 + 3 * (256 + 256) + 256 + 128 + 128)      # x in, h out/in per layer, mix in, add in, out
    print('P-GCN stack (4 layers + mix, both hands) B=%5d: %8.1f us  %6.1f MB  %6.2f TB/s' % (B, t, (w_bytes + act) / 1e6, (w_bytes + act) / t / 1e6))
    y = torch.empty(B, 42, 64, device='cuda')
    t = timeit(lambda: _capi.check(L.dir_ste_forward(C.byref(st.ste), _capi.ptr(tok), None, _capi.ptr(y), B, sp), 'ste'))
    print('STE (bf16 Linears)                      B=%5d: %8.1f us  %6.1f GFLOP/s-equivalent %6.1f TFLOP/s' % (B, t, 0, 36e6 * B / t / 1e6))
    pl = torch.randn(B, 64, device='cuda') * 0.1; pr = torch.randn(B, 64, device='cuda') * 0.1
    t = timeit(lambda: E.run_mano_pair(st.mano, pl, pr, B))
    print('MANO pair                               B=%5d: %8.1f us  %6.2f M (sample,hand)/s' % (B, t, 2 * B / t))
