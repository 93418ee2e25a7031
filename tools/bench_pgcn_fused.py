"""P-GCN stack of both hands: the five launches of rounds 2-3 (dir_pgcn_stack_forward_pair) against the single launch of round 4
(dir_pgcn_stack_forward_fused), per batch size and split count, back to back on one stream (HIP events).  python tools/bench_pgcn_fused.py [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import _capi, engine, synth

wdt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else torch.bfloat16
L = _capi.lib()
shapes = {}
for i in range(4):
    p = 'gconv_layers.%d.' % i
    shapes.update({p + 'gconv.W': (2, 21, 128, 128), p + 'gconv.e_0': (1, 21), p + 'gconv.e_1': (1, 40), p + 'gconv.bias': (128,), p + 'bn.weight': (128,),
                   p + 'bn.bias': (128,), p + 'bn.running_mean': (128,), p + 'bn.running_var': (128,), p + 'bn.num_batches_tracked': ()})
keep = []
lay = [engine.pack_pgcn({('gcn.' + k): torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in synth.synth_state_dict(shapes, 1234 + h).items()}, 'gcn', keep, weight_dtype=wdt)
       for h in range(2)]
sync = torch.zeros(int(L.dir_pgcn_fused_sync_bytes()) // 4, dtype=torch.int32, device='cuda')
wes = 2 if wdt == torch.bfloat16 else 4


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (32, 64, 128, 256, 1024, 4096):
    x = torch.randn(2, B, 21, 128, device='cuda')
    add = torch.randn(2, B, 21, 128, device='cuda')
    tok = torch.empty(B, 42, 128, device='cuda')
    scratch = torch.empty(4, B, 21, 256, device='cuda')
    sp = _capi.stream_ptr()
    alg = 4 * 2 * (2 * 21 * 128 * 128 * wes + 2 * B * 21 * 128 * 4)
    t_pair = timeit(lambda: L.dir_pgcn_stack_forward_pair(lay[0], lay[1], 4, _capi.ptr(x), _capi.ptr(add), _capi.ptr(tok), _capi.ptr(scratch), B, sp))
    row = 'B=%-5d alg %.1f MB   five launches %7.1f us (%.3f of 8 TB/s)   one launch, splits:' % (B, alg / 1e6, t_pair, alg / t_pair / 8e6)
    for S in (1, 2, 3, 4, 8):
        t = timeit(lambda: L.dir_pgcn_stack_forward_fused(lay[0], lay[1], 4, _capi.ptr(x), _capi.ptr(add), _capi.ptr(tok), _capi.ptr(scratch), _capi.ptr(sync), S, B, sp))
        row += '  %d: %6.1f us (%.3f)' % (S, t, alg / t / 8e6)
    print(row, flush=True)
assert int(sync[-4]) == 0
