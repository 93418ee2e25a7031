"""The P-GCN stack (4 PGraphConv layers + mix, both hands: dir_pgcn_stack_forward_pair) alone at one batch size, N calls -- the command
tools/pgcn_sweep.sh wraps in rocprofv3 (kernel trace, then --pmc FETCH_SIZE / WRITE_SIZE passes).  python tools/pgcn_sweep.py B [calls] [f32]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import _capi, engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else torch.bfloat16
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in synth.synth_state_dict(shapes, 1234).items() if k.startswith('decoder.projecter_3.gcn_')}
keep = []
gcn = (E.pack_pgcn(sd, 'decoder.projecter_3.gcn_left', keep, weight_dtype=dt), E.pack_pgcn(sd, 'decoder.projecter_3.gcn_right', keep, weight_dtype=dt))
L = _capi.lib()
x0 = torch.randn(2, B, 21, 128, device='cuda'); gp = torch.randn(2, B, 21, 128, device='cuda')
tok = torch.empty(B, 42, 128, device='cuda'); scratch = torch.empty(4, B, 21, 256, device='cuda')
for _ in range(calls):
    _capi.check(L.dir_pgcn_stack_forward_pair(gcn[0], gcn[1], 4, _capi.ptr(x0), _capi.ptr(gp), _capi.ptr(tok), _capi.ptr(scratch), B, _capi.stream_ptr()), 'pgcn')
torch.cuda.synchronize()
wes = 2 if dt == torch.bfloat16 else 4
print('B=%d weights %s: algorithmic bytes per stack call = weights %.2f MB + activations %.2f MB' % (
    B, 'bf16' if wes == 2 else 'f32', 4 * 2 * 2 * 21 * 128 * 128 * wes / 1e6, 4 * 2 * 2 * B * 21 * 128 * 4 / 1e6))
