"""Scratch: two forwards in flight, each confined to half of the CUs (hipExtStreamCreateWithCUMask) -- XCD-aligned halves keep each
forward's working set in its own four L2s.  Does partitioning beat free sharing?"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='xcd', choices=['none', 'xcd', 'halves', 'xcd_pairs'])
ap.add_argument('--autotune-cache', default='/tmp/at.json')
args = ap.parse_args()
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = 64
hip = C.CDLL('libamdhip64.so')


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


if args.mode == 'none':
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
else:
    if args.mode == 'xcd':          # CU bit i belongs to XCD i % 8 (round-robin numbering): XCDs 0-3 | 4-7
        sel = [lambda i: (i % 8) < 4, lambda i: (i % 8) >= 4]
    elif args.mode == 'xcd_pairs':  # even | odd XCDs
        sel = [lambda i: (i % 2) == 0, lambda i: (i % 2) == 1]
    else:                           # contiguous halves of the bit vector
        sel = [lambda i: i < 128, lambda i: i >= 128]
    streams = [masked_stream([f(i) for i in range(256)]) for f in sel]
eng = E.DirEngine(sd, dtype=torch.bfloat16)
imgs = [torch.randn(B, 3, 256, 256, device='cuda') for _ in range(2)]
eng.forward(imgs[0]); torch.cuda.synchronize()
if os.path.exists(args.autotune_cache):
    eng.import_tuning(imgs[0], json.load(open(args.autotune_cache)))
else:
    eng.autotune(imgs[0]); json.dump(eng.export_tuning(B), open(args.autotune_cache, 'w'))
graphs = []
for s, im in zip(streams, imgs):
    with torch.cuda.stream(s):
        eng.forward(im); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.forward(im)
    graphs.append(g)
torch.cuda.synchronize()
for n in (1, 2):
    for w in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(60):
            with torch.cuda.stream(streams[k % n]):
                graphs[k % n].replay()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('mode %-9s in flight %d: %.3f ms per forward' % (args.mode, n, dt / 60 * 1e3))
