"""Scratch (build first: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/_ubench/canary_vgpr.so tools/canary_vgpr.hip): VGPR canary beside the suspected aggressor (skip4.dual, conv variant 19 = 64x128 tile with the 3-buffer DMA ring)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = 64
eng = E.DirEngine(sd, dtype=torch.bfloat16)
lib = C.CDLL(os.path.join(ROOT, 'tools', '_ubench', 'canary_vgpr.so'))
g = torch.Generator(device='cuda').manual_seed(3)
c3 = torch.randn(B, 16, 16, 1024, device='cuda', generator=g).to(torch.bfloat16)
res4 = eng.res['skip_layer4']
y2 = res4.c2(res4.c1(c3))
sa, sv = torch.cuda.Stream(), torch.cuda.Stream()
MAXR = 4096
for v in (3, 19, 18):
    res4.dual.variant[B] = v
    rep = torch.zeros(1 + 4 * MAXR, device='cuda', dtype=torch.int32)
    torch.cuda.synchronize()
    for it in range(20):
        with torch.cuda.stream(sa):
            for _ in range(20): res4.dual(y2, c3)
        with torch.cuda.stream(sv):
            lib.canary_vgpr_launch(C.c_void_p(rep.data_ptr()), 2048, 40, MAXR, C.c_void_p(sv.cuda_stream))
    torch.cuda.synchronize()
    r = rep.cpu().numpy().astype(np.uint32)
    n = int(r[0])
    print('variant %d: %d lanes with corrupted VGPRs' % (v, n))
    if n:
        ent = r[1:1 + 4 * min(n, MAXR)].reshape(-1, 4)
        print('   lanes hit: %d; first bad register index histogram: %s' % (len(ent), sorted(set(ent[:, 1].tolist()))[:40]))
        print('   lane ids (mod 64):', sorted(set((ent[:, 0] % 64).tolist()))[:64])
        print('   waves (tid/64 mod 4):', sorted(set(((ent[:, 0] // 64) % 4).tolist())), ' bad regs per lane:', sorted(set(ent[:, 3].tolist()))[:20])
        print('   sample values:', [hex(x) for x in ent[:8, 2]])
