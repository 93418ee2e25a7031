"""Scratch: at B = 64 (bf16), are (a) autotuned variants bit-identical to the heuristic ones, (b) concurrent pipeline slots
bit-identical to one-at-a-time replays?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import engine as E, synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
B = 64
gen = torch.Generator(device='cuda').manual_seed(64)
imgs = [torch.randn(B, 3, 256, 256, device='cuda', generator=gen) for _ in range(2)]
KEYS = ('pd_mesh_xyz_left', 'pd_joint_uv_right', 'pd_offset')


def snap(o):
    return [o[s][k].clone() for s in range(3) for k in KEYS] + [o[3]['seg'].clone()]


def same(a, b):
    return [i for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)]


eng = E.DirEngine(sd, dtype=torch.bfloat16)
if os.environ.get('NO_OVERLAP'): eng.overlap = False
print('overlap', eng.overlap, 'factorised', eng.factorised_fusion)
base = [snap(eng.forward(im)) for im in imgs]
torch.cuda.synchronize()
for trial in range(int(os.environ.get('TRIALS', '2'))):
    if not os.environ.get('NO_TUNE'): eng.autotune(imgs[0])
    tuned = [snap(eng.forward(im)) for im in imgs]
    torch.cuda.synchronize()
    bad = [same(t, b) for t, b in zip(tuned, base)]
    print('trial %d: tuned eager vs heuristic eager: mismatching tensors %s' % (trial, bad))
    if any(bad):
        table = eng.export_tuning(B)
        json.dump(table, open(os.path.join(ROOT, 'gpurun_out', 'bad_tuning_%d.json' % trial), 'w'))
    pipe = E.ForwardPipeline(eng, imgs)
    for rep in range(int(os.environ.get('REPS', '10'))):
        pipe.launch(0); pipe.launch(1)
        o = [snap(pipe.wait(s)) for s in (0, 1)]
        bad = [same(x, t) for x, t in zip(o, tuned)]
        if any(bad):
            print('   rep %d: concurrent slots vs tuned eager: %s' % (rep, bad))
    src = [im.clone() for im in imgs]
    for rep in range(10):                      # refill on the slot's own stream, host wait for the results
        for s in (0, 1):
            pipe.refill(s, src[s]); pipe.launch(s)
        o = [snap(pipe.wait(s)) for s in (0, 1)]
        torch.cuda.synchronize()
        bad = [same(x, t) for x, t in zip(o, tuned)]
        if any(bad):
            print('   rep %d: refill + launch vs tuned eager: %s' % (rep, bad))
    for rep in range(3):
        for s in (0, 1):
            pipe.launch(s)
            x = snap(pipe.wait(s))
            b = same(x, tuned[s])
            if b:
                print('   rep %d slot %d: serial replay vs tuned eager: %s' % (rep, s, b))
print('done')
