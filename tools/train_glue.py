"""Which torch operators (host glue between the library calls) one training step launches, and from where: torch.profiler over the third step
of tools/bench_train.py's set-up.  python tools/train_glue.py [batch]"""
import os, sys, json, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dir_amd import synth
from dir_amd.optim import FlatAdamW
from dir_amd.train import step as TSTEP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json'))).items()}
sd = synth.synth_state_dict(shapes, 1234)
is_buf = lambda k: any(t in k for t in ('running_', 'num_batches', 'mano_layer', 'img_gird', 'seg_loss.weight'))
params = {k: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).cuda()) for k, v in sd.items() if not is_buf(k)}
buffers = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sd.items() if is_buf(k) and 'num_batches' not in k}
opt = FlatAdamW(list(params.values()), lr=1e-5)
opt.set_inactive(TSTEP.inactive_parameters(params))
rng = np.random.RandomState(0)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
img = dv(synth.synth_input('train.img.0', (B, 3, 256, 256), 1234))
target, meta = {}, {}
for s in ('left', 'right'):
    target['joint_2d_' + s] = dv(rng.uniform(-1, 1, (B, 21, 3)).astype(np.float32))
    target['mesh_2d_' + s] = dv(rng.uniform(-1, 1, (B, 778, 3)).astype(np.float32))
    target['joint_3d_' + s] = dv(rng.normal(0, 0.05, (B, 21, 3)).astype(np.float32))
    target['mesh_3d_' + s] = dv(rng.normal(0, 0.05, (B, 778, 3)).astype(np.float32))
    meta['center_' + s] = dv(rng.normal(0, 0.1, (B, 1, 3)).astype(np.float32))
target['seg'] = dv(rng.randint(0, 3, (B, 1, 256, 256)).astype(np.float32))
target['dense'] = dv(rng.rand(B, 3, 256, 256).astype(np.float32))
faces = tuple(dv(synth.loss_faces(s, 1234).astype(np.int64)) for s in ('left', 'right'))
for _ in range(2):
    TSTEP.train_step(params, buffers, img, target, meta, faces, opt)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    TSTEP.train_step(params, buffers, img, target, meta, faces, opt)
    torch.cuda.synchronize()
ev = prof.events()
# kernels by name
kern = collections.Counter(); ktime = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:60]] += 1; ktime[e.name[:60]] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
print('device activities: %d, %.1f ms' % (sum(kern.values()), sum(ktime.values()) / 1e3))
for k, n in kern.most_common(25):
    print('  %5d %9.1f us  %s' % (n, ktime[k], k))
# aten ops that are leaves (launch something) by python call site
site = collections.Counter(); stime = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and e.kernels:
        st = [s for s in (e.stack or []) if 'dir_amd' in s or 'tools/' in s]
        where = st[0].split('/root/repo/')[-1] if st else '?'
        key = (e.name, where[:90])
        site[key] += len(e.kernels); stime[key] += sum(k.duration for k in e.kernels)
print('torch operators that launch device work, by call site (launches, device us):')
for k, n in sorted(site.items(), key=lambda kv: -stime[kv[0]])[:60]:
    print('  %5d %9.1f us  %-22s %s' % (n, stime[k], k[0], k[1]))
print('total torch-operator launches %d, %.1f us' % (sum(site.values()), sum(stime.values())))
