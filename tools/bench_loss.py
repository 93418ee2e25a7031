"""Timing of the training-objective kernels (dir_stage_losses_forward x 3 stages + dir_dense_losses_forward) at the reference's
batch sizes (32 per GPU when training, 64 for validation): HIP events, 50 repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from dir_amd import synth
from dir_amd.models import loss as ML
from test_gpu_loss import _random_stage, cuda


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (32, 64, 256):
    rng = np.random.RandomState(B)
    pred, gt = _random_stage(rng, B)
    faces = [torch.from_numpy(synth.loss_faces(s)).cuda() for s in ('left', 'right')]
    target = cuda({k: v for k, v in gt.items() if not k.startswith('center')})
    meta = cuda({k: v for k, v in gt.items() if k.startswith('center')})
    p = cuda(pred)
    seg = torch.randn(B, 3, 32, 32, device='cuda'); dense = torch.rand(B, 3, 32, 32, device='cuda')
    gs = torch.randint(0, 3, (B, 1, 256, 256), device='cuda').float(); gd = torch.rand(B, 3, 256, 256, device='cuda')
    t_stage = timeit(lambda: ML.stage_losses(p, target, meta, faces))
    t_dense = timeit(lambda: ML.dense_losses(seg, dense, gs, gd))
    csr = [ML.vertex_face_csr(f) for f in faces]
    t_sb = timeit(lambda: ML.stage_loss_grads(p, target, meta, faces, csr=csr))
    t_db = timeit(lambda: ML.dense_loss_grads(seg, dense, gs, gd))
    print('B=%3d  backward: stage gradients %.1f us, dense + lovasz gradients %.1f us' % (B, t_sb, t_db))
    # algorithmic bytes: stage = both hands' pred + gt meshes (xyz + uv), joints; dense = logits + sampled gt + sort traffic
    stage_bytes = B * 2 * (778 * (3 + 2 + 3 + 3) + 21 * (3 + 2 + 3 + 3)) * 4
    print('B=%3d  stage losses %.1f us (%.1f MB algorithmic, %.0f GB/s)   dense + lovasz %.1f us' % (
        B, t_stage, stage_bytes / 1e6, stage_bytes / t_stage / 1e3, t_dense))
