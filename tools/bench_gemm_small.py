"""The joint-token path's small exact-fp32 products (dir_gemm_f32) as a serial chain of launches: us per call.  DIR_GEMM_DEEPK=0 | 1 A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd.train import ops as O

SHAPES = [  # M, N, K, batch, ta, tb   (from profiles/r04_h_train_step_library_calls.txt)
    (32, 128, 128, 21, 0, 1), (32, 128, 128, 21, 0, 0), (128, 128, 32, 21, 1, 0), (672, 128, 128, 1, 0, 0), (1344, 128, 384, 1, 0, 0),
    (1344, 128, 256, 1, 0, 1), (1344, 384, 128, 1, 0, 1), (1344, 256, 128, 1, 0, 0), (672, 128, 256, 1, 0, 1)]
for M, N, K, batch, ta, tb in SHAPES:
    A = torch.randn(batch, *((K, M) if ta else (M, K)), device='cuda')
    B = torch.randn(batch, *((N, K) if tb else (K, N)), device='cuda')
    C = torch.empty(batch, M, N, device='cuda')
    f = lambda: O.gemm_strided(A, B, C, M, N, K, A.shape[2], B.shape[2], N, ta=bool(ta), tb=bool(tb), batch=batch, sa=A[0].numel(), sb=B[0].numel(), sc=M * N)  # noqa: E731
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        f()
    e1.record()
    torch.cuda.synchronize()
    print('M=%d N=%d K=%d batch=%d ta=%d tb=%d: %.2f us per call in a chain' % (M, N, K, batch, ta, tb, e0.elapsed_time(e1) / 300 * 1e3))
