"""CPU restatement (numpy, fp32) of the optimiser the reference trains with -- TEST INFRASTRUCTURE.

train.py:227-230: torch.optim.AdamW (defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01) + CosineAnnealingLR(T_max, eta_min=0).
The algorithm lives in PyTorch (README.md:77,95 pins torch 1.11; this image has 2.10), not under /root/reference: the restatement
follows torch's single-tensor AdamW update and is pinned against torch.optim.AdamW itself (tests/test_oracle_golden.py, CPU)."""
import math

import numpy as np

F32 = np.float32


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """one update of fp32 arrays (returned, not in place); `step` is the 1-based count of this update"""
    p, g, m, v = (np.asarray(t, F32) for t in (p, g, m, v))
    p = (p * F32(1.0 - lr * weight_decay)).astype(F32)
    m = (m + F32(1.0 - beta1) * (g - m)).astype(F32)
    v = (v * F32(beta2) + (F32(1.0 - beta2) * g) * g).astype(F32)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    denom = (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)).astype(F32)
    p = (p - F32(lr / bc1) * (m / denom)).astype(F32)
    return p, m, v


def cosine_lr(base_lr, epoch, T_max, eta_min=0.0):
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / T_max)) / 2
