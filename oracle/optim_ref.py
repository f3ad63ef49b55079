"""CPU restatement of the optimizer arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

adamw_step(): torch.optim.AdamW single-tensor update as configured by the reference
  (/root/reference/torch_em/segmentation.py:543: lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2).
ema(): SPOCOTrainer._momentum_update /root/reference/torch_em/trainer/spoco_trainer.py:45-47.
"""
import numpy as np


def adamw_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    p = p * np.float32(1.0 - lr * weight_decay)
    m = m + np.float32(1 - beta1) * (g - m)
    v = np.float32(beta2) * v + np.float32(1 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
    p = p - np.float32(lr / bc1) * (m / denom)
    return p.astype("float32"), m.astype("float32"), v.astype("float32")


def ema(theta_k, theta_q, momentum):
    return (theta_k * np.float32(momentum) + theta_q * np.float32(1.0 - momentum)).astype("float32")
