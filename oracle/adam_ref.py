"""CPU restatement of torch.optim.Adam as the reference configures it
(scene/gaussian_model.py:163: lr per group, eps = 1e-15, betas (0.9, 0.999), no
weight decay, no amsgrad; stepped at train.py:126-128).  Test infrastructure
only: the product's update is the HIP kernel behind gsr_adam_step.

Pinned against torch.optim.Adam itself (the reference's optimiser, importable
here) by tests/test_adam_oracle.py.
"""
import math

import numpy as np


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-15, grad_scale=1.0):
    """One in-place update of fp32 arrays p, m, v from gradient g; ``step`` counts from 1."""
    f = np.float32
    g = (g * f(grad_scale)).astype(f)
    m += f(1.0 - beta1) * (g - m)                          # lerp; the weight is formed in double, as Python does
    v *= f(beta2)
    v += f(1.0 - beta2) * g * g                            # addcmul
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
    step_size = f(f(lr) / f(bc1))
    denom = np.sqrt(v) * f(1.0 / bc2_sqrt) + f(eps)
    p -= step_size * (m / denom)
    return p, m, v
