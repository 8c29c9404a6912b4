"""numpy restatement of the host-side steps that follow the render in the reference's loops --
TEST INFRASTRUCTURE (checker for the device kernels), not product code.

  adam_modified_step        exp_bunny/adam_modified.py:62-107 (float32 state, like torch)
  create_weighting_function exp_bunny/rendering.py:208-217
  weighted_l2               exp_bunny/rendering.py:360-364

Pinned against the reference's own optimiser class: tests/golden/adam_modified.npz holds
trajectories produced by importing exp_bunny/adam_modified.py in the build container
(tests/golden/make_golden.py: make_adam_modified).
"""
import math

import numpy as np


class AdamModifiedState:
    def __init__(self, shape, amsgrad=False):
        self.step = 0
        self.exp_avg = np.zeros(shape, np.float32)
        self.exp_avg_sq = np.zeros(shape, np.float32)
        self.max_exp_avg_sq = np.zeros(shape, np.float32) if amsgrad else None


def adam_modified_step(p, grad, st, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One step on float32 `p` [rows, cols] in place; `grad` float32/float64 (narrowed first)."""
    f = np.float32
    g = grad.astype(np.float32)
    b1, b2 = betas
    st.step += 1
    if weight_decay != 0:
        g = g + f(weight_decay) * p                                   # adam_modified.py:83-84
    st.exp_avg[...] = st.exp_avg * f(b1) + f(1 - b1) * g              # :87
    st.exp_avg_sq[...] = st.exp_avg_sq * f(b2) + f(1 - b2) * g * g    # :88
    if st.max_exp_avg_sq is not None:
        np.maximum(st.max_exp_avg_sq, st.exp_avg_sq, out=st.max_exp_avg_sq)   # :91
        denom = np.sqrt(st.max_exp_avg_sq) + f(eps)                   # :93
    else:
        denom = np.sqrt(st.exp_avg_sq) + f(eps)                       # :95
    new_denom = denom.mean(axis=1, keepdims=True, dtype=np.float32)   # :96
    bc1 = 1 - b1 ** st.step
    bc2 = 1 - b2 ** st.step
    step_size = lr * math.sqrt(bc2) / bc1                             # :102-104
    p += f(-step_size) * (st.exp_avg / new_denom)                     # :105
    return p


def create_weighting_function(data, gamma=1):
    eps = 0.1
    i_max = np.max(data)
    weight = (data / i_max + eps) ** gamma
    weight = weight / np.sum(weight)
    return weight * (data.shape[0] * data.shape[1])


def weighted_l2(transient, data, weight=None):
    d = (transient - data) * (1.0 if weight is None else np.sqrt(weight))
    return np.linalg.norm(d) ** 2 / d.shape[0]
