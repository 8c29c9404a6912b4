#!/usr/bin/env python3
"""Run-to-run determinism of one forward + gradient render (bunny, 16 sources): eager vs eager, and a captured HIP
graph's replay vs eager.  Rows and gradient must agree to fp64 summation order (~1e-13).
    python tools/determinism_check.py [iterations]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nlos_surface_optimization_amd import device as nd  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = np.load(os.path.join(ROOT, "tests", "golden", "bunny_5k.npz"))
dev = torch.device("cuda", 0)
g = np.linspace(-0.25, 0.25, 4)
o = torch.tensor([[x, y, 0] for y in g for x in g], dtype=torch.float32, device=dev)
n = torch.tensor([[0, 0, 1.0]] * 16, dtype=torch.float32, device=dev)
tv = torch.from_numpy(d["v"]).to(dev)
tf = torch.from_numpy(d["f"]).to(dev)
T = 512
data = torch.zeros((16, T), dtype=torch.float64, device=dev)
w = torch.ones_like(data)


def rel(a, b):
    return float((a - b).norm() / b.norm())


worst = {"eager_rows": 0.0, "eager_grad": 0.0, "accum_grad": 0.0}
r = nd.TransientRenderer(dev, seed=1)
t0, g0, _ = r.render_gradient(o, n, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9, data=data, weight=w)
for it in range(n_it):
    if it % 5 == 4:
        r = nd.TransientRenderer(dev, seed=1)          # fresh scratch (recycled, uninitialised device memory)
    junk = torch.empty(int(64e6 // 8), dtype=torch.float64, device=dev).normal_()     # dirty the allocator's pool
    del junk
    t1, g1, _ = r.render_gradient(o, n, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9, data=data, weight=w)
    acc = torch.zeros_like(g0)
    t2, g2, _ = r.render_gradient(o, n, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9, data=data, weight=w, gradient=acc)
    torch.cuda.synchronize()
    e = (rel(t1, t0), rel(g1, g0), rel(g2, g0))
    for k, v in zip(worst, e):
        worst[k] = max(worst[k], v)
    if max(e) > 1e-11:
        dg = (g1 - g0).abs()
        i = int(dg.argmax())
        print("iter %d: rows %.2e  grad(zeroed by the render) %.2e  grad(accumulated into zeros) %.2e | worst entry %d: %.6e vs %.6e; "
              "entries differing > 1e-12 rel: %d / %d" % (it, e[0], e[1], e[2], i, float(g1.flatten()[i]), float(g0.flatten()[i]),
                                                           int((dg > 1e-12 * g0.abs().max()).sum()), g0.numel()))
print("worst over %d iterations:" % n_it, worst)
sys.exit(1 if max(worst.values()) > 1e-11 else 0)
