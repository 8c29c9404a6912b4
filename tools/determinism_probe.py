#!/usr/bin/env python3
"""Where does the run-to-run difference of the vertex gradient come from?  Renders the same forward + gradient many
times and compares, against the first run: the rows (bitwise), the visibility cache (bitwise), the gradient."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nlos_surface_optimization_amd import device as nd  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = np.load(os.path.join(ROOT, "tests", "golden", "bunny_5k.npz"))
dev = torch.device("cuda", 0)
g = np.linspace(-0.25, 0.25, 4)
o = torch.tensor([[x, y, 0] for y in g for x in g], dtype=torch.float32, device=dev)
n = torch.tensor([[0, 0, 1.0]] * 16, dtype=torch.float32, device=dev)
tv = torch.from_numpy(d["v"]).to(dev)
tf = torch.from_numpy(d["f"]).to(dev)
F = d["f"].shape[0]
data = torch.zeros((16, 512), dtype=torch.float64, device=dev)
w = torch.ones_like(data)
r = nd.TransientRenderer(dev, seed=1)
ref = None
for it in range(n_it):
    t, gr, _ = r.render_gradient(o, n, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9, data=data, weight=w)
    torch.cuda.synchronize()
    vis, fid = r.debug_visibility(16, 5, F)
    cur = (t.cpu().numpy(), vis.copy(), gr.cpu().numpy())
    if ref is None:
        ref = cur
        continue
    rows_same = np.array_equal(cur[0], ref[0])
    nb = int((cur[0] != ref[0]).sum())
    vis_diff = np.argwhere(cur[1] != ref[1])
    gd = np.abs(cur[2] - ref[2])
    rel = float(np.linalg.norm(cur[2] - ref[2]) / np.linalg.norm(ref[2]))
    if rel > 1e-12 or len(vis_diff):
        verts = np.unique(np.argwhere(gd > 1e-12 * np.abs(ref[2]).max())[:, 0])
        print("iter %d: gradient rel %.2e | rows bitwise equal: %s (%d bins differ, max rel %.1e) | visibility words differing: %d %s | vertices touched: %d %s"
              % (it, rel, rows_same, nb, float(np.abs(cur[0] - ref[0]).max() / ref[0].max()), len(vis_diff), vis_diff[:4].tolist(), len(verts), verts[:8].tolist()))
        if nb:
            idx = np.argwhere(cur[0] != ref[0])
            l, b = idx[0]
            print("    first differing bin: source %d bin %d: %.17g vs %.17g; as float(-2d): %r vs %r" % (
                l, b, cur[0][l, b], ref[0][l, b], np.float32(2 * cur[0][l, b]), np.float32(2 * ref[0][l, b])))
            fl = np.float32(2 * cur[0]) != np.float32(2 * ref[0])
            print("    bins whose float(2 t) differs: %d %s" % (int(fl.sum()), np.argwhere(fl)[:4].tolist()))
print("done")
