#!/usr/bin/env python3
"""Side measurement (not the metric): row N as a PRODUCT, n x n lasers times n x n sensors on the benchmark mesh --
the record + combine kernels against the same measurements rendered as enumerated pairs (two grid passes per pair).
A sample of the pairs is checked against the CPU oracle before anything is timed.
    python tools/product_bench.py [n=8] [bins=512] [steps=20] [shading=0]      ->  one JSON line
shading=1: shading normals (perturbed vertex normals) + per-vertex albedo -- the extended records of round 6."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker of the gate below, never the thing timed)
from nlos_surface_optimization_amd import device as nd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    shading = len(sys.argv) > 4 and int(sys.argv[4]) != 0
    d = np.load(os.path.join(ROOT, "tests", "golden", "bunny_5k.npz"))
    v, f = np.ascontiguousarray(d["v"], np.float32), np.ascontiguousarray(d["f"], np.int32)
    g = np.linspace(-0.25, 0.25, n)
    la = np.array([[x, y, 0] for y in g for x in g], np.float32)
    sb = la + np.array([0.5 / (2 * max(n - 1, 1)), 0.5 / (2 * max(n - 1, 1)), 0], np.float32)      # the sensors sit between the lasers
    nl, ns_ = np.tile(np.array([0, 0, 1], np.float32), (la.shape[0], 1)), np.tile(np.array([0, 0, 1], np.float32), (sb.shape[0], 1))
    lb, ub, res, num_sample = 0.625, 1.625, 1.0 / T, 20000
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev, seed=0)
    r.enable_timing(True)
    tl, tln, ts, tsn, tv, tf = (torch.from_numpy(x).to(dev) for x in (la, nl, sb, ns_, v, f))
    okw, gkw = {}, {}
    if shading:
        p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        fn = np.cross(p1 - p0, p2 - p0).astype(np.float64)
        vn = np.zeros((v.shape[0], 3))
        for k in range(3):
            np.add.at(vn, f[:, k], fn)
        rs0 = np.random.RandomState(8)
        vn = vn / np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30) + 0.1 * rs0.standard_normal(vn.shape)
        vn = np.ascontiguousarray(vn / np.linalg.norm(vn, axis=1, keepdims=True), np.float32)
        alb = np.ascontiguousarray(0.3 + 0.7 * rs0.random_sample(v.shape[0]), np.float32)
        okw = dict(vnormal=vn, albedo=alb)
        gkw = dict(vertex_normal=torch.from_numpy(vn).to(dev), albedo=torch.from_numpy(alb).to(dev))
    L, S, F = la.shape[0], sb.shape[0], f.shape[0]
    spt = 1 + (num_sample - 1) // F
    data, _, _ = r.render_product(tl, tln, ts, tsn, tv, tf, num_sample, lb, ub, res, seed=1, **gkw)
    data = data * 1.1
    # gate: 16 of the pairs, rows and (on those pairs alone) the gradient, against the oracle
    rs = np.random.RandomState(0)
    li, sj = rs.randint(0, L, 16), rs.randint(0, S, 16)
    t_gpu, g_all, _ = r.render_product(tl, tln, ts, tsn, tv, tf, num_sample, lb, ub, res, data=data, **gkw)
    t_ref, _, _ = oracle.render_nonconfocal(la[li], nl[li], sb[sj], ns_[sj], v, f, num_sample, lb, ub, res, refine=1, accel=1,
                                            seed=0, shared_samples=1, **okw)
    e_rows = float(np.linalg.norm(t_gpu.cpu().numpy()[li, sj] - t_ref) / np.linalg.norm(t_ref))
    sub_l, sub_s = tl[:2].contiguous(), ts[:3].contiguous()
    _, g_gpu, _ = r.render_product(sub_l, tln[:2].contiguous(), sub_s, tsn[:3].contiguous(), tv, tf, num_sample, lb, ub, res,
                                   data=data[:2, :3].contiguous(), **gkw)
    _, g_ref, _ = oracle.render_product(la[:2], nl[:2], sb[:3], ns_[:3], v, f, num_sample, lb, ub, res,
                                        data=data[:2, :3].cpu().numpy(), accel=1, seed=0, **okw)
    e_grad = float(np.linalg.norm(g_gpu.cpu().numpy() - g_ref) / np.linalg.norm(g_ref))
    if not (e_rows <= 1e-5 and e_grad <= 1e-4):
        sys.exit("product_bench: PARITY GATE FAILED rows %.3e gradient %.3e" % (e_rows, e_grad))
    grad = torch.zeros((v.shape[0], 3), dtype=torch.float64, device=dev)
    out = {"workload": "SIDE MEASUREMENT (row N as a product, not the metric): %d lasers x %d sensors x %d bins, bunny_5k F=%d, spt=%d, "
                       "forward + vertex gradient, scene rebuilt every step%s" % (L, S, T, F, spt, ", SHADING NORMALS + ALBEDO" if shading else ""),
           "pair_samples_per_step": L * S * F * spt, "parity": {"rows_rel_l2_16_pairs": e_rows, "gradient_rel_l2_2x3": e_grad}}
    for name, pairs in (("record_and_combine", False), ("enumerated_pairs", True)):
        def step():
            r.render_product(tl, tln, ts, tsn, tv, tf, num_sample, lb, ub, res, data=data, gradient=grad, zero_gradient=True, pairs=pairs, **gkw)
        # (pre-warm: the GPU idled while the CPU oracle ran the gate and its clocks take tens of milliseconds to come back --
        # without this the first variant's 20 - 30 ms of timed steps read anything between 1.0 and 4.3 ms per step)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < 0.5:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        r.timing_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        kt = r.timing_mean_ms()[0]
        out[name] = {"ms_per_step": ms, "kernel_ms": dict(zip(["bvh_build", "forward", "residual", "gradient"], [float(x) for x in kt])),
                     "path": r.last_path(), "G_pair_samples_per_s": L * S * F * spt / ms / 1e6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
