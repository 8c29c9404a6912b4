#!/usr/bin/env python3
"""Re-run one case of tools/fuzz_parity.py and show, per occlusion back-end, where the rows differ from the oracle.
Usage: python tools/fuzz_case.py <case> <seed>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fz  # noqa: E402
import oracle as orc  # noqa: E402
from nlos_surface_optimization_amd import device as nd  # noqa: E402

case, seed = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(seed * 100003 + case)
while True:      # the sweep draws a case again while its oracle rows are all zero
    sc = fz.draw_scene(rs, case)
    v, f, o, nrm, ns, lb, ub, res, vn = (sc[k] for k in ("v", "f", "o", "nrm", "ns", "lb", "ub", "res", "vn"))
    F, L, spt, T, use_vn = f.shape[0], o.shape[0], sc["spt"], sc["T"], vn is not None
    if orc.render_transient(o, nrm, v, f, ns, lb, ub, res, accel=1, seed=case, vnormal=vn)[0].sum() > 0 or "--empty" in sys.argv:
        break
print("F", F, "L", L, "spt", spt, "ns", ns, "T", T, "res", res, "lb", lb, "vn", use_vn, "\no", o)
t_ref, _ = orc.render_transient(o, nrm, v, f, ns, lb, ub, res, accel=1, seed=case, vnormal=vn)
t_bf, _ = orc.render_transient(o, nrm, v, f, ns, lb, ub, res, accel=0, seed=case, vnormal=vn)
print("oracle bvh vs brute force:", np.abs(t_ref - t_bf).max())
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev, seed=case)
tv, tf_, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, nrm))
tvn = None if vn is None else torch.from_numpy(vn).to(dev)
out = {}
vis = {}
for fb in (0, 1, 2):
    t, _ = r.render_transient(to, tn, tv, tf_, ns, lb, ub, res, vertex_normal=tvn, force_bvh=fb, keep_visibility=True)
    vis[fb], fid = r.debug_visibility(L, 1 + (ns - 1) // F, F)
    out[fb] = t.cpu().numpy()
    d = np.abs(out[fb] - t_bf)
    i = np.unravel_index(d.argmax(), d.shape)
    print("force_bvh=%d: max |diff| %.3e at %s (ref %.6e gpu %.6e), bins differing > 1e-16*max: %d, row sums gpu-ref %s" % (
        fb, d.max(), i, t_bf[i], out[fb][i], (d > 1e-16 * t_bf.max()).sum(), (out[fb].sum(1) - t_bf.sum(1))))

# samples accepted by one back-end only (sorted-face order -> original face ids)
for fb in (0, 2):
    d = vis[fb] ^ vis[1]
    idx = np.argwhere(d != 0)
    print("force_bvh=%d vs 1: %d (source, word, face) entries differ" % (fb, len(idx)))
    for l, w, j in idx[:8]:
        bits = int(d[l, w, j])
        fo = int(fid[j])
        print("  source %d sorted face %d (original %d) sample bits %s  grid-accepted %s bvh-accepted %s" % (
            l, j, fo, bin(bits), bin(int(vis[fb][l, w, j])), bin(int(vis[1][l, w, j]))))
        p0, p1, p2 = v[f[fo]]
        print("    face vertices", p0, p1, p2)
        # who occludes?  brute-force closest hit of the ray towards the face centroid-ish sample
        for s_ in range(32):
            if bits >> s_ & 1:
                S, Tq = orc.sample(case, ((l * F + fo) * (1 + (ns - 1) // F)) + (w * 32 + s_))
                sq = np.sqrt(np.float32(Tq))
                pt = (1 - sq) * p0 + (1 - np.float32(S)) * sq * p1 + np.float32(S) * sq * p2
                dvec = (pt - o[l]).astype(np.float32)
                dvec /= np.linalg.norm(dvec)
                hit = orc.intersect(o[l][None], dvec[None], v, f, accel=0)
                print("    sample", s_, "point", pt, "dir", dvec, "closest hit face", hit[0, 0], "u,v", hit[0, 1:])
                fh = int(hit[0, 0])
                if fh >= 0:
                    print("    occluder vertices", v[f[fh]])
