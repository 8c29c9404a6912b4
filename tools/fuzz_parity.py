#!/usr/bin/env python3
"""Randomised parity sweep: GPU (all occlusion back-ends) vs the CPU oracle on random meshes, sources,
windows and sample counts.  The accept/reject decisions must be identical, so the forward rows agree
to fp64 summation order (~1e-15).  The sweep FAILS on any case above 1e-12, i.e. on a single differing sample
(with the grazing rule of DESIGN.md section 2 every back-end enumerates exactly the hits of the all-faces
definition); gradient 1e-4.  On meshes of up to 3000 faces the oracle's own BVH mode is also checked against
its brute-force mode (the definition).  Every logged case is a NON-EMPTY scene (empty draws are checked for all-zero
GPU rows and drawn again); 15 % of the cases have 64 ... 1024 sources.  Usage: python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402
from nlos_surface_optimization_amd import mesh_io, renderer  # noqa: E402


def rel_l2(a, b):
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))


def random_mesh(rs):
    kind = rs.randint(0, 4)
    if kind == 0:      # noisy height field (open sheet), wall-facing
        n = rs.randint(6, 60)
        xs, ys = np.meshgrid(np.linspace(-0.3, 0.3, n), np.linspace(-0.3, 0.3, n))
        z = 0.45 + 0.08 * np.sin(7 * xs + rs.rand()) * np.cos(5 * ys) + rs.normal(0, rs.choice([0.0, 0.004, 0.02]), xs.shape)
        v = np.stack([xs.ravel(), ys.ravel(), z.ravel()], 1)
        idx = np.arange(n * n).reshape(n, n)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
        f = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)])
    elif kind == 1:    # bunny, randomly scaled / shifted / subdivided
        d = np.load(os.path.join(ROOT, "tests", "golden", "bunny_5k.npz"))
        v, f = d["v"].astype(np.float64), d["f"]
        if rs.rand() < 0.3:
            keep = rs.rand(f.shape[0]) < rs.uniform(0.05, 0.9)
            f = f[keep]
        v = (v - v.mean(0)) * rs.uniform(0.5, 1.5) + np.array([rs.uniform(-0.1, 0.1), rs.uniform(-0.1, 0.1), rs.uniform(0.4, 0.6)])
        if rs.rand() < 0.25:
            v, f = mesh_io.subdivide(v.astype(np.float32), f, 1)          # ~20k faces: tiled grid
    elif kind == 2:    # triangle soup: random sizes and orientations, heavy occlusion
        m = rs.randint(64, 3000)
        c = np.stack([rs.uniform(-0.3, 0.3, m), rs.uniform(-0.3, 0.3, m), rs.uniform(0.3, 0.7, m)], 1)
        s = rs.choice([0.01, 0.05, 0.2]) * rs.rand(m, 1, 1)
        tri = c[:, None, :] + s * rs.normal(size=(m, 3, 3))
        v = tri.reshape(-1, 3)
        f = np.arange(3 * m).reshape(m, 3)
    else:              # sphere with an inner sphere (closed surfaces, depth layers)
        def sph(r, cz, n):
            th, ph = np.meshgrid(np.linspace(0.05, np.pi - 0.05, n), np.linspace(0, 2 * np.pi, 2 * n, endpoint=False), indexing="ij")
            vv = np.stack([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), cz + r * np.cos(th)], -1).reshape(-1, 3)
            idx = np.arange(n * 2 * n).reshape(n, 2 * n)
            a, b = idx[:-1, :], np.roll(idx[:-1, :], -1, 1)
            c, d = np.roll(idx[1:, :], -1, 1), idx[1:, :]
            ff = np.concatenate([np.stack([a.ravel(), b.ravel(), c.ravel()], 1), np.stack([a.ravel(), c.ravel(), d.ravel()], 1)])
            return vv, ff
        n = rs.randint(5, 30)
        v1, f1 = sph(0.15, 0.5, n)
        v2, f2 = sph(0.07, 0.42, max(4, n // 2))
        v = np.concatenate([v1, v2])
        f = np.concatenate([f1, f2 + v1.shape[0]])
    if rs.rand() < 0.5:
        f = f[:, [0, 2, 1]]
    return np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int32)


def draw_scene(rs, case):
    """One random scene: mesh, sources, window, sample count.  15 % of the scenes are MANY-source scenes (L in
    [64, 1024], regular grid or scattered, F <= 5000, a short window): thousands of workgroups in flight -- tickets,
    the need_tree flag raised by many sources at once, big-LDS relaunches, persistent gradient workgroups striding sources."""
    many = rs.rand() < 0.15
    while True:
        v, f = random_mesh(rs)
        if not many or f.shape[0] <= 5000:
            break
    F = f.shape[0]
    if many:
        L = int(rs.randint(64, 1025))
        if rs.rand() < 0.5:
            n = int(np.ceil(np.sqrt(L)))
            gx, gy = np.meshgrid(np.linspace(-0.5, 0.5, n), np.linspace(-0.5, 0.5, n))
            o = np.zeros((n * n, 3), np.float32)
            o[:, 0], o[:, 1] = gx.ravel(), gy.ravel()
            o = np.ascontiguousarray(o[:L])
        else:
            o = np.zeros((L, 3), np.float32)
            o[:, :2] = rs.uniform(-0.6, 0.6, (L, 2))
    else:
        L = int(rs.randint(1, 6))
        o = np.zeros((L, 3), np.float32)
        o[:, :2] = rs.uniform(-0.6, 0.6, (L, 2))
    if rs.rand() < 0.15:
        o[rs.randint(0, L), 2] = rs.uniform(0.3, 0.6)            # a wall point inside the scene's depth range
    nrm = np.tile(np.array([0, 0, 1], np.float32), (L, 1))
    spt = int(rs.choice([1, 2, 5] if many else [1, 2, 5, 9, 33]))
    ns = max(spt * F - rs.randint(0, F), 1)
    T = int(rs.choice([64, 128] if many else [64, 512, 1000]))
    res = float(np.float32(rs.choice([2.0 ** -7, 5e-3, 2.0 ** -5] if many else [2.0 ** -9, 2.0 ** -7, 1.2e-3, 5e-3])))
    # mostly windows that contain (part of) the object, sometimes one that misses it
    dmin = 2 * float(np.min(np.linalg.norm(v[None, ::7, :] - o[::max(1, L // 8), None, :], axis=2)))
    lb = float(np.float32(max(0.0, dmin + rs.uniform(-0.5, 0.3) * T * res))) if rs.rand() < 0.85 else float(np.float32(rs.uniform(0.0, 0.8)))
    ub = float(np.float32(np.float32(lb) + np.float32(T) * np.float32(res)))
    vn = None
    if rs.rand() < 0.25:
        from conftest import vertex_normals
        vn = vertex_normals(v, f)
    return dict(v=v, f=f, o=o, nrm=nrm, ns=int(ns), spt=spt, T=T, lb=lb, ub=ub, res=res, vn=vn, many=many)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import torch
    from nlos_surface_optimization_amd import device as nd
    dev = torch.device("cuda", 0)
    worst_t, worst_g, bad, differing, empties, empties_bad = 0.0, 0.0, 0, 0, 0, 0
    for case in range(n_cases):
        rs = np.random.RandomState(seed * 100003 + case)
        # A scene whose oracle rows are all zero (window misses the object, mesh wound away from the wall) passes
        # vacuously: the GPU rows are checked to be all zero as well (a non-zero one IS a mismatch), the draw is counted
        # under "empty draws" and the case is drawn again -- every logged case is a non-empty scene.
        while True:
            sc = draw_scene(rs, case)
            v, f, o, nrm, ns, lb, ub, res, vn = (sc[k] for k in ("v", "f", "o", "nrm", "ns", "lb", "ub", "res", "vn"))
            F, L, spt, T, use_vn = f.shape[0], o.shape[0], sc["spt"], sc["T"], vn is not None
            kw = dict(accel=1, seed=case, vnormal=vn)
            t_ref, _ = orc.render_transient(o, nrm, v, f, ns, lb, ub, res, **kw)
            r = nd.TransientRenderer(dev, seed=case)
            tv, tf_, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, nrm))
            tvn = None if vn is None else torch.from_numpy(vn).to(dev)
            if t_ref.sum() > 0:
                break
            empties += 1
            for fb in (0, 1):
                t, _ = r.render_transient(to, tn, tv, tf_, ns, lb, ub, res, vertex_normal=tvn, force_bvh=fb)
                if float(t.abs().sum()) != 0.0:
                    empties_bad += 1
                    print("case %3d: EMPTY scene (F=%d L=%d) but the GPU rows are not zero (force_bvh=%d)  <-- MISMATCH" % (case, F, L, fb), flush=True)
            r.close()
        e_def = 0.0
        if F <= 3000 and L <= 8:
            # the definition: closest hit over ALL faces (no acceleration structure)
            t_def, _ = orc.render_transient(o, nrm, v, f, ns, lb, ub, res, accel=0, seed=case, vnormal=vn)
            e_def = rel_l2(t_ref, t_def)
        errs = []
        for fb in (0, 1) + ((2,) if F > 6200 else ()):
            t, _ = r.render_transient(to, tn, tv, tf_, ns, lb, ub, res, vertex_normal=tvn, force_bvh=fb)
            errs.append(rel_l2(t.cpu().numpy(), t_ref))
        et = max(errs + [e_def])
        eg = 0.0
        if rs.rand() < 0.6:
            data = t_ref * (1 + 0.3 * rs.standard_normal(t_ref.shape))
            w = 0.5 + rs.random_sample(t_ref.shape)
            sb = int(rs.choice([1, 1, 2, 5]))
            rf = int(rs.choice([4, 10]))
            tf0 = int(rs.randint(0, 2))
            _, g_ref, _ = orc.render_gradient(o, nrm, v, f, ns, lb, ub, res, data, w, refine=rf, sigma_bin=sb,
                                              testing_flag=tf0, **kw)
            _, g, _ = r.render_gradient(to, tn, tv, tf_, ns, lb, ub, res, data=torch.from_numpy(data).to(dev),
                                        weight=torch.from_numpy(w).to(dev), refine_scale=rf, sigma_bin=sb,
                                        testing_flag=tf0, vertex_normal=tvn)
            eg = rel_l2(g.cpu().numpy(), g_ref)
        en = 0.0
        if rs.rand() < 0.35 and not use_vn:
            # row N: random sensor point per laser, both back-ends
            b = o.copy()
            b[:, :2] += rs.uniform(-0.3, 0.3, (L, 2)).astype(np.float32)
            tn_ref, _, _ = orc.render_nonconfocal(o, nrm, b, nrm, v, f, ns, lb, ub, res, refine=1, accel=1, seed=case)
            tb = torch.from_numpy(b).to(dev)
            for fb in (0, 1):
                t, _ = r.render_transient(to, tn, tv, tf_, ns, lb, ub, res, sensor=tb, sensor_normal=tn, force_bvh=fb)
                en = max(en, rel_l2(t.cpu().numpy(), tn_ref))
            et = max(et, en)
        ep = 0.0
        if rs.rand() < 0.3:
            # row N as a product (round 4): a few lasers x a few sensors on shared samples -- the record + combine kernels
            # where the scene allows them, the enumerated pairs otherwise -- against the pair oracle on the enumerated pairs
            La, Sb = int(rs.randint(1, 5)), int(rs.randint(1, 5))
            pl = np.ascontiguousarray(o[rs.randint(0, L, La)])
            ps = np.zeros((Sb, 3), np.float32)
            ps[:, :2] = rs.uniform(-0.6, 0.6, (Sb, 2))
            if rs.rand() < 0.3:
                ps[0] = pl[0]                              # a wall point that is laser and sensor
            pn, sn = np.tile(np.array([0, 0, 1], np.float32), (La, 1)), np.tile(np.array([0, 0, 1], np.float32), (Sb, 1))
            lbp = float(np.float32(max(0.0, lb - 0.1)))
            ubp = float(np.float32(np.float32(lbp) + np.float32(T) * np.float32(res)))
            tp_ref, _, _ = orc.render_product(pl, pn, ps, sn, v, f, ns, lbp, ubp, res, accel=1, seed=case)
            tpl, tpn, tps, tsn = (torch.from_numpy(x).to(dev) for x in (pl, pn, ps, sn))
            tp, _, _ = r.render_product(tpl, tpn, tps, tsn, tv, tf_, ns, lbp, ubp, res)
            ep = rel_l2(tp.cpu().numpy(), tp_ref)
            if tp_ref.sum() > 0 and rs.rand() < 0.5:
                dp = tp_ref * (1 + 0.3 * rs.standard_normal(tp_ref.shape))
                _, gp_ref, _ = orc.render_product(pl, pn, ps, sn, v, f, ns, lbp, ubp, res, data=dp, accel=1, seed=case)
                _, gp, _ = r.render_product(tpl, tpn, tps, tsn, tv, tf_, ns, lbp, ubp, res, data=torch.from_numpy(dp).to(dev))
                eg = max(eg, rel_l2(gp.cpu().numpy(), gp_ref))
            et = max(et, ep)
        ex = 0.0
        if rs.rand() < 0.3 and F <= 6200:
            # other rows on the same scene: GGX branch, SPAD jitter gradient, v1 driver, per-face intensity
            alpha = float(rs.uniform(0.05, 0.9))
            tg_ref, _ = orc.render_transient(o, nrm, v, f, ns, lb, ub, res, ggx_alpha=alpha, vnormal=vn, accel=1, seed=case)
            tg, _ = r.render_transient(to, tn, tv, tf_, ns, lb, ub, res, alpha=alpha, vertex_normal=tvn)
            ex = max(ex, rel_l2(tg.cpu().numpy(), tg_ref))
            jk = np.load(os.path.join(ROOT, "tests", "golden", "jitter_info.npz"))
            jw, jg, jo = jk["jitter_weight"], jk["jitter_grad"], int(jk["jitter_offset"])
            tj_ref, _, _ = orc.render_jitter(o, nrm, v, f, ns, lb, ub, res, jw, jo, accel=1, seed=case)
            dj = tj_ref * 1.25
            _, gj_ref, _ = orc.render_jitter(o, nrm, v, f, ns, lb, ub, res, jw, jo, jitter_grad=jg, data=dj,
                                             weight=np.ones_like(dj), testing_flag=1, accel=1, seed=case)
            tjw = torch.from_numpy(np.ascontiguousarray(jw.ravel())).to(dev)
            tjg = torch.from_numpy(np.ascontiguousarray(jg.ravel())).to(dev)
            tj, gj, _ = r.render_gradient(to, tn, tv, tf_, ns, lb, ub, res, data=torch.from_numpy(dj).to(dev),
                                          weight=torch.ones(dj.shape, dtype=torch.float64, device=dev),
                                          jitter_weight=tjw, jitter_grad=tjg, jitter_offset=jo, testing_flag=1)
            ex = max(ex, rel_l2(tj.cpu().numpy(), tj_ref))
            eg = max(eg, rel_l2(gj.cpu().numpy(), gj_ref))
            it_ref = orc.render_intensity(o, nrm, v, f, ns, lb, ub, accel=1, seed=case)
            it = r.render_intensity(to, tn, tv, tf_, ns, lb, ub)
            ex = max(ex, rel_l2(it.cpu().numpy(), it_ref))
            et = max(et, ex)
        r.close()
        # rel_l2 against an all-zero reference is the absolute norm: a non-zero GPU result against a zero oracle
        # result fails like any other difference
        ok = et <= 1e-12 and eg <= 1e-4         # 1e-12: fp64 summation order only -- a single differing sample fails
        differing += int(et > 1e-12)
        bad += int(not ok)
        worst_t, worst_g = max(worst_t, et), max(worst_g, eg)
        print("case %3d F=%6d L=%4d spt=%2d T=%4d vn=%d sum=%.3e  transient %.2e (grid/bvh%s, oracle bvh vs all-faces %.1e)  gradient %.2e  nc %.2e  product %.2e  ggx/jitter/intensity %.2e %s" % (
            case, F, L, spt, T, int(use_vn), t_ref.sum(), et, "/ovf" if F > 6200 else "", e_def, eg, en, ep, ex, "" if ok else "  <-- MISMATCH"),
            flush=True)
    print("empty draws (resampled, GPU rows checked to be zero): %d, of which non-zero on the GPU: %d" % (empties, empties_bad))
    print("cases with a differing sample (see tools/fuzz_case.py): %d" % (differing + empties_bad))
    print("worst transient %.3e, worst gradient %.3e, mismatches %d / %d" % (worst_t, worst_g, bad + empties_bad, n_cases))
    return 1 if (bad or empties_bad or differing) else 0


if __name__ == "__main__":
    sys.exit(main())
