#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the BUILD container only).

Reads /root/reference (read-only, absent on the GPU box) and writes DATA only:
  bunny_5k.npz        decimated exp_bunny/gt_bunny.obj (~5k faces), the cfg2/3 mesh
  mannequin.npz       exp_mannequin/cnlos_mannequin_threshold.obj (620 v / 1055 f), cfg4 mesh
  mannequin_measurement.npz  exp_mannequin/transient.mat: the 4096 wall points and the measured photon counts, folded to
                      1024 bins (cfg4's real inputs)
  pyref_angular.npz   inputs + outputs of the reference's numpy prototype
                      transient_rendering_python/rendering.py:angular_sampling (imported from
                      /root/reference) on the cfg1 plane and on the toy mesh of
                      transient_rendering_python/test_autograd.py:35-36 -- pins the oracle's
                      closest-hit / distance / binning primitives against reference code
  pyref_angular_nc.npz  the same prototype with lighting != sensor (row N, non-confocal pairs), incl. a
                      blocker that hides paths from one end point only
  pyref_radiometry.npz  the same prototype with 2 000 000 hemisphere directions per source on wall-parallel patches, in 40
                      batches: the statistical pin of the v2 radiometry (A ff^2 / spt, 1 / h^4, binning origin)
  jitter_info.npz     jitter/jitter_info.mat (the reference's measured SPAD jitter kernel) as npz
  adam_modified.npz   parameter trajectories of the reference's own optimiser class
                      (exp_bunny/adam_modified.py, imported and run on CPU) on fixed gradients
  oracle_cfg1.npz     oracle transient + gradient for BASELINE config 1 (regression pin)
  oracle_bunny16.npz  oracle transient + gradient, bunny_5k, 16 sources (regression pin), plus the same render under
                      the reference's rule-free hit test (all faces, no grazing rule): the contract may not drift from it
  ggx_table.npz       oracle GGX eval / eval_adiff / eval_nwdiff over an (alpha, n.w) grid

No reference source text is copied: the prototype is imported and executed, and only its
numeric inputs/outputs are stored.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from nlos_surface_optimization_amd import mesh_io  # noqa: E402
import oracle as orc  # noqa: E402


def cfg1():
    v = np.array([[-.25, -.25, .38], [.25, -.25, .38], [.25, .25, .38], [-.25, .25, .38]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2]], np.int32)
    g = np.linspace(-.25, .25, 2)
    origin = np.array([[x, y, 0] for y in g for x in g], np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (4, 1))
    return v, f, origin, normal


def grid_sources(n, half):
    g = np.linspace(-half, half, n)
    origin = np.array([[x, y, 0] for y in g for x in g], np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (origin.shape[0], 1))
    return origin, normal


def make_meshes():
    v, f = mesh_io.read_obj(os.path.join(REF, "transient_rendering_cython/exp_bunny/gt_bunny.obj"))
    nv, nf = mesh_io.decimate_to(v, f, 5000)
    np.savez_compressed(os.path.join(HERE, "bunny_5k.npz"), v=nv, f=nf)
    print("bunny_5k", nv.shape, nf.shape)
    mv, mf = mesh_io.read_obj(os.path.join(REF, "transient_rendering_cython/exp_mannequin/cnlos_mannequin_threshold.obj"))
    np.savez_compressed(os.path.join(HERE, "mannequin.npz"), v=mv, f=mf)
    print("mannequin", mv.shape, mf.shape)
    return nv, nf


def make_mannequin_measurement():
    """BASELINE configuration 4's real inputs: the reference's measured mannequin transient
    (exp_mannequin/transient.mat: `lighting` [4096, 3] on +-0.35 m, uint8 photon counts [4096, 2048] at 1.2 mm) read
    with the package's own reader and folded pairwise to 1024 bins of 2.4 mm, as exp_s/test.py:20-36 consumes such
    files.  Data only."""
    m = mesh_io.read_transient_mat(os.path.join(REF, "transient_rendering_cython/exp_mannequin/transient.mat"), fold=2)
    counts = m["transient"]
    assert counts.shape == (4096, 1024) and counts.max() < 256 and np.array_equal(counts, np.round(counts))
    np.savez_compressed(os.path.join(HERE, "mannequin_measurement.npz"), lighting=m["lighting"],
                        counts=counts.astype(np.uint8), lb=np.float32(0.0), res=np.float32(2.4e-3))
    print("mannequin measurement", counts.shape, "photons", int(counts.sum()), "max", int(counts.max()))


def make_pyref():
    """Run the reference's numpy prototype (imported, not copied) on fixed inputs."""
    sys.path.insert(0, os.path.join(REF, "transient_rendering_python"))
    import rendering as pyref  # noqa: E402  (the reference module)
    import mesh_intersection as pyint  # noqa: E402  (the reference module)

    out = {}
    rs = np.random.RandomState(0)
    cases = {}
    v, f, _, _ = cfg1()
    cases["plane"] = (v.astype(np.float64), f.astype(np.int64))
    tv = np.array([[-1, -1, .9], [1, -1, 1], [1, 1, 1.2], [-1, 1, 1], [-2, -2, 1], [2, 1, 1]], np.float64)
    tf = np.array([[0, 2, 1], [0, 3, 2], [4, 3, 0], [1, 2, 5]], np.int64)
    cases["toy"] = (tv, tf)
    for name, (mv, mf) in cases.items():
        n = 256
        # directions on the upper hemisphere (z > 0), fixed seed
        d = rs.normal(size=(n, 3))
        d[:, 2] = np.abs(d[:, 2]) + 0.2
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pairs = [((0.1, 0, 0), (0.1, 0, 0)), ((-0.1, 0, 0), (-0.1, 0, 0)),
                 ((0, 0.1, 0), (0, 0.1, 0)), ((0, -0.1, 0), (0, -0.1, 0))]
        mesh = types.SimpleNamespace()
        mesh.v, mesh.f = mv, mf
        p1, p2, p3 = mv[mf[:, 0]], mv[mf[:, 1]], mv[mf[:, 2]]
        fn = np.cross(p2 - p1, p3 - p1)
        fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        mesh.fn = fn
        opt = types.SimpleNamespace()
        opt.sample_num = n
        opt.max_distance_bin = 64 if name == "plane" else 400
        opt.distance_resolution = 0.02
        opt.epsilon = 1e-9
        opt.normal = "fn"
        res, prims, tnear = [], [], []
        for lighting, sensor in pairs:
            lighting = np.array(lighting, np.float64)
            sensor = np.array(sensor, np.float64)
            t = pyref.angular_sampling(mesh, d, lighting, sensor, np.array([0, 0, 1.0]), np.array([0, 0, 1.0]), opt)
            res.append(np.array(t))
            # per-ray nearest hit as the prototype selects it (rendering.py:37-52)
            hit, tt, _, _ = pyint.intersect_ray_mesh_batch_directions(lighting, d, mesh, opt.epsilon)
            tabs = np.where(hit, np.abs(tt), np.inf)
            pr = np.where(hit.any(axis=0), np.argmin(tabs, axis=0), -1)
            prims.append(pr)
            tnear.append(np.where(pr >= 0, tabs.min(axis=0), np.nan))
        out[name + "_prim"] = np.stack(prims)
        out[name + "_tnear"] = np.stack(tnear)
        out[name + "_fn"] = fn
        out[name + "_v"] = mv
        out[name + "_f"] = mf
        out[name + "_dir"] = d
        out[name + "_pairs"] = np.array([p[0] for p in pairs], np.float64)
        out[name + "_nbin"] = np.int64(opt.max_distance_bin)
        out[name + "_res"] = np.float64(opt.distance_resolution)
        out[name + "_transient"] = np.stack(res)
        print("pyref", name, np.stack(res).sum(axis=1))
    np.savez_compressed(os.path.join(HERE, "pyref_angular.npz"), **out)


def make_pyref_nc():
    """Reference numpy prototype with lighting != sensor (non-confocal pairs), data only."""
    sys.path.insert(0, os.path.join(REF, "transient_rendering_python"))
    import rendering as pyref  # noqa: E402  (the reference module)

    out = {}
    rs = np.random.RandomState(11)
    tv = np.array([[-1, -1, .9], [1, -1, 1], [1, 1, 1.2], [-1, 1, 1], [-2, -2, 1], [2, 1, 1]], np.float64)
    tf = np.array([[0, 2, 1], [0, 3, 2], [4, 3, 0], [1, 2, 5]], np.int64)
    # a wall-facing plane with a small blocker floating in front of it: some paths are visible from the
    # laser but hidden from the sensor and vice versa
    ov = np.array([[-1, -1, 1.0], [1, -1, 1.0], [1, 1, 1.0], [-1, 1, 1.0],
                   [-.15, -.2, .55], [.25, -.1, .5], [.05, .3, .6]], np.float64)
    of = np.array([[0, 2, 1], [0, 3, 2], [4, 6, 5]], np.int64)
    cases = {"toy": (tv, tf), "occluder": (ov, of)}
    pairs = [((0.1, 0, 0), (-0.3, 0.2, 0)), ((-0.4, -0.1, 0), (0.35, 0.3, 0)),
             ((0, 0.5, 0), (0, -0.5, 0)), ((0.2, -0.3, 0), (0.2, 0.3, 0))]
    for name, (mv, mf) in cases.items():
        n = 512
        d = rs.normal(size=(n, 3))
        d[:, 2] = np.abs(d[:, 2]) + 0.4
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        mesh = types.SimpleNamespace()
        mesh.v, mesh.f = mv, mf
        p1, p2, p3 = mv[mf[:, 0]], mv[mf[:, 1]], mv[mf[:, 2]]
        fn = np.cross(p2 - p1, p3 - p1)
        fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        mesh.fn = fn
        opt = types.SimpleNamespace()
        opt.sample_num = n
        opt.max_distance_bin = 400
        opt.distance_resolution = 0.02
        opt.epsilon = 1e-9
        opt.normal = "fn"
        res = []
        for lighting, sensor in pairs:
            t = pyref.angular_sampling(mesh, d, np.array(lighting, np.float64), np.array(sensor, np.float64),
                                       np.array([0, 0, 1.0]), np.array([0, 0, 1.0]), opt)
            res.append(np.array(t))
        out[name + "_fn"] = fn
        out[name + "_v"] = mv
        out[name + "_f"] = mf
        out[name + "_dir"] = d
        out[name + "_laser"] = np.array([p[0] for p in pairs], np.float64)
        out[name + "_sensor"] = np.array([p[1] for p in pairs], np.float64)
        out[name + "_nbin"] = np.int64(opt.max_distance_bin)
        out[name + "_res"] = np.float64(opt.distance_resolution)
        out[name + "_transient"] = np.stack(res)
        print("pyref nc", name, np.stack(res).sum(axis=1))
    np.savez_compressed(os.path.join(HERE, "pyref_angular_nc.npz"), **out)


def make_pyref_grad():
    """Central finite differences of the reference's numpy forward (rendering.py, imported) with respect to every
    vertex coordinate, of the functional sum_b angular_transient[b] (what `backward(ones)` of the torch prototype
    differentiates), on the test_autograd.py:35-36 toy mesh and the cfg-1 plane.  Face normals are recomputed
    from the displaced vertices, as rendering_grad.py:100-107 does."""
    sys.path.insert(0, os.path.join(REF, "transient_rendering_python"))
    import rendering as pyref  # noqa: E402  (the reference module)

    def forward(mv, mf, d, lighting, sensor, opt):
        mesh = types.SimpleNamespace()
        mesh.v, mesh.f = mv, mf
        p1, p2, p3 = mv[mf[:, 0]], mv[mf[:, 1]], mv[mf[:, 2]]
        fn = np.cross(p2 - p1, p3 - p1)
        mesh.fn = fn / np.linalg.norm(fn, axis=1, keepdims=True)
        return np.array(pyref.angular_sampling(mesh, d, lighting, sensor, np.array([0, 0, 1.0]), np.array([0, 0, 1.0]), opt))

    out = {}
    rs = np.random.RandomState(5)
    v, f, _, _ = cfg1()
    cases = {"plane": (v.astype(np.float64), f.astype(np.int64), 64),
             "toy": (np.array([[-1, -1, .9], [1, -1, 1], [1, 1, 1.2], [-1, 1, 1], [-2, -2, 1], [2, 1, 1]], np.float64),
                     np.array([[0, 2, 1], [0, 3, 2], [4, 3, 0], [1, 2, 5]], np.int64), 400)}
    for name, (mv, mf, nbin) in cases.items():
        n = 256
        d = rs.normal(size=(n, 3))
        d[:, 2] = np.abs(d[:, 2]) + 0.2
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        opt = types.SimpleNamespace(sample_num=n, max_distance_bin=nbin, distance_resolution=0.02, epsilon=1e-9, normal="fn")
        pairs = [((0.1, 0, 0), (0.1, 0, 0)), ((-0.5, 0, 0), (0.5, 0, 0)) if name == "toy" else ((-0.1, 0, 0), (0.05, 0.1, 0))]
        rows, grads = [], []
        for lighting, sensor in pairs:
            lighting, sensor = np.array(lighting, np.float64), np.array(sensor, np.float64)
            rows.append(forward(mv, mf, d, lighting, sensor, opt))
            g = np.zeros_like(mv)
            h = 1e-6
            for i in range(mv.shape[0]):
                for k in range(3):
                    vp, vm = mv.copy(), mv.copy()
                    vp[i, k] += h
                    vm[i, k] -= h
                    g[i, k] = (forward(vp, mf, d, lighting, sensor, opt).sum() - forward(vm, mf, d, lighting, sensor, opt).sum()) / (2 * h)
            grads.append(g)
        out[name + "_v"], out[name + "_f"], out[name + "_dir"] = mv, mf, d
        out[name + "_lighting"] = np.array([p[0] for p in pairs], np.float64)
        out[name + "_sensor"] = np.array([p[1] for p in pairs], np.float64)
        out[name + "_nbin"], out[name + "_res"] = np.int64(nbin), np.float64(0.02)
        out[name + "_transient"] = np.stack(rows)
        out[name + "_grad_fd"] = np.stack(grads)
        print("pyref grad", name, np.stack(rows).sum(axis=1), np.abs(np.stack(grads)).max())
    np.savez_compressed(os.path.join(HERE, "pyref_angular_grad.npz"), **out)


def radiometry_scenes():
    """Scenes of the statistical radiometry pin: wall-PARALLEL patches, so that the wall cosine of a path of length 2h
    is z_patch / h -- a function of the time bin -- and can be divided out of the v2 rows bin by bin.
      plane  the cfg-1 plane (z = 0.38, +-0.25 m), two wall-facing triangles
      steps  a far square (z = 0.60, +-0.30 m) partly hidden behind a near one (z = 0.40): the near patch's paths end
             at 2h <= 0.98 m, the far patch's start at 1.2 m, so every bin belongs to one depth
    Sources: the four confocal points of the prototype fixtures."""
    v, f, _, _ = cfg1()
    sv = np.array([[-.30, -.30, .60], [.30, -.30, .60], [.30, .30, .60], [-.30, .30, .60],
                   [-.10, -.12, .40], [.15, -.12, .40], [.15, .08, .40], [-.10, .08, .40]], np.float64)
    sf = np.array([[0, 2, 1], [0, 3, 2], [4, 6, 5], [4, 7, 6]], np.int64)
    src = np.array([[0.1, 0, 0], [-0.1, 0, 0], [0, 0.1, 0], [0, -0.1, 0]], np.float64)
    return {"plane": dict(v=v.astype(np.float64), f=f.astype(np.int64), nbin=80, zsplit=1e9, z=(0.38, 0.38), src=src, moved=True),
            "steps": dict(v=sv, f=sf, nbin=104, zsplit=1.1, z=(0.40, 0.60), src=src, moved=False)}


def make_pyref_radiometry():
    """Statistical pin of the v2 radiometry against data the REFERENCE produced (round 4).  The reference's numpy
    prototype (transient_rendering_python/rendering.py:angular_sampling, imported) is an angular estimator: with
    directions uniform on the hemisphere it integrates cos(theta_2) / d_2^2 over solid angle, i.e. the surface
    integral of cos(theta_1) cos(theta_2) / (d_1^2 d_2^2) over the visible surface (rendering.py:82-93).  The v2
    renderer the oracle restates is an AREA estimator of the same surface term times the two wall cosines
    (smoothed_transient/transient_and_gradient.cpp:224-232: A ff^2 / spt, ff = cos_surface cos_wall / h^2).  On
    wall-parallel patches cos_wall = z_patch / h is known per bin, so the two must agree bin by bin within the
    Monte-Carlo error: that pins A / spt, the two-way 1 / h^4, the clamped cosines and the binning origin
    (ceil(d / res) - 1 there, floor((2h - lb) / res) here) against reference output.
    2 000 000 directions per source, in 40 batches of 50 000 (the batch spread is the error estimate the test uses);
    directions: numpy RandomState(4242), uniform on the upper hemisphere (z = U(0, 1), phi = U(0, 2 pi)), drawn batch
    by batch in the order scene -> batch.  Only the histograms are stored."""
    sys.path.insert(0, os.path.join(REF, "transient_rendering_python"))
    import rendering as pyref  # noqa: E402  (the reference module)

    rs = np.random.RandomState(4242)
    nb, per = 40, 50000
    res = 2.0 ** -6
    delta = 0.005
    out = {"res": np.float64(res), "batches": np.int64(nb), "per_batch": np.int64(per), "delta": np.float64(delta)}

    def mesh_of(mv, mf):
        mesh = types.SimpleNamespace()
        mesh.v, mesh.f = mv, mf
        p1, p2, p3 = mv[mf[:, 0]], mv[mf[:, 1]], mv[mf[:, 2]]
        fn = np.cross(p2 - p1, p3 - p1)
        mesh.fn = fn / np.linalg.norm(fn, axis=1, keepdims=True)
        return mesh

    for name, sc in radiometry_scenes().items():
        mv, mf = sc["v"], sc["f"]
        opt = types.SimpleNamespace(sample_num=per, max_distance_bin=sc["nbin"], distance_resolution=res, epsilon=1e-9, normal="fn")
        # Finite-difference pin of the GRADIENT: every patch (4 vertices) is moved as a whole, keeping it wall-parallel --
        # translations along x, y, z and an in-plane scaling about its centre -- by +-delta, with the SAME directions
        # (common random numbers); the test differentiates functionals of the rows along these motions.  Only on the
        # scene WITHOUT occlusion: the reference's analytic gradient holds visibility fixed (no silhouette term,
        # smoothed_transient/transient_and_gradient.cpp:944-1001), so where a moving patch drags its shadow over another
        # the finite difference of the true forward is a different quantity (measured on `steps`: 40 % apart).
        npatch = mv.shape[0] // 4 if sc["moved"] else 0
        motions = []
        for k in range(npatch):
            idx = np.arange(4 * k, 4 * k + 4)
            c = mv[idx].mean(axis=0)
            for m in range(4):
                step = np.zeros_like(mv)
                if m < 3:
                    step[idx, m] = 1.0
                else:
                    step[idx, :2] = (mv[idx] - c)[:, :2]
                motions.append(step)
        motions = np.stack(motions) if motions else np.zeros((0,) + mv.shape)   # [npatch * 4, V, 3]: d vertex / d theta
        rows = np.zeros((nb, sc["src"].shape[0], sc["nbin"]))
        moved = np.zeros((motions.shape[0], 2, nb, sc["src"].shape[0], sc["nbin"]), np.float32)
        for b in range(nb):
            z = rs.uniform(0.0, 1.0, per)
            ph = rs.uniform(0.0, 2 * np.pi, per)
            r = np.sqrt(1.0 - z * z)
            d = np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)
            for i, o in enumerate(sc["src"]):
                up = np.array([0, 0, 1.0])
                rows[b, i] = pyref.angular_sampling(mesh_of(mv, mf), d, o, o, up, up, opt)
                for q in range(motions.shape[0]):
                    for sgn in (0, 1):
                        mq = mesh_of(mv + (delta if sgn == 0 else -delta) * motions[q], mf)
                        moved[q, sgn, b, i] = pyref.angular_sampling(mq, d, o, o, up, up, opt)
        out[name + "_v"], out[name + "_f"], out[name + "_src"] = mv, mf, sc["src"]
        out[name + "_nbin"] = np.int64(sc["nbin"])
        out[name + "_z"] = np.array(sc["z"], np.float64)
        out[name + "_zsplit"] = np.float64(sc["zsplit"])
        out[name + "_rows"] = rows
        if sc["moved"]:
            out[name + "_motions"] = motions
            out[name + "_moved"] = moved                                  # [motion, +/-, batch, source, bin]
        m = rows.sum(axis=2)
        print("pyref radiometry", name, "mass", m.mean(axis=0), "+-", m.std(axis=0, ddof=1) / np.sqrt(nb))
    np.savez_compressed(os.path.join(HERE, "pyref_radiometry.npz"), **out)


def make_jitter_info():
    """The reference's measured SPAD jitter kernel (a data file its own jitter/test.py loads)."""
    import scipy.io
    j = scipy.io.loadmat(os.path.join(REF, "transient_rendering_cython/jitter/jitter_info.mat"))
    np.savez_compressed(os.path.join(HERE, "jitter_info.npz"), jitter_weight=j["jitter_weight"],
                        jitter_grad=j["jitter_grad"], jitter_offset=np.int64(j["jitter_offset"][0, 0]),
                        jitter_time=j["jitter_time"])


def make_adam_modified():
    """Trajectory of the reference's own optimiser (exp_bunny/adam_modified.py, imported, on CPU):
    fixed float64 gradients narrowed to float32 as exp_bunny/test.py:212-213 does."""
    import warnings
    import torch
    sys.path.insert(0, os.path.join(REF, "transient_rendering_cython/exp_bunny"))
    from adam_modified import Adam_Modified  # noqa: E402  (the reference class)
    rs = np.random.RandomState(21)
    p0 = rs.standard_normal((257, 3)).astype(np.float32) * 0.1
    grads = rs.standard_normal((6, 257, 3)) * np.logspace(-6, -2, 257)[None, :, None]
    out = {"p0": p0, "grads": grads}
    for name, kw in (("plain", dict(lr=1e-4 / 3)), ("amsgrad_wd", dict(lr=2e-3, amsgrad=True, weight_decay=0.01,
                                                                       betas=(0.8, 0.99), eps=1e-6))):
        p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = Adam_Modified([p], **kw)
        traj = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for g in grads:
                p.grad = torch.from_numpy(g).float()
                opt.step()
                traj.append(p.data.numpy().copy())
        out[name + "_traj"] = np.stack(traj)
        out[name + "_kw"] = np.array([kw["lr"], kw.get("betas", (0.9, 0.999))[0], kw.get("betas", (0.9, 0.999))[1],
                                      kw.get("eps", 1e-8), kw.get("weight_decay", 0.0), float(kw.get("amsgrad", False))])
    np.savez_compressed(os.path.join(HERE, "adam_modified.npz"), **out)
    print("adam_modified", out["plain_traj"].shape)


def make_oracle_cfg1():
    v, f, origin, normal = cfg1()
    lb, ub, res = 0.0, 2.0, 2.0 ** -5
    tr, path = orc.render_transient(origin, normal, v, f, 256, lb, ub, res, seed=0)
    data = np.zeros_like(tr)
    weight = np.ones_like(tr)
    tr2, grad, _ = orc.render_gradient(origin, normal, v, f, 256, lb, ub, res, data, weight,
                                       refine=10, sigma_bin=1, testing_flag=1, loss_flag=0, seed=0)
    assert np.array_equal(tr, tr2)
    np.savez_compressed(os.path.join(HERE, "oracle_cfg1.npz"), v=v, f=f, origin=origin, normal=normal,
                        lb=lb, ub=ub, res=res, num_sample=256, transient=tr, pathlengths=path,
                        gradient=grad)
    print("cfg1 rows", tr.sum(axis=1), "grad", np.abs(grad).max())


def make_oracle_bunny(v, f):
    origin, normal = grid_sources(4, 0.25)
    lb, ub, res = 0.625, 1.625, 2.0 ** -9
    tr, path = orc.render_transient(origin, normal, v, f, 20000, lb, ub, res, seed=0, accel=1, threads=1)
    rs = np.random.RandomState(1)
    data = tr * (1.0 + 0.3 * rs.standard_normal(tr.shape))
    weight = 0.5 + rs.random_sample(tr.shape)
    _, grad, _ = orc.render_gradient(origin, normal, v, f, 20000, lb, ub, res, data, weight, refine=10,
                                     sigma_bin=1, testing_flag=1, loss_flag=0, seed=0, accel=1, threads=1)
    # the same render under the reference's RULE-FREE hit test (Embree accepts every den != 0,
    # SMO/transient_and_gradient.cpp:199-206), all faces, brute force: pins the contract's grazing rule against drift
    # (tests/test_oracle.py::test_contract_stays_within_tolerance_of_the_rule_free_fixture)
    with orc.rule_free():
        tr_rf, _ = orc.render_transient(origin, normal, v, f, 20000, lb, ub, res, seed=0, accel=0)
        _, grad_rf, _ = orc.render_gradient(origin, normal, v, f, 20000, lb, ub, res, data, weight, refine=10,
                                            sigma_bin=1, testing_flag=1, loss_flag=0, seed=0, accel=0)
    np.savez_compressed(os.path.join(HERE, "oracle_bunny16.npz"), origin=origin, normal=normal, lb=lb, ub=ub,
                        res=res, num_sample=20000, transient=tr, data=data, weight=weight, gradient=grad,
                        transient_rule_free=tr_rf, gradient_rule_free=grad_rf)
    print("bunny16 rows", tr.sum(axis=1)[:4], "grad", np.abs(grad).max(),
          "| vs rule-free: rows %.2e grad %.2e" % (np.linalg.norm(tr - tr_rf) / np.linalg.norm(tr_rf),
                                                   np.linalg.norm(grad - grad_rf) / np.linalg.norm(grad_rf)))


def make_ggx_table():
    alphas = np.array([0.05, 0.1, 0.3, 0.5, 0.9, 1.0], np.float32)
    nws = np.concatenate([np.linspace(-0.2, 1.0, 25), [1e-4, 0.9999, 1.0]]).astype(np.float32)
    ev = np.zeros((alphas.size, nws.size), np.float32)
    ad = np.zeros_like(ev)
    ns = np.zeros_like(ev)
    n = np.array([0, 0, 1], np.float32)
    for i, a in enumerate(alphas):
        for j, c in enumerate(nws):
            s = np.sqrt(max(0.0, 1.0 - float(c) ** 2))
            w = np.array([s, 0, c], np.float32)
            ev[i, j] = orc.ggx(a, n, w, "eval")
            ad[i, j] = orc.ggx(a, n, w, "adiff")
            ns[i, j] = orc.ggx(a, n, w, "nwsdiff")
    np.savez_compressed(os.path.join(HERE, "ggx_table.npz"), alpha=alphas, nw=nws, eval=ev, adiff=ad, nwsdiff=ns)


if __name__ == "__main__":
    if "--oracle-only" in sys.argv:
        # the oracle's own regression fixtures (after a change of the numeric contract); the reference is not needed
        d = np.load(os.path.join(HERE, "bunny_5k.npz"))
        make_oracle_cfg1()
        make_oracle_bunny(d["v"], d["f"])
        make_ggx_table()
        sys.exit(0)
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    if "--radiometry-only" in sys.argv:
        make_pyref_radiometry()
        sys.exit(0)
    if "--measurement-only" in sys.argv:
        make_mannequin_measurement()
        sys.exit(0)
    if "--meshes-only" in sys.argv:
        # the two fixture meshes and the oracle vectors that depend on them; the prototype / optimiser fixtures stay
        bv, bf = make_meshes()
        make_oracle_cfg1()
        make_oracle_bunny(bv, bf)
        sys.exit(0)
    bv, bf = make_meshes()
    make_mannequin_measurement()
    make_pyref()
    make_pyref_nc()
    make_pyref_grad()
    make_pyref_radiometry()
    make_jitter_info()
    make_adam_modified()
    make_oracle_cfg1()
    make_oracle_bunny(bv, bf)
    make_ggx_table()
