"""Context state of the device path: generations that guard reuse_bvh / reuse_visibility against stale
caches, the report of which kernels a render took (nothing falls back silently), chunked pass 1 when the
tile scratch is bounded, and the deferred bad-face-index status of the device-pointer path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, grid_sources, rel_l2

pytestmark = pytest.mark.gpu

LB, UB, RES, T = 0.625, 1.625, 2.0 ** -9, 512


def _dev_setup(bunny, n=3, seed=2):
    import torch
    from nlos_surface_optimization_amd import device as nd
    v, f = bunny
    o, nrm = grid_sources(n, 0.2)
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev, seed=seed)
    tv, tf, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, nrm))
    return r, tv, tf, to, tn, o, nrm


def test_backward_after_another_forward_differentiates_its_own_mesh(bunny, orc):
    """forward(v_a), forward(v_b) on the same renderer (same F, V, sources, seed), then backward of a: the gradient
    must be a's (smoothed_transient/transient_and_gradient.cpp:843-1007 evaluated at v_a), not a contraction against
    b's tree and visibility."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    r, tv, tf, to, tn, o, nrm = _dev_setup(bunny)
    v, f = bunny
    ns = 9000
    L = o.shape[0]
    rs = np.random.RandomState(3)
    vb_np = (v + 0.004 * rs.standard_normal(v.shape)).astype(np.float32)
    va = tv.clone().requires_grad_(True)
    vb = torch.from_numpy(vb_np).to(tv.device).requires_grad_(True)
    data = torch.zeros((L, T), dtype=torch.float64, device=tv.device)
    Ta = nd.render_transient_autograd(r, va, to, tn, tf, ns, LB, UB, RES, seed=2)
    gens_a = (r.mesh_generation(), r.visibility_generation())
    Tb = nd.render_transient_autograd(r, vb, to, tn, tf, ns, LB, UB, RES, seed=2)
    assert (r.mesh_generation(), r.visibility_generation()) != gens_a
    ((data - Ta) ** 2).sum().div(L).backward()
    ((data - Tb) ** 2).sum().div(L).backward()
    w = np.ones((L, T))
    _, ga, _ = orc.render_gradient(o, nrm, v, f, ns, LB, UB, RES, np.zeros((L, T)), w, accel=1, seed=2)
    _, gb, _ = orc.render_gradient(o, nrm, vb_np, f, ns, LB, UB, RES, np.zeros((L, T)), w, accel=1, seed=2)
    assert rel_l2(ga, gb) > 1e-2                                  # the two meshes have clearly different gradients
    assert rel_l2(va.grad.double().cpu().numpy(), ga) <= 1e-4
    assert rel_l2(vb.grad.double().cpu().numpy(), gb) <= 1e-4
    # the sum of two graphs through one renderer
    vc = tv.clone().requires_grad_(True)
    vd = torch.from_numpy(vb_np).to(tv.device).requires_grad_(True)
    tot = nd.render_transient_autograd(r, vc, to, tn, tf, ns, LB, UB, RES) + \
        nd.render_transient_autograd(r, vd, to, tn, tf, ns, LB, UB, RES)
    ((data - tot) ** 2).sum().backward()
    assert torch.isfinite(vc.grad).all() and torch.isfinite(vd.grad).all()
    r.close()


def test_reuse_flags_need_the_matching_generation(bunny):
    import torch
    from nlos_surface_optimization_amd import _lib
    r, tv, tf, to, tn, o, nrm = _dev_setup(bunny)
    ns = 9000
    L = o.shape[0]
    assert r.mesh_generation() == 0 and r.visibility_generation() == 0
    tr, _ = r.render_transient(to, tn, tv, tf, ns, LB, UB, RES, keep_visibility=True)
    mg, vg = r.mesh_generation(), r.visibility_generation()
    assert mg > 0 and vg > 0
    res = (0.1 * tr).contiguous()
    _, g_full, _ = r.render_gradient(to, tn, tv, tf, ns, LB, UB, RES, residual=res)
    mg2, vg2 = r.mesh_generation(), r.visibility_generation()
    assert mg2 > mg and vg2 > vg
    # reuse with the current generations: same gradient, no transient
    t_none, g_reuse, _ = r.render_gradient(to, tn, tv, tf, ns, LB, UB, RES, residual=res, reuse_visibility=True,
                                           reuse_bvh=True, mesh_generation=mg2, visibility_generation=vg2)
    assert t_none is None and rel_l2(g_reuse.cpu().numpy(), g_full.cpu().numpy()) <= 1e-12
    assert (r.mesh_generation(), r.visibility_generation()) == (mg2, vg2)      # a reuse records nothing new
    for kw in (dict(mesh_generation=mg, visibility_generation=vg2),             # stale tree
               dict(mesh_generation=mg2, visibility_generation=vg),             # stale cache
               dict()):                                                         # no generation at all
        with pytest.raises(_lib.NlosError):
            r.render_gradient(to, tn, tv, tf, ns, LB, UB, RES, residual=res, reuse_visibility=True, reuse_bvh=True, **kw)
    # a forward-only render that keeps no cache invalidates the visibility generation
    r.render_transient(to, tn, tv, tf, ns, LB, UB, RES)
    assert r.visibility_generation() == 0
    r.close()


def test_last_path_reports_backend_reason_and_workgroup_outcomes(bunny, mannequin):
    import torch
    from nlos_surface_optimization_amd import mesh_io
    r, tv, tf, to, tn, o, nrm = _dev_setup(bunny)
    ns = 9000
    r.render_transient(to, tn, tv, tf, ns, LB, UB, RES)
    p = r.last_path(count=True)
    assert p["backend"] == "grid" and p["reason"] == "" and p["tiles"] == 1 and p["chunks"] == 1 and p["rows_in_lds"]
    assert p["gradient_kernel"] == "none"
    assert p["workgroups"] == 9 and p["big_lds"] == 0 and p["bvh_queries"] == 0
    assert "rays_traced" not in p        # (a forward-only render keeps no item masks: nothing to count from)
    # what pass 1 did with the L * F * spt surface samples (round 5): the item-mask headers carry, per source, the rays that
    # went through the occlusion query and the samples that were binned; the second equals the set bits of the cache
    F = tf.shape[0]
    spt = 1 + (ns - 1) // F
    r.render_transient(to, tn, tv, tf, ns, LB, UB, RES, keep_visibility=True)
    p = r.last_path(count=True)
    vis, _ = r.debug_visibility(9, spt, F)
    accepted = int(sum(bin(int(w)).count("1") for w in vis.ravel()))
    assert p["samples_accepted"] == accepted and 0 < accepted <= p["rays_traced"] < 9 * F * spt
    r.render_transient(to, tn, tv, tf, ns, LB, UB, RES, force_bvh=True)
    p = r.last_path(count=True)
    assert p["backend"] == "bvh" and p["reason"] == "force_bvh" and "rays_traced" not in p
    # a wall point inside the scene's depth range: that workgroup traces through the in-kernel BVH query
    o2 = o.copy()
    o2[4, 2] = 0.45
    to2 = torch.from_numpy(o2).to(tv.device)
    data = torch.zeros((9, T), dtype=torch.float64, device=tv.device)
    r.render_gradient(to2, tn, tv, tf, ns, LB, UB, RES, data=data, weight=torch.ones_like(data))
    p = r.last_path(count=True)
    assert p["backend"] == "grid" and p["bvh_queries"] == 1 and p["gradient_kernel"] == "source-major, LDS accumulator"
    # 2048 bins: rows stay in global memory
    r.render_transient(to, tn, tv, tf, ns, 0.625, 1.625, 2.0 ** -11)
    assert not r.last_path()["rows_in_lds"]
    # tiny mesh -> BVH back-end, with the reason
    vq = np.array([[-.25, -.25, .38], [.25, -.25, .38], [.25, .25, .38], [-.25, .25, .38]], np.float32)
    fq = np.array([[0, 2, 1], [0, 3, 2]], np.int32)
    r.render_transient(to, tn, torch.from_numpy(vq).to(tv.device), torch.from_numpy(fq).to(tv.device), 256, 0.0, 2.0, 2.0 ** -5)
    p = r.last_path()
    assert p["backend"] == "bvh" and p["reason"] == "mesh below 64 faces"
    # the grid resolution grows with the LDS a small mesh leaves free -- without sending a fully visible height field
    # (every face reachable: the worst case of the entry estimate) into the coarsening path
    n = 32
    xs, ys = np.meshgrid(np.linspace(-0.3, 0.3, n), np.linspace(-0.3, 0.3, n))
    vh = np.stack([xs.ravel(), ys.ravel(), 0.45 + 0.05 * np.sin(7 * xs.ravel()) * np.cos(5 * ys.ravel())], 1).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    qa, qb, qc, qd = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    fh = np.concatenate([np.stack([qa, qc, qb], 1), np.stack([qa, qd, qc], 1)]).astype(np.int32)
    r.render_transient(to, tn, torch.from_numpy(vh).to(tv.device), torch.from_numpy(fh).to(tv.device), 4 * fh.shape[0], 0.3, 1.6, 2.0 ** -7)
    p = r.last_path(count=True)
    assert p["backend"] == "grid" and p["grid_R"] > int(round((fh.shape[0] / 2) ** 0.5)) and p["coarsened"] == 0 and p["big_lds"] == 0
    # large mesh -> tiled grid, face-major gradient
    v, f = bunny
    v2, f2 = mesh_io.subdivide(v, f, 1)
    tv2, tf2 = torch.from_numpy(v2).to(tv.device), torch.from_numpy(f2).to(tv.device)
    r.render_gradient(to, tn, tv2, tf2, 4 * f2.shape[0], LB, UB, RES, data=data, weight=torch.ones_like(data))
    p = r.last_path(count=True)
    assert p["backend"] == "tiled-grid" and p["tiles"] >= 4 and p["chunks"] == 1 and p["gradient_kernel"] == "face-major"
    assert p["workgroups"] == 9 * p["tiles"]
    r.close()


_CHUNK_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nlos_surface_optimization_amd import device as nd, mesh_io
d = np.load(%r)
v, f = mesh_io.subdivide(np.ascontiguousarray(d["v"], np.float32), np.ascontiguousarray(d["f"], np.int32), 1)
g = np.linspace(-0.22, 0.22, 3)
o = np.array([[x, y, 0] for y in g for x in g], np.float32)[:7]
n = np.tile(np.array([0, 0, 1], np.float32), (7, 1))
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev, seed=4)
tv, tf, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, n))
ns = 4 * f.shape[0]
t, _ = r.render_transient(to, tn, tv, tf, ns, 0.625, 1.625, 2.0 ** -9, keep_visibility=True)
p = r.last_path(count=True)
data = (1.3 * t).contiguous()
_, g1, _ = r.render_gradient(to, tn, tv, tf, ns, 0.625, 1.625, 2.0 ** -9, data=data, weight=torch.ones_like(data))
np.savez(sys.argv[1], t=t.cpu().numpy(), g=g1.cpu().numpy(), chunks=p["chunks"], backend=p["backend"])
"""


def test_bounded_tile_scratch_renders_the_sources_in_chunks(tmp_path):
    """The tiled grid's per-(source, tile) subsets are bounded (32 GB; here NLOS_TILE_SCRATCH_MAX forces 3 sources
    per chunk): pass 1 then runs in chunks of sources -- same rows, same gradient, still the tiled grid (round 1
    dropped to the BVH back-end, silently)."""
    golden = os.path.join(ROOT, "tests", "golden", "bunny_5k.npz")
    outs = []
    for tag, limit in (("one", None), ("chunked", None)):
        env = dict(os.environ)
        out = str(tmp_path / (tag + ".npz"))
        if tag == "chunked":
            # 9 tiles x (6 F / 9 + 512) slots x 6 B per source ~ 0.74 MB at F = 19 868: room for 3 sources
            env["NLOS_TILE_SCRATCH_MAX"] = str(3 * 800 * 1024)
        subprocess.run([sys.executable, "-c", _CHUNK_SCRIPT % (ROOT, golden), out], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    a, b = outs
    assert str(a["backend"]) == "tiled-grid" and str(b["backend"]) == "tiled-grid"
    assert int(a["chunks"]) == 1 and int(b["chunks"]) >= 2
    assert a["t"].sum() > 0 and rel_l2(b["t"], a["t"]) <= 1e-12
    assert rel_l2(b["g"], a["g"]) <= 1e-6          # (two renders: see test_render_step_is_hip_graph_capturable)


def test_device_path_surfaces_a_bad_face_index(bunny):
    """nlos_render on device pointers cannot validate indices on the host; the scene build flags them (and reads
    vertex 0 instead, so nothing faults).  The flag follows the build to pinned memory and is raised by
    nlos_ctx_check, or by the next render once it has arrived."""
    import torch
    from nlos_surface_optimization_amd import _lib
    r, tv, tf, to, tn, o, nrm = _dev_setup(bunny)
    ns = 9000
    bad = tf.clone()
    bad[5, 1] = tv.shape[0] + 3
    r.render_transient(to, tn, tv, bad, ns, LB, UB, RES)       # asynchronous: enqueued without an error
    with pytest.raises(_lib.NlosError, match="face index out of range"):
        r.check()
    r.check()                                                   # reported once
    t, _ = r.render_transient(to, tn, tv, tf, ns, LB, UB, RES)  # the context stays usable
    r.check()
    assert float(t.sum()) > 0
    # without an explicit check the next render reports it (the flag has arrived after a synchronise)
    r.render_transient(to, tn, tv, bad, ns, LB, UB, RES)
    torch.cuda.synchronize()
    with pytest.raises(_lib.NlosError, match="face index out of range"):
        r.render_transient(to, tn, tv, tf, ns, LB, UB, RES)
    t2, _ = r.render_transient(to, tn, tv, tf, ns, LB, UB, RES)
    r.check()
    assert rel_l2(t2.cpu().numpy(), t.cpu().numpy()) <= 1e-12
    # the synchronising entry points report it too: nobody has to remember check()
    r.render_transient(to, tn, tv, bad, ns, LB, UB, RES)
    with pytest.raises(_lib.NlosError, match="face index out of range"):
        r.last_path(count=True)
    r.enable_timing(True)
    r.timing_reset()
    r.render_transient(to, tn, tv, bad, ns, LB, UB, RES)
    with pytest.raises(_lib.NlosError, match="face index out of range"):
        r.timing_mean_ms()
    r.check()
    r.close()


_ITEMS_SCRIPT = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nlos_surface_optimization_amd import device as nd
d = np.load(%r)
v, f = d["v"], d["f"]
g = np.linspace(-0.2, 0.2, 3)
o = np.array([[x, y, 0] for y in g for x in g], np.float32)
n = np.tile(np.array([0, 0, 1], np.float32), (9, 1))
dev = torch.device("cuda", 0)
r = nd.TransientRenderer(dev, seed=4)
tv, tf, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, n))
rs = np.random.RandomState(2)
t, _ = r.render_transient(to, tn, tv, tf, 9000, 0.625, 1.625, 2.0 ** -9)
data = torch.from_numpy(t.cpu().numpy() * (1 + 0.3 * rs.standard_normal(t.shape))).to(dev)
t2, g2, _ = r.render_gradient(to, tn, tv, tf, 9000, 0.625, 1.625, 2.0 ** -9, data=data, weight=torch.ones_like(data))
torch.cuda.synchronize()
vis, fid = r.debug_visibility(9, 1 + (9000 - 1) // f.shape[0], f.shape[0])
np.savez(sys.argv[1], t=t2.cpu().numpy(), g=g2.cpu().numpy(), vis=vis, fid=fid)
"""


def test_item_mask_visibility_equals_per_face_words(bunny, tmp_path):
    """The visibility cache in its two layouts -- one 64-bit ballot per 64-ray item of the live list (what the grid kernel
    records for confocal renders) and a word per (face, 32 strata) (NLOS_VIS_ITEMS=0: everywhere) -- describes the same
    accepted samples: converted by k_items_to_words the caches are bitwise equal, and pass 2 gives the same gradient."""
    import subprocess
    import sys
    golden = os.path.join(ROOT, "tests", "golden", "bunny_5k.npz")
    outs = []
    for items in ("1", "0"):
        env = dict(os.environ)
        env["NLOS_VIS_ITEMS"] = items
        out = str(tmp_path / ("vis%s.npz" % items))
        subprocess.run([sys.executable, "-c", _ITEMS_SCRIPT % (ROOT, golden), out], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["fid"], b["fid"])
    assert np.array_equal(a["vis"], b["vis"]) and int(np.unpackbits(a["vis"].view(np.uint8)).sum()) > 1000
    assert rel_l2(a["t"], b["t"]) <= 1e-13 and rel_l2(a["g"], b["g"]) <= 1e-6


def test_face_major_gradient_reads_an_item_mask_cache(orc):
    """A mesh the single-workgroup grid renders (F <= 6.2 k) whose 3V-double accumulator does not fit LDS (a triangle soup:
    V = 3 F): pass 1 records item masks, the face-major gradient kernel indexes the cache by face -- k_items_to_words sits
    in between.  Against the oracle."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    rs = np.random.RandomState(12)
    m = 3000
    c = np.stack([rs.uniform(-0.25, 0.25, m), rs.uniform(-0.25, 0.25, m), rs.uniform(0.4, 0.6, m)], 1)
    tri = c[:, None, :] + 0.02 * rs.normal(size=(m, 3, 3))
    v = np.ascontiguousarray(tri.reshape(-1, 3), np.float32)
    f = np.ascontiguousarray(np.arange(3 * m).reshape(m, 3), np.int32)
    # wind every triangle towards the wall
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    flip = np.cross(p1 - p0, p2 - p0)[:, 2] > 0
    f[flip] = f[flip][:, [0, 2, 1]]
    o, n = grid_sources(3, 0.2)
    ns = 4 * m
    t_ref, _ = orc.render_transient(o, n, v, f, ns, LB, UB, RES, accel=1, seed=0)
    data = np.ascontiguousarray(t_ref * (1 + 0.3 * rs.standard_normal(t_ref.shape)))
    w = np.ones_like(data)
    _, g_ref, _ = orc.render_gradient(o, n, v, f, ns, LB, UB, RES, data, w, accel=1, seed=0)
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev, seed=0)
    tv, tf, to, tn, td, tw = (torch.from_numpy(x).to(dev) for x in (v, f, o, n, data, w))
    t, g, _ = r.render_gradient(to, tn, tv, tf, ns, LB, UB, RES, data=td, weight=tw)
    p = r.last_path()
    assert p["backend"] == "grid" and p["gradient_kernel"].startswith("face-major")
    assert t_ref.sum() > 0 and rel_l2(t.cpu().numpy(), t_ref) <= 1e-5 and rel_l2(g.cpu().numpy(), g_ref) <= 1e-4
    r.close()


def _morton_order(v, f):
    """Host restatement of the single-workgroup builder's key (bvh_build.hip, phase 2: fp32 centroid sums, 8 bits per axis)
    and the order a STABLE sort by it gives."""
    def expand(x):
        x = x.astype(np.uint64)
        x = (x * 0x00010001) & 0xFF0000FF
        x = (x * 0x00000101) & 0x0F00F00F
        x = (x * 0x00000011) & 0xC30C30C3
        x = (x * 0x00000005) & 0x49249249
        return x
    p = v[f]                                                        # [F, 3, 3] float32
    lo, hi = p.reshape(-1, 3).min(0), p.reshape(-1, 3).max(0)
    c = (np.float32(0) + p[:, 0]) + p[:, 1] + p[:, 2]
    ext = hi - lo
    inv = np.where(ext > 0, np.float32(1) / np.where(ext > 0, ext, np.float32(1)), np.float32(0)).astype(np.float32)
    nrm = ((c * np.float32(1.0 / 3.0) - lo) * inv).astype(np.float32)
    q = np.minimum(np.maximum(nrm * np.float32(256), np.float32(0)), np.float32(255)).astype(np.uint32)
    key = (expand(q[:, 0]) << 2) | (expand(q[:, 1]) << 1) | expand(q[:, 2])
    return key, np.argsort(key, kind="stable")


@pytest.mark.parametrize("mesh", ["bunny", "mannequin", "ragged"])
def test_scene_build_orders_the_faces_by_a_stable_morton_sort(bunny, mannequin, mesh):
    """The face order every kernel works in (debug_read 1) is the stable sort of the 24-bit Morton keys of the fp32
    centroid sums: the in-LDS radix sort of k_build_bvh (4 passes x 6 bits, one counter column per wave, ranks by
    match-any ballots) against numpy's stable argsort -- on the benchmark mesh, on a mesh whose face count is no multiple
    of anything (ragged last rounds of every wave), and with many equal keys (stability is what LSD radix relies on)."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    v, f = bunny if mesh != "mannequin" else mannequin
    if mesh == "ragged":
        f = np.ascontiguousarray(np.concatenate([f[:1237], f[:1237][::-1]]))       # 2 474 faces, every key twice
    o, n = grid_sources(2, 0.2)
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev)
    tv, tf, to, tn = (torch.from_numpy(x).to(dev) for x in (v, f, o, n))
    r.render_transient(to, tn, tv, tf, 2 * f.shape[0], 0.625, 1.625, 2.0 ** -9, keep_visibility=True)
    torch.cuda.synchronize()
    _, fid = r.debug_visibility(4, 2, f.shape[0])
    key, order = _morton_order(v, f)
    assert np.array_equal(np.sort(fid), np.arange(f.shape[0]))
    assert (np.diff(key[fid].astype(np.int64)) >= 0).all()
    assert np.array_equal(fid, order)
    r.close()


@pytest.mark.parametrize("env", [{"NLOS_GEO_CACHE": "0"}, {"NLOS_FUSE_RESIDUAL": "0"}, {"NLOS_GEO_CACHE": "0", "NLOS_FUSE_RESIDUAL": "0", "NLOS_VIS_ITEMS": "0"}])
def test_the_step_without_its_round_4_shortcuts_still_matches_the_oracle(env):
    """The vertex-gradient step reads pass 1's geometry cache and forms the residual inside pass 2 by default; the paths
    behind them (pass 2 regenerating its samples, the residual launch, per-face visibility words) stay in the library for
    pairs, jitter, the tiled grid and visibility reuse.  The switches are read once per process, so the smoke render --
    forward + gradient against the oracle -- runs in a child process with them off."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "__graft_entry__.py"), "smoke"], capture_output=True, text=True,
                         env=e, timeout=600)
    assert out.returncode == 0 and "smoke: transient rel-L2" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_geometry_cache_is_bounded_and_optional():
    """NLOS_GEO_CACHE_MAX_GB = 0: the cache is never allocated and the step regenerates its samples in pass 2 -- the same
    results (child process: the bound is read once)."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    e = dict(os.environ)
    e["NLOS_GEO_CACHE_MAX_GB"] = "0"
    out = subprocess.run([sys.executable, os.path.join(root, "__graft_entry__.py"), "smoke"], capture_output=True, text=True,
                         env=e, timeout=600)
    assert out.returncode == 0 and "smoke: transient rel-L2" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_source_order_does_not_change_results_and_follows_the_origin_array(bunny, orc):
    """Round 6: pass 1 takes its sources in Z-order of their wall positions (a cached permutation, recomputed when the origin
    array or its length changes).  Rows belong to their sources whatever the order; a different origin array of the same
    length, a shuffled one, and wall positions that are all negative must all render the oracle's rows."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    from conftest import grid_sources
    v, f = bunny
    dev = torch.device("cuda", 0)
    r = nd.TransientRenderer(dev, seed=0)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    o, n = grid_sources(24, 0.25)                                   # 576 sources (>= 512: the order is used)
    rs = np.random.RandomState(3)
    for name, oo in (("grid", o), ("shuffled", o[rs.permutation(o.shape[0])]), ("negative", o - np.array([0.6, 0.5, 0], np.float32))):
        oo = np.ascontiguousarray(oo, np.float32)
        to, tn = torch.from_numpy(oo).to(dev), torch.from_numpy(n).to(dev)
        t, _ = r.render_transient(to, tn, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9)
        sel = rs.choice(oo.shape[0], 12, replace=False)
        t_ref, _ = orc.render_transient(oo, n, v, f, 20000, 0.625, 1.625, 2.0 ** -9, accel=1, seed=0)
        assert rel_l2(t.cpu().numpy(), t_ref) <= 1e-12, name
        assert rel_l2(t.cpu().numpy()[sel], t_ref[sel]) <= 1e-12, name
        # the same array again (cached order), then mutated in place (stale order: still every source's own row)
        t2, _ = r.render_transient(to, tn, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9)
        assert rel_l2(t2.cpu().numpy(), t_ref) <= 1e-12, name
        to.copy_(torch.flip(to, [0]))
        t3, _ = r.render_transient(to, tn, tv, tf, 20000, 0.625, 1.625, 2.0 ** -9)
        t_ref3, _ = orc.render_transient(np.ascontiguousarray(oo[::-1]), n, v, f, 20000, 0.625, 1.625, 2.0 ** -9, accel=1, seed=0)
        assert rel_l2(t3.cpu().numpy(), t_ref3) <= 1e-12, name
    r.close()
