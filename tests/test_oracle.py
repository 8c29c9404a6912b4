"""CPU tests of the oracle (the checker itself): pinned against
  (1) fixtures produced by the reference's own numpy prototype (imported from /root/reference
      in the build container by tests/golden/make_golden.py; only numeric data is committed),
  (2) finite differences of its own forward model (formula check),
  (3) its committed golden vectors (regression pin for the GPU parity tests).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, grid_sources, rel_l2


def test_counter_rng_is_pure_and_in_unit_interval(orc):
    s = [orc.sample(7, k) for k in range(2000)]
    a = np.array(s)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert orc.sample(7, 123) == orc.sample(7, 123)
    assert orc.sample(7, 123) != orc.sample(8, 123)
    # 23-bit mantissa floats: k * 2^-23 exactly (stratified_transient_raytracer/rng_sse.h:33-42)
    assert np.all(a * 2 ** 23 == np.round(a * 2 ** 23))
    assert abs(a.mean() - 0.5) < 0.02


def test_num_bins_is_float32_ceil(orc):
    assert orc.num_bins(0.0, 2.0, 2.0 ** -5) == 64
    assert orc.num_bins(0.625, 1.625, 2.0 ** -9) == 512
    for T in (512, 1024, 1200, 2048):
        assert orc.num_bins(0.0, float(np.float32(T) * np.float32(1.2e-3)), 1.2e-3) == T


def test_closest_hit_matches_reference_prototype(orc):
    """Hit selection, distance, binning and cos/d^2 weighting of the reference's numpy
    prototype (transient_rendering_python/rendering.py:8-93) reproduced from the oracle's
    closest-hit primitive on the prototype's own inputs."""
    g = np.load(os.path.join(GOLDEN, "pyref_angular.npz"))
    for name in ("plane", "toy"):
        v, f, d = g[name + "_v"], g[name + "_f"], g[name + "_dir"]
        nbin, res = int(g[name + "_nbin"]), float(g[name + "_res"])
        fn = g[name + "_fn"]
        for k, p in enumerate(g[name + "_pairs"]):
            o = np.tile(p, (d.shape[0], 1))
            hit = orc.intersect(o, d, v, f, accel=0)
            prim = hit[:, 0].astype(int)
            assert np.array_equal(prim, g[name + "_prim"][k])          # same nearest triangle per ray
            ok = prim >= 0
            bary = hit.copy()
            pts = orc.barycentric_to_world(v, f, bary).astype(np.float64)
            d1 = np.linalg.norm(pts[ok] - p, axis=1)
            assert np.allclose(d1, g[name + "_tnear"][k][ok], rtol=0, atol=2e-6)   # unit directions: t == distance
            # confocal pair: d2 == d1, v2 = (sensor - x)/d2; prototype's bin and weight
            v2 = (p - pts[ok]) / d1[:, None]
            cos = np.einsum("ij,ij->i", fn[prim[ok]], v2)
            cos[cos < 0] = 0
            b = np.ceil((d1 + d1) / res) - 1
            keep = b <= nbin
            t = np.zeros(nbin)
            np.add.at(t, b[keep].astype(int), (cos / d1 ** 2)[keep])
            t *= 2 * np.pi / d.shape[0]
            ref = g[name + "_transient"][k]
            # fp32 hit points vs the prototype's fp64: a sample may cross a bin edge
            assert abs(t.sum() - ref.sum()) <= 1e-5 * ref.sum()
            assert np.abs(np.cumsum(t) - np.cumsum(ref)).max() <= 2.0 * (cos / d1 ** 2).max() * 2 * np.pi / d.shape[0]


def test_bvh_equals_brute_force_on_random_rays(orc, bunny):
    v, f = bunny
    rs = np.random.RandomState(5)
    n = 20000
    o = np.zeros((n, 3), np.float32)
    o[:, :2] = rs.uniform(-0.3, 0.3, (n, 2))
    tgt = v[rs.randint(0, v.shape[0], n)] + rs.normal(0, 0.01, (n, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    a = orc.intersect(o, d, v, f, accel=0)
    b = orc.intersect(o, d, v, f, accel=1)
    assert np.array_equal(np.nan_to_num(a, nan=-7), np.nan_to_num(b, nan=-7))
    assert (a[:, 0] >= 0).mean() > 0.5
    s = orc.intersect(o, d, v, f, accel=1, short=True)
    assert np.array_equal(s, a[:, 0])


def test_render_bvh_equals_brute_force(orc, bunny):
    v, f = bunny
    origin, normal = grid_sources(2, 0.2)
    a, _ = orc.render_transient(origin, normal, v, f, 9000, 0.625, 1.625, 2.0 ** -9, accel=0, threads=1)
    b, _ = orc.render_transient(origin, normal, v, f, 9000, 0.625, 1.625, 2.0 ** -9, accel=1, threads=1)
    assert np.array_equal(a, b)
    assert a.sum() > 0


def test_oracle_regression_vs_golden(orc, cfg1, bunny):
    c = cfg1
    g = np.load(os.path.join(GOLDEN, "oracle_cfg1.npz"))
    tr, gr, path = orc.render_gradient(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                       np.zeros((4, 64)), np.ones((4, 64)), seed=0)
    assert rel_l2(tr, g["transient"]) < 1e-13 and rel_l2(gr, g["gradient"]) < 1e-12
    assert np.array_equal(path, g["pathlengths"])
    # plane 2h in [0.76, 1.60] -> bins 24..51
    nz = np.nonzero(tr.sum(axis=0))[0]
    assert nz.min() >= 24 and nz.max() <= 51
    v, f = bunny
    g = np.load(os.path.join(GOLDEN, "oracle_bunny16.npz"))
    tr, gr, _ = orc.render_gradient(g["origin"], g["normal"], v, f, int(g["num_sample"]), float(g["lb"]),
                                    float(g["ub"]), float(g["res"]), g["data"], g["weight"], seed=0, accel=1)
    assert rel_l2(tr, g["transient"]) < 1e-13 and rel_l2(gr, g["gradient"]) < 1e-11


def test_contract_stays_within_tolerance_of_the_rule_free_fixture(orc, bunny):
    """The committed rule-free vectors (all faces, no grazing rule: the reference's definition,
    SMO/transient_and_gradient.cpp:199-206) pin the contract itself: whatever happens to the rule or the arithmetic,
    the contract's render may not leave them by more than a tenth of the stated tolerance."""
    v, f = bunny
    g = np.load(os.path.join(GOLDEN, "oracle_bunny16.npz"))
    tr, gr, _ = orc.render_gradient(g["origin"], g["normal"], v, f, int(g["num_sample"]), float(g["lb"]),
                                    float(g["ub"]), float(g["res"]), g["data"], g["weight"], seed=0, accel=1)
    assert rel_l2(tr, g["transient_rule_free"]) <= 1e-6
    assert np.abs(tr - g["transient_rule_free"]).max() <= 1e-7 * g["transient_rule_free"].max()
    assert rel_l2(gr, g["gradient_rule_free"]) <= 1e-6


@pytest.mark.timeout(2400)      # three all-faces brute-force renders of 100 sources: ~5 minutes on 8 cores
def test_grazing_rule_stays_inside_the_tolerance_of_the_rule_free_definition(orc, bunny, mannequin):
    """for every BASELINE mesh / window the contract (grazing rule of
    include/nlos_contract.h) against the oracle with the rule switched off, all faces, brute force -- the reference's
    semantics (Embree accepts every den != 0).  100 sources each.  (profiles/r03_graze_sweep.json holds the same
    table for cut-offs 2^-6 ... 2^-11; tools/graze_sweep.py.)"""
    bv, bf = bunny
    mv, mf = mannequin
    cases = [("bunny 512 bins", bv, bf, 0.25, 0.625, 1.625, 2.0 ** -9, {}),
             ("mannequin +-0.35 1024 bins", mv, mf, 0.35, 0.0, 1024 * 2.4e-3, 2.4e-3, {}),
             ("bunny GGX 1024 bins", bv, bf, 0.25, 0.625, 1.625, 2.0 ** -10, dict(ggx_alpha=0.3))]
    assert abs(orc.graze_ratio() - 2.0 ** -9) < 1e-12
    for name, v, f, half, lb, ub, res, kw in cases:
        origin, normal = grid_sources(10, half)
        t_con, _ = orc.render_transient(origin, normal, v, f, 20000, lb, ub, res, seed=0, accel=1, **kw)
        rs = np.random.RandomState(3)
        data = t_con * (1.0 + 0.3 * rs.standard_normal(t_con.shape))
        weight = 0.5 + rs.random_sample(t_con.shape)
        t_con2, g_con, _ = orc.render_gradient(origin, normal, v, f, 20000, lb, ub, res, data, weight, seed=0, accel=1, **kw)
        assert rel_l2(t_con, t_con2) <= 1e-13          # (thread reduction order)
        with orc.rule_free():       # one call: rows and gradient of the rule-free all-faces definition
            t_free, g_free, _ = orc.render_gradient(origin, normal, v, f, 20000, lb, ub, res, data, weight, seed=0, accel=0, **kw)
        rows = np.linalg.norm(t_con - t_free, axis=1) / np.linalg.norm(t_free, axis=1)
        assert rel_l2(t_con, t_free) <= 1e-6, name
        assert rows.max() <= 1e-5, name
        assert np.abs(t_con - t_free).max() <= 1e-6 * t_free.max(), name
        assert rel_l2(g_con, g_free) <= 1e-6, name
        assert t_free.sum() > 0


def test_row_e_prim_ids_against_the_rule_free_definition_and_direction_length(orc, bunny):
    """Row E (embree_intersector): how many primIDs does the grazing rule change on 1e5 random rays (the reference hands
    the rays to Embree, which has no rule), and -- ADVICE round 2 -- the rule must not depend on |d|: the reference
    passes directions of any length (EMB/c_embree_intersector.cpp:20-45)."""
    v, f = bunny
    rs = np.random.RandomState(17)
    n = 100000
    o = np.zeros((n, 3), np.float32)
    o[:, :2] = rs.uniform(-0.3, 0.3, (n, 2))
    tgt = v[rs.randint(0, v.shape[0], n)] + rs.normal(0, 0.01, (n, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    a = orc.intersect(o, d, v, f, accel=1, short=True)
    with orc.rule_free():
        b = orc.intersect(o, d, v, f, accel=0, short=True)
    differing = int((a != b).sum())
    assert differing <= 2, differing            # measured: 0 of 100 000 at the 2^-10 cut-off
    # powers of two scale every product exactly: identical decisions, identical barycentrics
    full = orc.intersect(o, d, v, f, accel=1)
    for s in (2.0 ** -7, 2.0 ** 7):
        sc = orc.intersect(o, (d * np.float32(s)).astype(np.float32), v, f, accel=1)
        assert np.array_equal(np.nan_to_num(sc, nan=-7), np.nan_to_num(full, nan=-7))
    # any other length: the same hits up to the rounding of the scaled direction (never "all rays miss")
    for s in (0.01, 100.0):
        sc = orc.intersect(o, (d * np.float32(s)).astype(np.float32), v, f, accel=1, short=True)
        assert (sc != a).mean() < 1e-3
        assert (sc >= 0).mean() > 0.5


def test_flipped_plane_is_dark(orc, cfg1):
    c = cfg1
    tr, _ = orc.render_transient(c["origin"], c["normal"], c["v"], np.ascontiguousarray(c["f"][:, [0, 2, 1]]), 256,
                                 c["lb"], c["ub"], c["res"])
    assert np.all(tr == 0)
    # v1 forward has no clamp: back faces contribute ff^2
    tr1, _ = orc.render_transient(c["origin"], c["normal"], c["v"], np.ascontiguousarray(c["f"][:, [0, 2, 1]]), 256,
                                  c["lb"], c["ub"], c["res"], clamp=0)
    tr2, _ = orc.render_transient(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"])
    # (the flipped winding permutes the stratified sample map, so only the mass is comparable)
    assert tr1.sum() > 0 and abs(tr1.sum() - tr2.sum()) < 0.25 * tr2.sum()


def test_source_sharding_is_exact(orc, bunny):
    v, f = bunny
    origin, normal = grid_sources(3, 0.2)
    kw = dict(refine=10, sigma_bin=1, accel=1, threads=1)
    rs = np.random.RandomState(0)
    data = rs.random_sample((9, 512)) * 1e-3
    w = np.ones((9, 512))
    t_all, g_all, _ = orc.render_gradient(origin, normal, v, f, 8000, 0.625, 1.625, 2.0 ** -9, data, w, **kw)
    g_sum = np.zeros_like(g_all)
    rows = []
    for lo, hi in ((0, 4), (4, 9)):
        t, g, _ = orc.render_gradient(origin[lo:hi], normal[lo:hi], v, f, 8000, 0.625, 1.625, 2.0 ** -9,
                                      data[lo:hi], w[lo:hi], source_offset=lo, total_sources=9, **kw)
        rows.append(t)
        g_sum += g
    assert np.array_equal(np.vstack(rows), t_all)
    assert rel_l2(g_sum, g_all) < 1e-13


def test_gradient_formula_against_finite_differences(orc):
    """With the normal term on and a large refine_scale the analytic gradient converges to the
    finite-difference gradient of the Gaussian-smoothed loss.  sigma_bin >= 5
    makes the forward transient use the same Gaussian as the gradient taps, so
    loss(v) = (1/L) sum w (data - T(v))^2 is exactly what the gradient differentiates."""
    v = np.array([[-.11, -.07, .42], [.12, -.09, .47], [.02, .13, .40]], np.float32)
    f = np.array([[0, 2, 1]], np.int32)
    origin = np.array([[0.05, -0.02, 0], [-0.15, 0.1, 0]], np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (2, 1))
    lb, ub, res, ns = 0.5, 1.5, 2.0 ** -6, 64
    R, SB = 24, 5
    rs = np.random.RandomState(2)
    data = rs.random_sample((2, 64)) * 0.02
    w = 0.5 + rs.random_sample((2, 64))

    def loss(vv):
        t, _ = orc.render_transient(origin, normal, vv, f, ns, lb, ub, res, refine=R, sigma_bin=SB, threads=1)
        return float(np.sum(w * (data - t) ** 2) / origin.shape[0])

    _, g, _ = orc.render_gradient(origin, normal, v, f, ns, lb, ub, res, data, w, refine=R, sigma_bin=SB,
                                  testing_flag=0, normal_term=1, threads=1)
    _, g_off, _ = orc.render_gradient(origin, normal, v, f, ns, lb, ub, res, data, w, refine=R, sigma_bin=SB,
                                      testing_flag=0, normal_term=0, threads=1)
    fd = np.zeros((3, 3))
    eps = 1e-3          # the reference's own step (paper_fig/finite_diff.py:86-98); smaller steps resolve the sub-bin staircase
    for i in range(3):
        for c in range(3):
            vp, vm = v.astype(np.float64).copy(), v.astype(np.float64).copy()
            vp[i, c] += eps
            vm[i, c] -= eps
            fd[i, c] = (loss(vp.astype(np.float32)) - loss(vm.astype(np.float32))) / (
                float(np.float32(vp[i, c])) - float(np.float32(vm[i, c])))
    err_on, err_off = rel_l2(g, fd), rel_l2(g_off, fd)
    assert err_on < 0.03, (err_on, err_off)
    assert err_on < err_off      # dropping the normal term (v2 default for face normals) is further from FD


def test_scalar_gradients_against_finite_differences(orc, cfg1):
    c = cfg1
    rs = np.random.RandomState(4)
    data = rs.random_sample((4, 64)) * 0.3
    w = np.ones((4, 64))
    kw = dict(refine=10, sigma_bin=5, threads=1)
    V = c["v"].shape[0]

    def loss_alb(a):
        t, _ = orc.render_transient(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                    refine=10, sigma_bin=5, albedo=np.full(V, a, np.float32), threads=1)
        return float(np.sum(w * (data - t) ** 2) / 4)

    _, ga = orc.render_gradient_scalar(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                       data, w, albedo=np.full(V, 0.8, np.float32), **kw)
    fd = (loss_alb(0.81) - loss_alb(0.79)) / (float(np.float32(0.81)) - float(np.float32(0.79)))
    assert abs(ga - fd) <= 0.02 * abs(fd)

    def loss_alpha(al):
        t, _ = orc.render_transient(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                    refine=10, sigma_bin=5, ggx_alpha=al, threads=1)
        return float(np.sum(w * (data - t) ** 2) / 4)

    _, gal = orc.render_gradient_scalar(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                        data, w, wrt_alpha=True, ggx_alpha=0.3, **kw)
    fd = (loss_alpha(0.305) - loss_alpha(0.295)) / (float(np.float32(0.305)) - float(np.float32(0.295)))
    assert abs(gal - fd) <= 0.03 * abs(fd)


def test_ggx_table_and_derivatives(orc):
    g = np.load(os.path.join(GOLDEN, "ggx_table.npz"))
    n = np.array([0, 0, 1], np.float32)
    for i, a in enumerate(g["alpha"]):
        for j, c in enumerate(g["nw"]):
            s = np.sqrt(max(0.0, 1.0 - float(c) ** 2))
            w = np.array([s, 0, c], np.float32)
            assert orc.ggx(a, n, w, "eval") == g["eval"][i, j]
            assert orc.ggx(a, n, w, "adiff") == g["adiff"][i, j]
            assert orc.ggx(a, n, w, "nwsdiff") == g["nwsdiff"][i, j]
    # early-outs (ggx_confocal.cpp:15-17): back side is black
    assert np.all(g["eval"][:, g["nw"] <= 0] == 0)
    # d/d alpha against finite differences of eval
    w = np.array([0.6, 0, 0.8], np.float32)
    for a in (0.2, 0.4, 0.7):
        fd = (orc.ggx(a + 1e-3, n, w) - orc.ggx(a - 1e-3, n, w)) / 2e-3
        assert abs(orc.ggx(a, n, w, "adiff") - fd) <= 2e-2 * abs(fd)


def test_v1_residual_box_filter_and_shapes(orc, cfg1):
    c = cfg1
    rs = np.random.RandomState(6)
    data = rs.random_sample((4, 64)) * 0.2
    t0, g0, _ = orc.render_gradient_v1(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"], data, 0)
    t2, g2, _ = orc.render_gradient_v1(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"], data, 2)
    assert np.array_equal(t0, t2) and g0.shape == (4, 3)
    assert rel_l2(g0, g2) > 1e-3          # smoothing the residual changes the gradient
    # with w_width = 0 v1 == v2 kernel with one unit tap and the normal term on: compare to the v2
    # oracle at sigma_bin -> tiny kernel is not available, so check the symmetric structure instead
    assert np.all(np.isfinite(g0)) and np.abs(g0).max() > 0


def test_intensity_equals_row_mass(orc, bunny):
    """sum_f intensity[f] == sum_{l,b} transient[l,b]: same samples, same acceptance, no binning."""
    v, f = bunny
    origin, normal = grid_sources(2, 0.2)
    t, _ = orc.render_transient(origin, normal, v, f, 9000, 0.625, 1.625, 2.0 ** -9, accel=1)
    inten = orc.render_intensity(origin, normal, v, f, 9000, 0.625, 1.625, accel=1)
    assert abs(inten.sum() - t.sum()) <= 1e-12 * t.sum()


def test_vertex_gradient_sums_to_full_gradient_structure(orc, cfg1):
    """The single-vertex per-bin gradient (paper_fig/finite_diff.py) contracted with -2*diff equals
    the corresponding row of the full vertex gradient when both use the normal term."""
    c = cfg1
    rs = np.random.RandomState(8)
    data = rs.random_sample((1, 64)) * 0.3
    w = np.ones((1, 64))
    o, n = c["origin"][:1], c["normal"][:1]
    t, g, _ = orc.render_gradient(o, n, c["v"], c["f"], 256, c["lb"], c["ub"], c["res"], data, w, refine=10,
                                  sigma_bin=1, normal_term=1, threads=1)
    diff = (data - t) * w
    for vert in range(4):
        pv = orc.render_vertex_gradient(vert, o, n, c["v"], c["f"], 256, c["lb"], c["ub"], c["res"], refine=10,
                                        sigma_bin=1, threads=1)
        contracted = (-2 * diff[0][:, None] * pv).sum(axis=0)
        assert np.allclose(contracted, g[vert], rtol=2e-5, atol=1e-9 * np.abs(g).max())


# ------------------------------------------------------------------ row N (non-confocal pairs)
def test_nonconfocal_reduces_to_confocal(orc, bunny):
    """laser == sensor must reproduce rows F/G: forward bit for bit, gradient to fp32 rounding."""
    v, f = bunny
    origin, normal = grid_sources(3, 0.2)
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -9, 20000
    rs = np.random.RandomState(5)
    t0, _ = orc.render_transient(origin, normal, v, f, ns, lb, ub, res, accel=1, threads=1)
    data = t0 * (1 + 0.3 * rs.standard_normal(t0.shape))
    w = 0.5 + rs.random_sample(t0.shape)
    _, g0, _ = orc.render_gradient(origin, normal, v, f, ns, lb, ub, res, data, w, accel=1, threads=1)
    t1, g1, _ = orc.render_nonconfocal(origin, normal, origin, normal, v, f, ns, lb, ub, res, data=data,
                                       weight=w, accel=1, threads=1)
    assert np.array_equal(t0, t1)
    assert rel_l2(g1, g0) < 1e-6


def test_nonconfocal_reciprocity_and_path_length(orc, bunny):
    """Swapping laser and sensor leaves the transient unchanged (Helmholtz reciprocity of the
    Lambertian three-bounce path) up to the fp32 rounding of the resampled hit point, and moving
    the sensor away lengthens the paths."""
    v, f = bunny
    a, n = grid_sources(3, 0.2)
    b = a.copy()
    b[:, 0] += 0.09
    b[:, 1] -= 0.06
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -9, 20000
    tab, _, path = orc.render_nonconfocal(a, n, b, n, v, f, ns, lb, ub, res, accel=1)
    tba, _, _ = orc.render_nonconfocal(b, n, a, n, v, f, ns, lb, ub, res, accel=1)
    assert tab.sum() > 0
    # same accepted set except where rounding flips an edge sample; compare cumulative rows
    assert np.abs(np.cumsum(tab, 1) - np.cumsum(tba, 1)).max() < 1e-3 * tab.sum(1).max()
    far = a.copy()
    far[:, 0] += 0.4
    tfar, _, _ = orc.render_nonconfocal(a, n, far, n, v, f, ns, lb, ub, res, accel=1)
    taa, _, _ = orc.render_nonconfocal(a, n, a, n, v, f, ns, lb, ub, res, accel=1)
    mean_bin = lambda t: (t * np.arange(t.shape[1])).sum() / t.sum()
    assert mean_bin(tfar) > mean_bin(taa)


def test_nonconfocal_matches_reference_prototype(orc):
    """The reference's numpy prototype (transient_rendering_python/rendering.py:8-93) run with
    lighting != sensor: nearest hit from the laser, second-segment visibility from the sensor,
    bin = ceil((d1+d2)/res)-1, weight cos(theta2)/d2^2 -- reproduced from the oracle's closest-hit
    primitive with the acceptance rule row N uses (closest hit from BOTH end points is the face)."""
    g = np.load(os.path.join(GOLDEN, "pyref_angular_nc.npz"))
    for name in ("toy", "occluder"):
        v, f, d = g[name + "_v"], g[name + "_f"], g[name + "_dir"]
        nbin, res = int(g[name + "_nbin"]), float(g[name + "_res"])
        fn = g[name + "_fn"]
        for k in range(g[name + "_laser"].shape[0]):
            a, b = g[name + "_laser"][k], g[name + "_sensor"][k]
            hit = orc.intersect(np.tile(a, (d.shape[0], 1)), d, v, f, accel=0)
            prim = hit[:, 0].astype(int)
            ok = prim >= 0
            pts = orc.barycentric_to_world(v, f, hit.copy()).astype(np.float64)
            d1 = np.linalg.norm(pts - a, axis=1)
            v2 = b - pts
            d2 = np.linalg.norm(v2, axis=1)
            v2 = v2 / d2[:, None]
            # sensor leg: accepted iff the closest hit from the sensor towards the point is the same face
            back = orc.intersect(np.tile(b, (d.shape[0], 1)), -v2, v, f, accel=0)
            vis = ok & (back[:, 0].astype(int) == prim)
            cos = np.einsum("ij,ij->i", fn[np.maximum(prim, 0)], v2)
            cos[cos < 0] = 0
            bn = np.ceil((d1 + d2) / res) - 1
            keep = vis & (bn <= nbin)
            t = np.zeros(nbin + 1)
            np.add.at(t, bn[keep].astype(int), (cos / d2 ** 2)[keep])
            t = t[:nbin] * 2 * np.pi / d.shape[0]
            ref = g[name + "_transient"][k]
            wmax = (cos / d2 ** 2)[keep].max() * 2 * np.pi / d.shape[0]
            # a sample exactly on an occluder's silhouette / bin edge may differ (fp32 vs fp64 points)
            assert abs(t.sum() - ref.sum()) <= 2.0 * wmax + 1e-5 * ref.sum(), (name, k, t.sum(), ref.sum())
            assert np.abs(np.cumsum(t) - np.cumsum(ref)).max() <= 3.0 * wmax
        assert g[name + "_transient"].sum() > 0


def test_nonconfocal_gradient_against_finite_differences(orc):
    """Same construction as the confocal FD test: sigma_bin >= 5 smooths the forward rows with the
    Gaussian the gradient taps use, so the analytic gradient must match central differences."""
    v = np.array([[-.11, -.07, .42], [.12, -.09, .47], [.02, .13, .40]], np.float32)
    f = np.array([[0, 2, 1]], np.int32)
    a = np.array([[0.05, -0.02, 0], [-0.15, 0.1, 0]], np.float32)
    b = np.array([[-0.12, 0.08, 0], [0.2, -0.05, 0]], np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (2, 1))
    lb, ub, res, ns = 0.5, 1.5, 2.0 ** -6, 1024
    R, SB = 24, 5
    rs = np.random.RandomState(3)
    data = rs.random_sample((2, 64)) * 0.02
    w = 0.5 + rs.random_sample((2, 64))

    def loss(vv):
        t, _, _ = orc.render_nonconfocal(a, n, b, n, vv, f, ns, lb, ub, res, refine=R, sigma_bin=SB, threads=1)
        return float(np.sum(w * (data - t) ** 2) / a.shape[0])

    _, g, _ = orc.render_nonconfocal(a, n, b, n, v, f, ns, lb, ub, res, data=data, weight=w, refine=R,
                                     sigma_bin=SB, testing_flag=0, normal_term=1, threads=1)
    fd = np.zeros((3, 3))
    eps = 1e-3
    for i in range(3):
        for c in range(3):
            vp, vm = v.astype(np.float64).copy(), v.astype(np.float64).copy()
            vp[i, c] += eps
            vm[i, c] -= eps
            fd[i, c] = (loss(vp.astype(np.float32)) - loss(vm.astype(np.float32))) / (
                float(np.float32(vp[i, c])) - float(np.float32(vm[i, c])))
    assert rel_l2(g, fd) < 0.02, (rel_l2(g, fd), g, fd)     # 0.9 % here; 0.3 % at refine 48, sigma_bin 8


def test_nonconfocal_ggx_reduces_to_confocal_and_matches_finite_differences(orc, bunny):
    """GGX for (laser, sensor) pairs: brdf = D(n.h) G1(n.wa) G1(n.wb) / 4 with the half vector h.  (a) sensor ==
    laser gives the confocal GGX rows (ggx/transient_and_gradient.cpp:236; h = w up to rounding); (b) the analytic
    vertex gradient matches central differences of the Gaussian-smoothed loss (the confocal GGX gradient mirrors the reference's BRDF_dx, whose first term lacks the 1/h of the chain
    rule, so it is the pair formula -- derived here, no reference exists -- that is checked against differences)."""
    v, f = bunny
    origin, normal = grid_sources(2, 0.2)
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -9, 20000
    for alpha in (0.15, 0.5):
        t0, _ = orc.render_transient(origin, normal, v, f, ns, lb, ub, res, ggx_alpha=alpha, accel=1)
        t1, _, _ = orc.render_nonconfocal(origin, normal, origin, normal, v, f, ns, lb, ub, res, refine=1, ggx_alpha=alpha, accel=1)
        assert t0.sum() > 0 and rel_l2(t1, t0) < 1e-5
    v3 = np.array([[-.11, -.07, .42], [.12, -.09, .47], [.02, .13, .40]], np.float32)
    f3 = np.array([[0, 2, 1]], np.int32)
    a = np.array([[0.05, -0.02, 0], [-0.15, 0.1, 0]], np.float32)
    b = np.array([[-0.12, 0.08, 0], [0.2, -0.05, 0]], np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (2, 1))
    lb, ub, res, ns = 0.5, 1.5, 2.0 ** -6, 1024
    R, SB = 24, 5
    rs = np.random.RandomState(3)
    data = rs.random_sample((2, 64)) * 0.02
    w = 0.5 + rs.random_sample((2, 64))
    for alpha in (0.3, 0.8):         # face normals, normal term on: the setting in which differences validate the formulas
        kw = dict(ggx_alpha=alpha, threads=1)

        def loss(vv):
            t, _, _ = orc.render_nonconfocal(a, n, b, n, vv, f3, ns, lb, ub, res, refine=R, sigma_bin=SB, **kw)
            return float(np.sum(w * (data - t) ** 2) / a.shape[0])

        _, g, _ = orc.render_nonconfocal(a, n, b, n, v3, f3, ns, lb, ub, res, data=data, weight=w, refine=R,
                                         sigma_bin=SB, testing_flag=0, normal_term=1, **kw)
        fd = np.zeros((3, 3))
        eps = 1e-3
        for i in range(3):
            for c in range(3):
                vp, vm = v3.astype(np.float64).copy(), v3.astype(np.float64).copy()
                vp[i, c] += eps
                vm[i, c] -= eps
                fd[i, c] = (loss(vp.astype(np.float32)) - loss(vm.astype(np.float32))) / (
                    float(np.float32(vp[i, c])) - float(np.float32(vm[i, c])))
        assert np.abs(fd).max() > 0 and rel_l2(g, fd) < 0.02, (alpha, rel_l2(g, fd), g, fd)    # 0.8 % / 0.6 %


# ------------------------------------------------------------------ jitter/ module
def test_jitter_forward_is_histogram_convolved_with_kernel(orc, bunny):
    """jitter/transient_and_gradient.cpp:331-347 on the reference's own measured kernel
    (jitter/jitter_info.mat, committed as data): row = full conv(histogram, w)[offset : offset+T]."""
    v, f = bunny
    j = np.load(os.path.join(GOLDEN, "jitter_info.npz"))
    jw, jo = j["jitter_weight"], int(j["jitter_offset"])
    assert jw.shape == (40, 1) and jo == 8
    # what jitter_grad is: the kernel's derivative per tap (central differences, to ~1 % of its peak)
    assert np.abs(j["jitter_grad"].ravel() - np.gradient(jw.ravel()))[1:-1].max() < 0.02 * np.abs(j["jitter_grad"]).max()
    o, n = grid_sources(2, 0.1)
    lb, ub, res = 0.0, float(np.float32(1200 * 0.0012)), 0.0012                           # jitter/test.py:41-45
    t0, _ = orc.render_transient(o, n, v, f, 20000, lb, ub, res, accel=1)
    tj, _, path = orc.render_jitter(o, n, v, f, 20000, lb, ub, res, jw, jo, accel=1)
    assert t0.shape == (4, 1200)
    for l in range(4):
        y = np.convolve(t0[l], jw.ravel())
        assert np.allclose(tj[l], y[jo:jo + 1200], rtol=0, atol=1e-15)
    td, _, _ = orc.render_jitter(o, n, v, f, 20000, lb, ub, res, np.array([[1.0]]), 0, accel=1)
    assert np.abs(td - t0).max() <= 1e-16                                                  # delta kernel (jitter/test.py:60-62)


def test_jitter_gradient_against_finite_differences(orc):
    """With a wide smooth kernel w and jitter_grad = dw/d(tap) the analytic gradient matches central
    differences of loss = sum w (data - conv(hist, w))^2 / L; without the jitter_grad term it does not."""
    v = np.array([[-.11, -.07, .42], [.12, -.09, .47], [.02, .13, .40]], np.float32)
    f = np.array([[0, 2, 1]], np.int32)
    a = np.array([[0.05, -0.02, 0], [-0.15, 0.1, 0]], np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (2, 1))
    lb, ub, res, ns = 0.5, 1.5, 2.0 ** -8, 8192
    T = orc.num_bins(lb, ub, res)
    rs = np.random.RandomState(3)
    data = rs.random_sample((2, T)) * 0.02
    w = 0.5 + rs.random_sample((2, T))
    sig = 24
    x = np.arange(8 * sig + 1) - 4 * sig
    jw = np.exp(-x ** 2 / (2 * sig ** 2)) / (sig * np.sqrt(2 * np.pi))
    jg = -x / sig ** 2 * jw

    def loss(vv):
        t, _, _ = orc.render_jitter(a, n, vv, f, ns, lb, ub, res, jw, 4 * sig)
        return float(np.sum(w * (data - t) ** 2) / 2)

    _, g, _ = orc.render_jitter(a, n, v, f, ns, lb, ub, res, jw, 4 * sig, jitter_grad=jg, data=data, weight=w,
                                testing_flag=0, normal_term=1)
    _, g0, _ = orc.render_jitter(a, n, v, f, ns, lb, ub, res, jw, 4 * sig, jitter_grad=0 * jg, data=data, weight=w,
                                 testing_flag=0, normal_term=1)
    fd = np.zeros((3, 3))
    eps = 1e-3
    for i in range(3):
        for c in range(3):
            vp, vm = v.astype(np.float64).copy(), v.astype(np.float64).copy()
            vp[i, c] += eps
            vm[i, c] -= eps
            fd[i, c] = (loss(vp.astype(np.float32)) - loss(vm.astype(np.float32))) / (
                float(np.float32(vp[i, c])) - float(np.float32(vm[i, c])))
    assert rel_l2(g, fd) < 0.02, rel_l2(g, fd)                 # 0.4 % here
    assert rel_l2(g0, fd) > 4 * rel_l2(g, fd)


# ------------------------------------------------------------------ mesh regularisers
def test_regularisers_against_finite_differences(orc, bunny):
    """Accumulated per-face terms are exact gradients: the "curvature" gradient is d(total area)/dv,
    and the normal-smoothing gradient is d/dv sum_f A_f (1 - nbar_f . n_f) with nbar held fixed."""
    from nlos_surface_optimization_amd import mesh_io
    v, f = bunny
    aff = mesh_io.face_affinity(f)
    assert aff.shape == f.shape and aff.dtype == np.int32
    g = aff[aff >= 0]
    assert g.size > 0.9 * aff.size                      # mostly manifold
    k = np.nonzero(aff[:, 0] >= 0)[0][:200]
    assert all(i in aff[aff[i, 0]] for i in k)          # neighbour relation is symmetric
    vd = v.astype(np.float64)

    def face_terms(vv):
        p0, p1, p2 = vv[f[:, 0]], vv[f[:, 1]], vv[f[:, 2]]
        nr = np.cross(p1 - p0, p2 - p0)
        A = 0.5 * np.linalg.norm(nr, axis=1)
        return nr / (2 * A[:, None]), A

    n0, A0 = face_terms(vd)
    nbar = n0 * A0[:, None]
    wsum = A0.copy()
    for c in range(3):
        ok = aff[:, c] >= 0
        nbar[ok] += (n0 * A0[:, None])[aff[ok, c]]
        wsum[ok] += A0[aff[ok, c]]
    ln = np.linalg.norm(nbar, axis=1)
    live = ln > 1e-3 * wsum                 # cancelling neighbourhoods (flipped twin faces of the fixture) are skipped
    nbar[live] /= ln[live, None]

    def smooth(vv):
        n, A = face_terms(vv)
        return float((A * (1 - np.einsum("ij,ij->i", nbar, n)))[live].sum())

    val, gs = orc.mesh_regulariser(v, f, aff)
    zero, ga = orc.mesh_regulariser(v, f, None)
    assert zero == 0.0 and abs(val - smooth(vd)) < 1e-6 * val
    rs = np.random.RandomState(0)
    for i in rs.randint(0, v.shape[0], 12):
        for c in range(3):
            vp, vm = vd.copy(), vd.copy()
            vp[i, c] += 1e-6
            vm[i, c] -= 1e-6
            fa = (face_terms(vp)[1].sum() - face_terms(vm)[1].sum()) / 2e-6
            fs = (smooth(vp) - smooth(vm)) / 2e-6
            assert abs(fa - ga[i, c]) < 2e-6 * np.abs(ga).max() + 1e-9
            assert abs(fs - gs[i, c]) < 1e-4 * np.abs(gs).max() + 1e-9
    # the reference's `=` stores: every vertex keeps the term of its highest incident face only
    _, go = orc.mesh_regulariser(v, f, None, overwrite=True)
    last = np.full(v.shape[0], -1)
    for c in range(3):
        np.maximum.at(last, f[:, c], np.arange(f.shape[0]))
    i = 77
    fi = last[i]
    j = list(f[fi]).index(i)
    e = [vd[f[fi, 2]] - vd[f[fi, 1]], vd[f[fi, 0]] - vd[f[fi, 2]], vd[f[fi, 1]] - vd[f[fi, 0]]][j]
    assert np.allclose(go[i], np.cross(n0[fi], e / 2), rtol=0, atol=1e-7)
    assert not np.allclose(go, ga)


def test_product_is_its_pairs_on_shared_samples(orc, mannequin):
    """Row N as a product (north_star's L x S x T): defined as the enumerated pairs on sample points shared by all wall
    points (include/nlos_hip.h, nlos_render_args.n_sensors).  With shared samples a wall point's leg does not depend on
    who it is paired with, so: the diagonal of a set paired with itself is its confocal render, bit for bit; exchanging
    laser and sensor changes a row only by the fp32 rounding of the resampled hit point; and a plain (unshared)
    render differs -- the keys really are shared."""
    v, f = mannequin
    la = np.array([[0.1, 0, 0], [-0.2, 0.1, 0]], np.float32)
    sb = np.array([[0.1, 0, 0], [0.3, -0.1, 0], [-0.2, 0.1, 0]], np.float32)
    n2, n3 = np.tile(np.array([0, 0, 1], np.float32), (2, 1)), np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    lb, ub, res, ns = 0.0, 2.0, 2.0 ** -9, 4000
    t, g, path = orc.render_product(la, n2, sb, n3, v, f, ns, lb, ub, res, accel=1, threads=1)
    assert t.shape == (2, 3, 1024) and g is None and path.shape == (1024,) and (t.sum(axis=2) > 0).all()
    # pair by pair
    for i in range(2):
        for j in range(3):
            tp, _, _ = orc.render_nonconfocal(la[i:i + 1], n2[:1], sb[j:j + 1], n3[:1], v, f, ns, lb, ub, res, refine=1,
                                              accel=1, shared_samples=1, threads=1)
            assert np.array_equal(tp[0], t[i, j])
    # laser 0 == sensor 0 and laser 1 == sensor 2: the confocal rows of those wall points (shared samples)
    tc, _ = orc.render_transient(la, n2, v, f, ns, lb, ub, res, accel=1, shared_samples=1, threads=1)
    assert np.array_equal(t[0, 0], tc[0]) and np.array_equal(t[1, 2], tc[1])
    tc_plain, _ = orc.render_transient(la, n2, v, f, ns, lb, ub, res, accel=1, threads=1)
    assert np.array_equal(tc_plain[0], tc[0]) and not np.array_equal(tc_plain[1], tc[1])      # source 0's keys are the shared ones
    # reciprocity: (laser 0, sensor 2) is (laser 1, sensor 0) with the legs exchanged
    assert rel_l2(t[0, 2], t[1, 0]) < 1e-6 and not np.array_equal(t[0, 2], t[0, 1])
    # gradient: normalised by the L * S measurements, the sum of the pairs' contributions
    data = t * 1.25
    _, g, _ = orc.render_product(la, n2, sb, n3, v, f, ns, lb, ub, res, data=data, accel=1)
    acc = np.zeros_like(g)
    for i in range(2):
        for j in range(3):
            _, gp, _ = orc.render_nonconfocal(la[i:i + 1], n2[:1], sb[j:j + 1], n3[:1], v, f, ns, lb, ub, res,
                                              data=data[i, j][None], accel=1, shared_samples=1, total_sources=6)
            acc += gp
    assert np.abs(g).max() > 0 and rel_l2(acc, g) < 1e-12
