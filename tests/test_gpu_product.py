"""Row N as a product (north_star's L x S x T histogram): every (laser, sensor) combination of two sets of wall points,
rendered on sample points shared by all wall points (include/nlos_hip.h, nlos_render_args.n_sensors).  The product is
DEFINED as its pairs, so the checker is the pair oracle on the enumerated pairs with shared samples
(oracle.render_product); the record + combine kernels, the enumerated-pairs fallback and the numpy drop-in must all
agree with it: rows to fp64 summation order, gradient to the pair kernels' tolerance."""
import numpy as np
import pytest

from conftest import plane_cfg1, rel_l2

pytestmark = pytest.mark.gpu

LB, UB, RES, T = 0.625, 1.625, 2.0 ** -9, 512


def _wall(points):
    p = np.ascontiguousarray(np.array(points, np.float32))
    return p, np.tile(np.array([0, 0, 1], np.float32), (p.shape[0], 1))


def _dev(*arrays):
    import torch
    return [None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in arrays]


@pytest.fixture()
def renderer():
    import torch
    from nlos_surface_optimization_amd import device as nd
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=11)
    yield r
    r.close()


def test_record_and_combine_kernels_match_the_pair_oracle(orc, bunny, renderer):
    v, f = bunny
    # four lasers, three sensors (one of them also a laser; one wall point INSIDE the scene's depth range: its record
    # pass runs the in-kernel BVH query)
    la, lan = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0], [0.05, -0.25, 0], [0.3, 0.0, 0.45]])
    sb, sbn = _wall([[0.1, 0.0, 0], [0.2, 0.2, 0], [-0.15, -0.05, 0]])
    ns = 3 * f.shape[0]
    t_ref, _, p_ref = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, accel=1, seed=11)
    assert t_ref.shape == (4, 3, T) and (t_ref.sum(axis=2) > 0).sum() >= 9
    tl, tln, ts, tsn, tv, tf = _dev(la, lan, sb, sbn, v, f)
    t, g, p = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES)
    assert g is None and renderer.last_path()["backend"] == "grid"
    assert rel_l2(t.cpu().numpy(), t_ref) < 1e-12 and np.array_equal(p.cpu().numpy(), p_ref)
    # ... and the enumerated pairs (the definition) give the same rows
    t2, _, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, pairs=True)
    assert rel_l2(t2.cpu().numpy(), t_ref) < 1e-12

    # gradient of sum w (data - T)^2 over the 12 measurements
    rs = np.random.RandomState(3)
    data = t_ref * (1 + 0.3 * rs.standard_normal(t_ref.shape))
    w = 0.5 + rs.random_sample(t_ref.shape)
    _, g_ref, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, data=data, weight=w, accel=1, seed=11,
                                     sigma_bin=1, testing_flag=1)
    td, tw = _dev(data, w)
    t3, g3, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, data=td, weight=tw)
    assert rel_l2(t3.cpu().numpy(), t_ref) < 1e-12
    assert rel_l2(g3.cpu().numpy(), g_ref) < 1e-4
    _, g4, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, data=td, weight=tw, pairs=True)
    assert rel_l2(g4.cpu().numpy(), g_ref) < 1e-4 and rel_l2(g4.cpu().numpy(), g3.cpu().numpy()) < 1e-6
    # accumulation into a caller's buffer (v2 semantics)
    import torch
    acc = torch.full_like(g3, 0.5)
    renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, data=td, weight=tw, gradient=acc)
    assert rel_l2((acc - 0.5).cpu().numpy(), g_ref) < 1e-4


def test_one_set_as_lasers_and_sensors_reduces_to_confocal_on_the_diagonal(orc, bunny, renderer):
    """lasers and sensors the SAME array: one record pass serves both roles; the diagonal is the confocal render of
    those wall points on shared samples, bit for bit in the oracle and to summation order on the GPU; the rows are
    symmetric under exchanging the two wall points up to the fp32 rounding of the resampled hit (reciprocity)."""
    v, f = bunny
    w, wn = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0], [0.0, 0.2, 0], [0.25, -0.2, 0], [-0.05, -0.1, 0]])
    ns = 5 * f.shape[0]
    # (threads=1: the confocal oracle reduces per-thread rows, so only a one-thread run is bit-reproducible)
    t_ref, _, _ = orc.render_product(w, wn, w, wn, v, f, ns, LB, UB, RES, accel=1, seed=11, threads=1)
    c_ref, _ = orc.render_transient(w, wn, v, f, ns, LB, UB, RES, accel=1, seed=11, shared_samples=1, threads=1)
    for i in range(5):
        assert np.array_equal(t_ref[i, i], c_ref[i])
    tw_, twn, tv, tf = _dev(w, wn, v, f)
    t, _, _ = renderer.render_product(tw_, twn, tw_, twn, tv, tf, ns, LB, UB, RES)
    t = t.cpu().numpy()
    assert rel_l2(t, t_ref) < 1e-12
    c, _ = renderer.render_transient(tw_, twn, tv, tf, ns, LB, UB, RES, shared_samples=True)
    assert rel_l2(np.stack([t[i, i] for i in range(5)]), c.cpu().numpy()) < 1e-12
    assert rel_l2(t.transpose(1, 0, 2), t) < 1e-5 and not np.array_equal(t[0, 1], t[0, 0])


@pytest.mark.parametrize("case", ["tiny_mesh", "many_strata", "refined_rows", "large_mesh"])
def test_scenes_outside_the_record_pass_are_rendered_as_enumerated_pairs(orc, bunny, renderer, case):
    """F < 64, spt > 32, sigma_bin >= 5 (refined forward rows) and meshes beyond the single-workgroup grid: the same
    entry, the same definition, through the pair path."""
    from nlos_surface_optimization_amd import mesh_io
    v, f = bunny
    la, lan = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0]])
    sb, sbn = _wall([[0.2, 0.2, 0], [-0.15, -0.05, 0], [0.0, 0.0, 0]])
    lb, ub, res, kw, gkw = LB, UB, RES, {}, {}
    if case == "tiny_mesh":
        c1 = plane_cfg1()
        v, f, ns, lb, ub, res = c1["v"], c1["f"], 256, 0.0, 2.0, 2.0 ** -5
    elif case == "many_strata":
        keep = np.arange(0, f.shape[0], 9)
        f = np.ascontiguousarray(f[keep])
        ns = 33 * f.shape[0]
    elif case == "refined_rows":
        ns = 2 * f.shape[0]
        gkw = dict(sigma_bin=5, refine=4)
    else:
        v, f = mesh_io.subdivide(v, f, 1)
        ns = f.shape[0]
    t_ref, _, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, lb, ub, res, accel=1, seed=11)
    tl, tln, ts, tsn, tv, tf = _dev(la, lan, sb, sbn, v, f)
    t, _, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, lb, ub, res)
    assert t_ref.sum() > 0 and rel_l2(t.cpu().numpy(), t_ref) < 1e-12
    if case in ("refined_rows", "tiny_mesh"):
        rs = np.random.RandomState(4)
        data = t_ref * (1 + 0.3 * rs.standard_normal(t_ref.shape))
        tg_ref, g_ref, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, lb, ub, res, data=data, accel=1, seed=11, **gkw)
        td, = _dev(data)
        tg, g, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, lb, ub, res, data=td,
                                           sigma_bin=gkw.get("sigma_bin", 1), refine_scale=gkw.get("refine", 10))
        assert rel_l2(tg.cpu().numpy(), tg_ref) < 1e-12 and rel_l2(g.cpu().numpy(), g_ref) < 1e-4


def test_numpy_drop_in_of_the_product(orc, bunny):
    from nlos_surface_optimization_amd import _lib, renderer as rn
    v, f = bunny
    la, lan = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0], [0.0, 0.2, 0]])
    sb, sbn = _wall([[0.2, 0.2, 0], [-0.15, -0.05, 0]])
    ns = 2 * f.shape[0]
    _lib.lib().nlos_set_default_seed(11)
    try:
        t = np.full((3, 2, T), 7.0)
        path = np.zeros(T)
        rn.renderNonConfocalProductTransient(la, lan, sb, sbn, v, f, ns, LB, UB, RES, t, path)
        t_ref, _, p_ref = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, accel=1, seed=11)
        assert rel_l2(t, t_ref) < 1e-12 and np.array_equal(path, p_ref)
        data, w = t_ref * 1.2, np.ones_like(t_ref)
        g = np.zeros((v.shape[0], 3))
        rn.renderNonConfocalProductGradient(la, lan, sb, sbn, v, f, ns, LB, UB, RES, t, path, g, data, w, 10, 1, 1, 0)
        _, g_ref, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, data=data, weight=w, accel=1, seed=11)
        assert rel_l2(g, g_ref) < 1e-4
        with pytest.raises(AssertionError):
            rn.renderNonConfocalProductTransient(la, lan, sb, sbn, v, f, ns, LB, UB, RES, np.zeros((2, 3, T)), path)
        with pytest.raises(ValueError):
            rn.renderNonConfocalProductTransient(la, lan, sb, sbn, v, f, ns, LB, UB, RES, np.zeros((6, T)), path)
    finally:
        _lib.lib().nlos_set_default_seed(0)


def test_product_of_a_grid_of_wall_points_full_rows(orc, bunny, renderer):
    """8 x 8 wall points as lasers and as sensors (64 x 64 = 4 096 measurements x 512 bins): every row against the pair
    oracle on a sample of the pairs, total mass and symmetry on all of them."""
    v, f = bunny
    g = np.linspace(-0.25, 0.25, 8)
    w, wn = _wall([[x, y, 0] for y in g for x in g])
    ns = 20000
    tw_, twn, tv, tf = _dev(w, wn, v, f)
    t, _, _ = renderer.render_product(tw_, twn, tw_, twn, tv, tf, ns, LB, UB, RES)
    t = t.cpu().numpy()
    assert t.shape == (64, 64, T) and (t.sum(axis=2) > 0).all()
    rs = np.random.RandomState(8)
    li, sj = rs.randint(0, 64, 24), rs.randint(0, 64, 24)
    t_ref, _, _ = orc.render_nonconfocal(w[li], wn[li], w[sj], wn[sj], v, f, ns, LB, UB, RES, refine=1, accel=1, seed=11,
                                         shared_samples=1)
    assert rel_l2(t[li, sj], t_ref) < 1e-12
    assert rel_l2(t.transpose(1, 0, 2), t) < 1e-5


@pytest.mark.parametrize("bad", [dict(refine_scale=0), dict(sigma_bin=0), dict(refine_scale=-3), dict(sigma_bin=-1),
                                 dict(refine_scale=600), dict(lower_bound=1.625, upper_bound=0.625),
                                 dict(resolution=-2.0 ** -9), dict(transient=None), dict(data=None)])
def test_product_gradient_rejects_what_nlos_render_rejects(bunny, renderer, bad):
    """The record + combine fast path does not go through nlos_render's argument checks (round-4 advice): a refine_scale
    or sigma_bin below 1 (division by zero in the tap tables, a negative table size that threw across the C boundary), an
    over-long temporal kernel, an inverted window or a missing output must come back as NLOS_ERR_ARG from the product
    entry as well -- straight through the C ABI, so that nothing on the Python side can catch it first."""
    import ctypes
    import torch
    from nlos_surface_optimization_amd import _lib
    v, f = bunny
    la, lan = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0]])
    sb, sbn = _wall([[0.1, 0.0, 0], [0.2, 0.2, 0]])
    tl, tln, ts, tsn, tv, tf = _dev(la, lan, sb, sbn, v, f)
    a = renderer._args(_lib.MODE_GRADIENT, tl, tln, tv, tf, 3 * f.shape[0], LB, UB, RES, 10, 1)
    trans = torch.zeros((2, 2, T), dtype=torch.float64, device="cuda:0")
    data = torch.zeros((2, 2, T), dtype=torch.float64, device="cuda:0")
    grad = torch.zeros((v.shape[0], 3), dtype=torch.float64, device="cuda:0")
    a.sensor, a.sensor_normal, a.n_sensors = ts.data_ptr(), tsn.data_ptr(), 2
    a.transient, a.data, a.gradient, a.testing_flag = trans.data_ptr(), data.data_ptr(), grad.data_ptr(), 1
    for k, val in bad.items():
        setattr(a, k, val)
    rc = renderer._lib.nlos_render(renderer._h, ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1, (bad, rc)         # NLOS_ERR_ARG
    assert b"nlos_render" in renderer._lib.nlos_last_error()
    # the context is still usable, and the same arguments without the defect render
    t, _, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, 3 * f.shape[0], LB, UB, RES)
    assert float(t.sum()) > 0


@pytest.mark.parametrize("case", ["vertex_normals", "albedo", "both", "both_testing_flag_0"])
def test_shading_normals_and_albedo_through_the_extended_records(orc, bunny, renderer, case):
    """Round 6: the pair's normal and albedo are interpolated at the LASER leg's hit, so the sensor's form factor depends on
    the laser; the records now carry what makes the legs separable again (laser: normal + albedo + its own form factor,
    sensor: unit direction + length) and such scenes stay on the record + combine kernels -- same rows as the enumerated
    pairs (the definition) and as the pair oracle, gradient to the pair kernels' tolerance."""
    from conftest import vertex_normals
    v, f = bunny
    la, lan = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0], [0.05, -0.25, 0], [0.3, 0.0, 0.45]])
    sb, sbn = _wall([[0.1, 0.0, 0], [0.2, 0.2, 0], [-0.15, -0.05, 0]])
    ns = 3 * f.shape[0]
    rs = np.random.RandomState(8)
    vn = vertex_normals(v, f) if case != "albedo" else None
    if vn is not None:       # shading normals that differ from the geometric ones
        vn = vn + 0.15 * rs.standard_normal(vn.shape).astype(np.float32)
        vn = np.ascontiguousarray(vn / np.linalg.norm(vn, axis=1, keepdims=True), np.float32)
    alb = np.ascontiguousarray(0.3 + 0.7 * rs.random_sample(v.shape[0]), np.float32) if case != "vertex_normals" else None
    tf_flag = 0 if case == "both_testing_flag_0" else 1
    okw = dict(vnormal=vn, albedo=alb)
    t_ref, _, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, accel=1, seed=11, **okw)
    t_plain, _, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, accel=1, seed=11)
    assert t_ref.sum() > 0 and rel_l2(t_ref, t_plain) > 1e-3          # the features do change the rows
    tl, tln, ts, tsn, tv, tf, tvn, talb = _dev(la, lan, sb, sbn, v, f, vn, alb)
    gkw = dict(vertex_normal=tvn, albedo=talb)
    t, _, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, **gkw)
    p = renderer.last_path()
    assert p["backend"] == "grid"
    assert rel_l2(t.cpu().numpy(), t_ref) < 1e-12
    t2, _, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, pairs=True, **gkw)
    assert rel_l2(t2.cpu().numpy(), t_ref) < 1e-12
    data = t_ref * (1 + 0.3 * rs.standard_normal(t_ref.shape))
    w = 0.5 + rs.random_sample(t_ref.shape)
    _, g_ref, _ = orc.render_product(la, lan, sb, sbn, v, f, ns, LB, UB, RES, data=data, weight=w, accel=1, seed=11, sigma_bin=1,
                                     testing_flag=tf_flag, **okw)
    td, tw = _dev(data, w)
    t3, g3, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, data=td, weight=tw, testing_flag=tf_flag, **gkw)
    assert rel_l2(t3.cpu().numpy(), t_ref) < 1e-12 and rel_l2(g3.cpu().numpy(), g_ref) < 1e-4
    _, g4, _ = renderer.render_product(tl, tln, ts, tsn, tv, tf, ns, LB, UB, RES, data=td, weight=tw, testing_flag=tf_flag, pairs=True, **gkw)
    assert rel_l2(g4.cpu().numpy(), g3.cpu().numpy()) < 1e-6
    # one set in both roles: a single record pass serves lasers and sensors
    ws, wsn = _wall([[0.1, 0.0, 0], [-0.2, 0.1, 0], [0.0, 0.2, 0]])
    s_ref, _, _ = orc.render_product(ws, wsn, ws, wsn, v, f, ns, LB, UB, RES, accel=1, seed=11, **okw)
    tws, twsn = _dev(ws, wsn)
    s_gpu, _, _ = renderer.render_product(tws, twsn, tws, twsn, tv, tf, ns, LB, UB, RES, **gkw)
    assert rel_l2(s_gpu.cpu().numpy(), s_ref) < 1e-12
