"""BASELINE.json's configurations at their REAL sizes against the live CPU oracle (all host threads), through the
device-pointer C ABI.  These are the renders where the paths that only matter at large L run for real -- thousands
of workgroups taking tickets, in-workgroup coarsening restarts, the big-LDS second launch, persistent gradient
workgroups striding over sources, LDS-accumulator flushes -- and every row and the vertex gradient are compared
(reference rows: smoothed_transient/transient_and_gradient.cpp:122-237 forward, :843-1007 gradient).

Tolerances: transient rel-L2 <= 1e-5 and max-abs <= 1e-6 * max, gradient rel-L2 <= 1e-4.
The oracle needs 5-40 s per configuration on the GPU box's host cores."""
import numpy as np
import pytest

from conftest import grid_sources, rel_l2

pytestmark = pytest.mark.gpu


def _check(t_gpu, t_ref, g_gpu=None, g_ref=None):
    assert t_ref.sum() > 0
    assert rel_l2(t_gpu, t_ref) <= 1e-5
    assert np.abs(t_gpu - t_ref).max() <= 1e-6 * np.abs(t_ref).max()
    # per row, too: a single wrong source would drown in the L2 norm of 4096 rows
    num = np.linalg.norm(t_gpu - t_ref, axis=1)
    den = np.linalg.norm(t_ref, axis=1)
    assert (num <= 1e-5 * np.maximum(den, 1e-30) + 1e-18).all(), "worst row %d" % int(np.argmax(num / np.maximum(den, 1e-30)))
    if g_ref is not None:
        assert np.abs(g_ref).max() > 0 and rel_l2(g_gpu, g_ref) <= 1e-4


def _noisy_data(t_ref, seed):
    rs = np.random.RandomState(seed)
    return np.ascontiguousarray(t_ref * (1 + 0.25 * rs.standard_normal(t_ref.shape)))


def _tensors(*arrays):
    import torch
    dev = torch.device("cuda", 0)
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]


def test_config2_and_3_bunny_32x32_512_bins_vs_oracle(bunny, orc):
    """cfg 2 (forward only) and cfg 3 (forward + vertex gradient, then one Adam_Modified step) at 32x32 sources."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    from nlos_surface_optimization_amd.adam_modified import Adam_Modified
    v, f = bunny
    o, n = grid_sources(32, 0.25)
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -9, 20000
    t_ref, _ = orc.render_transient(o, n, v, f, ns, lb, ub, res, accel=1, seed=0)
    data = _noisy_data(t_ref, 1)
    w = np.ones_like(data)
    t_ref2, g_ref, _ = orc.render_gradient(o, n, v, f, ns, lb, ub, res, data, w, accel=1, seed=0)
    assert rel_l2(t_ref2, t_ref) <= 1e-13        # (the oracle's threads sum their buffers in any order)
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=0)
    to, tn, tv, tf, td, tw = _tensors(o, n, v, f, data, w)
    t_fwd, _ = r.render_transient(to, tn, tv, tf, ns, lb, ub, res)                    # cfg 2
    _check(t_fwd.cpu().numpy(), t_ref)
    t_g, grad, _ = r.render_gradient(to, tn, tv, tf, ns, lb, ub, res, data=td, weight=tw)   # cfg 3
    _check(t_g.cpu().numpy(), t_ref, grad.cpu().numpy(), g_ref)
    p = r.last_path(count=True)
    assert p["backend"] == "grid" and p["workgroups"] == 1024 and p["bvh_queries"] == 0
    # one Adam step on the device == the same step taken from the oracle's gradient
    # (exp_bunny/test.py:56,212-214: lr = 1e-4 / 3, grad narrowed to float32)
    from oracle import optim_ref
    pv = tv.clone().requires_grad_(True)
    opt = Adam_Modified([pv], lr=1e-4 / 3)
    Adam_Modified.assign_grad(pv, grad)
    opt.step()
    p_ref = v.copy()
    st = optim_ref.AdamModifiedState(p_ref.shape)
    optim_ref.adam_modified_step(p_ref, g_ref.astype(np.float32), st, lr=1e-4 / 3)
    moved = np.abs(p_ref - v).max()
    assert moved > 0 and np.abs(pv.detach().cpu().numpy() - p_ref).max() <= 1e-3 * moved
    r.close()


def test_metric_config_bunny_64x64_512_bins_vs_oracle(bunny, orc):
    """The configuration BASELINE.json's metric is quoted on: 64x64 sources x 512 bins, F = 4967, spt = 5,
    forward + gradient: 101.7 M surface samples, all 4096 rows and the gradient against the oracle."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    v, f = bunny
    o, n = grid_sources(64, 0.25)
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -9, 20000
    L = o.shape[0]
    # data as in bench.py: rows of a slightly displaced mesh (rough surface: the overflow / coarsening paths run)
    rs = np.random.RandomState(0)
    v_gt = (v + 0.002 * rs.standard_normal(v.shape)).astype(np.float32)
    d_ref, _ = orc.render_transient(o, n, v_gt, f, ns, lb, ub, res, accel=1, seed=1)
    w = np.ones_like(d_ref)
    t_ref, g_ref, _ = orc.render_gradient(o, n, v, f, ns, lb, ub, res, d_ref, w, accel=1, seed=0)
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=0)
    to, tn, tv, tf, tvg, tw = _tensors(o, n, v, f, v_gt, w)
    data, _ = r.render_transient(to, tn, tvg, tf, ns, lb, ub, res, seed=1)
    _check(data.cpu().numpy(), d_ref)                                     # the displaced mesh: coarsened sources
    pd = r.last_path(count=True)
    assert pd["backend"] == "grid" and pd["workgroups"] == L
    td = torch.from_numpy(d_ref).to(to.device)
    t_g, grad, _ = r.render_gradient(to, tn, tv, tf, ns, lb, ub, res, data=td, weight=tw)
    _check(t_g.cpu().numpy(), t_ref, grad.cpu().numpy(), g_ref)
    p = r.last_path(count=True)
    assert p["backend"] == "grid" and p["gradient_kernel"] == "source-major, LDS accumulator" and p["bvh_queries"] == 0
    r.close()


def test_config4_mannequin_64x64_1024_bins_confocal_and_pairs_two_blocks(mannequin, orc):
    """cfg 4's shape: exp_mannequin mesh (1055 faces), 64x64 wall points on [-0.35, 0.35]^2, 1024 bins of 2.4 mm from
    0, num_sample 20000 (spt 19).  Confocal = the reference-parity case; the non-confocal pairs are row N.
    Both also as two source blocks (what two ranks render, RNG keyed on the global index) summed like the all-reduce."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    from nlos_surface_optimization_amd.dist import shard_bounds
    v, f = mannequin
    o, n = grid_sources(64, 0.35)
    L = o.shape[0]
    lb, res = 0.0, 2.4e-3
    ub = float(np.float32(1024) * np.float32(res))
    ns = 20000
    t_ref, _ = orc.render_transient(o, n, v, f, ns, lb, ub, res, accel=1, seed=0)
    assert t_ref.shape == (L, 1024)
    data = _noisy_data(t_ref, 2)
    w = np.ones_like(data)
    _, g_ref, _ = orc.render_gradient(o, n, v, f, ns, lb, ub, res, data, w, accel=1, seed=0)
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=0)
    to, tn, tv, tf, td, tw = _tensors(o, n, v, f, data, w)
    t_g, grad, _ = r.render_gradient(to, tn, tv, tf, ns, lb, ub, res, data=td, weight=tw)
    _check(t_g.cpu().numpy(), t_ref, grad.cpu().numpy(), g_ref)
    gsum, rows = torch.zeros_like(grad), []
    for rank in range(2):
        lo, hi = shard_bounds(L, rank, 2)
        t, g, _ = r.render_gradient(to[lo:hi].contiguous(), tn[lo:hi].contiguous(), tv, tf, ns, lb, ub, res,
                                    data=td[lo:hi].contiguous(), weight=tw[lo:hi].contiguous(),
                                    source_offset=lo, total_sources=L)
        rows.append(t)
        gsum += g
    _check(torch.cat(rows).cpu().numpy(), t_ref, gsum.cpu().numpy(), g_ref)
    # non-confocal: sensor = laser shifted by one and a half grid steps (stays on the wall)
    b = o.copy()
    b[:, 0] = np.clip(b[:, 0] + 0.0167, -0.35, 0.35)
    b[:, 1] = np.clip(b[:, 1] - 0.0111, -0.35, 0.35)
    tn_ref, _, _ = orc.render_nonconfocal(o, n, b, n, v, f, ns, lb, ub, res, refine=1, accel=1, seed=0)
    dn = _noisy_data(tn_ref, 3)
    _, gn_ref, _ = orc.render_nonconfocal(o, n, b, n, v, f, ns, lb, ub, res, data=dn, weight=w, accel=1, seed=0)
    tb, tdn = _tensors(b, dn)
    gsum, rows = torch.zeros_like(grad), []
    for rank in range(2):
        lo, hi = shard_bounds(L, rank, 2)
        t, g, _ = r.render_gradient(to[lo:hi].contiguous(), tn[lo:hi].contiguous(), tv, tf, ns, lb, ub, res,
                                    data=tdn[lo:hi].contiguous(), weight=tw[lo:hi].contiguous(),
                                    sensor=tb[lo:hi].contiguous(), sensor_normal=tn[lo:hi].contiguous(),
                                    source_offset=lo, total_sources=L)
        rows.append(t)
        gsum += g
    _check(torch.cat(rows).cpu().numpy(), tn_ref, gsum.cpu().numpy(), gn_ref)
    r.close()


def test_config4_on_the_references_own_measurement(mannequin, orc):
    """cfg 4 on its REAL inputs: the 4096 wall points and the measured photon counts of the reference's
    exp_mannequin/transient.mat (tests/golden/mannequin_measurement.npz: `lighting` on +-0.35 m, uint8 counts folded
    pairwise to 1024 bins of 2.4 mm, consumed as exp_s/test.py:20-36 consumes such files: raw counts as `data`, the
    loss weighting of exp_bunny/rendering.py:208-217 at gamma = 0).  Confocal (the reference-parity case), whole and as
    two source blocks summed like the all-reduce, against the live oracle."""
    import os
    import torch
    from conftest import GOLDEN
    from nlos_surface_optimization_amd import device as nd, rendering
    from nlos_surface_optimization_amd.dist import shard_bounds
    v, f = mannequin
    m = np.load(os.path.join(GOLDEN, "mannequin_measurement.npz"))
    o = np.ascontiguousarray(m["lighting"], np.float32)
    n = np.ascontiguousarray(np.tile(np.array([0, 0, 1], np.float32), (o.shape[0], 1)))
    data = np.ascontiguousarray(m["counts"], np.float64)
    L = o.shape[0]
    assert o.shape == (4096, 3) and data.shape == (4096, 1024) and abs(float(np.abs(o[:, :2]).max()) - 0.35) < 1e-6
    lb, res = float(m["lb"]), float(m["res"])
    ub = float(np.float32(1024) * np.float32(res))
    ns = 20000
    w = np.ascontiguousarray(rendering.create_weighting_function(data, 0))
    t_ref, g_ref, _ = orc.render_gradient(o, n, v, f, ns, lb, ub, res, data, w, accel=1, seed=0)
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=0)
    to, tn, tv, tf, td, tw = _tensors(o, n, v, f, data, w)
    t_g, grad, _ = r.render_gradient(to, tn, tv, tf, ns, lb, ub, res, data=td, weight=tw)
    _check(t_g.cpu().numpy(), t_ref, grad.cpu().numpy(), g_ref)
    gsum, rows = torch.zeros_like(grad), []
    for rank in range(2):
        lo, hi = shard_bounds(L, rank, 2)
        t, g, _ = r.render_gradient(to[lo:hi].contiguous(), tn[lo:hi].contiguous(), tv, tf, ns, lb, ub, res,
                                    data=td[lo:hi].contiguous(), weight=tw[lo:hi].contiguous(),
                                    source_offset=lo, total_sources=L)
        rows.append(t)
        gsum += g
    _check(torch.cat(rows).cpu().numpy(), t_ref, gsum.cpu().numpy(), g_ref)
    # the measured rows really are what the render is compared with: the residual's mass is the data's
    assert data.sum() > 1e6 and t_ref.sum() < 1e-3 * data.sum()
    r.close()


def test_config5_bunny_ggx_64x64_1024_bins_poisson_noised(bunny, orc):
    """cfg 5: GGX branch (alpha 0.3), 64x64 sources x 1024 bins, measurement = Poisson-noised clean transient +
    background (exp_noise/noise/addNoiseExample.m:9; numpy default_rng(0)), vertex gradient and d/d alpha."""
    import torch
    from nlos_surface_optimization_amd import device as nd
    v, f = bunny
    o, n = grid_sources(64, 0.25)
    lb, ub, res, ns = 0.625, 1.625, 2.0 ** -10, 20000
    clean, _ = orc.render_transient(o, n, v, f, ns, lb, ub, res, ggx_alpha=0.3, accel=1, seed=0)
    rng = np.random.default_rng(0)
    c = 2e4 / clean.sum(axis=1, keepdims=True)
    data = np.ascontiguousarray(rng.poisson(c * clean) / c + rng.poisson(0.05, clean.shape) / c)
    w = np.ones_like(data)
    t_ref, g_ref, _ = orc.render_gradient(o, n, v, f, ns, lb, ub, res, data, w, ggx_alpha=0.3, testing_flag=1, accel=1, seed=0)
    r = nd.TransientRenderer(torch.device("cuda", 0), seed=0)
    to, tn, tv, tf, td, tw = _tensors(o, n, v, f, data, w)
    t_g, grad, _ = r.render_gradient(to, tn, tv, tf, ns, lb, ub, res, data=td, weight=tw, alpha=0.3)
    _check(t_g.cpu().numpy(), t_ref, grad.cpu().numpy(), g_ref)
    a_ref = orc.render_gradient_scalar(o[:512], n[:512], v, f, ns, lb, ub, res, data[:512], w[:512], wrt_alpha=True,
                                       ggx_alpha=0.3, accel=1, seed=0)[1]
    _, ga = r.render_gradient_scalar(to[:512].contiguous(), tn[:512].contiguous(), tv, tf, ns, lb, ub, res,
                                     td[:512].contiguous(), tw[:512].contiguous(), alpha=0.3)
    assert abs(float(ga) - a_ref) <= 1e-4 * abs(a_ref)
    r.close()
