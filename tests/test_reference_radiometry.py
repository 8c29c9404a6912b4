"""Statistical pin of the v2 radiometry and of the vertex gradient against output of the REFERENCE's own code.

The native v2 renderer cannot be built here (Embree / TBB / MKL / Boost), and the reference ships no golden vectors,
so until round 4 the oracle's `A ff^2 / spt` (smoothed_transient/transient_and_gradient.cpp:224-232) and its gradient
vectors `t1`, `t2` and taps (:944-1001) were checked only against the builder's own restatement.  The reference's numpy
prototype (transient_rendering_python/rendering.py:angular_sampling) does run here, and it integrates the SAME surface
term by a different estimator: with directions uniform on the hemisphere,
    angular_transient[b] = (2 pi / N) sum cos(theta_2) / d_2^2   over the directions whose path length falls in bin b
is the surface integral of cos(theta_1) cos(theta_2) / (d_1^2 d_2^2) over the visible surface (rendering.py:82-93),
while the v2 row is the AREA estimator of that integrand times the two wall cosines (ff = cos_surface cos_wall / h^2).
On wall-PARALLEL patches cos_wall = z_patch / h is a function of the bin, so it can be divided out bin by bin.
tests/golden/make_golden.py:make_pyref_radiometry ran the imported prototype with 2 000 000 directions per source in
40 batches (the batch spread is the Monte-Carlo error used below) and stored only the histograms.

What agreement within that error pins: the area weight and 1 / spt, the two-way 1 / h^4, both clamped surface
cosines, visibility (scene `steps`: a far square partly hidden behind a near one) and the binning origin
(ceil(d / res) - 1 there, floor((2h - lb) / res) here).  Stated errors (1 sigma, relative): row mass 0.17 % (plane),
0.22 % (steps); mean bin 0.004 / 0.03 bins; gradient functionals 1 - 2 %."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

FINE = 16                                  # oracle sub-bins per prototype bin: cos_wall is evaluated per sub-bin


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "pyref_radiometry.npz"))


def _scene(fx, name):
    v = np.ascontiguousarray(fx[name + "_v"], np.float32)
    f = np.ascontiguousarray(fx[name + "_f"], np.int32)
    src = np.ascontiguousarray(fx[name + "_src"], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (src.shape[0], 1))
    return v, f, src, nrm, int(fx[name + "_nbin"]), fx[name + "_z"], float(fx[name + "_zsplit"])


def _cos2(nbin, sub, res, z_near, z_far, zsplit):
    """cos_wall^2 at the centre of every (sub-)bin of path length: the patch in front of `zsplit` metres of path is at
    depth z_near, the one behind it at z_far"""
    path = (np.arange(nbin * sub) + 0.5) * (res / sub)
    zp = np.where(path < zsplit, z_near, z_far)
    return np.minimum(zp / (path / 2), 1.0) ** 2


@pytest.mark.parametrize("name", ["plane", "steps"])
def test_rows_match_the_reference_prototype_within_monte_carlo_error(orc, fx, name):
    v, f, src, nrm, nbin, z, zsplit = _scene(fx, name)
    res, nb = float(fx["res"]), int(fx["batches"])
    rows = fx[name + "_rows"]                                       # [batch, source, bin], the reference's output
    # oracle: 4 M samples per face (its own Monte-Carlo error is then a fifth of the prototype's), 16 sub-bins per bin
    tr, _ = orc.render_transient(src, nrm, v, f, f.shape[0] * 4000000, 0.0, nbin * res, res / FINE, seed=3, accel=0)
    assert tr.shape == (src.shape[0], nbin * FINE)
    surf = (tr / _cos2(nbin, FINE, res, z[0], z[1], zsplit)).reshape(src.shape[0], nbin, FINE).sum(axis=2)
    bins = np.arange(nbin)
    for i in range(src.shape[0]):
        p = rows[:, i]                                              # [batch, bin]
        mass = p.sum(axis=1)
        mean = (p * bins).sum(axis=1) / mass
        var = (p * bins * bins).sum(axis=1) / mass - mean ** 2
        o_mass = surf[i].sum()
        o_mean = (surf[i] * bins).sum() / o_mass
        o_var = (surf[i] * bins * bins).sum() / o_mass - o_mean ** 2
        sem = lambda x: x.std(ddof=1) / np.sqrt(nb)                 # noqa: E731
        assert sem(mass) / mass.mean() < 3e-3                       # the stated error: 0.2 % of the row mass
        # 4 sigma of the prototype's error + 5e-4 for the oracle's own (and the sub-bin evaluation of cos_wall)
        assert abs(mass.mean() - o_mass) <= 4 * sem(mass) + 5e-4 * o_mass, (name, i, mass.mean(), o_mass, sem(mass))
        assert abs(mean.mean() - o_mean) <= 4 * sem(mean) + 2e-3, (name, i, mean.mean(), o_mean, sem(mean))
        assert abs(var.mean() - o_var) <= 4 * sem(var) + 2e-3 * o_var, (name, i, var.mean(), o_var, sem(var))
        # bin by bin: chi^2 over the bins the prototype populated (40 batches: its sigma estimate is itself 11 % noisy)
        sd = p.std(axis=0, ddof=1) / np.sqrt(nb)
        ok = sd > 0
        chi2 = ((((p.mean(axis=0) - surf[i])[ok]) / sd[ok]) ** 2).sum()
        dof = int(ok.sum())
        assert dof >= 20 and chi2 < dof + 5 * np.sqrt(2 * dof), (name, i, chi2, dof)
        assert np.all(surf[i][~ok] <= 1e-9 * o_mass)                # nothing where the reference saw nothing


def test_vertex_gradient_matches_finite_differences_of_the_reference_forward(orc, fx):
    """d/d(motion) of Phi = sum_b row[b] along rigid motions of the unoccluded plane that keep it wall-parallel
    (translations x, y, z; in-plane scaling about its centre).  Reference side: central differences of the imported
    prototype's rows with common directions, times the known cos_wall^2 of the moved patch.  Oracle side: its analytic
    vertex gradient (t1, t2 x e, tap loop, 1 / L) contracted with the motion -- with data = row - 1/2, weight = 1 the
    residual factor -2 w (data - row) is exactly 1 in every bin, so the gradient IS dPhi/dv.
    Translations contract the t2 x e terms away (the three edges of a face sum to zero) and pin t1 = dI/dp and the
    taps; the scaling changes the area and pins t2 as well."""
    name = "plane"
    v, f, src, nrm, nbin, z, zsplit = _scene(fx, name)
    res, nb, delta = float(fx["res"]), int(fx["batches"]), float(fx["delta"])
    motions, moved = fx[name + "_motions"], fx[name + "_moved"].astype(np.float64)     # [Q, V, 3], [Q, +-, batch, source, bin]
    worst = 0.0
    for i in range(src.shape[0]):
        o, n = src[i:i + 1], nrm[i:i + 1]
        ns = f.shape[0] * 2000000
        tr, _ = orc.render_transient(o, n, v, f, ns, 0.0, nbin * res, res / FINE, seed=3, accel=0)
        _, g, _ = orc.render_gradient(o, n, v, f, ns, 0.0, nbin * res, res / FINE, tr - 0.5, np.ones_like(tr), refine=10,
                                      sigma_bin=1, testing_flag=1, seed=3, accel=0)
        for q in range(motions.shape[0]):
            dz = delta if q % 4 == 2 else 0.0                       # the z translation changes the patch's wall cosine
            fd = ((moved[q, 0, :, i] * _cos2(nbin, 1, res, z[0] + dz, z[1] + dz, zsplit)).sum(axis=1) -
                  (moved[q, 1, :, i] * _cos2(nbin, 1, res, z[0] - dz, z[1] - dz, zsplit)).sum(axis=1)) / (2 * delta)
            sem = fd.std(ddof=1) / np.sqrt(nb)
            analytic = float((g * motions[q]).sum())
            # 4 sigma + 1.5 % (central differences at delta = 5 mm, cos_wall per whole bin on this side)
            assert abs(fd.mean() - analytic) <= 4 * sem + 0.015 * abs(analytic), (i, "xyzs"[q % 4], fd.mean(), sem, analytic)
            if abs(analytic) > 1.0:
                worst = max(worst, abs(fd.mean() - analytic) / abs(analytic))
                assert sem / abs(analytic) < 0.03                   # the stated error of the pin: <= 3 % per functional
    assert worst < 0.03


def test_one_call_over_all_sources_carries_the_reference_one_over_L(orc, fx):
    """The same functionals with all sources in ONE call: the driver averages the per-source gradients,
    gradient = (1 / L) sum_l g_l (smoothed_transient/transient_and_gradient.cpp:561-565), so the contraction with a motion
    must equal the MEAN over the sources of the reference's finite differences.  (The test above renders one source per
    call, L = 1, and cannot see a wrong normalisation.)"""
    name = "plane"
    v, f, src, nrm, nbin, z, zsplit = _scene(fx, name)
    res, nb, delta = float(fx["res"]), int(fx["batches"]), float(fx["delta"])
    motions, moved = fx[name + "_motions"], fx[name + "_moved"].astype(np.float64)
    L = src.shape[0]
    assert L >= 2
    ns = f.shape[0] * 1000000
    tr, _ = orc.render_transient(src, nrm, v, f, ns, 0.0, nbin * res, res / FINE, seed=3, accel=0)
    _, g, _ = orc.render_gradient(src, nrm, v, f, ns, 0.0, nbin * res, res / FINE, tr - 0.5, np.ones_like(tr), refine=10,
                                  sigma_bin=1, testing_flag=1, seed=3, accel=0)
    checked = 0
    for q in range(motions.shape[0]):
        dz = delta if q % 4 == 2 else 0.0
        fd = np.stack([((moved[q, 0, :, i] * _cos2(nbin, 1, res, z[0] + dz, z[1] + dz, zsplit)).sum(axis=1) -
                        (moved[q, 1, :, i] * _cos2(nbin, 1, res, z[0] - dz, z[1] - dz, zsplit)).sum(axis=1)) / (2 * delta)
                       for i in range(L)])                              # [source, batch]
        mean_fd = fd.mean(axis=1).mean()                                # (1 / L) sum_l dPhi_l / dq
        sem = np.sqrt(((fd.std(axis=1, ddof=1) / np.sqrt(nb)) ** 2).sum()) / L
        analytic = float((g * motions[q]).sum())
        assert abs(mean_fd - analytic) <= 4 * sem + 0.015 * abs(analytic), ("xyzs"[q % 4], mean_fd, sem, analytic)
        if abs(analytic) > 1.0:
            checked += 1
            # a missing 1 / L would be off by a factor L >= 2, a 1 / (L - 1) or 1 / (L + 1) by >= 20 %: far outside 4 sigma + 1.5 %
            assert abs(L * mean_fd - analytic) > 10 * (4 * sem + 0.015 * abs(analytic))
    assert checked >= 2
