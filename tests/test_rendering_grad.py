"""BASELINE configuration 1 ("plumbing", CPU): the counterpart of the reference's torch prototype
transient_rendering_python/rendering_grad.py (angular sampling of one (lighting, sensor) pair with a vertex
gradient).  Forward rows against fixtures produced by the reference's own numpy twin (rendering.py, imported in
the build container by tests/golden/make_golden.py); gradients against central finite differences of that
numpy forward (pyref_angular_grad.npz) on the cfg-1 plane and the test_autograd.py:35-36 toy mesh."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from nlos_surface_optimization_amd import rendering_grad


def _mesh(v, f, grad=False):
    m = types.SimpleNamespace()
    m.v = torch.from_numpy(np.array(v, np.float64)).requires_grad_(grad)
    m.f = torch.from_numpy(np.array(f, np.int64))
    return m


def _opt(n, nbin, res):
    return types.SimpleNamespace(sample_num=int(n), max_distance_bin=int(nbin), distance_resolution=float(res), epsilon=1e-9)


@pytest.mark.parametrize("name", ["plane", "toy"])
def test_forward_matches_the_reference_prototype(name):
    up = np.array([0, 0, 1.0])
    cases = [("pyref_angular.npz", name, True)]
    if name == "toy":
        cases += [("pyref_angular_nc.npz", "toy", False), ("pyref_angular_nc.npz", "occluder", False)]   # lighting != sensor
    for fixture, name, same in cases:
        d = np.load(os.path.join(GOLDEN, fixture))
        mesh = _mesh(d[name + "_v"], d[name + "_f"])
        dirs = d[name + "_dir"]
        opt = _opt(dirs.shape[0], d[name + "_nbin"], d[name + "_res"])
        lights = d[name + "_pairs"] if same else d[name + "_laser"]
        sensors = lights if same else d[name + "_sensor"]
        for k in range(lights.shape[0]):
            t = rendering_grad.angular_sampling(mesh, dirs, lights[k], sensors[k], up, up, opt)
            ref = d[name + "_transient"][k]
            assert ref.sum() > 0 and t.shape == ref.shape
            assert np.abs(t.numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("name", ["plane", "toy"])
def test_vertex_gradient_matches_finite_differences_of_the_reference_forward(name):
    d = np.load(os.path.join(GOLDEN, "pyref_angular_grad.npz"))
    up = np.array([0, 0, 1.0])
    dirs = d[name + "_dir"]
    opt = _opt(dirs.shape[0], d[name + "_nbin"], d[name + "_res"])
    for k in range(d[name + "_lighting"].shape[0]):
        mesh = _mesh(d[name + "_v"], d[name + "_f"], grad=True)
        t = rendering_grad.angular_sampling(mesh, dirs, d[name + "_lighting"][k], d[name + "_sensor"][k], up, up, opt)
        assert np.abs(t.detach().numpy() - d[name + "_transient"][k]).max() <= 1e-12 * d[name + "_transient"][k].max()
        t.backward(torch.ones_like(t))                  # test_autograd.py: backward(ones) -> v.grad
        g, g_fd = mesh.v.grad.numpy(), d[name + "_grad_fd"][k]
        assert np.abs(g_fd).max() > 0
        assert np.linalg.norm(g - g_fd) <= 1e-5 * np.linalg.norm(g_fd)


def test_config1_four_pairs_64_bins_256_samples():
    """The configuration BASELINE.json names: 2-triangle plane, 4 (laser, sensor) pairs, 64 bins, 256 directions."""
    d = np.load(os.path.join(GOLDEN, "pyref_angular.npz"))
    mesh = _mesh(d["plane_v"], d["plane_f"], grad=True)
    dirs, up = d["plane_dir"], np.array([0, 0, 1.0])
    assert dirs.shape[0] == 256 and int(d["plane_nbin"]) == 64 and d["plane_pairs"].shape[0] == 4
    opt = _opt(256, 64, d["plane_res"])
    rows = torch.stack([rendering_grad.angular_sampling(mesh, dirs, p, p, up, up, opt) for p in d["plane_pairs"]])
    assert rows.shape == (4, 64) and np.abs(rows.detach().numpy() - d["plane_transient"]).max() <= 1e-12
    (rows ** 2).sum().backward()
    assert torch.isfinite(mesh.v.grad).all() and float(mesh.v.grad.abs().sum()) > 0
