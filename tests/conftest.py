import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def grid_sources(n, half):
    g = np.linspace(-half, half, n)
    origin = np.array([[x, y, 0] for y in g for x in g], np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (origin.shape[0], 1))
    return np.ascontiguousarray(origin), np.ascontiguousarray(normal)


def plane_cfg1():
    """BASELINE config 1: 2-triangle wall-facing plane, 2x2 sources, 64 bins, 256 samples."""
    v = np.array([[-.25, -.25, .38], [.25, -.25, .38], [.25, .25, .38], [-.25, .25, .38]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2]], np.int32)
    origin, normal = grid_sources(2, 0.25)
    return dict(v=v, f=f, origin=origin, normal=normal, lb=0.0, ub=2.0, res=2.0 ** -5, num_sample=256)


@pytest.fixture(scope="session")
def cfg1():
    return plane_cfg1()


@pytest.fixture(scope="session")
def bunny():
    d = np.load(os.path.join(GOLDEN, "bunny_5k.npz"))
    return np.ascontiguousarray(d["v"], np.float32), np.ascontiguousarray(d["f"], np.int32)


@pytest.fixture(scope="session")
def mannequin():
    d = np.load(os.path.join(GOLDEN, "mannequin.npz"))
    return np.ascontiguousarray(d["v"], np.float32), np.ascontiguousarray(d["f"], np.int32)


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))


def vertex_normals(v, f):
    """Area-weighted per-vertex normals (numpy; stands in for cgal_api.per_vertex_normal)."""
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0).astype(np.float64)
    vn = np.zeros((v.shape[0], 3))
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    n = np.linalg.norm(vn, axis=1, keepdims=True)
    n[n == 0] = 1
    return np.ascontiguousarray(vn / n, np.float32)
