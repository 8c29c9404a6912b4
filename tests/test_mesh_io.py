import os

import numpy as np

from conftest import GOLDEN


def test_obj_roundtrip_and_decimation_is_deterministic(tmp_path, bunny):
    from nlos_surface_optimization_amd import mesh_io
    v, f = bunny
    p = os.path.join(str(tmp_path), "m.obj")
    mesh_io.write_obj(p, v, f)
    v2, f2 = mesh_io.read_obj(p)
    assert np.array_equal(f, f2) and np.allclose(v, v2, rtol=0, atol=1e-7)
    a = mesh_io.cluster_decimate(v, f, 0.02)
    b = mesh_io.cluster_decimate(v, f, 0.02)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert 0 < a[1].shape[0] < f.shape[0]
    assert a[1].min() >= 0 and a[1].max() < a[0].shape[0]
    # winding preserved: outward normals -> positive signed volume
    c = a[0].mean(0)
    p0, p1, p2 = (a[0][a[1][:, k]].astype(np.float64) - c for k in range(3))
    assert np.einsum("ij,ij->i", p0, np.cross(p1, p2)).sum() > 0


def test_fixture_meshes_are_wall_facing_closed_enough(bunny, mannequin):
    for v, f in (bunny, mannequin):
        assert f.min() >= 0 and f.max() < v.shape[0]
        assert v[:, 2].min() > 0.2          # in front of the wall z = 0
        # no two faces over the same three vertices (coincident twins make visibility a coin toss of the last bit of t)
        assert np.unique(np.sort(f, axis=1), axis=0).shape[0] == f.shape[0]
        # no zero-area face, in the renderer's precision
        p0, p1, p2 = (v[f[:, k]] for k in range(3))
        n = np.cross(p1 - p0, p2 - p0)
        assert (np.linalg.norm(n, axis=1) > 0).all()


def test_coincident_faces_are_dropped_keeping_the_wall_facing_one():
    from nlos_surface_optimization_amd import mesh_io
    v = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1.2], [2, 2, 1]], np.float32)
    f = np.array([[0, 1, 2],        # normal +z (faces away from the wall z = 0)
                  [1, 2, 3],
                  [2, 1, 0],        # the twin of face 0, normal -z: this one stays
                  [1, 2, 0],        # a rotation of face 0
                  [0, 1, 4], [4, 1, 0],      # another pair
                  [1, 1, 3]], np.int32)      # zero area
    g = mesh_io.drop_coincident_faces(v, f)
    assert g.tolist() == [[1, 2, 3], [2, 1, 0], [4, 1, 0]]


def test_read_transient_mat_layouts(tmp_path):
    """The two layouts of the reference's shipped measurement files (MAT v5 written here with scipy)."""
    import scipy.io
    from nlos_surface_optimization_amd import mesh_io
    rs = np.random.RandomState(0)
    tr = rs.randint(0, 255, (16, 64)).astype(np.uint8)
    li = rs.uniform(-0.35, 0.35, (16, 3))
    li[:, 2] = 0
    p = str(tmp_path / "transient.mat")
    scipy.io.savemat(p, {"transient": tr, "lighting": li})
    d = mesh_io.read_transient_mat(p, fold=2)
    assert d["transient"].shape == (16, 32) and d["transient"].dtype == np.float64 and d["transient"].flags.c_contiguous
    assert np.array_equal(d["transient"], tr[:, 0::2].astype(np.float64) + tr[:, 1::2])
    assert d["lighting"].dtype == np.float32 and np.allclose(d["lighting"], li, atol=1e-7)
    assert np.array_equal(d["lighting_normal"][3], [0, 0, 1])
    rect = rs.random_sample((4, 4, 20))
    q = str(tmp_path / "rect.mat")
    scipy.io.savemat(q, {"rect_data": rect})
    e = mesh_io.read_transient_mat(q)
    assert e["transient"].shape == (16, 20) and np.array_equal(e["transient"][5], rect[1, 1]) and "lighting" not in e
    import pytest
    with pytest.raises(ValueError):
        mesh_io.read_transient_mat(p, fold=5)
