"""Drop-in for the reference's `embree_intersector` extension module
(transient_rendering_cython/embree_intersector/embree_intersector.pyx): closest-hit
queries for N arbitrary rays against a triangle mesh, used by space carving
(transient_rendering_cython/rendering.py:11-23) and Delaunay validity checks
(exp_bunny/rendering.py:103-179).  The BVH build and traversal run on the GPU.

Output convention (c_embree_intersector.cpp:39-45): column 0 = primID as float
(-1 on a miss), columns 1,2 = barycentrics (u, v) with
hit = (1-u-v)*V[f0] + u*V[f1] + v*V[f2]; u, v are left untouched on a miss.
"""
import numpy as np

from . import _lib
from ._check import f32, i32, ptr


def _check_rays(origin, direction):
    f32(origin, 2, "origin"); f32(direction, 2, "direction")
    assert origin.shape[0] == direction.shape[0], "Origin and Direction need to be Nx3"
    assert origin.shape[1] == 3, "Origin needs to be Nx3"
    assert direction.shape[1] == 3, "Direction needs to be Nx3"


def _check_mesh(v, f):
    f32(v, 2, "v"); i32(f, 2, "f")
    assert v.shape[1] == 3, "vertex should be Vx3"
    assert f.shape[1] == 3, "face should be Fx3"


def embree3_tbb_intersection(origin, direction, v, f, barycoord):
    """embree_intersector.pyx:85-94 -> embree3_tbb_line_intersection."""
    _check_mesh(v, f)
    _check_rays(origin, direction)
    f32(barycoord, 2, "barycoord")
    assert barycoord.shape[0] == origin.shape[0], "barycoord needs to be Nx1 or Nx3"
    assert barycoord.shape[1] == 3, "barycoord needs to be Nx3"
    rc = _lib.lib().nlos_embree3_tbb_line_intersection(
        ptr(origin), ptr(direction), direction.shape[0], ptr(v), v.shape[0], ptr(f), f.shape[0],
        ptr(barycoord))
    _lib.check(rc, "embree3_tbb_line_intersection")


def embree3_tbb_short_intersection(origin, direction, v, f, barycoord):
    """embree_intersector.pyx:73-82 -> embree3_tbb_short_line_intersection."""
    _check_mesh(v, f)
    _check_rays(origin, direction)
    f32(barycoord, 1, "barycoord")
    assert barycoord.shape[0] == origin.shape[0], "barycoord needs to be Nx1"
    rc = _lib.lib().nlos_embree3_tbb_short_line_intersection(
        ptr(origin), ptr(direction), direction.shape[0], ptr(v), v.shape[0], ptr(f), f.shape[0],
        ptr(barycoord))
    _lib.check(rc, "embree3_tbb_short_line_intersection")


def barycoord_to_world(v, f, barycoord, intersection_p):
    """embree_intersector.pyx:62-70 -> barycentric_to_world."""
    _check_mesh(v, f)
    f32(barycoord, 2, "barycoord"); f32(intersection_p, 2, "intersection_p")
    assert barycoord.shape[0] == intersection_p.shape[0], "barycoord and intersection_p should be Nx3"
    assert barycoord.shape[1] == 3, "barycoord should be Nx3"
    assert intersection_p.shape[1] == 3, "intersection_p should be Nx3"
    rc = _lib.lib().nlos_barycentric_to_world_n(ptr(v), v.shape[0], ptr(f), f.shape[0], ptr(barycoord),
                                                barycoord.shape[0], ptr(intersection_p))
    _lib.check(rc, "barycentric_to_world")


class PyMesh:
    """embree_intersector.pyx:8-59: a mesh object with the same query methods."""

    def __init__(self, v, f):
        f32(v, 2, "v"); i32(f, 2, "f")
        assert v.shape[1] == 3, "Vertices needs to be Vx3"
        assert f.shape[1] == 3, "Face needs to be Tx3"
        self._v = np.array(v, dtype=np.float32, order="C")
        self._f = np.array(f, dtype=np.int32, order="C")
        self._vn = None
        self._fn = None
        self._area = None

    def test(self):
        print("PyMesh: %d vertices, %d faces" % (self._v.shape[0], self._f.shape[0]))

    def embree3_tbb_intersection(self, origin, direction, barycoord):
        embree3_tbb_intersection(origin, direction, self._v, self._f, barycoord)

    def embree3_tbb_short_intersection(self, origin, direction, barycoord):
        embree3_tbb_short_intersection(origin, direction, self._v, self._f, barycoord)

    def set_vn(self, vn):
        f32(vn, 2, "vn")
        assert vn.shape[1] == 3, "vn needs to be #vertices x 3"
        assert vn.shape[0] == self._v.shape[0], "vn nees to be #vertices x 3"
        self._vn = np.array(vn, dtype=np.float32, order="C")

    def set_fn_and_face_area(self, fn, area):
        f32(fn, 2, "fn"); f32(area, 1, "area")
        assert fn.shape[1] == 3, "fn needs to be #face x 3"
        assert fn.shape[0] == self._f.shape[0], "fn needs to be #face x 3"
        assert area.shape[0] == self._f.shape[0], "barycoord needs to be #face x 1"
        self._fn = np.array(fn, dtype=np.float32, order="C")
        self._area = np.array(area, dtype=np.float32, order="C")

    def barycoord_to_world(self, barycoord, intersection_p):
        barycoord_to_world(self._v, self._f, barycoord, intersection_p)
