"""Drop-in for the reference's v1 `renderer` module
(transient_rendering_cython/stratified_transient_raytracer/renderer.pyx, imported by
transient_rendering_cython/rendering.py:2,8).

v1 differs from v2 (renderer.py) by: unclamped forward form factor
(stratifiedStreamedTransientRenderer.cpp:130-137), a one-tap gradient with the normal
term always on, an optional double box filter of the residual (`w_width`,
stratifiedStreamedGradientRenderer.cpp:447-462), and a zeroed (not accumulated)
gradient output.  The index typos at stratifiedStreamedGradientRenderer.cpp:276-278,
289-291 are not reproduced.
"""
from . import _lib
from ._check import f32, f64, i32, ptr
from .renderer import _check_grad, _check_tp, _common, _num_bins


def renderStreamedCurvatureGradient(vertices, faces, gradient):
    """stratified_transient_raytracer/renderer.pyx:13-18 -> streamed_render_curvature_grad
    (stratifiedStreamedGradientRenderer.cpp:298-348): the gradient of the total surface area, the same body as
    the v2 module's (per-vertex terms accumulated over the incident faces; see renderer.set_regulariser_overwrite)."""
    f32(vertices, 2, "vertices"); i32(faces, 2, "faces"); f64(gradient, 2, "gradient")
    assert vertices.shape[1] == 3, "vertices needs to be Vx3"
    assert faces.shape[1] == 3, "faces needs to be Fx3"
    assert gradient.shape[0] == vertices.shape[0], "gradient dimension should be Vx3"
    assert gradient.shape[1] == 3, "gradient dimension should be Vx3"
    rc = _lib.lib().nlos_streamed_render_curvature_grad(
        ptr(vertices), vertices.shape[0], ptr(faces), faces.shape[0], ptr(gradient))
    _lib.check(rc, "v1 streamed_render_curvature_grad")


def renderTransient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound, resolution,
                    transient, pathlengths):
    """stratified_transient_raytracer/renderer.pyx:93-102 -> render_transient (stratifiedTransientRenderer.cpp:
    132-218): ONE wall point (origin, normal are [3]), rows are [numBins]; v1 estimator (unclamped form factor)."""
    f32(origin, 1, "origin"); f32(normal, 1, "normal"); f32(vertices, 2, "vertices"); i32(faces, 2, "faces")
    f64(transient, 1, "transient"); f64(pathlengths, 1, "pathlengths")
    assert origin.shape[0] == 3, "origin needs to be 1x3"
    assert normal.shape[0] == 3, "normal needs to be 1x3"
    assert vertices.shape[1] == 3, "vertices needs to be Vx3"
    assert faces.shape[1] == 3, "faces needs to be Fx3"
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    assert transient.shape[0] == numBins, \
        "transient dimension should match number of bins = math.ceil((upper_bound-lower_bound)/resolution)"
    assert pathlengths.shape[0] == numBins, \
        "pathlength dimension should match number of bins = math.ceil((upper_bound-lower_bound)/resolution)"
    rc = _lib.lib().nlos_v1_render_transient(
        ptr(origin), ptr(normal), ptr(vertices), vertices.shape[0], ptr(faces), faces.shape[0], int(num_sample),
        lower_bound, upper_bound, resolution, ptr(transient), ptr(pathlengths))
    _lib.check(rc, "v1 render_transient")


def renderStreamedTransient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                            resolution, transient, pathlengths):
    L = _common(origin, normal, vertices, faces)
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    rc = _lib.lib().nlos_v1_streamed_render_transient(
        ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], None, None, ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, ptr(transient),
        ptr(pathlengths))
    _lib.check(rc, "v1 streamed_render_transient")


def renderStreamedTransientShading(origin, normal, vertices, vertexNormal, faces, num_sample,
                                   lower_bound, upper_bound, resolution, transient, pathlengths):
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    rc = _lib.lib().nlos_v1_streamed_render_transient(
        ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], ptr(vertexNormal), None,
        ptr(faces), faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution,
        ptr(transient), ptr(pathlengths))
    _lib.check(rc, "v1 streamed_render_transient")


def renderStreamedTransientwAlbedo(origin, normal, vertices, albedo, faces, num_sample, lower_bound,
                                   upper_bound, resolution, transient, pathlengths):
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    rc = _lib.lib().nlos_v1_streamed_render_transient(
        ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], None, ptr(albedo), ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, ptr(transient),
        ptr(pathlengths))
    _lib.check(rc, "v1 streamed_render_transient")


def renderStreamedGradient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                           resolution, w_width, transient, pathlengths, gradient, data):
    """stratified_transient_raytracer/renderer.pyx:22-37."""
    L = _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    f64(data, 2, "data")
    assert data.shape[0] == L and data.shape[1] == numBins, \
        "data transient dimension should  be LxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    rc = _lib.lib().nlos_v1_streamed_render_gradient(
        ptr(data), ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, int(w_width),
        ptr(transient), ptr(pathlengths), ptr(gradient))
    _lib.check(rc, "v1 streamed_render_gradient")
