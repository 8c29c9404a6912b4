"""Drop-in for the reference's v2 `renderer` extension module.

Mirrors transient_rendering_cython/smoothed_transient/renderer.pyx (the module every
`exp_*` script imports through `exp_bunny/rendering.py:3,18`): same function names,
positional signatures, typed-array checks, `assert` shape validation, in-place
outputs and `None` return (scalar-gradient variants return a Python float).  The
work happens in libnlos_hip.so on an MI355X; nothing here computes on the CPU.

renderStreamedNormalSmoothing / renderStreamedCurvatureGradient accumulate
the per-face terms over a vertex's incident faces; the reference stores them with `=` (last writer
wins) -- `set_regulariser_overwrite(True)` selects that behaviour (highest incident face wins).
"""
import ctypes

import numpy as np

from . import _lib
from ._check import f32, f64, i32, ptr


def set_regulariser_overwrite(flag):
    """False (default): accumulate per-vertex regulariser terms; True: the reference's `=` stores."""
    _lib.lib().nlos_set_regulariser_overwrite(1 if flag else 0)


def renderStreamedNormalSmoothing(vertices, faces, f_affinity, gradient):
    """renderer.pyx:13-21 -> streamed_render_normal_smoothing; returns the smoothing value."""
    f32(vertices, 2, "vertices"); i32(faces, 2, "faces"); i32(f_affinity, 2, "f_affinity"); f64(gradient, 2, "gradient")
    assert vertices.shape[1] == 3, "vertices needs to be Vx3"
    assert faces.shape[1] == 3, "faces needs to be Fx3"
    assert f_affinity.shape[1] == 3, "face affinity needs to be Fx3"
    assert f_affinity.shape[0] == faces.shape[0], "face affinity needs to be Fx3"
    assert gradient.shape[0] == vertices.shape[0], "gradient dimension should be Vx3"
    assert gradient.shape[1] == 3, "gradient dimension should be Vx3"
    val = ctypes.c_double(0.0)
    rc = _lib.lib().nlos_streamed_render_normal_smoothing(
        ptr(vertices), vertices.shape[0], ptr(faces), faces.shape[0], ptr(f_affinity), ptr(gradient),
        ctypes.cast(ctypes.byref(val), ctypes.c_void_p))
    _lib.check(rc, "streamed_render_normal_smoothing")
    return val.value


def renderStreamedCurvatureGradient(vertices, faces, gradient):
    """renderer.pyx:26-31 -> streamed_render_curvature_grad (gradient of the total surface area)."""
    f32(vertices, 2, "vertices"); i32(faces, 2, "faces"); f64(gradient, 2, "gradient")
    assert vertices.shape[1] == 3, "vertices needs to be Vx3"
    assert faces.shape[1] == 3, "faces needs to be Fx3"
    assert gradient.shape[0] == vertices.shape[0], "gradient dimension should be Vx3"
    assert gradient.shape[1] == 3, "gradient dimension should be Vx3"
    rc = _lib.lib().nlos_streamed_render_curvature_grad(
        ptr(vertices), vertices.shape[0], ptr(faces), faces.shape[0], ptr(gradient))
    _lib.check(rc, "streamed_render_curvature_grad")


def _num_bins(lower_bound, upper_bound, resolution):
    """Row length the native side writes: ceil((ub - lb) / res) in float32
    (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:514-515).  renderer.pyx:101 validates the
    caller's arrays against the same expression in Python doubles; where the two disagree (a quotient within
    float32 rounding of an integer) the reference strides its rows by one count and checks them against the
    other.  Shapes are validated against the count the kernels use, so such a call fails the shape assertion
    instead of writing mis-strided rows."""
    return _lib.num_bins(lower_bound, upper_bound, resolution)


def _common(origin, normal, vertices, faces):
    f32(origin, 2, "origin"); f32(normal, 2, "normal"); f32(vertices, 2, "vertices"); i32(faces, 2, "faces")
    L = origin.shape[0]
    assert origin.shape[1] == 3, "origin needs to be Lx3"
    assert normal.shape[0] == L, "normal needs to be Lx3"
    assert normal.shape[1] == 3, "normal needs to be Lx3"
    assert vertices.shape[1] == 3, "vertices needs to be Vx3"
    assert faces.shape[1] == 3, "faces needs to be Fx3"
    return L


def _check_tp(transient, pathlengths, L, numBins):
    f64(transient, 2, "transient"); f64(pathlengths, 1, "pathlengths")
    msg = "transient dimension should  be LxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert transient.shape[0] == L, msg
    assert transient.shape[1] == numBins, msg
    assert pathlengths.shape[0] == numBins, \
        "pathlength dimension should be Bx1 (B = math.ceil((upper_bound-lower_bound)/resolution))"


def _check_dw(data, weight, L, numBins):
    f64(data, 2, "data"); f64(weight, 2, "weight")
    msg = "data transient dimension should  be LxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert data.shape[0] == L, msg
    assert data.shape[1] == numBins, msg
    assert weight.shape[0] == L, "weighting should be LxB"
    assert weight.shape[1] == numBins, "weighting should be LxB"


def _check_grad(gradient, vertices):
    f64(gradient, 2, "gradient")
    assert gradient.shape[0] == vertices.shape[0], "gradient dimension should be Vx3"
    assert gradient.shape[1] == 3, "gradient dimension should be Vx3"


def _transient(origin, normal, vertices, vnormal, albedo, faces, num_sample, lb, ub, res,
               transient, pathlengths, refine_scale, sigma_bin):
    rc = _lib.lib().nlos_streamed_render_transient(
        ptr(origin), origin.shape[0], ptr(normal), ptr(vertices), vertices.shape[0], ptr(vnormal),
        ptr(albedo), ptr(faces), faces.shape[0], int(num_sample), lb, ub, res, ptr(transient),
        ptr(pathlengths), int(refine_scale), int(sigma_bin))
    _lib.check(rc, "streamed_render_transient")


def renderStreamedTransient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                            resolution, transient, pathlengths, refine_scale, sigma_bin):
    """renderer.pyx:175-187 -> streamed_render_transient(vertexNormal=NULL, albedo=NULL)."""
    L = _common(origin, normal, vertices, faces)
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, None, faces, num_sample, lower_bound, upper_bound,
               resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTransientShading(origin, normal, vertices, vertexNormal, faces, num_sample,
                                   lower_bound, upper_bound, resolution, transient, pathlengths,
                                   refine_scale, sigma_bin):
    """renderer.pyx:141-156 (shading normals)."""
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, vertexNormal, None, faces, num_sample, lower_bound,
               upper_bound, resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTransientwAlbedo(origin, normal, vertices, albedo, faces, num_sample, lower_bound,
                                   upper_bound, resolution, transient, pathlengths, refine_scale,
                                   sigma_bin):
    """renderer.pyx:158-172 (per-vertex albedo)."""
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, albedo, faces, num_sample, lower_bound, upper_bound,
               resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTriangleIntensity(origin, normal, vertices, faces, num_sample, lower_bound,
                                    upper_bound, intensity):
    """renderer.pyx:189-200 -> streamed_render_intensity."""
    _common(origin, normal, vertices, faces)
    f64(intensity, 1, "intensity")
    assert intensity.shape[0] == faces.shape[0], "intensity should be (F,)"
    rc = _lib.lib().nlos_streamed_render_intensity(
        ptr(origin), origin.shape[0], ptr(normal), ptr(vertices), vertices.shape[0], None, ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, ptr(intensity))
    _lib.check(rc, "streamed_render_intensity")


def _gradient(fn_name, origin, normal, vertices, extra, faces, num_sample, lb, ub, res, transient,
              pathlengths, gradient, data, weight, refine_scale, sigma_bin, testing_flag, loss_flag):
    fn = getattr(_lib.lib(), fn_name)
    rc = fn(ptr(data), ptr(weight), ptr(origin), origin.shape[0], ptr(normal), ptr(vertices),
            vertices.shape[0], ptr(extra), ptr(faces), faces.shape[0], int(num_sample), lb, ub, res,
            ptr(transient), ptr(pathlengths), ptr(gradient), int(refine_scale), int(sigma_bin),
            int(testing_flag), int(loss_flag))
    _lib.check(rc, fn_name)


def renderStreamedGradient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                           resolution, transient, pathlengths, gradient, data, weight, refine_scale,
                           sigma_bin, testing_flag, loss_flag):
    """renderer.pyx:94-111 -> streamed_render_gradient(vertexNormal=NULL)."""
    L = _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    _gradient("nlos_streamed_render_gradient", origin, normal, vertices, None, faces, num_sample,
              lower_bound, upper_bound, resolution, transient, pathlengths, gradient, data, weight,
              refine_scale, sigma_bin, testing_flag, loss_flag)


def renderStreamedShadingGradient(origin, normal, vertices, faces, vertexNormal, num_sample,
                                  lower_bound, upper_bound, resolution, transient, pathlengths,
                                  gradient, data, weight, refine_scale, sigma_bin, testing_flag,
                                  loss_flag):
    """renderer.pyx:114-138 -> streamed_render_gradient(vertexNormal)."""
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    _gradient("nlos_streamed_render_gradient", origin, normal, vertices, vertexNormal, faces,
              num_sample, lower_bound, upper_bound, resolution, transient, pathlengths, gradient,
              data, weight, refine_scale, sigma_bin, testing_flag, loss_flag)


def renderStreamedGradientWithAlbedo(origin, normal, vertices, faces, albedo, num_sample, lower_bound,
                                     upper_bound, resolution, transient, pathlengths, gradient, data,
                                     weight, refine_scale, sigma_bin, testing_flag, loss_flag):
    """renderer.pyx:55-75 -> streamed_render_gradient_w_albedo."""
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert albedo.shape[0] == vertices.shape[0], "albedo needs to be Vx1"
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    _gradient("nlos_streamed_render_gradient_w_albedo", origin, normal, vertices, albedo, faces,
              num_sample, lower_bound, upper_bound, resolution, transient, pathlengths, gradient,
              data, weight, refine_scale, sigma_bin, testing_flag, loss_flag)


def renderStreamedGradientAlbedo(origin, normal, vertices, faces, albedo, num_sample, lower_bound,
                                 upper_bound, resolution, transient, pathlengths, data, weight,
                                 refine_scale, sigma_bin, testing_flag, loss_flag):
    """renderer.pyx:33-52 -> streamed_render_gradient_albedo; returns d loss / d albedo."""
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert albedo.shape[0] == vertices.shape[0], "albedo needs to be Vx1"
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_dw(data, weight, L, numBins)
    out = ctypes.c_double(0.0)
    rc = _lib.lib().nlos_streamed_render_gradient_albedo(
        ptr(data), ptr(weight), ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0],
        ptr(albedo), ptr(faces), faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution,
        ptr(transient), ptr(pathlengths), int(refine_scale), int(sigma_bin), int(testing_flag),
        int(loss_flag), ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
    _lib.check(rc, "streamed_render_gradient_albedo")
    return out.value


def renderStreamedVertexGradient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                                 resolution, gradient, vertex_num, refine_scale, sigma_bin):
    """renderer.pyx:78-88 -> streamed_render_vertex_gradient (measurement is hard-wired to 1)."""
    _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    f64(gradient, 2, "gradient")
    assert gradient.shape[0] == numBins, "gradient dimension should be Vx3"
    assert gradient.shape[1] == 3, "gradient dimension should be Vx3"
    rc = _lib.lib().nlos_streamed_render_vertex_gradient(
        int(vertex_num), ptr(origin), 1, ptr(normal), ptr(vertices), vertices.shape[0], ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, ptr(gradient),
        int(refine_scale), int(sigma_bin))
    _lib.check(rc, "streamed_render_vertex_gradient")


# ---------------------------------------------------------------------------------------------
# Row N: non-confocal (laser, sensor) pairs.  The reference has no native
# function for it (prototypes only: transient_rendering_python/rendering.py:8-93,
# mesh_optimization/rendering.py:739-797); these two follow renderStreamedTransient /
# renderStreamedGradient with the sensor arrays inserted after the laser's.
def _pairs(laser, laser_normal, sensor, sensor_normal, vertices, faces):
    L = _common(laser, laser_normal, vertices, faces)
    f32(sensor, 2, "sensor"); f32(sensor_normal, 2, "sensor_normal")
    assert sensor.shape[0] == L and sensor.shape[1] == 3, "sensor needs to be Lx3"
    assert sensor_normal.shape[0] == L and sensor_normal.shape[1] == 3, "sensor normal needs to be Lx3"
    return L


def renderNonConfocalTransient(laser, laser_normal, sensor, sensor_normal, vertices, faces, num_sample,
                               lower_bound, upper_bound, resolution, transient, pathlengths,
                               refine_scale=1, sigma_bin=1, vertexNormal=None, albedo=None, alpha=None):
    """transient[i] = three-bounce histogram of pair (laser[i], sensor[i]); bins floor((d1+d2-lb)/res)."""
    L = _pairs(laser, laser_normal, sensor, sensor_normal, vertices, faces)
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    if vertexNormal is not None:
        f32(vertexNormal, 2, "vertexNormal")
        assert vertexNormal.shape == vertices.shape, "vertex normal needs to be Vx3"
    if albedo is not None:
        f32(albedo, 1, "albedo")
        assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    if alpha is not None:            # GGX BRDF of the `ggx` module, half-vector form for the pair
        rc = _lib.lib().nlos_ggx_nonconfocal_render_transient(
            ptr(laser), ptr(laser_normal), ptr(sensor), ptr(sensor_normal), L, ptr(vertices), vertices.shape[0],
            ptr(vertexNormal), ptr(albedo), ptr(faces), faces.shape[0], float(alpha), int(num_sample), lower_bound,
            upper_bound, resolution, ptr(transient), ptr(pathlengths), int(refine_scale), int(sigma_bin))
        _lib.check(rc, "ggx_nonconfocal_render_transient")
        return
    rc = _lib.lib().nlos_nonconfocal_render_transient(
        ptr(laser), ptr(laser_normal), ptr(sensor), ptr(sensor_normal), L, ptr(vertices), vertices.shape[0],
        ptr(vertexNormal), ptr(albedo), ptr(faces), faces.shape[0], int(num_sample), lower_bound, upper_bound,
        resolution, ptr(transient), ptr(pathlengths), int(refine_scale), int(sigma_bin))
    _lib.check(rc, "nonconfocal_render_transient")


def renderNonConfocalGradient(laser, laser_normal, sensor, sensor_normal, vertices, faces, num_sample,
                              lower_bound, upper_bound, resolution, transient, pathlengths, gradient, data,
                              weight, refine_scale, sigma_bin, testing_flag, loss_flag, vertexNormal=None,
                              albedo=None, alpha=None):
    """Vertex gradient of sum w (data - T)^2 / L over the pairs; accumulated into `gradient` (v2 semantics)."""
    L = _pairs(laser, laser_normal, sensor, sensor_normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_dw(data, weight, L, numBins)
    _check_grad(gradient, vertices)
    if vertexNormal is not None:
        f32(vertexNormal, 2, "vertexNormal")
        assert vertexNormal.shape == vertices.shape, "vertex normal needs to be Vx3"
    if albedo is not None:
        f32(albedo, 1, "albedo")
        assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    if alpha is not None:
        rc = _lib.lib().nlos_ggx_nonconfocal_render_gradient(
            ptr(data), ptr(weight), ptr(laser), ptr(laser_normal), ptr(sensor), ptr(sensor_normal), L, ptr(vertices),
            vertices.shape[0], ptr(vertexNormal), ptr(albedo), ptr(faces), faces.shape[0], float(alpha), int(num_sample),
            lower_bound, upper_bound, resolution, ptr(transient), ptr(pathlengths), ptr(gradient),
            int(refine_scale), int(sigma_bin), int(testing_flag), int(loss_flag))
        _lib.check(rc, "ggx_nonconfocal_render_gradient")
        return
    rc = _lib.lib().nlos_nonconfocal_render_gradient(
        ptr(data), ptr(weight), ptr(laser), ptr(laser_normal), ptr(sensor), ptr(sensor_normal), L, ptr(vertices),
        vertices.shape[0], ptr(vertexNormal), ptr(albedo), ptr(faces), faces.shape[0], int(num_sample),
        lower_bound, upper_bound, resolution, ptr(transient), ptr(pathlengths), ptr(gradient),
        int(refine_scale), int(sigma_bin), int(testing_flag), int(loss_flag))
    _lib.check(rc, "nonconfocal_render_gradient")


# ---- row N as a product: every (laser, sensor) combination of two sets of wall points (north_star's L x S x T
# histogram).  The reference has prototype formulas only (transient_rendering_python/mesh_optimization/rendering.py:739-797);
# the product is defined as its pairs, rendered on sample points shared by all wall points (include/nlos_hip.h,
# nlos_render_args.n_sensors).  transient / data / weight are [L, S, B] float64 arrays.
def _product(laser, laser_normal, sensor, sensor_normal, vertices, faces):
    L = _common(laser, laser_normal, vertices, faces)
    f32(sensor, 2, "sensor"); f32(sensor_normal, 2, "sensor_normal")
    S = sensor.shape[0]
    assert S > 0 and sensor.shape[1] == 3, "sensor needs to be Sx3"
    assert sensor_normal.shape[0] == S and sensor_normal.shape[1] == 3, "sensor normal needs to be Sx3"
    return L, S


def renderNonConfocalProductTransient(laser, laser_normal, sensor, sensor_normal, vertices, faces, num_sample,
                                      lower_bound, upper_bound, resolution, transient, pathlengths):
    """transient[i, j] = three-bounce histogram of the pair (laser[i], sensor[j]); bins floor((d1 + d2 - lb) / res)."""
    L, S = _product(laser, laser_normal, sensor, sensor_normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    f64(transient, 3, "transient"); f64(pathlengths, 1, "pathlengths")
    assert transient.shape == (L, S, numBins), "transient dimension should  be LxSxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert pathlengths.shape[0] == numBins, "pathlength dimension should be Bx1 (B = math.ceil((upper_bound-lower_bound)/resolution))"
    rc = _lib.lib().nlos_nonconfocal_product_render_transient(
        ptr(laser), ptr(laser_normal), L, ptr(sensor), ptr(sensor_normal), S, ptr(vertices), vertices.shape[0], ptr(faces),
        faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, ptr(transient), ptr(pathlengths))
    _lib.check(rc, "nonconfocal_product_render_transient")


def renderNonConfocalProductGradient(laser, laser_normal, sensor, sensor_normal, vertices, faces, num_sample,
                                     lower_bound, upper_bound, resolution, transient, pathlengths, gradient, data,
                                     weight, refine_scale, sigma_bin, testing_flag, loss_flag):
    """Vertex gradient of sum w (data - T)^2 / (L S) over the L x S measurements; accumulated into `gradient`."""
    L, S = _product(laser, laser_normal, sensor, sensor_normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    f64(transient, 3, "transient"); f64(pathlengths, 1, "pathlengths"); f64(data, 3, "data"); f64(weight, 3, "weight")
    assert transient.shape == (L, S, numBins), "transient dimension should  be LxSxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert pathlengths.shape[0] == numBins, "pathlength dimension should be Bx1 (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert data.shape == (L, S, numBins), "data transient dimension should  be LxSxB   (B = math.ceil((upper_bound-lower_bound)/resolution))"
    assert weight.shape == (L, S, numBins), "weighting should be LxSxB"
    _check_grad(gradient, vertices)
    rc = _lib.lib().nlos_nonconfocal_product_render_gradient(
        ptr(data), ptr(weight), ptr(laser), ptr(laser_normal), L, ptr(sensor), ptr(sensor_normal), S, ptr(vertices),
        vertices.shape[0], ptr(faces), faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution,
        ptr(transient), ptr(pathlengths), ptr(gradient), int(refine_scale), int(sigma_bin), int(testing_flag), int(loss_flag))
    _lib.check(rc, "nonconfocal_product_render_gradient")
