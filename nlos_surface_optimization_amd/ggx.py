"""Drop-in for the reference's `ggx` extension module
(transient_rendering_cython/ggx/ggx.pyx): the v2 renderer with an isotropic GGX
microfacet BRDF of scalar roughness `alpha` (ggx/ggx_confocal.cpp:13-232).
Same names, positional signatures, checks and in-place semantics as the reference.
"""
import ctypes

from . import _lib
from ._check import f32, ptr
from .renderer import _check_dw, _check_grad, _check_tp, _common, _num_bins


def _transient(origin, normal, vertices, vnormal, albedo, faces, alpha, num_sample, lb, ub, res,
               transient, pathlengths, refine_scale, sigma_bin):
    rc = _lib.lib().nlos_ggx_streamed_render_transient(
        ptr(origin), origin.shape[0], ptr(normal), ptr(vertices), vertices.shape[0], ptr(vnormal),
        ptr(albedo), ptr(faces), faces.shape[0], alpha, int(num_sample), lb, ub, res, ptr(transient),
        ptr(pathlengths), int(refine_scale), int(sigma_bin))
    _lib.check(rc, "ggx streamed_render_transient")


def renderStreamedTransient(origin, normal, vertices, faces, alpha, num_sample, lower_bound,
                            upper_bound, resolution, transient, pathlengths, refine_scale, sigma_bin):
    """ggx.pyx:118-130."""
    L = _common(origin, normal, vertices, faces)
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, None, faces, alpha, num_sample, lower_bound,
               upper_bound, resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTransientShading(origin, normal, vertices, vertexNormal, faces, alpha, num_sample,
                                   lower_bound, upper_bound, resolution, transient, pathlengths,
                                   refine_scale, sigma_bin):
    """ggx.pyx:82-96."""
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, vertexNormal, None, faces, alpha, num_sample, lower_bound,
               upper_bound, resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTransientwAlbedo(origin, normal, vertices, albedo, faces, alpha, num_sample,
                                   lower_bound, upper_bound, resolution, transient, pathlengths,
                                   refine_scale, sigma_bin):
    """ggx.pyx:100-113."""
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, albedo, faces, alpha, num_sample, lower_bound,
               upper_bound, resolution, transient, pathlengths, refine_scale, sigma_bin)


def renderStreamedTriangleIntensity(origin, normal, vertices, faces, alpha, num_sample, lower_bound,
                                    upper_bound, intensity):
    """ggx.pyx:134-143."""
    from ._check import f64
    _common(origin, normal, vertices, faces)
    f64(intensity, 1, "intensity")
    assert intensity.shape[0] == faces.shape[0], "intensity should be (F,)"
    rc = _lib.lib().nlos_ggx_streamed_render_intensity(
        ptr(origin), origin.shape[0], ptr(normal), ptr(vertices), vertices.shape[0], None, ptr(faces),
        faces.shape[0], alpha, int(num_sample), lower_bound, upper_bound, ptr(intensity))
    _lib.check(rc, "ggx streamed_render_intensity")


def _gradient(origin, normal, vertices, vnormal, faces, alpha, num_sample, lb, ub, res, transient,
              pathlengths, gradient, data, weight, refine_scale, sigma_bin, testing_flag):
    rc = _lib.lib().nlos_ggx_streamed_render_gradient(
        ptr(data), ptr(weight), ptr(origin), origin.shape[0], ptr(normal), ptr(vertices),
        vertices.shape[0], ptr(vnormal), ptr(faces), faces.shape[0], alpha, int(num_sample), lb, ub,
        res, ptr(transient), ptr(pathlengths), ptr(gradient), int(refine_scale), int(sigma_bin),
        int(testing_flag))
    _lib.check(rc, "ggx streamed_render_gradient")


def renderStreamedGradient(origin, normal, vertices, faces, alpha, num_sample, lower_bound,
                           upper_bound, resolution, transient, pathlengths, gradient, data, weight,
                           refine_scale, sigma_bin, testing_flag):
    """ggx.pyx:37-54."""
    L = _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    _gradient(origin, normal, vertices, None, faces, alpha, num_sample, lower_bound, upper_bound,
              resolution, transient, pathlengths, gradient, data, weight, refine_scale, sigma_bin,
              testing_flag)


def renderStreamedShadingGradient(origin, normal, vertices, faces, vertexNormal, alpha, num_sample,
                                  lower_bound, upper_bound, resolution, transient, pathlengths,
                                  gradient, data, weight, refine_scale, sigma_bin, testing_flag):
    """ggx.pyx:59-78."""
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    _gradient(origin, normal, vertices, vertexNormal, faces, alpha, num_sample, lower_bound,
              upper_bound, resolution, transient, pathlengths, gradient, data, weight, refine_scale,
              sigma_bin, testing_flag)


def renderStreamedGradientAlpha(origin, normal, vertices, faces, alpha, num_sample, lower_bound,
                                upper_bound, resolution, transient, pathlengths, data, weight,
                                refine_scale, sigma_bin):
    """ggx.pyx:12-29 -> streamed_render_gradient_alpha; returns d loss / d alpha."""
    L = _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_dw(data, weight, L, numBins)
    out = ctypes.c_double(0.0)
    rc = _lib.lib().nlos_ggx_streamed_render_gradient_alpha(
        ptr(data), ptr(weight), ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], None,
        ptr(faces), faces.shape[0], alpha, int(num_sample), lower_bound, upper_bound, resolution,
        ptr(transient), ptr(pathlengths), int(refine_scale), int(sigma_bin),
        ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
    _lib.check(rc, "ggx streamed_render_gradient_alpha")
    return out.value
