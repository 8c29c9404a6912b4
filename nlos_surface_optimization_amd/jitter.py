"""Drop-in for the reference's `jitter` extension module.

Mirrors transient_rendering_cython/jitter/jitter.pyx (imported by exp_bunny/rendering.py:8,22 and
used when `opt.jitter` is set, exp_bunny/rendering.py:262-263): the measured SPAD jitter kernel
replaces the Gaussian of the `renderer` module -- the forward rows are the plain histogram
convolved with `weight` (jitter/transient_and_gradient.cpp:331-347), the gradient uses one tap per
kernel entry with `jitter_grad` carrying the kernel's time derivative (jitter/...:944-969).  Same
positional signatures, typed-array checks, assert messages and in-place outputs as the .pyx (its
commented-out functions are not provided).  The work happens in libnlos_hip.so on an MI355X.
"""
from . import _lib
from ._check import f32, f64, ptr
from .renderer import _check_dw, _check_grad, _check_tp, _common, _num_bins


def _kernel(weight, name):
    f64(weight, 2, name)
    assert weight.shape[0] >= 1, "%s needs at least one tap" % name
    return weight.shape[0]


def _transient(origin, normal, vertices, vnormal, albedo, faces, num_sample, lb, ub, res, transient,
               pathlengths, weight, weight_offset):
    K = _kernel(weight, "weight")
    rc = _lib.lib().nlos_jitter_streamed_render_transient(
        ptr(origin), origin.shape[0], ptr(normal), ptr(vertices), vertices.shape[0], ptr(vnormal),
        ptr(albedo), ptr(faces), faces.shape[0], int(num_sample), lb, ub, res, ptr(weight),
        int(weight_offset), K, ptr(transient), ptr(pathlengths))
    _lib.check(rc, "jitter streamed_render_transient")


def renderStreamedTransient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                            resolution, transient, pathlengths, weight, weight_offset):
    """jitter.pyx:140-152."""
    L = _common(origin, normal, vertices, faces)
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, None, faces, num_sample, lower_bound, upper_bound, resolution,
               transient, pathlengths, weight, weight_offset)


def renderStreamedTransientShading(origin, normal, vertices, vertexNormal, faces, num_sample, lower_bound,
                                   upper_bound, resolution, transient, pathlengths, weight, weight_offset):
    """jitter.pyx:104-118."""
    L = _common(origin, normal, vertices, faces)
    f32(vertexNormal, 2, "vertexNormal")
    assert vertexNormal.shape[1] == 3, "vertex normal needs to be Vx3"
    assert vertices.shape[0] == vertexNormal.shape[0], "vertex normal needs to be Vx3"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, vertexNormal, None, faces, num_sample, lower_bound, upper_bound,
               resolution, transient, pathlengths, weight, weight_offset)


def renderStreamedTransientwAlbedo(origin, normal, vertices, albedo, faces, num_sample, lower_bound,
                                   upper_bound, resolution, transient, pathlengths, weight, weight_offset):
    """jitter.pyx:122-136."""
    L = _common(origin, normal, vertices, faces)
    f32(albedo, 1, "albedo")
    assert vertices.shape[0] == albedo.shape[0], "albedo nees to be Vx1"
    _check_tp(transient, pathlengths, L, _num_bins(lower_bound, upper_bound, resolution))
    _transient(origin, normal, vertices, None, albedo, faces, num_sample, lower_bound, upper_bound, resolution,
               transient, pathlengths, weight, weight_offset)


def renderStreamedGradient(origin, normal, vertices, faces, num_sample, lower_bound, upper_bound, resolution,
                           jitter_weight, jitter_grad, jitter_offset, transient, pathlengths, gradient, data,
                           weight, testing_flag):
    """jitter.pyx:59-77 -> streamed_render_gradient(vertexNormal=NULL, ...)."""
    L = _common(origin, normal, vertices, faces)
    numBins = _num_bins(lower_bound, upper_bound, resolution)
    _check_tp(transient, pathlengths, L, numBins)
    _check_grad(gradient, vertices)
    _check_dw(data, weight, L, numBins)
    K = _kernel(jitter_weight, "jitter_weight")
    f64(jitter_grad, 2, "jitter_grad")
    assert jitter_grad.shape[0] == K, "jitter_grad needs one entry per jitter_weight tap"
    rc = _lib.lib().nlos_jitter_streamed_render_gradient(
        ptr(data), ptr(weight), ptr(origin), L, ptr(normal), ptr(vertices), vertices.shape[0], None,
        ptr(faces), faces.shape[0], int(num_sample), lower_bound, upper_bound, resolution, ptr(jitter_weight),
        ptr(jitter_grad), int(jitter_offset), K, ptr(transient), ptr(pathlengths), ptr(gradient),
        int(testing_flag))
    _lib.check(rc, "jitter streamed_render_gradient")
