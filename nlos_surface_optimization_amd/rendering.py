"""Facade with the reference's `rendering.py` call signatures.

Mirrors the hot-path entry points of transient_rendering_cython/exp_bunny/rendering.py
(v2, the living API: SURVEY.md Q13): allocate the numpy outputs, dispatch on the
`opt` flags, return `(transient, gradient, pathlengths)` -- so an Adam loop written
against the reference (`p.grad.data = torch.from_numpy(grad).float(); optimizer.step()`,
exp_bunny/test.py:212-214) runs unchanged.  `opt` / `mesh` are the reference's ad-hoc
attribute bags (exp_bunny/test.py:16-46): opt.lighting, opt.lighting_normal,
opt.sample_num, opt.max_distance_bin, opt.distance_resolution, opt.bin_refine_resolution,
opt.sigma_bin, opt.testing_flag, opt.loss_flag, opt.alpha_flag, opt.albedo_flag,
opt.jitter, opt.normal; mesh.v, mesh.f, mesh.vn, mesh.alpha, mesh.albedo, mesh.f_affinity.

Out of scope here: CGAL / El Topo remeshing.
"""
import numpy as np

from . import embree_intersector, ggx, jitter, renderer


def _flag(opt, name, default=0):
    return getattr(opt, name, default)


def _alloc(opt, mesh, with_gradient=True):
    measurement_num = opt.lighting.shape[0]
    transient = np.zeros((measurement_num, opt.max_distance_bin), dtype=np.double, order='C')
    pathlengths = np.zeros(opt.max_distance_bin, dtype=np.double, order='C')
    gradient = np.zeros(mesh.v.shape, dtype=np.double, order='C') if with_gradient else None
    return transient, pathlengths, gradient


def inverseRendering(mesh, data, weight, opt):
    """exp_bunny/rendering.py:252-269."""
    transient, pathlengths, gradient = _alloc(opt, mesh)
    ub = opt.max_distance_bin * opt.distance_resolution
    if _flag(opt, 'alpha_flag'):
        ggx.renderStreamedGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, mesh.alpha,
                                   opt.sample_num, 0, ub, opt.distance_resolution, transient, pathlengths,
                                   gradient, data, weight, opt.bin_refine_resolution, opt.sigma_bin,
                                   opt.testing_flag)
    elif _flag(opt, 'jitter'):
        jitter.renderStreamedGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0, ub,
                                      opt.distance_resolution, opt.jitter_weight, opt.jitter_grad,
                                      int(opt.jitter_offset), transient, pathlengths, gradient, data, weight,
                                      opt.testing_flag)
    elif _flag(opt, 'albedo_flag'):
        albedo = np.ones(mesh.v.shape[0], dtype=np.float32, order='C') * mesh.albedo
        renderer.renderStreamedGradientWithAlbedo(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, albedo,
                                                  opt.sample_num, 0, ub, opt.distance_resolution, transient,
                                                  pathlengths, gradient, data, weight,
                                                  opt.bin_refine_resolution, opt.sigma_bin, opt.testing_flag,
                                                  _flag(opt, 'loss_flag'))
    else:
        renderer.renderStreamedGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                        ub, opt.distance_resolution, transient, pathlengths, gradient, data,
                                        weight, opt.bin_refine_resolution, opt.sigma_bin, opt.testing_flag,
                                        _flag(opt, 'loss_flag'))
    return transient, gradient, pathlengths


def inverseShadingRendering(mesh, data, weight, opt):
    """exp_bunny/rendering.py:219-229; mesh.vn must be set (the reference fills it with CGAL)."""
    transient, pathlengths, gradient = _alloc(opt, mesh)
    renderer.renderStreamedShadingGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, mesh.vn,
                                           opt.sample_num, 0, opt.max_distance_bin * opt.distance_resolution,
                                           opt.distance_resolution, transient, pathlengths, gradient, data,
                                           weight, opt.bin_refine_resolution, opt.sigma_bin, opt.testing_flag,
                                           _flag(opt, 'loss_flag'))
    return transient, gradient, pathlengths


def inverseRenderingAlpha(mesh, data, weight, opt):
    """exp_bunny/rendering.py:232-238."""
    transient, pathlengths, _ = _alloc(opt, mesh, False)
    g = ggx.renderStreamedGradientAlpha(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, mesh.alpha,
                                        opt.sample_num, 0, opt.max_distance_bin * opt.distance_resolution,
                                        opt.distance_resolution, transient, pathlengths, data, weight,
                                        opt.bin_refine_resolution, opt.sigma_bin)
    return transient, g


def inverseRenderingAlbedo(mesh, data, weight, opt):
    """exp_bunny/rendering.py:241-250."""
    transient, pathlengths, _ = _alloc(opt, mesh, False)
    albedo = np.ones(mesh.v.shape[0], dtype=np.float32, order='C') * mesh.albedo
    g = renderer.renderStreamedGradientAlbedo(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, albedo,
                                              opt.sample_num, 0, opt.max_distance_bin * opt.distance_resolution,
                                              opt.distance_resolution, transient, pathlengths, data, weight,
                                              opt.bin_refine_resolution, opt.sigma_bin, opt.testing_flag,
                                              _flag(opt, 'loss_flag'))
    return transient, g


def forwardRendering(mesh, opt):
    """exp_bunny/rendering.py:280-297."""
    transient, pathlengths, _ = _alloc(opt, mesh, False)
    ub = opt.max_distance_bin * opt.distance_resolution
    fn = getattr(opt, 'normal', 'fn') == 'fn'
    if _flag(opt, 'alpha_flag'):
        if fn:
            ggx.renderStreamedTransient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, mesh.alpha,
                                        opt.sample_num, 0, ub, opt.distance_resolution, transient, pathlengths, 1, 1)
        else:
            ggx.renderStreamedTransientShading(opt.lighting, opt.lighting_normal, mesh.v, mesh.vn, mesh.f,
                                               mesh.alpha, opt.sample_num, 0, ub, opt.distance_resolution,
                                               transient, pathlengths, 1, 1)
    else:
        if fn:
            renderer.renderStreamedTransient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                             ub, opt.distance_resolution, transient, pathlengths, 1, 1)
        else:
            renderer.renderStreamedTransientShading(opt.lighting, opt.lighting_normal, mesh.v, mesh.vn, mesh.f,
                                                    opt.sample_num, 0, ub, opt.distance_resolution, transient,
                                                    pathlengths, 1, 1)
    return transient, pathlengths


def vertex_gradient(mesh, vertex_num, opt):
    """exp_bunny/rendering.py:26-30."""
    gradient = np.zeros((opt.max_distance_bin, 3), dtype=np.double, order='C')
    renderer.renderStreamedVertexGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                          opt.max_distance_bin * opt.distance_resolution,
                                          opt.distance_resolution, gradient, vertex_num,
                                          opt.bin_refine_resolution, opt.sigma_bin)
    return gradient


def removeTriangle(mesh, opt):
    """exp_bunny/rendering.py:271-278: drop border faces that receive no light."""
    intensity = np.zeros(mesh.f.shape[0], dtype=np.double, order='C')
    renderer.renderStreamedTriangleIntensity(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                             opt.max_distance_bin * opt.distance_resolution, intensity)
    threshold = 0
    keep_face = np.logical_or((intensity > threshold), np.sum(mesh.f_affinity < 0, axis=1) == 0)
    print('remove #face:%d' % (mesh.f.shape[0] - np.sum(keep_face)))
    mesh.f = mesh.f[keep_face, :]


def space_carving_projection(v, space_carving_mesh):
    """exp_bunny/rendering.py:193-206: push vertices behind the carved surface along +z."""
    direction = np.array([0, 0, 1], dtype=np.float32, order='C')
    direction = np.ascontiguousarray(np.tile(direction, (v.shape[0], 1)))
    barycoord = np.ndarray((v.shape[0], 3), dtype=np.float32, order='C')
    new_v = np.array(v)
    new_v[:, 2] = 0
    embree_intersector.embree3_tbb_intersection(new_v, direction, space_carving_mesh.v, space_carving_mesh.f,
                                                barycoord)
    intersection_p = np.ndarray((v.shape[0], 3), dtype=np.float32, order='C')
    embree_intersector.barycoord_to_world(space_carving_mesh.v, space_carving_mesh.f, barycoord, intersection_p)
    index = barycoord[:, 0] >= 0
    v[index, 2] = np.maximum(intersection_p[index, 2], v[index, 2])


def renderStreamedNormalSmoothing(mesh):
    """exp_bunny/rendering.py:298-301; mesh.f_affinity is the int32 [F,3] neighbour table
    (cgal_api.face_affinity in the reference, mesh_io.face_affinity here)."""
    gradient = np.zeros(mesh.v.shape, dtype=np.double, order='C')
    val = renderer.renderStreamedNormalSmoothing(mesh.v, mesh.f, mesh.f_affinity, gradient)
    return val, gradient


def renderStreamedCurvatureGradient(mesh):
    """exp_bunny/rendering.py:303-306."""
    gradient = np.zeros(mesh.v.shape, dtype=np.double, order='C')
    renderer.renderStreamedCurvatureGradient(mesh.v, mesh.f, gradient)
    return gradient


def create_weighting_function(data, gamma=1):
    """Per-bin loss weights of the measured transient `data` [L, T] (the interface of exp_bunny/rendering.py:208-217):
    bins are weighted by (data / max(data) + 0.1) ** gamma and the weights rescaled to mean 1, so gamma = 0 gives
    all ones.  Host numpy like the reference's; the device twin is device.TransientRenderer.create_weighting_function
    (csrc/optimiser.hip)."""
    data = np.asarray(data)
    w = np.power(data / data.max() + 0.1, gamma)
    return w * (w.size / w.sum())


def evaluate_loss_with_normal_smoothness(gt_transient, weight, transient, smoothing_val, mesh, render_opt):
    """exp_bunny/rendering.py:360-367 -> (L1 + smooth_weight * smoothing_val, L1)."""
    difference = transient - gt_transient
    difference = difference * np.sqrt(weight)
    L1 = np.linalg.norm(difference) ** 2 / difference.shape[0]
    L2 = render_opt.smooth_weight * smoothing_val
    return L1 + L2, L1
