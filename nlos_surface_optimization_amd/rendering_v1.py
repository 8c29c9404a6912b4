"""v1 facade (transient_rendering_cython/rendering.py:11-47, the CY/main.py driver's API).

`inverseRendering(mesh, data, opt)` returns a gradient of shape (V,3): the reference's
facade allocates (L,3V) but its own extension asserts (V,3) (SURVEY.md section 3.2
"Stale-API warning"); the V x 3 contract is the one every working script uses.
"""
import numpy as np

from . import renderer_v1 as renderer
from .rendering import space_carving_projection  # noqa: F401  (same code in both facades)


def inverseRendering(mesh, data, opt):
    measurement_num = opt.lighting.shape[0]
    transient = np.zeros((measurement_num, opt.max_distance_bin), dtype=np.double, order='C')
    pathlengths = np.zeros(opt.max_distance_bin, dtype=np.double, order='C')
    gradient = np.zeros(mesh.v.shape, dtype=np.double, order='C')
    renderer.renderStreamedGradient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                    opt.max_distance_bin * opt.distance_resolution, opt.distance_resolution,
                                    opt.w_width, transient, pathlengths, gradient, data)
    return transient, gradient, pathlengths


def forwardRendering(mesh, opt):
    measurement = opt.lighting.shape[0]
    transient = np.zeros((measurement, opt.max_distance_bin), dtype=np.double, order='C')
    pathlengths = np.zeros(opt.max_distance_bin, dtype=np.double, order='C')
    if opt.normal == 'fn':
        renderer.renderStreamedTransient(opt.lighting, opt.lighting_normal, mesh.v, mesh.f, opt.sample_num, 0,
                                         opt.max_distance_bin * opt.distance_resolution,
                                         opt.distance_resolution, transient, pathlengths)
    else:
        renderer.renderStreamedTransientShading(opt.lighting, opt.lighting_normal, mesh.v, mesh.vn, mesh.f,
                                                opt.sample_num, 0,
                                                opt.max_distance_bin * opt.distance_resolution,
                                                opt.distance_resolution, transient, pathlengths)
    return transient, pathlengths
