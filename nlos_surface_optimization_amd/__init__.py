"""nlos_surface_optimization_amd -- MI355X-native differentiable transient renderer.

The one data-parallel hot path of cmu-ci-lab/nlos_surface_optimization (stratified
confocal transient rendering + analytic per-vertex gradient + closest-hit queries) as
hand-written HIP kernels for gfx950 behind the reference's own module names:

    from nlos_surface_optimization_amd import renderer, ggx, jitter, embree_intersector, rendering

`renderer` / `ggx` / `embree_intersector` mirror the reference's Cython extension modules
(numpy in, in-place numpy out); `rendering` mirrors its facade; `device` is the additive
torch-tensor / autograd path and `dist` the multi-GPU source sharding.  There is no CPU
fallback: without libnlos_hip.so and an AMD GPU every render call raises.
"""
from . import _lib  # noqa: F401
from . import embree_intersector, ggx, jitter, renderer, renderer_v1, rendering, rendering_v1  # noqa: F401

__all__ = ["renderer", "renderer_v1", "ggx", "jitter", "embree_intersector", "rendering", "rendering_v1",
           "device", "dist", "mesh_io", "adam_modified"]


def __getattr__(name):
    # torch-dependent submodules are imported lazily
    if name in ("device", "dist", "mesh_io", "adam_modified"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
