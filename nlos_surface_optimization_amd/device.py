"""Device-resident path: torch tensors in HBM -> libnlos_hip.so -> torch tensors.

PyTorch is plumbing here (device memory, streams, torch.distributed); the render is
the C ABI of include/nlos_hip.h section 2 (`nlos_render` on device pointers, enqueued
on torch's current HIP stream, no host synchronisation).

The reference has no device path and no autograd.Function (it injects numpy gradients:
transient_rendering_cython/main.py:114-115); `TransientFunction` is the additive
wrapper SURVEY.md section 7 step 2 asks for: forward = rendered transient, backward =
the reference's analytic vertex gradient (smoothed_transient/transient_and_gradient.cpp:
843-1007) contracted with the upstream gradient.
"""
import ctypes

import torch

from . import _lib


def _dptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _want(t, dtype, name, ndim=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP torch tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must have dtype %s (got %s)" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dimensions" % (name, ndim))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


class TransientRenderer:
    """One render context per GPU (BVH, visibility cache and residual scratch live in it)."""

    def __init__(self, device=None, seed=0):
        if not torch.cuda.is_available():
            raise _lib.NlosError("no AMD GPU visible to torch: the transient renderer has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.seed = int(seed)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().nlos_ctx_create(self.device.index or 0, ctypes.byref(h)), "nlos_ctx_create")
        self._h = h
        self._lib = _lib.lib()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nlos_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def num_bins(self, lower_bound, upper_bound, resolution):
        return _lib.num_bins(lower_bound, upper_bound, resolution)

    def scratch_bytes(self):
        return int(self._lib.nlos_ctx_scratch_bytes(self._h))

    def enable_timing(self, on=True):
        self._lib.nlos_ctx_enable_timing(self._h, 1 if on else 0)

    def last_timing_ms(self):
        """(bvh build, forward, residual, gradient) of the last render; synchronises the stream."""
        torch.cuda.current_stream(self.device).synchronize()
        buf = (ctypes.c_float * 4)()
        _lib.check(self._lib.nlos_ctx_last_timing(self._h, ctypes.cast(buf, ctypes.c_void_p)), "nlos_ctx_last_timing")
        return tuple(float(x) for x in buf)

    def timing_reset(self):
        self._lib.nlos_ctx_timing_reset(self._h)

    def timing_mean_ms(self):
        """Per-stage mean over the renders since timing_reset() (ring of 256); synchronises."""
        torch.cuda.current_stream(self.device).synchronize()
        buf = (ctypes.c_float * 4)()
        cnt = ctypes.c_int(0)
        _lib.check(self._lib.nlos_ctx_timing_mean(self._h, ctypes.cast(buf, ctypes.c_void_p),
                                                  ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p)), "nlos_ctx_timing_mean")
        return tuple(float(x) for x in buf), cnt.value

    def mesh_generation(self):
        """Generation of the scene tree the context holds (changes with every build)."""
        return int(self._lib.nlos_ctx_mesh_generation(self._h))

    def visibility_generation(self):
        """Generation of the visibility cache the context holds (0: none; changes with every recorded pass 1)."""
        return int(self._lib.nlos_ctx_visibility_generation(self._h))

    def last_path(self, count=False):
        """Which kernels the last render took and why, as a dict (`count=True` synchronises and adds the
        per-workgroup outcomes of the grid launches: coarsened / big-LDS / in-kernel BVH query)."""
        info = _lib.PathInfo()
        _lib.check(self._lib.nlos_ctx_last_path(self._h, ctypes.byref(info), 1 if count else 0), "nlos_ctx_last_path")
        d = {"backend": _lib.PATH_NAMES.get(info.backend, str(info.backend)),
             "reason": _lib.REASON_NAMES.get(info.reason, str(info.reason)),
             "grid_R": info.grid_R, "tiles": info.tiles, "tile_cap": info.tile_cap, "chunks": info.chunks,
             "rows_in_lds": bool(info.rows_in_lds),
             "gradient_kernel": _lib.GRADIENT_KERNEL_NAMES.get(info.gradient_kernel, str(info.gradient_kernel))}
        if count and info.workgroups >= 0:
            d.update(workgroups=int(info.workgroups), coarsened=int(info.coarsened), big_lds=int(info.big_lds),
                     bvh_queries=int(info.bvh_queries))
        if count and info.rays_traced >= 0:
            d.update(rays_traced=int(info.rays_traced), samples_accepted=int(info.samples_accepted))
        return d

    def check(self):
        """Synchronise and raise if an earlier render on this context met a face index out of range."""
        _lib.check(self._lib.nlos_ctx_check(self._h), "nlos_ctx_check")

    def _args(self, mode, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
              resolution, refine_scale=1, sigma_bin=1, vertex_normal=None, albedo=None,
              source_offset=0, total_sources=0, alpha=None, seed=None, force_bvh=False, source_stride=1,
              shared_samples=False, sensor=None, sensor_normal=None, jitter_weight=None, jitter_grad=None, jitter_offset=0):
        a = _lib.RenderArgs()
        self._lib.nlos_render_args_init(ctypes.byref(a))
        _want(origin, torch.float32, "origin", 2); _want(normal, torch.float32, "normal", 2)
        _want(vertices, torch.float32, "vertices", 2); _want(faces, torch.int32, "faces", 2)
        _want(vertex_normal, torch.float32, "vertex_normal", 2); _want(albedo, torch.float32, "albedo", 1)
        assert origin.shape[1] == 3 and normal.shape == origin.shape, "origin/normal need to be Lx3"
        assert vertices.shape[1] == 3, "vertices needs to be Vx3"
        assert faces.shape[1] == 3, "faces needs to be Fx3"
        a.mode = mode
        a.origin, a.normal, a.L = _dptr(origin), _dptr(normal), origin.shape[0]
        a.source_offset, a.total_sources = int(source_offset), int(total_sources)
        a.source_stride = int(source_stride)
        a.shared_samples = 1 if shared_samples else 0
        a.vertices, a.V = _dptr(vertices), vertices.shape[0]
        a.faces, a.F = _dptr(faces), faces.shape[0]
        a.vertex_normal, a.albedo = _dptr(vertex_normal), _dptr(albedo)
        a.num_samples = int(num_sample)
        a.lower_bound, a.upper_bound, a.resolution = lower_bound, upper_bound, resolution
        a.refine_scale, a.sigma_bin = int(refine_scale), int(sigma_bin)
        a.seed = self.seed if seed is None else int(seed)
        if alpha is not None:
            a.use_ggx, a.ggx_alpha = 1, float(alpha)
        a.force_bvh = int(force_bvh)       # False/0 default, True/1 BVH only, 2 diagnostic (tile overflow fallback)
        if sensor is not None:
            # row N: measurement l is the pair (laser origin[l], sensor[l]); default sensor wall normal = laser's
            if sensor_normal is None:
                sensor_normal = normal
            _want(sensor, torch.float32, "sensor", 2); _want(sensor_normal, torch.float32, "sensor_normal", 2)
            assert sensor.shape == origin.shape and sensor_normal.shape == origin.shape, "sensor/sensor_normal need to be Lx3"
            a.sensor, a.sensor_normal = _dptr(sensor), _dptr(sensor_normal)
        if jitter_weight is not None:
            # SPAD jitter kernel (reference module `jitter`): f64 [K] or [K,1] device tensors
            _want(jitter_weight, torch.float64, "jitter_weight"); _want(jitter_grad, torch.float64, "jitter_grad")
            a.jitter_weight, a.jitter_grad = _dptr(jitter_weight), _dptr(jitter_grad)
            a.jitter_offset, a.jitter_length = int(jitter_offset), int(jitter_weight.shape[0])
            assert jitter_grad is None or jitter_grad.shape[0] == jitter_weight.shape[0], "jitter_grad needs one entry per tap"
        return a

    def _run(self, a, keep):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self._lib.nlos_render(self._h, ctypes.byref(a), ctypes.c_void_p(stream))
        _lib.check(rc, "nlos_render")
        del keep

    # ------------------------------------------------------------------ renders
    def render_transient(self, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                         resolution, refine_scale=1, sigma_bin=1, vertex_normal=None, albedo=None,
                         alpha=None, clamp=True, keep_visibility=False, out=None, **kw):
        """Rows S,I,F,FD. Returns (transient [L,T] f64, pathlengths [T] f64)."""
        a = self._args(_lib.MODE_TRANSIENT, origin, normal, vertices, faces, num_sample, lower_bound,
                       upper_bound, resolution, refine_scale, sigma_bin, vertex_normal, albedo,
                       alpha=alpha, **kw)
        T = self.num_bins(lower_bound, upper_bound, resolution)
        transient = out if out is not None else torch.empty((origin.shape[0], T), dtype=torch.float64, device=self.device)
        path = torch.empty(T, dtype=torch.float64, device=self.device)
        a.transient, a.pathlengths = _dptr(transient), _dptr(path)
        a.clamp = 1 if clamp else 0
        a.keep_visibility = 1 if keep_visibility else 0
        self._run(a, (origin, normal, vertices, faces, vertex_normal, albedo))
        return transient, path

    def render_gradient(self, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                        resolution, data=None, weight=None, refine_scale=10, sigma_bin=1, testing_flag=1,
                        loss_flag=0, vertex_normal=None, albedo=None, alpha=None, gradient=None,
                        normal_term=-1, residual=None, reuse_visibility=False, reuse_bvh=False,
                        mesh_generation=0, visibility_generation=0, zero_gradient=False, **kw):
        """Rows D,G,GD. Returns (transient, gradient [V,3] f64 (accumulated into if given -- overwritten with
        zero_gradient=True: the render clears it itself, no fill operation of its own --), pathlengths).
        `reuse_bvh` / `reuse_visibility` need the generations read after the render whose tree / cache they
        reuse (mesh_generation(), visibility_generation()); with reuse_visibility pass 1 is skipped and the
        returned transient is None."""
        a = self._args(_lib.MODE_GRADIENT, origin, normal, vertices, faces, num_sample, lower_bound,
                       upper_bound, resolution, refine_scale, sigma_bin, vertex_normal, albedo,
                       alpha=alpha, **kw)
        L, T = origin.shape[0], self.num_bins(lower_bound, upper_bound, resolution)
        _want(data, torch.float64, "data", 2); _want(weight, torch.float64, "weight", 2)
        _want(residual, torch.float64, "residual", 2)
        for t, n in ((data, "data"), (weight, "weight"), (residual, "residual")):
            if t is not None:
                assert tuple(t.shape) == (L, T), "%s should be LxB" % n
        transient = None if reuse_visibility else torch.empty((L, T), dtype=torch.float64, device=self.device)
        path = torch.empty(T, dtype=torch.float64, device=self.device)
        if gradient is None:
            gradient = torch.empty((vertices.shape[0], 3), dtype=torch.float64, device=self.device)
            zero_gradient = True
        else:
            _want(gradient, torch.float64, "gradient", 2)
            assert tuple(gradient.shape) == (vertices.shape[0], 3), "gradient dimension should be Vx3"
        a.zero_gradient = 1 if zero_gradient else 0
        a.data, a.weight, a.residual = _dptr(data), _dptr(weight), _dptr(residual)
        a.transient, a.pathlengths, a.gradient = _dptr(transient), _dptr(path), _dptr(gradient)
        a.testing_flag, a.loss_test, a.normal_term = int(testing_flag), int(loss_flag), int(normal_term)
        a.reuse_visibility = 1 if reuse_visibility else 0
        a.reuse_bvh = 1 if reuse_bvh else 0
        a.mesh_generation, a.visibility_generation = int(mesh_generation), int(visibility_generation)
        self._run(a, (origin, normal, vertices, faces, vertex_normal, albedo, data, weight, residual))
        return transient, gradient, path

    def render_product(self, laser, laser_normal, sensor, sensor_normal, vertices, faces, num_sample, lower_bound,
                       upper_bound, resolution, data=None, weight=None, refine_scale=None, sigma_bin=1, testing_flag=1,
                       loss_flag=0, gradient=None, zero_gradient=False, pairs=False, seed=None, total_lasers=0,
                       out=None, vertex_normal=None, albedo=None):
        """Row N as a product (include/nlos_hip.h, nlos_render_args.n_sensors): every (laser[i], sensor[j]) combination on
        sample points shared by all wall points.  Returns (transient [L, S, T] f64, gradient [V, 3] f64 or None,
        pathlengths); with `data` ([L, S, T]; `weight` likewise, default 1) the vertex gradient of
        sum w (data - T)^2 / (L S) is computed (accumulated into `gradient` if given).  pairs=True renders the enumerated
        pairs instead of the record + combine kernels (the definition; same results).  vertex_normal / albedo: shading
        normals and per-vertex albedo, interpolated at the LASER leg's hit as in the pair renderer (record + combine kernels
        with extended records since round 6)."""
        grad = data is not None
        if refine_scale is None:
            refine_scale = 10 if grad else 1
        a = self._args(_lib.MODE_GRADIENT if grad else _lib.MODE_TRANSIENT, laser, laser_normal, vertices, faces, num_sample,
                       lower_bound, upper_bound, resolution, refine_scale, sigma_bin, vertex_normal=vertex_normal, albedo=albedo,
                       seed=seed, total_sources=total_lasers)
        _want(sensor, torch.float32, "sensor", 2); _want(sensor_normal, torch.float32, "sensor_normal", 2)
        assert sensor.shape[1] == 3 and sensor_normal.shape == sensor.shape and sensor.shape[0] > 0, "sensor/sensor_normal need to be Sx3"
        L, S, T = laser.shape[0], sensor.shape[0], self.num_bins(lower_bound, upper_bound, resolution)
        a.sensor, a.sensor_normal, a.n_sensors = _dptr(sensor), _dptr(sensor_normal), S
        a.product_pairs = 1 if pairs else 0
        transient = out if out is not None else torch.empty((L, S, T), dtype=torch.float64, device=self.device)
        assert tuple(transient.shape) == (L, S, T) and transient.is_contiguous(), "transient should be LxSxB"
        path = torch.empty(T, dtype=torch.float64, device=self.device)
        a.transient, a.pathlengths = _dptr(transient), _dptr(path)
        if grad:
            if weight is None:
                weight = torch.ones_like(data)
            _want(data, torch.float64, "data", 3); _want(weight, torch.float64, "weight", 3)
            assert tuple(data.shape) == (L, S, T) and tuple(weight.shape) == (L, S, T), "data / weight should be LxSxB"
            if gradient is None:
                gradient = torch.empty((vertices.shape[0], 3), dtype=torch.float64, device=self.device)
                zero_gradient = True
            else:
                _want(gradient, torch.float64, "gradient", 2)
                assert tuple(gradient.shape) == (vertices.shape[0], 3), "gradient dimension should be Vx3"
            a.zero_gradient = 1 if zero_gradient else 0
            a.data, a.weight, a.gradient = _dptr(data), _dptr(weight), _dptr(gradient)
            a.testing_flag, a.loss_test = int(testing_flag), int(loss_flag)
        self._run(a, (laser, laser_normal, sensor, sensor_normal, vertices, faces, data, weight, vertex_normal, albedo))
        return transient, (gradient if grad else None), path

    def render_gradient_scalar(self, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                               resolution, data, weight, refine_scale=10, sigma_bin=1, loss_flag=0,
                               albedo=None, alpha=None, vertex_normal=None, **kw):
        """Row A (d/d albedo) or, with `alpha`, the GGX d/d alpha. Returns (transient, 0-dim f64 tensor)."""
        mode = _lib.MODE_GRAD_ALPHA if alpha is not None else _lib.MODE_GRAD_ALBEDO
        a = self._args(mode, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                       resolution, refine_scale, sigma_bin, vertex_normal, albedo, alpha=alpha, **kw)
        L, T = origin.shape[0], self.num_bins(lower_bound, upper_bound, resolution)
        _want(data, torch.float64, "data", 2); _want(weight, torch.float64, "weight", 2)
        transient = torch.empty((L, T), dtype=torch.float64, device=self.device)
        path = torch.empty(T, dtype=torch.float64, device=self.device)
        out = torch.zeros(1, dtype=torch.float64, device=self.device)
        a.data, a.weight = _dptr(data), _dptr(weight)
        a.transient, a.pathlengths, a.scalar_out = _dptr(transient), _dptr(path), _dptr(out)
        a.loss_test = int(loss_flag)
        self._run(a, (origin, normal, vertices, faces, albedo, data, weight))
        return transient, out[0]

    def render_intensity(self, origin, normal, vertices, faces, num_sample, lower_bound, upper_bound,
                         vertex_normal=None, alpha=None, intensity=None, **kw):
        """Row X. Returns intensity [F] f64 (accumulated into if given)."""
        a = self._args(_lib.MODE_INTENSITY, origin, normal, vertices, faces, num_sample, lower_bound,
                       upper_bound, 1.0, 1, 1, vertex_normal, None, alpha=alpha, **kw)
        if intensity is None:
            intensity = torch.zeros(faces.shape[0], dtype=torch.float64, device=self.device)
        a.intensity = _dptr(intensity)
        self._run(a, (origin, normal, vertices, faces, vertex_normal))
        return intensity

    def mesh_regulariser(self, vertices, faces, face_affinity=None, overwrite=False):
        """SURVEY 8f-2 on device tensors: returns (value 0-dim f64 tensor | None, gradient [V,3] f64).
        face_affinity None -> area ("curvature") gradient, else normal smoothing."""
        _want(vertices, torch.float32, "vertices", 2); _want(faces, torch.int32, "faces", 2)
        _want(face_affinity, torch.int32, "face_affinity", 2)
        grad = torch.empty((vertices.shape[0], 3), dtype=torch.float64, device=self.device)
        val = torch.zeros(1, dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self._lib.nlos_mesh_regulariser(self._h, _dptr(vertices), vertices.shape[0], _dptr(faces),
                                                 faces.shape[0], _dptr(face_affinity), _dptr(grad), _dptr(val),
                                                 1 if overwrite else 0, ctypes.c_void_p(stream))
        _lib.check(rc, "nlos_mesh_regulariser")
        return (val[0] if face_affinity is not None else None), grad

    def debug_visibility(self, L, spt, F):
        """Diagnostics: (visibility cache uint32 [L, words, F] in sorted-face order, original face id per
        sorted slot) of the last render that kept it (keep_visibility=True or a gradient render)."""
        import numpy as np
        words = (spt + 31) // 32
        vis = np.zeros((L, words, F), np.uint32)
        fid = np.zeros(F, np.int32)
        for what, arr in ((0, vis), (1, fid)):
            n = self._lib.nlos_ctx_debug_read(self._h, what, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes)
            if n != arr.nbytes:
                raise _lib.NlosError("nlos_ctx_debug_read(%d) returned %d, expected %d" % (what, n, arr.nbytes))
        return vis, fid

    def debug_visibility_digest(self):
        """Diagnostics: 2 x 64-bit digest of the accepted-sample words of the last render that kept them, computed on the
        device (tools/soak.py: equal inputs must give equal digests, whatever order the workgroups ran in)."""
        import numpy as np
        dg = np.zeros(2, np.uint64)
        n = self._lib.nlos_ctx_debug_read(self._h, 3, dg.ctypes.data_as(ctypes.c_void_p), dg.nbytes)
        if n != 16:
            raise _lib.NlosError("nlos_ctx_debug_read(3) returned %d" % n)
        return int(dg[0]), int(dg[1])

    def debug_grid_paths(self, L):
        """Diagnostics: int32 [L] path code per source of the last single-workgroup grid launch (0 normal,
        0x100 + R coarsened to R x R after a cell-list overflow, 1 redone with the whole CU's LDS)."""
        import numpy as np
        codes = np.zeros(L, np.int32)
        n = self._lib.nlos_ctx_debug_read(self._h, 2, codes.ctypes.data_as(ctypes.c_void_p), codes.nbytes)
        if n != codes.nbytes:
            raise _lib.NlosError("nlos_ctx_debug_read(2) returned %d, expected %d" % (n, codes.nbytes))
        return codes

    def create_weighting_function(self, data, gamma=1.0):
        """exp_bunny/rendering.py:208-217 on a device tensor: (data/max + 0.1)^gamma, rescaled to sum to data.numel()."""
        _want(data, torch.float64, "data", 2)
        weight = torch.empty_like(data)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self._lib.nlos_create_weighting(self._h, _dptr(data), data.shape[0], data.shape[1], float(gamma),
                                                 _dptr(weight), ctypes.c_void_p(stream))
        _lib.check(rc, "nlos_create_weighting")
        return weight

    def weighted_l2(self, transient, data, weight=None):
        """L1 term of evaluate_loss_with_* (exp_bunny/rendering.py:360-364): sum w (T - data)^2 / rows, 0-dim f64."""
        _want(transient, torch.float64, "transient", 2); _want(data, torch.float64, "data", 2)
        _want(weight, torch.float64, "weight", 2)
        assert transient.shape == data.shape and (weight is None or weight.shape == data.shape), "shapes should be LxB"
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self._lib.nlos_weighted_l2(self._h, _dptr(transient), _dptr(data), _dptr(weight), data.shape[0],
                                            data.shape[1], _dptr(out), ctypes.c_void_p(stream))
        _lib.check(rc, "nlos_weighted_l2")
        return out[0]

    def intersect(self, origins, directions, vertices, faces, short=False):
        """Row E on device tensors: [N,3] (primID,u,v; NaN u,v on a miss) or [N] primIDs."""
        _want(origins, torch.float32, "origins", 2); _want(directions, torch.float32, "directions", 2)
        _want(vertices, torch.float32, "vertices", 2); _want(faces, torch.int32, "faces", 2)
        n = origins.shape[0]
        if short:
            out = torch.empty(n, dtype=torch.float32, device=self.device)
            o3, o1 = None, out
        else:
            out = torch.full((n, 3), float("nan"), dtype=torch.float32, device=self.device)
            o3, o1 = out, None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self._lib.nlos_intersect(self._h, _dptr(origins), _dptr(directions), n, _dptr(vertices),
                                          vertices.shape[0], _dptr(faces), faces.shape[0], _dptr(o3), _dptr(o1),
                                          ctypes.c_void_p(stream))
        _lib.check(rc, "nlos_intersect")
        return out


class TransientFunction(torch.autograd.Function):
    """transient = f(vertices); backward = analytic vertex gradient (Gaussian-smoothed in time).

    d loss / d v = sum_{l,b} G[l,b] * d T_smooth[l,b] / d v with G the upstream gradient: the
    reference's kernel evaluates -2 * difference * dT/dv / L, so difference := -G/2 and
    total_sources := 1 give exactly that contraction.
    """

    @staticmethod
    def forward(ctx, vertices, renderer, origin, normal, faces, num_sample, lower_bound, upper_bound,
                resolution, refine_scale, sigma_bin, normal_term, seed):
        v = vertices.detach().contiguous()
        # the function being differentiated: the reference's gradient driver renders the forward unsmoothed
        # for sigma_bin < 5 and Gaussian-smoothed at refine_scale otherwise
        # (smoothed_transient/stratifiedStreamedGradientRenderer.cpp:521-524)
        fwd_refine = refine_scale if sigma_bin >= 5 else 1
        transient, _ = renderer.render_transient(origin, normal, v, faces, num_sample, lower_bound,
                                                 upper_bound, resolution, fwd_refine, sigma_bin, keep_visibility=True,
                                                 total_sources=1, seed=seed)
        # backward may reuse this render's tree and visibility cache only while the context still holds them
        ctx.generations = (renderer.mesh_generation(), renderer.visibility_generation())
        ctx.renderer = renderer
        ctx.save_for_backward(v, origin, normal, faces)
        ctx.params = (num_sample, lower_bound, upper_bound, resolution, refine_scale, sigma_bin, normal_term, seed)
        return transient

    @staticmethod
    def backward(ctx, grad_out):
        v, origin, normal, faces = ctx.saved_tensors
        num_sample, lb, ub, res, refine, sigma_bin, normal_term, seed = ctx.params
        residual = (-0.5 * grad_out).to(torch.float64).contiguous()
        r = ctx.renderer
        mesh_gen, vis_gen = ctx.generations
        if (r.mesh_generation(), r.visibility_generation()) == (mesh_gen, vis_gen):
            _, grad, _ = r.render_gradient(origin, normal, v, faces, num_sample, lb, ub, res, residual=residual,
                                           refine_scale=refine, sigma_bin=sigma_bin, normal_term=normal_term,
                                           reuse_visibility=True, reuse_bvh=True, mesh_generation=mesh_gen,
                                           visibility_generation=vis_gen, total_sources=1, seed=seed)
        else:
            # another render or scene build used this context since forward(): build and trace again
            _, grad, _ = r.render_gradient(origin, normal, v, faces, num_sample, lb, ub, res, residual=residual,
                                           refine_scale=refine, sigma_bin=sigma_bin, normal_term=normal_term,
                                           total_sources=1, seed=seed)
        return (grad.to(v.dtype),) + (None,) * 12


def render_transient_autograd(renderer, vertices, origin, normal, faces, num_sample, lower_bound,
                              upper_bound, resolution, refine_scale=10, sigma_bin=1, normal_term=0, seed=0):
    """Differentiable transient: gradients flow to `vertices` (float32 [V,3] on the GPU)."""
    return TransientFunction.apply(vertices, renderer, origin, normal, faces, num_sample, lower_bound,
                                   upper_bound, resolution, refine_scale, sigma_bin, normal_term, seed)
