"""ctypes bindings of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

`oracle/nlos_oracle.c` is a plain-C restatement of the reference's transient
renderer (see `oracle/nlos_oracle.h` for the file:line citations and the parity
status).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
``cpu_baseline`` leg import this package, and only as the checker / reported CPU
baseline.  Nothing under ``nlos_surface_optimization_amd/`` imports it.
"""
import contextlib
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnlos_oracle.so")
_lib = None


class Opts(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("source_offset", ctypes.c_int64),
        ("total_sources", ctypes.c_int32),
        ("accel", ctypes.c_int32),
        ("threads", ctypes.c_int32),
        ("use_ggx", ctypes.c_int32),
        ("ggx_alpha", ctypes.c_float),
        ("normal_term", ctypes.c_int32),
        ("clamp", ctypes.c_int32),
        ("source_stride", ctypes.c_int32),
        ("shared_samples", ctypes.c_int32),
        ("sampled_point", ctypes.c_int32),
    ]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    deps = [os.path.join(_HERE, "nlos_oracle.c"), os.path.join(_HERE, "nlos_oracle.h"),
            os.path.join(_HERE, "..", "include", "nlos_contract.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.nlos_oracle_render_gradient_scalar.restype = ctypes.c_double
        for name in ("nlos_oracle_ggx_eval", "nlos_oracle_ggx_eval_adiff", "nlos_oracle_ggx_eval_nwsdiff"):
            getattr(_lib, name).restype = ctypes.c_float
            getattr(_lib, name).argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def set_graze_ratio(ratio):
    """Grazing rule of the numeric contract (include/nlos_contract.h): ratio < 0 restores the contract's value,
    0 switches the rule off (with accel=0: the reference's rule-free all-faces definition).  Process-global."""
    f = lib().nlos_oracle_set_graze_ratio
    f.argtypes = [ctypes.c_float]
    f.restype = None
    f(float(ratio))


def graze_ratio():
    f = lib().nlos_oracle_graze_ratio
    f.restype = ctypes.c_float
    return float(f())


@contextlib.contextmanager
def rule_free():
    """with oracle.rule_free(): ... -- renders inside use the reference's rule-free hit test (Embree accepts every
    den != 0, SMO/transient_and_gradient.cpp:199-206); call the renders with accel=0 for the all-faces definition."""
    set_graze_ratio(0.0)
    try:
        yield
    finally:
        set_graze_ratio(-1.0)


def make_opts(seed=0, source_offset=0, total_sources=0, accel=0, threads=0,
              ggx_alpha=None, normal_term=-1, clamp=1, sampled_point=0, source_stride=1, shared_samples=0):
    o = Opts()
    o.seed = seed
    o.source_offset = source_offset
    o.total_sources = total_sources
    o.accel = accel
    o.threads = threads
    o.use_ggx = 0 if ggx_alpha is None else 1
    o.ggx_alpha = 0.0 if ggx_alpha is None else float(ggx_alpha)
    o.normal_term = normal_term
    o.clamp = clamp
    o.sampled_point = sampled_point
    o.source_stride = int(source_stride)
    o.shared_samples = int(shared_samples)
    return o


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_bins(lb, ub, res):
    f = lib().nlos_oracle_num_bins
    f.argtypes = [ctypes.c_float] * 3
    return int(f(lb, ub, res))


def sample(seed, k):
    S = ctypes.c_float()
    T = ctypes.c_float()
    f = lib().nlos_oracle_sample
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    f(seed, k, ctypes.byref(S), ctypes.byref(T))
    return S.value, T.value


def intersect(origins, dirs, v, f, accel=0, threads=0, short=False):
    """Row E. Returns float32 [N,3] (primID,u,v; untouched cols stay NaN on miss) or [N]."""
    o, d, v, f = _f32(origins), _f32(dirs), _f32(v), _i32(f)
    n = o.shape[0]
    if short:
        out = np.empty(n, dtype=np.float32)
        rc = lib().nlos_oracle_intersect(_p(o), _p(d), n, _p(v), v.shape[0], _p(f), f.shape[0],
                                         None, _p(out), accel, threads)
    else:
        out = np.full((n, 3), np.nan, dtype=np.float32)
        rc = lib().nlos_oracle_intersect(_p(o), _p(d), n, _p(v), v.shape[0], _p(f), f.shape[0],
                                         _p(out), None, accel, threads)
    if rc:
        raise ValueError("oracle intersect failed rc=%d" % rc)
    return out


def barycentric_to_world(v, f, bary):
    v, f, bary = _f32(v), _i32(f), _f32(bary)
    out = np.zeros((bary.shape[0], 3), dtype=np.float32)
    lib().nlos_oracle_barycentric_to_world(_p(v), _p(f), _p(bary), bary.shape[0], _p(out))
    return out


def render_transient(origin, normal, v, f, num_sample, lb, ub, res, refine=1, sigma_bin=1,
                     vnormal=None, albedo=None, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    vnormal, albedo = _f32(vnormal), _f32(albedo)
    L, T = origin.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((L, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_transient
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rc = fn(_p(origin), L, _p(normal), _p(v), v.shape[0], _p(vnormal), _p(albedo), _p(f), f.shape[0],
            int(num_sample), lb, ub, res, _p(transient), _p(path), refine, sigma_bin, ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_transient failed rc=%d" % rc)
    return transient, path


def render_gradient(origin, normal, v, f, num_sample, lb, ub, res, data, weight, refine=10,
                    sigma_bin=1, testing_flag=1, loss_flag=0, vnormal=None, albedo=None,
                    gradient=None, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    vnormal, albedo = _f32(vnormal), _f32(albedo)
    data, weight = _f64(data), _f64(weight)
    L, T = origin.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((L, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    if gradient is None:
        gradient = np.zeros((v.shape[0], 3), dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_gradient
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rc = fn(_p(data), _p(weight), _p(origin), L, _p(normal), _p(v), v.shape[0], _p(vnormal),
            _p(albedo), _p(f), f.shape[0], int(num_sample), lb, ub, res, _p(transient), _p(path),
            _p(gradient), refine, sigma_bin, testing_flag, loss_flag, ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_gradient failed rc=%d" % rc)
    return transient, gradient, path


def mesh_regulariser(v, f, affinity=None, overwrite=False):
    """Returns (value, gradient [V,3]); affinity None -> area ("curvature") gradient, value 0."""
    v, f = _f32(v), _i32(f)
    aff = None if affinity is None else _i32(affinity)
    grad = np.zeros((v.shape[0], 3), dtype=np.float64)
    fn = lib().nlos_oracle_mesh_regulariser
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int]
    val = fn(_p(v), v.shape[0], _p(f), f.shape[0], _p(aff), _p(grad), 1 if overwrite else 0)
    return float(val), grad


def render_jitter(origin, normal, v, f, num_sample, lb, ub, res, jitter_weight, jitter_offset,
                  jitter_grad=None, data=None, weight=None, testing_flag=1, vnormal=None, albedo=None,
                  gradient=None, **kw):
    """jitter/ module: forward (data None) or forward + gradient. Returns (transient, gradient|None, path)."""
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    vnormal, albedo = _f32(vnormal), _f32(albedo)
    jw = _f64(np.asarray(jitter_weight).ravel())
    jg = None if jitter_grad is None else _f64(np.asarray(jitter_grad).ravel())
    L, T = origin.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((L, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    if data is not None:
        data = _f64(data)
        weight = _f64(np.ones_like(data) if weight is None else weight)
        if gradient is None:
            gradient = np.zeros((v.shape[0], 3), dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_jitter
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    rc = fn(_p(data), _p(weight), _p(origin), L, _p(normal), _p(v), v.shape[0], _p(vnormal), _p(albedo),
            _p(f), f.shape[0], int(num_sample), lb, ub, res, _p(jw), _p(jg), int(jitter_offset), jw.shape[0],
            _p(transient), _p(path), _p(gradient) if data is not None else None, testing_flag, ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_jitter failed rc=%d" % rc)
    return transient, (gradient if data is not None else None), path


def render_nonconfocal(laser, laser_normal, sensor, sensor_normal, v, f, num_sample, lb, ub, res,
                       data=None, weight=None, refine=10, sigma_bin=1, testing_flag=1, loss_flag=0,
                       vnormal=None, albedo=None, gradient=None, **kw):
    """Row N: (laser[i], sensor[i]) pairs. Returns (transient, gradient or None, pathlengths)."""
    laser, laser_normal, sensor, sensor_normal = _f32(laser), _f32(laser_normal), _f32(sensor), _f32(sensor_normal)
    v, f, vnormal, albedo = _f32(v), _i32(f), _f32(vnormal), _f32(albedo)
    P, T = laser.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((P, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    if data is not None:
        data = _f64(data)
        weight = _f64(np.ones_like(data) if weight is None else weight)
        if gradient is None:
            gradient = np.zeros((v.shape[0], 3), dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_nonconfocal
    fn.argtypes = ([ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p])
    rc = fn(_p(data), _p(weight), _p(laser), _p(laser_normal), _p(sensor), _p(sensor_normal), P,
            _p(v), v.shape[0], _p(vnormal), _p(albedo), _p(f), f.shape[0], int(num_sample), lb, ub, res,
            _p(transient), _p(path), _p(gradient) if data is not None else None, refine, sigma_bin,
            testing_flag, loss_flag, ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_nonconfocal failed rc=%d" % rc)
    return transient, (gradient if data is not None else None), path


def render_product(laser, laser_normal, sensor, sensor_normal, v, f, num_sample, lb, ub, res, data=None, weight=None,
                   refine=None, **kw):
    """Row N as a product: every (laser[i], sensor[j]) combination, on sample points shared by all wall points.  The
    product is DEFINED as its pairs (include/nlos_hip.h, nlos_render_args.n_sensors), and that is how the checker
    computes it: the La * Sb pairs are enumerated and handed to the pair renderer with shared_samples = 1.
    Returns (transient [La, Sb, T], gradient or None, pathlengths); data / weight are [La, Sb, T]; the gradient is
    normalised by the La * Sb measurements."""
    laser, laser_normal, sensor, sensor_normal = _f32(laser), _f32(laser_normal), _f32(sensor), _f32(sensor_normal)
    La, Sb = laser.shape[0], sensor.shape[0]
    T = num_bins(lb, ub, res)
    li, sj = np.repeat(np.arange(La), Sb), np.tile(np.arange(Sb), La)
    if data is not None:
        data = np.ascontiguousarray(_f64(data).reshape(La * Sb, T))
        weight = None if weight is None else np.ascontiguousarray(_f64(weight).reshape(La * Sb, T))
    kw = dict(kw)
    kw["shared_samples"] = 1
    kw["refine"] = (1 if data is None else 10) if refine is None else refine     # (forward-only calls smooth when refine > 1)
    t, g, path = render_nonconfocal(laser[li], laser_normal[li], sensor[sj], sensor_normal[sj], v, f, num_sample, lb, ub, res,
                                    data=data, weight=weight, **kw)
    return t.reshape(La, Sb, T), g, path


def render_gradient_scalar(origin, normal, v, f, num_sample, lb, ub, res, data, weight,
                           refine=10, sigma_bin=1, loss_flag=0, albedo=None, wrt_alpha=False, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    albedo = _f32(albedo)
    data, weight = _f64(data), _f64(weight)
    L, T = origin.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((L, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_gradient_scalar
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p]
    g = fn(_p(data), _p(weight), _p(origin), L, _p(normal), _p(v), v.shape[0], _p(albedo), _p(f),
           f.shape[0], int(num_sample), lb, ub, res, _p(transient), _p(path), refine, sigma_bin,
           loss_flag, 1 if wrt_alpha else 0, ctypes.byref(o))
    return transient, float(g)


def render_intensity(origin, normal, v, f, num_sample, lb, ub, vnormal=None, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    vnormal = _f32(vnormal)
    inten = np.zeros(f.shape[0], dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_intensity
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(_p(origin), origin.shape[0], _p(normal), _p(v), v.shape[0], _p(vnormal), _p(f),
            f.shape[0], int(num_sample), lb, ub, _p(inten), ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_intensity failed rc=%d" % rc)
    return inten


def render_vertex_gradient(vertex_num, origin, normal, v, f, num_sample, lb, ub, res,
                           refine=10, sigma_bin=1, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    T = num_bins(lb, ub, res)
    grad = np.zeros((T, 3), dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_vertex_gradient
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p]
    rc = fn(int(vertex_num), _p(origin), origin.shape[0], _p(normal), _p(v), v.shape[0], _p(f),
            f.shape[0], int(num_sample), lb, ub, res, _p(grad), refine, sigma_bin, ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_vertex_gradient failed rc=%d" % rc)
    return grad


def render_gradient_v1(origin, normal, v, f, num_sample, lb, ub, res, data, w_width=0, **kw):
    origin, normal, v, f = _f32(origin), _f32(normal), _f32(v), _i32(f)
    data = _f64(data)
    L, T = origin.shape[0], num_bins(lb, ub, res)
    transient = np.zeros((L, T), dtype=np.float64)
    path = np.zeros(T, dtype=np.float64)
    grad = np.zeros((v.shape[0], 3), dtype=np.float64)
    o = make_opts(**kw)
    fn = lib().nlos_oracle_render_gradient_v1
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(_p(data), _p(origin), L, _p(normal), _p(v), v.shape[0], _p(f), f.shape[0],
            int(num_sample), lb, ub, res, w_width, _p(transient), _p(path), _p(grad),
            ctypes.byref(o))
    if rc:
        raise ValueError("oracle render_gradient_v1 failed rc=%d" % rc)
    return transient, grad, path


def ggx(alpha, n, w, which="eval"):
    n, w = _f32(n), _f32(w)
    fn = getattr(lib(), {"eval": "nlos_oracle_ggx_eval", "adiff": "nlos_oracle_ggx_eval_adiff",
                         "nwsdiff": "nlos_oracle_ggx_eval_nwsdiff"}[which])
    return float(fn(ctypes.c_float(alpha), _p(n), _p(w)))


def trace_sample(origin_l, normal_l, l_global, face, s, spt, v, f, lb, ub, res, seed=0):
    origin_l, normal_l, v, f = _f32(origin_l), _f32(normal_l), _f32(v), _i32(f)
    out = np.zeros(7, dtype=np.float64)
    fn = lib().nlos_oracle_trace_sample
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_void_p]
    ok = fn(_p(origin_l), _p(normal_l), int(l_global), int(face), int(s), int(spt), _p(v),
            v.shape[0], _p(f), f.shape[0], lb, ub, res, int(seed), _p(out))
    return int(ok), out
