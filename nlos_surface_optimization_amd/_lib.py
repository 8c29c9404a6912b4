"""ctypes binding of libnlos_hip.so (the C ABI declared in include/nlos_hip.h).

There is no CPU fallback: if the shared library is missing, or no AMD GPU is
visible when a render is requested, the call raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnlos_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None

c_f32p = ctypes.c_void_p
c_f64p = ctypes.c_void_p
c_i32p = ctypes.c_void_p


class NlosError(RuntimeError):
    """A libnlos_hip call returned a non-zero status."""


class RenderArgs(ctypes.Structure):
    """Mirror of `nlos_render_args` (include/nlos_hip.h, section 2)."""
    _fields_ = [
        ("mode", ctypes.c_int32),
        ("origin", ctypes.c_void_p),
        ("normal", ctypes.c_void_p),
        ("L", ctypes.c_int32),
        ("source_offset", ctypes.c_int64),
        ("total_sources", ctypes.c_int32),
        ("vertices", ctypes.c_void_p),
        ("V", ctypes.c_int32),
        ("faces", ctypes.c_void_p),
        ("F", ctypes.c_int32),
        ("vertex_normal", ctypes.c_void_p),
        ("albedo", ctypes.c_void_p),
        ("num_samples", ctypes.c_int32),
        ("lower_bound", ctypes.c_float),
        ("upper_bound", ctypes.c_float),
        ("resolution", ctypes.c_float),
        ("refine_scale", ctypes.c_int32),
        ("sigma_bin", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("data", ctypes.c_void_p),
        ("weight", ctypes.c_void_p),
        ("transient", ctypes.c_void_p),
        ("pathlengths", ctypes.c_void_p),
        ("gradient", ctypes.c_void_p),
        ("intensity", ctypes.c_void_p),
        ("scalar_out", ctypes.c_void_p),
        ("testing_flag", ctypes.c_int32),
        ("loss_test", ctypes.c_int32),
        ("normal_term", ctypes.c_int32),
        ("clamp", ctypes.c_int32),
        ("use_ggx", ctypes.c_int32),
        ("ggx_alpha", ctypes.c_float),
        ("vertex_num", ctypes.c_int32),
        ("w_width", ctypes.c_int32),
        ("reuse_bvh", ctypes.c_int32),
        ("residual", ctypes.c_void_p),
        ("keep_visibility", ctypes.c_int32),
        ("reuse_visibility", ctypes.c_int32),
        ("force_bvh", ctypes.c_int32),
        ("sensor", ctypes.c_void_p),
        ("sensor_normal", ctypes.c_void_p),
        ("jitter_weight", ctypes.c_void_p),
        ("jitter_grad", ctypes.c_void_p),
        ("jitter_offset", ctypes.c_int32),
        ("jitter_length", ctypes.c_int32),
        ("mesh_generation", ctypes.c_int64),
        ("visibility_generation", ctypes.c_int64),
        ("zero_gradient", ctypes.c_int32),
        ("v1_sampled_point", ctypes.c_int32),
        ("source_stride", ctypes.c_int32),
        ("shared_samples", ctypes.c_int32),
        ("n_sensors", ctypes.c_int32),
        ("product_pairs", ctypes.c_int32),
    ]


class PathInfo(ctypes.Structure):
    """Mirror of `nlos_path_info` (include/nlos_hip.h): the kernels the last render took and why."""
    _fields_ = [
        ("backend", ctypes.c_int32),
        ("reason", ctypes.c_int32),
        ("grid_R", ctypes.c_int32),
        ("tiles", ctypes.c_int32),
        ("tile_cap", ctypes.c_int32),
        ("chunks", ctypes.c_int32),
        ("rows_in_lds", ctypes.c_int32),
        ("gradient_kernel", ctypes.c_int32),
        ("workgroups", ctypes.c_int64),
        ("coarsened", ctypes.c_int64),
        ("big_lds", ctypes.c_int64),
        ("bvh_queries", ctypes.c_int64),
        ("rays_traced", ctypes.c_int64),
        ("samples_accepted", ctypes.c_int64),
    ]


PATH_NAMES = {0: "none", 1: "grid", 2: "tiled-grid", 3: "bvh"}
REASON_NAMES = {0: "", 1: "force_bvh", 2: "mesh below 64 faces", 3: "rows and cell tables leave no LDS for the cell lists",
                4: "tile limits", 5: "non-confocal pairs in a mode the grid passes do not carry", 6: "mesh beyond one workgroup's grid",
                7: "time window outside the grid trace's arithmetic range"}
GRADIENT_KERNEL_NAMES = {0: "none", 1: "source-major, LDS accumulator", 2: "source-major, global atomics", 3: "face-major"}


MODE_TRANSIENT = 0
MODE_GRADIENT = 1
MODE_INTENSITY = 2
MODE_GRAD_ALBEDO = 3
MODE_GRAD_ALPHA = 4
MODE_VERTEX_GRADIENT = 5
MODE_GRADIENT_V1 = 6

# every symbol include/nlos_hip.h declares: name -> (restype, argtypes)
_I, _F, _P, _U64, _I64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64
SYMBOLS = {
    "nlos_last_error": (ctypes.c_char_p, []),
    "nlos_device_count": (_I, []),
    "nlos_version": (_I, []),
    "nlos_env_report": (_I, [ctypes.c_char_p, _I]),
    "nlos_streamed_render_transient": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _I, _I]),
    "nlos_streamed_render_intensity": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _F, _P]),
    "nlos_streamed_render_gradient": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I, _I]),
    "nlos_streamed_render_gradient_w_albedo": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I, _I]),
    "nlos_streamed_render_gradient_albedo": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _I, _I, _I, _I, _P]),
    "nlos_streamed_render_vertex_gradient": (_I, [_I, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _F, _P, _I, _I]),
    "nlos_ggx_streamed_render_transient": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _F, _I, _F, _F, _F, _P, _P, _I, _I]),
    "nlos_ggx_streamed_render_intensity": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _F, _I, _F, _F, _P]),
    "nlos_ggx_streamed_render_gradient": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _F, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I]),
    "nlos_ggx_streamed_render_gradient_alpha": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _F, _I, _F, _F, _F, _P, _P, _I, _I, _P]),
    "nlos_v1_streamed_render_gradient": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _I, _F, _F, _F, _I, _P, _P, _P]),
    "nlos_v1_streamed_render_transient": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P]),
    "nlos_v1_render_transient": (_I, [_P, _P, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P]),
    "nlos_streamed_render_normal_smoothing": (_I, [_P, _I, _P, _I, _P, _P, _P]),
    "nlos_streamed_render_curvature_grad": (_I, [_P, _I, _P, _I, _P]),
    "nlos_set_regulariser_overwrite": (None, [_I]),
    "nlos_ctx_debug_read": (_I64, [_P, _I, _P, _I64]),
    "nlos_adam_modified_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, _P]),
    "nlos_create_weighting": (_I, [_P, _P, _I, _I, ctypes.c_double, _P, _P]),
    "nlos_weighted_l2": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "nlos_mesh_regulariser": (_I, [_P, _P, _I, _P, _I, _P, _P, _P, _I, _P]),
    "nlos_jitter_streamed_render_transient": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _P, _I, _I, _P, _P]),
    "nlos_jitter_streamed_render_gradient": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F, _P, _P, _I, _I, _P, _P, _P, _I]),
    "nlos_nonconfocal_product_render_transient": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P]),
    "nlos_nonconfocal_product_render_gradient": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I, _I]),
    "nlos_nonconfocal_render_transient": (_I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _I, _I]),
    "nlos_nonconfocal_render_gradient": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I, _I]),
    "nlos_ggx_nonconfocal_render_transient": (_I, [_P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _F, _I, _F, _F, _F, _P, _P, _I, _I]),
    "nlos_ggx_nonconfocal_render_gradient": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _F, _I, _F, _F, _F, _P, _P, _P, _I, _I, _I, _I]),
    "nlos_embree3_tbb_line_intersection": (_I, [_P, _P, _I, _P, _I, _P, _I, _P]),
    "nlos_embree3_tbb_short_line_intersection": (_I, [_P, _P, _I, _P, _I, _P, _I, _P]),
    "nlos_barycentric_to_world_n": (_I, [_P, _I, _P, _I, _P, _I, _P]),
    "nlos_set_default_seed": (None, [_U64]),
    "nlos_set_default_device": (None, [_I]),
    "nlos_ctx_create": (_I, [_I, ctypes.POINTER(ctypes.c_void_p)]),
    "nlos_ctx_destroy": (None, [_P]),
    "nlos_ctx_scratch_bytes": (_I64, [_P]),
    "nlos_sizeof_render_args": (_I, []),
    "nlos_render_args_init": (None, [ctypes.POINTER(RenderArgs)]),
    "nlos_render": (_I, [_P, ctypes.POINTER(RenderArgs), _P]),
    "nlos_intersect": (_I, [_P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _P]),
    "nlos_num_bins": (_I, [_F, _F, _F]),
    "nlos_ctx_enable_timing": (None, [_P, _I]),
    "nlos_ctx_last_timing": (_I, [_P, _P]),
    "nlos_ctx_timing_reset": (None, [_P]),
    "nlos_ctx_timing_mean": (_I, [_P, _P, _P]),
    "nlos_ctx_mesh_generation": (_I64, [_P]),
    "nlos_ctx_visibility_generation": (_I64, [_P]),
    "nlos_ctx_last_path": (_I, [_P, ctypes.POINTER(PathInfo), _I]),
    "nlos_ctx_check": (_I, [_P]),
}


def build(force=False):
    """Compile libnlos_hip.so for gfx950 with hipcc (recipe: csrc/Makefile)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "nlos_hip.h"))
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", CSRC, "-j4", "all"])
    return LIB_PATH


def source_stamp():
    """Identity of the kernels a measurement belongs to: sha256 over the sources libnlos_hip.so is built from
    (csrc/*.hip, csrc/*.h, csrc/Makefile, include/*.h; file names and contents, sorted), first 16 hex digits, plus the
    same of the built library.  The SOURCE hash is what a profile is matched on: two builds of the same sources in
    different directories differ in a few hundred bytes (embedded paths), so the library hash is informative only.
    Used by tools/round_summary.py (stamps profiles/pmc_summary.json) and bench.py (refuses counters of another build)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(_HERE, "..", "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    for f in sorted(files, key=os.path.basename):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    out = {"source_sha256_16": h.hexdigest()[:16], "lib_sha256_16": None}
    if os.path.exists(LIB_PATH):
        with open(LIB_PATH, "rb") as fh:
            out["lib_sha256_16"] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


def _preload_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7, loaded by file name through RPATH $ORIGIN), while libnlos_hip.so
    asks for libamdhip64.so.7 and would otherwise bind /opt/rocm's copy: two runtimes in one
    process, and whichever initialises second sees no GPU.  Pre-loading torch's copy (when
    torch is installed) makes the dynamic linker resolve both requests to the same object,
    whatever the import order.  Without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
    return cand


def lib():
    """Load the shared library (never builds implicitly on a GPU box: ship the .so)."""
    global _lib
    if _lib is None:
        _preload_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise NlosError(
                "libnlos_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().nlos_last_error()
        raise NlosError("%s failed (status %d): %s" % (what or "libnlos_hip call", rc,
                                                      msg.decode() if msg else "?"))


def device_count():
    return int(lib().nlos_device_count())


def env_report():
    """The library's environment switches as {name: value-as-read} (nlos_env_report)."""
    buf = ctypes.create_string_buffer(2048)
    lib().nlos_env_report(buf, 2048)
    out = {}
    for line in buf.value.decode().splitlines():
        k, _, rest = line.partition("=")
        out[k] = rest.split(" ")[0]
    return out


def num_bins(lb, ub, res):
    return int(lib().nlos_num_bins(lb, ub, res))
