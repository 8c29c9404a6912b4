"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/nlos_hip.h declares, the ctypes mirror matches the C layout, the reference-shaped
Python modules validate arguments exactly like the reference's Cython signatures, and the
product fails loudly (no CPU fallback, no oracle import) when no GPU is present."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nlos_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nlos_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nlos_surface_optimization_amd import _lib
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libnlos_hip.so does not export %s" % n
        assert n in _lib.SYMBOLS, "ctypes table misses %s" % n
    assert set(_lib.SYMBOLS) <= set(names)


def test_render_args_layout_matches_c():
    from nlos_surface_optimization_amd import _lib
    assert ctypes.sizeof(_lib.RenderArgs) == _lib.lib().nlos_sizeof_render_args()
    a = _lib.RenderArgs()
    _lib.lib().nlos_render_args_init(ctypes.byref(a))
    assert (a.refine_scale, a.sigma_bin, a.normal_term, a.clamp, a.vertex_num) == (1, 1, -1, 1, -1)
    assert a.mode == 0 and a.reuse_visibility == 0 and not a.residual


def test_num_bins_matches_reference_float32_ceil():
    from nlos_surface_optimization_amd import _lib
    assert _lib.num_bins(0.0, 2.0, 2.0 ** -5) == 64
    assert _lib.num_bins(0.625, 1.625, 2.0 ** -9) == 512
    assert _lib.num_bins(0.0, 1200 * 1.2e-3, 1.2e-3) in (1200, 1201)   # python-side ub product, see renderer._num_bins


def _no_gpu():
    from nlos_surface_optimization_amd import _lib
    return _lib.device_count() == 0


def test_no_gpu_fails_loudly_no_fallback(cfg1):
    """Without a GPU every render raises; nothing silently computes on the CPU."""
    if not _no_gpu():
        pytest.skip("GPU present")
    from nlos_surface_optimization_amd import _lib, embree_intersector, renderer
    c = cfg1
    tr, path = np.zeros((4, 64)), np.zeros(64)
    with pytest.raises(_lib.NlosError, match="no HIP device"):
        renderer.renderStreamedTransient(c["origin"], c["normal"], c["v"], c["f"], 256, c["lb"], c["ub"], c["res"],
                                         tr, path, 1, 1)
    assert np.all(tr == 0)
    out = np.zeros((1, 3), np.float32)
    with pytest.raises(_lib.NlosError):
        embree_intersector.embree3_tbb_intersection(c["origin"][:1], c["normal"][:1], c["v"], c["f"], out)
    h = ctypes.c_void_p()
    assert _lib.lib().nlos_ctx_create(0, ctypes.byref(h)) == -3


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "nlos_surface_optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn
                assert "nlos_oracle" not in text, fn


def test_typed_array_checks_mirror_cython(cfg1):
    from nlos_surface_optimization_amd import embree_intersector, ggx, renderer, renderer_v1
    c = cfg1
    o, n, v, f = c["origin"], c["normal"], c["v"], c["f"]
    tr, path, grad = np.zeros((4, 64)), np.zeros(64), np.zeros((4, 3))
    data, w = np.zeros((4, 64)), np.ones((4, 64))
    args = (o, n, v, f, 256, c["lb"], c["ub"], c["res"], tr, path, 1, 1)
    # dtype / layout -> ValueError (Cython buffer mismatch)
    with pytest.raises(ValueError, match="dtype mismatch"):
        renderer.renderStreamedTransient(o.astype(np.float64), *args[1:])
    with pytest.raises(ValueError, match="dtype mismatch"):
        renderer.renderStreamedTransient(o, n, v, f.astype(np.int64), *args[4:])
    with pytest.raises(ValueError, match="C-contiguous"):
        renderer.renderStreamedTransient(np.asfortranarray(o), *args[1:])
    with pytest.raises(ValueError, match="dimensions"):
        renderer.renderStreamedTransient(o[0], *args[1:])
    with pytest.raises(TypeError):
        renderer.renderStreamedTransient(o.tolist(), *args[1:])
    # shapes -> AssertionError with the reference's messages (renderer.pyx:94-111)
    with pytest.raises(AssertionError, match="origin needs to be Lx3"):
        renderer.renderStreamedTransient(np.zeros((4, 2), np.float32), *args[1:])
    with pytest.raises(AssertionError, match="normal needs to be Lx3"):
        renderer.renderStreamedTransient(o, n[:3], *args[2:])
    with pytest.raises(AssertionError, match="transient dimension"):
        renderer.renderStreamedTransient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], np.zeros((4, 63)), path, 1, 1)
    with pytest.raises(AssertionError, match="pathlength dimension"):
        renderer.renderStreamedTransient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], tr, np.zeros(65), 1, 1)
    with pytest.raises(AssertionError, match="gradient dimension should be Vx3"):
        renderer.renderStreamedGradient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], tr, path, np.zeros((5, 3)),
                                        data, w, 10, 1, 1, 0)
    with pytest.raises(AssertionError, match="weighting should be LxB"):
        renderer.renderStreamedGradient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], tr, path, grad, data,
                                        np.ones((4, 60)), 10, 1, 1, 0)
    with pytest.raises(AssertionError, match="albedo"):
        renderer.renderStreamedTransientwAlbedo(o, n, v, np.ones(3, np.float32), f, 256, c["lb"], c["ub"], c["res"],
                                                tr, path, 1, 1)
    with pytest.raises(AssertionError, match="vertex normal needs to be Vx3"):
        ggx.renderStreamedTransientShading(o, n, v, np.zeros((3, 3), np.float32), f, 0.3, 256, c["lb"], c["ub"],
                                           c["res"], tr, path, 1, 1)
    with pytest.raises(AssertionError, match="intensity should be"):
        renderer.renderStreamedTriangleIntensity(o, n, v, f, 256, c["lb"], c["ub"], np.zeros(3))
    with pytest.raises(AssertionError, match="gradient dimension"):
        renderer.renderStreamedVertexGradient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], np.zeros((63, 3)), 0, 10, 1)
    with pytest.raises(AssertionError, match="data transient dimension"):
        renderer_v1.renderStreamedGradient(o, n, v, f, 256, c["lb"], c["ub"], c["res"], 1, tr, path, grad,
                                           np.zeros((4, 60)))
    with pytest.raises(AssertionError, match="Origin and Direction"):
        embree_intersector.embree3_tbb_intersection(o, n[:2], v, f, np.zeros((4, 3), np.float32))
    with pytest.raises(AssertionError, match="barycoord needs to be Nx3"):
        embree_intersector.embree3_tbb_intersection(o, n, v, f, np.zeros((4, 2), np.float32))
    with pytest.raises(AssertionError, match="barycoord needs to be Nx1"):
        embree_intersector.embree3_tbb_short_intersection(o, n, v, f, np.zeros(3, np.float32))


def test_facade_signatures_exist():
    import inspect
    from nlos_surface_optimization_amd import rendering, rendering_v1
    assert list(inspect.signature(rendering.inverseRendering).parameters) == ["mesh", "data", "weight", "opt"]
    assert list(inspect.signature(rendering.forwardRendering).parameters) == ["mesh", "opt"]
    assert list(inspect.signature(rendering_v1.inverseRendering).parameters) == ["mesh", "data", "opt"]
    assert list(inspect.signature(rendering.space_carving_projection).parameters) == ["v", "space_carving_mesh"]
    w = rendering.create_weighting_function(np.array([[1.0, 2.0], [3.0, 4.0]]), 0)
    assert np.allclose(w, 1.0)


def test_public_headers_are_plain_c(tmp_path):
    """The drop-in boundary must be bindable from C / cgo / JNI / ctypes: include/nlos_hip.h (and the
    oracle's header) compile as strict C99, no C++ or HIP types in the signatures."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "%s"\n#include "%s"\nint main(void) { nlos_render_args a; (void)a; return nlos_sizeof_render_args() > 0 ? 0 : 1; }\n'
                   % (os.path.join(ROOT, "include", "nlos_hip.h"), os.path.join(ROOT, "oracle", "nlos_oracle.h")))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_bench_contract_helpers_run_without_a_gpu():
    """bench.py's argument parsing and the `config` object of its JSON line (CPU-only parts)."""
    import json
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0 and a.bins == 512 and a.num_sample == 20000
    c = bench.workload_config(a, 64, 512, 4967, 2432, 5, 4096, 1)
    assert "64x64 confocal sources" in c["workload"] and "bunny_5k" in c["workload"] and "model" not in c
    json.dumps(c)
    for flags in (["--non-confocal"], ["--subdivide", "1"], ["--faces", "6000"], ["--mesh", "mannequin", "--bins", "1024"],
                  ["--forward-only"]):
        b = bench.parse_args(flags)
        w = bench.workload_config(b, 32, b.bins, 1000, 500, 4, 1024, 1)["workload"]
        assert ("side measurement" in w) == (flags[0] in ("--non-confocal", "--subdivide", "--faces"))
    assert bench.METRIC.startswith("surface samples/sec fwd+grad")


def test_environment_switches_are_listed_with_their_defaults():
    """Round 6: one accessor reads every behaviour-changing environment switch once per process; nlos_env_report lists them.
    Run in a child process with two of them set, so that this process's own (already read) values do not matter."""
    code = ("import json; from nlos_surface_optimization_amd import _lib; print(json.dumps(_lib.env_report()))")
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("NLOS_"):
            del env[k]
    env["NLOS_ROW_LDS_MAX"] = "4096"
    env["NLOS_FWD_ORDER"] = "0"
    import json
    import subprocess
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    want = {"NLOS_TILE_THRESHOLD": "6200", "NLOS_LAZY_TREE": "1", "NLOS_FUSE_RESIDUAL": "1", "NLOS_TILE_TRIS": "3000", "NLOS_VIS_ITEMS": "1",
            "NLOS_GEO_CACHE": "1", "NLOS_GEO_CACHE_MAX_GB": "-1", "NLOS_ROW_LDS_MAX": "4096", "NLOS_GRAD_WIDE": "1", "NLOS_GRAD_MIN_SOURCES": "1",
            "NLOS_FWD_ORDER": "0", "NLOS_GEO_MAX_SPT": "8"}
    for k, v in want.items():
        assert rep.get(k) == v, (k, rep.get(k))
    assert int(rep["NLOS_TILE_SCRATCH_MAX"]) == 32 << 30


def test_bench_config_presets_name_the_baseline_configurations():
    """bench.py --config: the presets of BASELINE.json's configurations and of the reference's experiment shape."""
    import bench
    a = bench.parse_args([])
    assert (a.config, a.grid, a.bins, a.mesh, a.half, a.res_m, a.alpha) == ("metric", 64, 512, "bunny_5k", 0.25, None, None)
    a = bench.parse_args(["--config", "2"])
    assert a.grid == 32 and a.forward_only and a.bins == 512
    a = bench.parse_args(["--config", "4"])
    assert (a.mesh, a.bins, a.half, a.lb, a.res_m, a.measurement) == ("mannequin", 1024, 0.35, 0.0, 2.4e-3, True)
    a = bench.parse_args(["--config", "4pairs"])
    assert a.non_confocal and not a.measurement and a.mesh == "mannequin"
    a = bench.parse_args(["--config", "5"])
    assert a.alpha == 0.3 and a.poisson and a.bins == 1024
    a = bench.parse_args(["--config", "exp"])
    assert (a.bins, a.lb, a.res_m) == (1200, 0.0, 1.2e-3)
    from nlos_surface_optimization_amd import _lib
    assert _lib.num_bins(0.0, 1200 * 1.2e-3, 1.2e-3) == 1200       # the window the reference's scripts pass gives their 1200 bins
    assert a.sustain_seconds == 6.0
