#!/usr/bin/env python3
"""Benchmark of the hot path: confocal transient forward + per-vertex gradient on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one `renderStreamedGradient`-equivalent call on device-resident inputs
(BVH build + pass 1 + residual + pass 2), plus -- for N > 1 -- the single
RCCL all-reduce of the 3V-double vertex gradient.  Workload: BASELINE.json's metric configuration,
64x64 confocal sources x 512 bins on the ~5k-face bunny, num_sample = 20000 (spt = 5), refine = 10,
sigma_bin = 1.  For N > 1 the 64x64 grid is split into N contiguous source blocks, one per rank
(`--scaling strong`, the default: north_star's curve is this one workload at 1/2/4/8 GPUs);
`--scaling weak` renders a 64 x 64N grid instead, one 64x64 block per rank.

Launching: `python bench.py --gpus N` with N > 1 starts the N ranks itself -- the parent process never
touches the GPU and runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py`
as a child (one process per GPU over RCCL).  Started under torch.distributed.run directly
(WORLD_SIZE in the environment) it is a rank.  Fewer than N visible devices, a WORLD_SIZE that
differs from --gpus, or an RCCL group that did not come up with N ranks is an error (exit code != 0),
never a silent one-GPU measurement.

Rank 0 prints ONE JSON line.  Before any timing EVERY rank runs the parity gate BASELINE.md section 3 promises, on
the workload that is timed (the metric's, or any of the side measurements: pairs, forward only, other meshes): the
CPU oracle renders the first sources of the rank's block (at N = 1 the same block the `cpu_baseline` leg times),
and rows and vertex gradient of a GPU render of that block must agree (transient rel-L2 <= 1e-5 and
max-abs <= 1e-6 * max, gradient rel-L2 <= 1e-4); the ranks' verdicts are MIN-all-reduced; on failure nothing is
timed and the exit code is 1.  At N = 1 the line also carries `strong_share`: the per-step time of every rank's
block of the 2-, 4- and 8-way strong split, measured on this one GPU (`--as-rank K --of N` times a single block).
`roofline` is measured live with HIP events on the launch stream (nlos_ctx_timing_mean); `cpu_baseline`
times the CPU oracle (a port of the reference algorithm, kind "port") on rank 0 at N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "surface samples/sec fwd+grad; 64x64 sensors x 512 bins, bunny mesh"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PARITY_TOL = {"transient_rel_l2": 1e-5, "transient_max_abs_over_max": 1e-6, "gradient_rel_l2": 1e-4}


# BASELINE.json's other configurations on SURVEY 8(d)'s inputs, plus the shape every experiment script of the reference runs
# (`--config`, round 6): side measurements, each gated against the oracle like the metric line and carrying its own roofline.
CONFIGS = {
    "metric": {},
    # cfg 2 / 3: exp_bunny mesh, 32x32 confocal, 512 bins; forward only / forward + gradient
    "2": dict(grid=32, forward_only=True),
    "3": dict(grid=32),
    # cfg 4: exp_mannequin mesh, 64x64 wall points on +-0.35 m, 1024 bins of 2.4 mm from 0 -- confocal on the reference's own
    # measurement (exp_mannequin/transient.mat: tests/golden/mannequin_measurement.npz), and its non-confocal pairs
    "4": dict(mesh="mannequin", bins=1024, half=0.35, lb=0.0, res=2.4e-3, measurement=True),
    "4pairs": dict(mesh="mannequin", bins=1024, half=0.35, lb=0.0, res=2.4e-3, non_confocal=True),
    # cfg 5: GGX branch (alpha 0.3) + Poisson-noised measured transient, 64x64x1024
    "5": dict(bins=1024, alpha=0.3, poisson=True),
    # the shape of the reference's experiment scripts: 64x64 sources per call, 1200 bins x 1.2 mm from 0, refine 10, sigma_bin 1
    # (exp_bunny/test.py:33-44,62-66, exp_ggx/test1.py:21-22)
    "exp": dict(bins=1200, lb=0.0, res=1.2e-3),
}


def grid_sources(nx, ny, half):
    gx = np.linspace(-half, half, nx)
    gy = np.linspace(-half, half, ny)
    origin = np.array([[x, y, 0] for y in gy for x in gx], np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (origin.shape[0], 1))
    return origin, normal


def load_pmc_summary():
    """Counters of the dominant kernel from the committed rocprofv3 PMC summary (profiles/pmc_summary.json;
    separate --pmc passes, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE units with the gfx950 correction of
    MI355X_MICROARCH.md section HBM).  None if absent."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(p):
        return None
    try:
        with open(p) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def counters_of_this_build(pmc, kernel, comparable, L, F):
    """(hbm bytes per launch or None, issue block or {"stale": ...} or None, stamp of this build, stamp of the profile).
    Counters are reported only from a profile of THIS build: profiles/pmc_summary.json carries the hash of the kernel
    sources it was taken on (tools/round_summary.py, _lib.source_stamp); a summary of other sources -- or one without
    a stamp -- gives traffic None and issue {"stale": ...}, never another binary's numbers."""
    from nlos_surface_optimization_amd import _lib
    here = _lib.source_stamp()
    if not (pmc and comparable and pmc.get("kernel") == kernel and pmc.get("L") == L and pmc.get("F") == F):
        return None, None, here, None
    stamp = pmc.get("stamp") or {}
    if stamp.get("source_sha256_16") != here["source_sha256_16"]:
        return None, {"stale": "profiles/pmc_summary.json was taken on kernel sources %s, this build is %s: counters not reported"
                               % (stamp.get("source_sha256_16"), here["source_sha256_16"])}, here, stamp
    return pmc.get("hbm_bytes_per_launch"), pmc.get("issue"), here, stamp


# ------------------------------------------------------------------------------------------------
# CPU side of rank 0: the oracle as parity checker and as the reported baseline (never the product)
# ------------------------------------------------------------------------------------------------
def _oracle_block(v, f, origin, normal, lb, ub, res, num_sample, data_rows, n, threads, source_offset=0,
                  forward_only=False, sensor=None, sensor_normal=None, source_stride=1, ggx_alpha=None):
    """Rows and vertex gradient of the first n sources of a block by the CPU oracle (total_sources = n; RNG keys
    of the block's global source indices); returns (transient, gradient or None, seconds).  The same call for every
    workload the bench can time: confocal forward + gradient, forward only, non-confocal pairs, any mesh."""
    import oracle
    oracle.build()
    o, nn = np.ascontiguousarray(origin[:n]), np.ascontiguousarray(normal[:n])
    d = np.ascontiguousarray(data_rows[:n])
    w = np.ones_like(d)
    kw = dict(accel=1, threads=threads, seed=0, source_offset=int(source_offset), source_stride=int(source_stride))
    if ggx_alpha is not None:
        kw["ggx_alpha"] = float(ggx_alpha)
    t0 = time.perf_counter()
    if sensor is not None:
        tr, g, _ = oracle.render_nonconfocal(o, nn, np.ascontiguousarray(sensor[:n]), np.ascontiguousarray(sensor_normal[:n]),
                                             v, f, num_sample, lb, ub, res, data=None if forward_only else d,
                                             weight=None if forward_only else w, refine=10, sigma_bin=1,
                                             testing_flag=1, loss_flag=0, **kw)
    elif forward_only:
        tr, _ = oracle.render_transient(o, nn, v, f, num_sample, lb, ub, res, **kw)
        g = None
    else:
        tr, g, _ = oracle.render_gradient(o, nn, v, f, num_sample, lb, ub, res, d, w, refine=10, sigma_bin=1,
                                          testing_flag=1, loss_flag=0, **kw)
    return tr, g, time.perf_counter() - t0


def cpu_reference(v, f, origin, normal, lb, ub, res, num_sample, data_rows, budget_s, source_offset=0, share=1, **wk):
    """Oracle (CPU port of the reference algorithm, own BVH, per-thread buffers, literal 41-tap loop) on a bounded
    sample of the same workload: the first `n` sources of this rank's block.  budget_s > 0: sized to about that much
    wall time and reported as `cpu_baseline`; budget_s == 0: a small block for the parity gate only (at most 48
    sources and about 3 M rays; `share` ranks of one host run their gates at the same time and share its cores).
    Returns (n, transient rows, gradient, cpu_baseline dict or None)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    F = f.shape[0]
    spt = 1 + (num_sample - 1) // F
    L = origin.shape[0]

    def run(n, threads):
        return _oracle_block(v, f, origin, normal, lb, ub, res, num_sample, data_rows, n, threads, source_offset, **wk)

    if budget_s <= 0:
        n = int(min(48, L, max(4, 3.0e6 // max(F * spt, 1))))
        tr, g, _ = run(n, max(1, avail // max(share, 1)))
        return n, tr, g, None
    # pick the thread count that gives the best throughput on a short probe (SMT siblings and
    # container CPU quotas make "all logical CPUs" the wrong choice on some hosts)
    n0 = min(max(32, avail // 2), L)
    cands = sorted({max(1, avail), max(1, avail // 2), max(1, avail // 4)}, reverse=True)
    run(min(8, L), cands[0])                                   # untimed: thread pool + page faults
    probe = {c: min(run(n0, c)[2], run(n0, c)[2]) for c in cands}
    best = min(probe.values())
    cores = next(c for c in cands if probe[c] <= 1.15 * best)   # most threads within 15 % of the best
    # size the sample to ~budget_s of wall time, re-sizing once if the estimate was off
    n = int(max(n0, min(L, n0 * budget_s / max(probe[cores], 1e-3))))
    tr, g, t = run(n, cores)
    if t < 0.6 * budget_s and n < L:
        n = int(min(L, n * budget_s / max(t, 1e-3)))
        tr, g, t = run(n, cores)
    base = {"value": n * F * spt / t, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d sources of the same workload (%.1f s wall on %d threads; %d logical "
                      "CPUs available), oracle with its own BVH, OpenMP over (source, face) with per-thread "
                      "buffers" % (n, L, t, cores, avail)}
    return n, tr, g, base


def parity_gate(render_block, n, t_ref, g_ref):
    """the GPU render of the oracle's block must agree before anything is timed."""
    t_gpu, g_gpu = render_block(n)
    den_t = float(np.linalg.norm(t_ref))
    out = {
        "rows": int(n),
        "transient_rel_l2": float(np.linalg.norm(t_gpu - t_ref) / den_t) if den_t > 0 else float("inf"),
        "transient_max_abs_over_max": float(np.abs(t_gpu - t_ref).max() / np.abs(t_ref).max()) if den_t > 0 else float("inf"),
        "tolerance": PARITY_TOL,
        "checker": "CPU oracle (oracle/nlos_oracle.c), same sample keys, block of the timed workload's first sources",
    }
    if g_ref is not None:          # (forward-only workloads have no gradient to compare)
        den_g = float(np.linalg.norm(g_ref))
        out["gradient_rel_l2"] = float(np.linalg.norm(g_gpu - g_ref) / den_g) if den_g > 0 else float("inf")
    out["pass"] = bool(all(out[k] <= PARITY_TOL[k] for k in PARITY_TOL if k in out))
    return out


def time_api_paths(args, r, backend, origin_np, normal_np, v_np, f_np, data_np, lb, ub, res, T,
                   origin, normal, verts, faces, data, weight, grad):
    """Side measurements on the metric workload, never `value`: (1) the host-pointer drop-in -- numpy arrays in and out
    through the C ABI's section 1 (uploads of data / weight and the download of the rows are part of every call, as in the
    reference); (2) the autograd pair -- TransientFunction.forward (pass 1, visibility kept) + a weighted-L2 loss in torch +
    backward (pass 2 on the kept visibility).  Results of the drop-in are compared with the device path's (same seed)."""
    import torch
    from nlos_surface_optimization_amd import renderer as np_renderer
    from nlos_surface_optimization_amd import device as nd
    L = origin_np.shape[0]
    V = v_np.shape[0]
    n = args.dropin_steps
    out = {}
    # ---- numpy drop-in
    w_np = np.ones_like(data_np)
    tr_np = np.zeros((L, T))
    path_np = np.zeros(T)
    g_np = np.zeros((V, 3))

    def call():
        g_np[:] = 0.0
        np_renderer.renderStreamedGradient(origin_np, normal_np, v_np, f_np, args.num_sample, lb, ub, res, tr_np, path_np,
                                           g_np, data_np, w_np, 10, 1, 1, 0)
    for _ in range(3):
        call()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    out["dropin_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / n
    # same results as the device path (both use seed 0; the drop-in scales by 1 / L of its own call, like the bench step)
    t_dev, g_dev, _ = r.render_gradient(origin, normal, verts, faces, args.num_sample, lb, ub, res, data=data, weight=weight,
                                        refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0, gradient=grad, zero_gradient=True)
    backend.sync()
    dt = float(np.abs(t_dev.cpu().numpy() - tr_np).max())
    dg = float(np.abs(g_dev.cpu().numpy() - g_np).max() / max(float(np.abs(g_np).max()), 1e-300))
    out["dropin"] = {"steps": n, "max_abs_row_difference_vs_device_path": dt, "max_rel_gradient_difference_vs_device_path": dg,
                     "host_bytes_per_call": int(3 * L * T * 8 + 2 * L * 12 + V * 12 + f_np.size * 4 + V * 24),
                     "entry": "renderer.renderStreamedGradient(numpy) -> nlos_streamed_render_gradient (include/nlos_hip.h section 1)"}
    # ---- autograd pair
    vt = verts.detach().clone().requires_grad_(True)

    def ag():
        if vt.grad is not None:
            vt.grad = None
        tr = nd.render_transient_autograd(r, vt, origin, normal, faces, args.num_sample, lb, ub, res, refine_scale=10, sigma_bin=1)
        loss = (((tr - data) ** 2) * weight).sum() / L
        loss.backward()
    for _ in range(3):
        ag()
    backend.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        ag()
    backend.sync()
    out["autograd_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / n
    ga = vt.grad.detach().to(torch.float64).cpu().numpy()
    out["autograd"] = {"steps": n, "max_rel_gradient_difference_vs_device_path": float(np.abs(ga - g_dev.cpu().numpy()).max() / max(float(np.abs(ga).max()), 1e-300)),
                       "entry": "device.render_transient_autograd: TransientFunction.forward (keep_visibility) + torch loss + backward (reuse_visibility)"}
    return out


def workload_config(args, g, T, F, V, spt, L_total, world):
    """The `config` object of the JSON line (names the workload; the side-measurement flags say so)."""
    return {
        "workload": ("ONE RANK'S SHARE (rank %d of %d) of the strong split, side measurement, not the metric: " % (args.as_rank, args.of)
                     if args.of > 1 else "") +
                    ("BASELINE config %s (side measurement, not the metric): " % args.config if args.config != "metric" else "") +
                    ("GGX alpha %.2f, Poisson-noised measurement, " % args.alpha if args.alpha is not None else "") +
                    ("the reference's measured photon counts and wall points (exp_mannequin/transient.mat), " if args.measurement else "") +
                    ("window [%g, %g) m in %g mm bins, " % (args.lb, args.lb + T * args.res_m, 1e3 * args.res_m) if args.res_m is not None else "") +
                    ("NON-CONFOCAL pairs (row N side measurement, not the metric) " if args.non_confocal else "") +
                    ("SUBDIVIDED mesh x4^%d (side measurement, not the metric) " % args.subdivide if args.subdivide else "") +
                    ("RE-DECIMATED mesh (side measurement, not the metric) " if args.faces else "") +
                    ("forward-only " if args.forward_only else "forward+gradient ") +
                    "%dx%d confocal sources%s x %d bins, %s (F=%d, V=%d), num_sample=%d "
                    "(spt=%d), refine=10, sigma_bin=1, BVH rebuilt every step" % (
                        g, g, "" if world == 1 else (" split over %d ranks" % world if args.scaling == "strong" else " per rank"),
                        T, args.mesh, F, V, args.num_sample, spt),
        "sources_total": L_total, "faces": F, "bins": T, "spt": spt,
        "parallelism": "source sharding x%d (%s) + one all-reduce of the 3V gradient (%s)" % (
            world, "one grid per rank" if args.scaling == "weak" else args.partition + " partition", args.all_reduce),
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=64, help="sources per side (64 -> 64x64)")
    ap.add_argument("--bins", type=int, default=512)
    ap.add_argument("--num-sample", type=int, default=20000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): one grid x grid block of sources split over the ranks; weak: grid x grid per rank")
    ap.add_argument("--forward-only", action="store_true", help="BASELINE config 2 (parity-run size, not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the ~13 s CPU leg (the parity gate still runs, on a small block)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed steps before the W warm-up steps, for at least this long: the GPU idles while the CPU oracle "
                         "runs the parity gate and its clocks take tens of milliseconds to come back (0 = skip)")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="after the timed steps: a loop of at least this long, reported as sustained_ms_per_step (0 = skip; "
                         "6 s by default so that a 5-s activity sampler always sees the GPU busy)")
    ap.add_argument("--diagnostic-no-gate", action="store_true",
                    help="kernel-variant experiments (tools/ab_pmc.sh builds variants whose results are deliberately wrong): a "
                         "failing parity gate does not stop the run, but NO JSON line is printed -- per-kernel ms go to stderr, exit code 3")
    ap.add_argument("--subdivide", type=int, default=0,
                    help="side measurement (not the metric): 1->4 midpoint subdivision passes of the mesh (F x 4^n)")
    ap.add_argument("--mesh", choices=["bunny_5k", "mannequin"], default="bunny_5k",
                    help="side measurement: exp_mannequin/cnlos_mannequin_threshold.obj (1055 faces) instead of the metric's bunny")
    ap.add_argument("--faces", type=int, default=0,
                    help="side measurement: subdivide once, then vertex-cluster down to about this many faces")
    ap.add_argument("--as-rank", type=int, default=0,
                    help="with --of N: side measurement on ONE GPU of what rank K of an N-rank strong split does per step "
                         "(its block of the grid, global source offsets and 1/L scaling, no collective)")
    ap.add_argument("--of", type=int, default=1, help="see --as-rank")
    ap.add_argument("--partition", choices=["contiguous", "strided"], default="strided",
                    help="strong scaling: which sources a rank owns -- a contiguous block of the grid (the reference's own "
                         "batching, exp_bunny/test.py:66-67) or every N-th source (l = rank mod N: an even sample of the wall, "
                         "so the ranks finish together; default).  Results do not depend on it (RNG keys are global).")
    ap.add_argument("--all-reduce", choices=["torch", "rccl-direct"], default="torch",
                    help="N > 1: the gradient all-reduce through torch.distributed (default), or enqueued on the render stream by a "
                         "communicator of the process's own (dist.RcclDirect; exercised at world size 1 only so far)")
    ap.add_argument("--share-steps", type=int, default=10,
                    help="N = 1, metric workload: after the timed steps, time every rank's block of the 2-, 4- and 8-way strong "
                         "split for this many steps each and report them as `strong_share` (0 = skip)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="metric",
                    help="which BASELINE.json configuration to time (default: the metric's).  2 / 3: bunny 32x32x512 forward / "
                         "forward+gradient; 4: mannequin 64x64x1024 on the reference's own measurement (+-0.35 m, 2.4 mm bins "
                         "from 0); 4pairs: its non-confocal pairs; 5: GGX alpha 0.3, 64x64x1024, Poisson-noised data; exp: the "
                         "reference's experiment shape, 64x64 x 1200 bins of 1.2 mm from 0.  Everything but `metric` is a side "
                         "measurement with its own parity gate and roofline")
    ap.add_argument("--dropin-steps", type=int, default=20,
                    help="N = 1, metric workload: after the timed steps, time the same step through the reference-shaped host "
                         "entry (renderer.renderStreamedGradient on numpy arrays: what main.py / exp_bunny/test.py call) and "
                         "through the autograd pair (TransientFunction forward + backward) for this many steps each; reported "
                         "as dropin_ms_per_step / autograd_ms_per_step (0 = skip)")
    ap.add_argument("--non-confocal", action="store_true",
                    help="row N side measurement (not the metric): every source becomes a (laser, sensor) pair, "
                         "sensor = laser + (0.05, -0.03, 0)")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config]
    args.half, args.lb, args.res_m, args.alpha = 0.25, None, None, None
    args.measurement = args.poisson = False
    for k, val in preset.items():
        setattr(args, {"res": "res_m"}.get(k, k), val)
    return args


# ------------------------------------------------------------------------------------------------
# launcher (parent process: never initialises the GPU)
# ------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_devices():
    """Number of GPUs torch can see.  torch.cuda.device_count() does not initialise the GPU on this image."""
    import torch
    return torch.cuda.device_count()


def launch(argv, n_ranks, script=None, require_devices=True, timeout=None):
    """Start `n_ranks` ranks of `script` (default: this file) under torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1, and return the child's exit code.  The parent has not touched the GPU."""
    if require_devices:
        have = visible_devices()
        if have < n_ranks:
            sys.stderr.write("bench.py: --gpus %d requested but only %d device(s) are visible; refusing to measure "
                             "fewer GPUs than asked for\n" % (n_ranks, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL (see the image notes)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, timeout=timeout)


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
class GpuBackend:
    """Device, process-group backend and renderer of a real rank."""
    name = "nccl"                      # "nccl" IS RCCL on ROCm

    def __init__(self, local_rank):
        import torch
        self.torch = torch
        have = torch.cuda.device_count()
        if local_rank >= have:
            raise SystemExit("bench.py: rank needs device %d but only %d device(s) are visible" % (local_rank, have))
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.device)

    def make_renderer(self):
        from nlos_surface_optimization_amd import device as nd
        r = nd.TransientRenderer(self.device, seed=0)
        r.enable_timing(True)
        return r

    def sync(self):
        self.torch.cuda.synchronize()


def run_rank(args, backend):
    """The benchmark body of one rank.  `backend` supplies device / process-group backend / renderer (tests drive
    this same function with a gloo backend and a CPU stand-in renderer)."""
    import torch
    import torch.distributed as dist
    from nlos_surface_optimization_amd import dist as ndist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    dev = backend.device
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend.name, rank=rank, world_size=world)
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)           # every rank contributes 1: the group really has N members
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group came up with %d ranks (all-reduce of ones = %d), wanted %d"
                             % (dist.get_world_size(), rccl_ranks, args.gpus))

    d = np.load(os.path.join(ROOT, "tests", "golden", args.mesh + ".npz"))
    v_np = np.ascontiguousarray(d["v"], np.float32)
    f_np = np.ascontiguousarray(d["f"], np.int32)
    if args.subdivide > 0:
        from nlos_surface_optimization_amd import mesh_io
        v_np, f_np = mesh_io.subdivide(v_np, f_np, args.subdivide)
    if args.faces > 0:
        from nlos_surface_optimization_amd import mesh_io
        v_np, f_np = mesh_io.subdivide(v_np, f_np, 1)
        v_np, f_np = mesh_io.decimate_to(v_np, f_np, args.faces)
        v_np, f_np = np.ascontiguousarray(v_np, np.float32), np.ascontiguousarray(f_np, np.int32)
    F, V = f_np.shape[0], v_np.shape[0]
    T = args.bins
    if args.res_m is None:
        lb, ub, res = 0.625, 1.625, 1.0 / T      # exact in fp32 for T = 512 / 1024
    else:
        # a window given by its bin width: ub the way the reference's scripts form it (max_distance_bin * distance_resolution,
        # exp_bunny/rendering.py:258), falling back to the fp32 product where that rounds to one bin more
        from nlos_surface_optimization_amd import _lib as _nl
        lb, res = float(args.lb), float(args.res_m)
        ub = lb + T * res
        if _nl.num_bins(lb, ub, res) != T:
            ub = float(np.float32(lb) + np.float32(T) * np.float32(res))
        if _nl.num_bins(lb, ub, res) != T:
            raise SystemExit("bench.py: the window of --config %s does not give %d bins" % (args.config, T))
    spt = 1 + (args.num_sample - 1) // F

    g = args.grid
    if args.scaling == "weak":
        origin_np, normal_np = grid_sources(g, g * world, args.half)    # 64 x 64N grid, one 64x64 block per rank
    else:
        origin_np, normal_np = grid_sources(g, g, args.half)            # one 64x64 grid, split over the ranks
    meas = None
    if args.measurement:
        # cfg 4 on its real inputs: the reference's wall points and photon counts (tests/golden/mannequin_measurement.npz)
        meas = np.load(os.path.join(ROOT, "tests", "golden", "mannequin_measurement.npz"))
        origin_np = np.ascontiguousarray(meas["lighting"], np.float32)
        normal_np = np.ascontiguousarray(np.tile(np.array([0, 0, 1], np.float32), (origin_np.shape[0], 1)))
        if world != 1 or args.scaling == "weak":
            raise SystemExit("bench.py: --config 4 (the measured wall points) is a one-GPU side measurement")
    L_total = origin_np.shape[0]
    partition = args.partition if args.scaling == "strong" else "contiguous"    # (weak: one whole grid per rank)
    own, lo, stride = ndist.shard_slice(L_total, rank, world, partition)
    if args.of > 1:
        if world != 1 or not (0 <= args.as_rank < args.of):
            raise SystemExit("bench.py: --as-rank K --of N is a one-GPU side measurement (0 <= K < N, --gpus 1)")
        own, lo, stride = ndist.shard_slice(L_total, args.as_rank, args.of, partition)
    origin_np, normal_np = np.ascontiguousarray(origin_np[own]), np.ascontiguousarray(normal_np[own])   # this rank's sources
    L = origin_np.shape[0]
    keys = {"source_offset": lo, "source_stride": stride}       # global index of local source l = lo + l * stride

    r = backend.make_renderer()
    origin = torch.from_numpy(origin_np).to(dev)
    normal = torch.from_numpy(normal_np).to(dev)
    faces = torch.from_numpy(f_np).to(dev)
    verts = torch.from_numpy(v_np).to(dev)
    # synthetic measurement: transient of a slightly displaced copy of the mesh, weight == 1
    rs = np.random.RandomState(0)
    v_gt = torch.from_numpy((v_np + 0.002 * rs.standard_normal(v_np.shape)).astype(np.float32)).to(dev)
    akw = {} if args.alpha is None else {"alpha": float(args.alpha)}          # GGX branch (cfg 5)
    if meas is not None:
        data = torch.from_numpy(np.ascontiguousarray(meas["counts"], np.float64)[own]).to(dev)
    elif args.poisson:
        # exp_noise/noise/addNoiseExample.m:9 -- Poisson-noised clean transient + background, numpy default_rng(0)
        clean, _ = r.render_transient(origin, normal, verts, faces, args.num_sample, lb, ub, res, total_sources=L_total, **keys, **akw)
        clean = clean.cpu().numpy()
        rng = np.random.default_rng(0)
        cs = 2e4 / np.maximum(clean.sum(axis=1, keepdims=True), 1e-300)
        data = torch.from_numpy(np.ascontiguousarray(rng.poisson(cs * clean) / cs + rng.poisson(0.05, clean.shape) / cs)).to(dev)
    else:
        data, _ = r.render_transient(origin, normal, v_gt, faces, args.num_sample, lb, ub, res,
                                     total_sources=L_total, seed=1, **keys, **akw)
    weight = torch.ones_like(data)
    grad = torch.zeros((V, 3), dtype=torch.float64, device=dev)
    nc = {}
    if args.non_confocal:
        nc = {"sensor": (origin + torch.tensor([0.05, -0.03, 0.0], device=dev)).contiguous(), "sensor_normal": normal}

    # ---- parity gate: EVERY rank checks the first sources of ITS block of the very workload that is timed (confocal or
    # pairs, forward-only or with the gradient, whatever the mesh); the flags are MIN-all-reduced, so a result line
    # means every GPU passed.  On rank 0 at N = 1 the CPU leg doubles as the reported baseline. ----
    plain = (not (args.forward_only or args.non_confocal or args.subdivide or args.faces or args.of > 1) and args.mesh == "bunny_5k"
             and args.config == "metric")
    parity, cpu_base = None, None
    budget = 15.0 if (world == 1 and plain and not args.no_cpu_baseline) else 0.0
    data_np = data.cpu().numpy()
    wk = {"forward_only": bool(args.forward_only)}
    if args.alpha is not None:
        wk["ggx_alpha"] = float(args.alpha)
    if args.non_confocal:
        wk.update(sensor=nc["sensor"].cpu().numpy(), sensor_normal=nc["sensor_normal"].cpu().numpy())
    n_ref, t_ref, g_ref, cpu_base = cpu_reference(v_np, f_np, origin_np, normal_np, lb, ub, res,
                                                  args.num_sample, data_np, budget, source_offset=lo, share=world,
                                                  source_stride=stride, **wk)

    def render_block(n):
        ncb = {k: t[:n].contiguous() for k, t in nc.items()}
        if args.forward_only:
            tb, _ = r.render_transient(origin[:n].contiguous(), normal[:n].contiguous(), verts, faces, args.num_sample,
                                       lb, ub, res, total_sources=n, **keys, **ncb, **akw)
            return tb.cpu().numpy(), None
        gb = torch.zeros((V, 3), dtype=torch.float64, device=dev)
        tb, gb, _ = r.render_gradient(origin[:n].contiguous(), normal[:n].contiguous(), verts, faces,
                                      args.num_sample, lb, ub, res, data=data[:n].contiguous(),
                                      weight=weight[:n].contiguous(), refine_scale=10, sigma_bin=1,
                                      testing_flag=1, loss_flag=0, gradient=gb, total_sources=n, **keys, **ncb, **akw)
        return tb.cpu().numpy(), gb.cpu().numpy()

    parity = parity_gate(render_block, n_ref, t_ref, g_ref)
    parity["rank"] = rank
    gate_ok = parity["pass"]
    if world > 1:
        flag = torch.tensor([1.0 if gate_ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity["all_ranks_pass"] = bool(flag.item() > 0.5)
        parity["ranks_gated"] = world
        if not gate_ok:
            sys.stderr.write("bench.py: rank %d: PARITY GATE FAILED: %s\n" % (rank, json.dumps(parity)))
        gate_ok = parity["all_ranks_pass"]
    diagnostic = bool(args.diagnostic_no_gate and not gate_ok)
    if not gate_ok and not diagnostic:
        if rank == 0:
            sys.stderr.write("bench.py: PARITY GATE FAILED, nothing timed: %s\n" % json.dumps(parity))
        if world > 1:
            dist.destroy_process_group()
        return 1

    direct = None
    if world > 1 and args.all_reduce == "rccl-direct":
        direct = ndist.RcclDirect(rank, world, dev)

    def step():
        if args.forward_only:
            r.render_transient(origin, normal, verts, faces, args.num_sample, lb, ub, res,
                               total_sources=L_total, **keys, **nc, **akw)
        else:
            r.render_gradient(origin, normal, verts, faces, args.num_sample, lb, ub, res, data=data,
                              weight=weight, refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0,
                              gradient=grad, zero_gradient=True, total_sources=L_total, **keys, **nc, **akw)
            if direct is not None:
                direct.all_reduce_sum_(grad)
            elif world > 1:
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)

    def timed(n_steps):
        if world > 1:
            dist.barrier()
        backend.sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        if world > 1:
            dist.barrier()
        backend.sync()
        return time.perf_counter() - t0

    if args.prewarm_seconds > 0:
        # step() holds a collective at N > 1, so every rank must run the SAME number of batches: the decision to go
        # on is itself all-reduced (MAX of the elapsed time -- the loop ends for everybody once the slowest rank's
        # clock says so); ranks that decided on their own clocks could leave the loop one batch apart and pair a
        # leftover gradient all-reduce with the others' barrier
        t_pw = time.perf_counter()
        while True:
            for _ in range(8):
                step()
            backend.sync()
            el = time.perf_counter() - t_pw
            if world > 1:
                te_pw = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(te_pw, op=dist.ReduceOp.MAX)
                el = float(te_pw.item())
            if el >= args.prewarm_seconds:
                break
    for _ in range(args.warmup):
        step()
    backend.sync()
    if hasattr(r, "timing_reset"):
        r.timing_reset()
    elapsed_local = timed(args.steps)
    per_rank_ms = [1e3 * elapsed_local / args.steps]
    elapsed = elapsed_local
    if world > 1:
        te = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(allt, te)
        per_rank_ms = [1e3 * float(x.item()) / args.steps for x in allt]
        elapsed = max(float(x.item()) for x in allt)                 # MAX over ranks
    # per-kernel durations: HIP events recorded on the launch stream inside the timed region
    kt = np.array(r.timing_mean_ms()[0]) if hasattr(r, "timing_mean_ms") else np.zeros(4)
    path = r.last_path(count=True) if hasattr(r, "last_path") else None
    # what one step does with its L * F * spt surface samples (pass 1's own counters, nlos_path_info): faces a wall point
    # sees from behind are dropped before sampling, so fewer rays are traced than samples are counted in `value`
    rays = [(path or {}).get("rays_traced", -1), (path or {}).get("samples_accepted", -1)]
    if world > 1:
        tr_ = torch.tensor([float(x) for x in rays] + [1.0 if rays[0] >= 0 else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tr_, op=dist.ReduceOp.SUM)
        rays = [int(tr_[0].item()), int(tr_[1].item())] if int(tr_[2].item()) == world else [-1, -1]

    # sustained figure: the K timed steps are a short burst at boost clock; a loop of >= sustain-seconds shows
    # what a long optimisation sees (DVFS give-back, MI355X_MICROARCH.md)
    sustained_ms, sustained_steps = None, 0
    if args.sustain_seconds > 0:
        per = max(elapsed / max(args.steps, 1), 1e-5)
        sustained_steps = int(min(max(args.sustain_seconds / per, args.steps), 200000))
        t_s = timed(sustained_steps)
        if world > 1:
            ts = torch.tensor([t_s], dtype=torch.float64, device=dev)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            t_s = float(ts.item())
        sustained_ms = 1e3 * t_s / sustained_steps

    # ---- the paths a user of the reference's API calls (round 6): the numpy drop-in (host arrays in, host arrays out, every
    # call: transient_rendering_cython/main.py:114-115, exp_bunny/rendering.py:252-269) and the autograd pair ----
    api_paths = None
    if world == 1 and plain and args.dropin_steps > 0 and not diagnostic and backend.name == "nccl":     # (a real GPU rank)
        api_paths = time_api_paths(args, r, backend, origin_np, normal_np, v_np, f_np, data_np, lb, ub, res, T,
                                   origin, normal, verts, faces, data, weight, grad)

    # ---- strong scaling, the part one GPU can time: every rank's block of the 2-, 4- and 8-way split of this very grid
    # (global source offsets, 1/L_total scaling, replicated scene build -- what a rank does per step, without the
    # all-reduce) ----
    share = None
    if world == 1 and plain and args.share_steps > 0 and not diagnostic:
        share = {}
        for part in ndist.PARTITIONS:
            share[part] = {}
            for n_split in (2, 4, 8):
                per, per_k = [], []
                for k in range(n_split):
                    ksl, klo, kstride = ndist.shard_slice(L_total, k, n_split, part)
                    o_k, n_k = origin[ksl].contiguous(), normal[ksl].contiguous()
                    d_k, w_k = data[ksl].contiguous(), weight[ksl].contiguous()

                    def step_k():
                        r.render_gradient(o_k, n_k, verts, faces, args.num_sample, lb, ub, res, data=d_k, weight=w_k,
                                          refine_scale=10, sigma_bin=1, testing_flag=1, loss_flag=0, gradient=grad,
                                          zero_gradient=True, source_offset=klo, source_stride=kstride, total_sources=L_total)
                    for _ in range(3):
                        step_k()
                    backend.sync()
                    if hasattr(r, "timing_reset"):
                        r.timing_reset()
                    t0 = time.perf_counter()
                    for _ in range(args.share_steps):
                        step_k()
                    backend.sync()
                    per.append(1e3 * (time.perf_counter() - t0) / args.share_steps)
                    if hasattr(r, "timing_mean_ms"):
                        per_k.append([round(float(x), 5) for x in r.timing_mean_ms()[0]])
                share[part][str(n_split)] = {"per_rank_ms": per, "max_ms": max(per), "spread": (max(per) - min(per)) / min(per),
                                             "efficiency_without_collective": (1e3 * elapsed / args.steps) / (n_split * max(per)),
                                             # HIP-event means per rank block: [bvh_build, k_forward, k_residual, k_gradient] ms
                                             "per_rank_kernel_ms": per_k}

    if diagnostic:
        if rank == 0:
            sys.stderr.write("bench.py: DIAGNOSTIC run (parity gate failed, no result line): ms/step %.3f kernel_ms %s\n" % (
                1e3 * elapsed / args.steps, json.dumps(dict(zip(["bvh_build", "k_forward", "k_residual", "k_gradient"], [float(x) for x in kt])))))
        if world > 1:
            dist.destroy_process_group()
        return 3
    if rank == 0:
        samples_per_step = (L if args.of > 1 else L_total) * F * spt          # all ranks (--as-rank: this block only)
        ms = 1e3 * elapsed / args.steps
        cfg = workload_config(args, g, T, F, V, spt, L_total, world)
        if path is not None:
            cfg["path"] = path
        out = {
            # schema 5 (round 5): + rays_traced_per_step / accepted_per_step / traced_rays_per_s, roofline.step_model,
            # roofline.valu_issue_frac, roofline.pass2, strong_share[partition][N].per_rank_kernel_ms.  Schema 4 (round 4):
            # strong_share nested as [partition][N] (default partition strided), as_rank.sources a count.
            # schema 6 (round 6): + dropin_ms_per_step / autograd_ms_per_step (+ `dropin`, `autograd`), roofline.frac_on_traced_rays,
            # roofline.issue.in_situ (measured marginal issue costs), --config side lines, sustained loop >= 6 s by default.
            "schema": 6,
            "metric": METRIC,
            "value": samples_per_step * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "per_rank_ms_per_step": per_rank_ms,
            "sustained_ms_per_step": sustained_ms,
            "sustained_steps": sustained_steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32 per-sample math, f64 accumulation",
            "data": "synthetic",
            "config": cfg,
        }
        if parity is not None:
            out["parity"] = parity
        if rays[0] >= 0:
            # `value` counts SURFACE SAMPLES (the metric's unit: sources x faces x strata, the reference's loop count); these say
            # how many of them pass 1 traces a ray for and how many end up in a bin
            out["rays_traced_per_step"] = rays[0]
            out["accepted_per_step"] = rays[1]
            out["traced_rays_per_s"] = rays[0] * args.steps / elapsed
            out["samples_note"] = ("of the %d surface samples of a step, %d (%.1f %%) are traced: the faces a wall point sees from "
                                   "behind contribute exactly 0 under the clamped form factor and are dropped per (source, face) "
                                   "before sampling; %d (%.1f %%) are found visible and binned" % (
                                       samples_per_step, rays[0], 100.0 * rays[0] / samples_per_step, rays[1],
                                       100.0 * rays[1] / samples_per_step))
        if args.of > 1:
            out["as_rank"] = {"rank": args.as_rank, "of": args.of, "partition": partition, "first_source": lo, "source_stride": stride, "sources": L,
                              "note": "one GPU timing what rank %d of a %d-rank strong split does per step (no collective); "
                                      "`value` counts this block's samples only" % (args.as_rank, args.of)}
        if api_paths is not None:
            out.update(api_paths)
        if share is not None:
            share["note"] = ("measured on ONE GPU: every rank's sources of the N-way strong split of this grid, for both partitions "
                             "(contiguous blocks / every N-th source), %d steps each, "
                             "scene build replicated, NO all-reduce (a 3V-double RCCL all-reduce per step comes on top); "
                             "efficiency_without_collective = ms_per_step / (N x slowest block)" % args.share_steps)
            out["strong_share"] = share
        # roofline of the dominant kernel, measured live (HIP events, rank 0, this rank's launches)
        names = ["bvh_build", "k_forward", "k_residual", "k_gradient"]
        dom = int(np.argmax(kt))
        per_sample = {"k_forward": 16.0 + 36.0 / spt, "k_gradient": 184.0 + 36.0 / spt}.get(names[dom], 0.0)
        local_samples = L * F * spt
        if kt[dom] > 0 and per_sample > 0:
            achieved = per_sample * local_samples / (kt[dom] * 1e-3) / 1e9
            traffic, issue, here_stamp, pmc_stamp = counters_of_this_build(load_pmc_summary(), names[dom], plain and world == 1, L, F)
            roof = {
                # the contract's figure: ALGORITHMIC bytes of SURVEY 8(d) per launch / measured kernel time, against
                # HBM peak.  It is a model of the reference's traffic, not this kernel's: rows, grid and accumulators
                # live in LDS here (see `traffic` / `hbm_measured_GBps` for the real bytes).
                # `bound` names the roofline `achieved` / `peak` / `frac` are priced against (the contract's: HBM);
                # `binds_in_practice` what the counters say limits the kernel (see `issue`)
                "bound": "hbm", "binds_in_practice": "valu-issue",
                "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_sample": per_sample,
                "kernel_ms": {n: float(x) for n, x in zip(names, kt)},
            }
            if rays[0] >= 0 and world == 1:
                # the same contract figure priced on the rays pass 1 really traces (the samples of faces seen from behind are
                # dropped before sampling: `frac` counts them, this does not)
                roof["frac_on_traced_rays"] = per_sample * rays[0] / (kt[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if traffic:
                roof["hbm_measured_GBps"] = traffic / (kt[dom] * 1e-3) / 1e9
                roof["hbm_measured_frac"] = roof["hbm_measured_GBps"] / HBM_PEAK_GBS
            roof["build"] = here_stamp
            # the same model over the whole step (16 + 36/spt B per sample forward, 184 + 36/spt B gradient): the
            # reference's traffic for this work would exceed the HBM peak at this step time -- the histogram
            # read-modify-writes, the residual window and the nine gradient accumulations it counts live in LDS and
            # registers here -- so the step is not priced against HBM at all (see valu_issue_frac)
            step_bytes = (200.0 + 72.0 / spt) * local_samples
            roof["step_model"] = {"algorithmic_bytes_per_step": step_bytes, "GBps": step_bytes / (ms * 1e-3) / 1e9,
                                  "frac_of_hbm": step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "note": "exceeds peak where > 1: a model of the reference's traffic, not of this path's"}
            # the ceiling that binds the dominant kernel, first-class: VALU issue slots used / available (central estimate of
            # the weighted issue model, profiles/pmc_summary.json of THIS build; None when the committed counters are stale)
            # (round 6: priced with the instruction costs measured inside this kernel where the profile carries them --
            # `issue.valu_busy_in_situ`; `issue.valu_busy` keeps the back-to-back costs of rounds 3-5 for comparison)
            vb = (issue or {}).get("valu_busy_in_situ") or (issue or {}).get("valu_busy")
            roof["valu_issue_frac"] = vb.get("central") if isinstance(vb, dict) else None
            pmc_all = load_pmc_summary() or {}
            g2 = pmc_all.get("k_gradient") if traffic else None
            if g2 and kt[3] > 0:
                roof["pass2"] = {"kernel": "k_gradient", "ms": float(kt[3]), "hbm_bytes_per_launch": g2.get("hbm_bytes_per_launch"),
                                 "hbm_GBps": (g2.get("hbm_bytes_per_launch") or 0.0) / (kt[3] * 1e-3) / 1e9,
                                 "valu_busy": (g2.get("valu_busy") or {}).get("central"),
                                 "active_lane_frac": g2.get("active_lane_frac")}
            if issue:
                # the ceiling that binds: VALU issue slots (from the committed PMC passes of this same command on this
                # same build), or {"stale": ...} when the committed counters belong to other kernel sources
                roof["issue"] = issue
            if traffic:
                roof["profile_stamp"] = pmc_stamp
            out["roofline"] = roof
        try:      # every behaviour-changing environment switch as this process read it (nlos_env_report)
            from nlos_surface_optimization_amd import _lib as _nl2
            out["env"] = _nl2.env_report()
        except Exception:
            pass
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None, backend_factory=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # parent: start the ranks as a child process group; nothing here has touched the GPU
        return launch(argv, args.gpus)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = (backend_factory or GpuBackend)(local_rank)
    return run_rank(args, backend)


if __name__ == "__main__":
    sys.exit(main())
