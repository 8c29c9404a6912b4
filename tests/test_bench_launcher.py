"""bench.py's multi-process path on CPU: `python bench.py --gpus N` must start N ranks itself (never measure one GPU
and label it N), prove the group's membership, split the sources, and print one JSON line from rank 0.  The ranks
run bench.py's own body on a gloo group with the CPU oracle as stand-in renderer (tests/_bench_standin.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

STANDIN = os.path.join(ROOT, "tests", "_bench_standin.py")
SMALL = ["--grid", "4", "--num-sample", "5000", "--steps", "1", "--warmup", "0", "--sustain-seconds", "0", "--prewarm-seconds", "0",
         "--share-steps", "0"]


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_launcher_starts_two_ranks_and_reports_them(scaling):
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch(%r, 2, script=%r, require_devices=False))"
            % (ROOT, ["--gpus", "2", "--scaling", scaling] + SMALL, STANDIN))
    p = _run([sys.executable, "-c", code])
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["scaling"] == scaling
    assert out["config"]["sources_total"] == (16 if scaling == "strong" else 32)
    assert out["value"] > 0 and out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-9
    assert out["parity"]["pass"] and out["parity"]["rows"] == (8 if scaling == "strong" else 16)
    assert out["parity"]["ranks_gated"] == 2 and out["parity"]["all_ranks_pass"]      # every rank gates its own block
    assert "cpu_baseline" not in out                    # N > 1: no CPU leg


def test_prewarm_runs_the_same_number_of_collectives_on_every_rank():
    """the prewarm loop is bounded by wall time and every step holds an all-reduce -- the decision to
    leave the loop must be collective, or two ranks that straddle the threshold hang.  Two gloo ranks, prewarm on."""
    small = [a for a in SMALL]
    small[small.index("--prewarm-seconds") + 1] = "0.3"
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch(%r, 2, script=%r, require_devices=False, timeout=500))"
            % (ROOT, ["--gpus", "2"] + small, STANDIN))
    p = _run([sys.executable, "-c", code])
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["parity"]["all_ranks_pass"]


def test_one_ranks_share_and_the_share_table():
    p = _run([sys.executable, STANDIN, "--gpus", "1", "--no-cpu-baseline", "--as-rank", "3", "--of", "4"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    ar = out["as_rank"]      # default partition: strided -- sources 3, 7, 11, 15 of the 16
    assert (ar["partition"], ar["first_source"], ar["source_stride"], ar["sources"]) == ("strided", 3, 4, 4)
    assert out["parity"]["rows"] == 4 and out["parity"]["pass"]
    assert "ONE RANK'S SHARE" in out["config"]["workload"] and "strong_share" not in out
    p = _run([sys.executable, STANDIN, "--gpus", "1", "--no-cpu-baseline", "--as-rank", "3", "--of", "4", "--partition", "contiguous"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    ar = _json_line(p.stdout)["as_rank"]
    assert (ar["partition"], ar["first_source"], ar["source_stride"], ar["sources"]) == ("contiguous", 12, 1, 4)
    small = [a for a in SMALL]
    small[small.index("--share-steps") + 1] = "1"
    p = _run([sys.executable, STANDIN, "--gpus", "1", "--no-cpu-baseline"] + small)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    for part in ("contiguous", "strided"):
        sh = out["strong_share"][part]
        assert [len(sh[k]["per_rank_ms"]) for k in ("2", "4", "8")] == [2, 4, 8] and sh["8"]["max_ms"] > 0


def test_side_workloads_are_gated_too():
    for extra in (["--forward-only"], ["--mesh", "mannequin"]):
        p = _run([sys.executable, STANDIN, "--gpus", "1", "--no-cpu-baseline"] + extra + SMALL)
        assert p.returncode == 0, p.stderr[-2000:]
        out = _json_line(p.stdout)
        assert out["parity"]["pass"] and out["parity"]["rows"] >= 4
        assert ("gradient_rel_l2" in out["parity"]) == (extra != ["--forward-only"])


def test_single_rank_line_has_parity_and_no_launcher():
    p = _run([sys.executable, STANDIN, "--gpus", "1", "--no-cpu-baseline"] + SMALL)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["parity"]["pass"] and out["parity"]["rows"] == 16
    assert out["metric"].startswith("surface samples/sec fwd+grad")


def test_world_size_mismatch_and_missing_devices_are_errors():
    # a rank started with WORLD_SIZE != --gpus must not fall back to a one-GPU measurement
    p = _run([sys.executable, STANDIN, "--gpus", "2"] + SMALL, env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # the real launcher refuses when fewer devices are visible than ranks asked for (this container has no GPU)
    import torch
    if torch.cuda.device_count() < 2:
        p = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL)
        assert p.returncode != 0 and "device" in p.stderr
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_parity_gate_failure_blocks_the_timing(tmp_path):
    """A renderer that disagrees with the oracle: no JSON line, exit code 1."""
    bad = tmp_path / "bad_standin.py"
    bad.write_text(
        "import sys\nsys.path.insert(0, %r)\nimport _bench_standin as s\nimport bench\n"
        "class Bad(s.OracleStandIn):\n"
        "    def render_gradient(self, *a, **k):\n"
        "        t, g, p = super().render_gradient(*a, **k)\n"
        "        return t * 1.001, g, p\n"
        "class B(s.CpuBackend):\n"
        "    def make_renderer(self):\n        return Bad(seed=0)\n"
        "sys.exit(bench.main(backend_factory=B))\n" % os.path.join(ROOT, "tests"))
    p = _run([sys.executable, str(bad), "--gpus", "1", "--no-cpu-baseline"] + SMALL)
    assert p.returncode == 1 and "PARITY GATE FAILED" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
