#!/usr/bin/env python3
"""profiles/pmc_summary.json (what bench.py loads into `roofline.traffic` / `roofline.issue`) from a profile
directory written by tools/profile_round.sh:  python tools/round_summary.py gpurun_out/profile_<tag> <L> <F>

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB units; FETCH doubled: the gfx950 correction of
MI355X_MICROARCH.md, calibrated on wide streaming reads -- an upper bound for this gathering kernel).

issue block (round 3): a WEIGHTED model instead of round 2's uniform "4 cycles per VALU instruction".  Issue costs per
instruction class were measured on this chip (tools/issue_rate.hip -> profiles/r03_issue_rates.json: 2.3 cycles per wave64
instruction per SIMD for v_add/mul/fma_f32, v_add_u32, v_and/xor, v_mov; 4.1 for compares, selects, min/max, shifts,
integer multiplies, conversions, fp64, lane ops; 8.1 for v_sqrt/rcp/rsq_f32 -- the guide's "2 cycles" is right for the
first class, round 2's "4" for the second), the class counts come from the SQ_INSTS_VALU_* counters of the same profile
(tools/issue_model.py explains the bounds: INT32 and the unclassified rest mix both rates)."""
import json
import os
import subprocess
import sys

root, L, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from nlos_surface_optimization_amd import _lib  # noqa: E402  (source_stamp only: no GPU call)
pmc = json.load(open(os.path.join(root, "pmc_summary_all.json")))
trace = json.load(open(os.path.join(root, "kernel_trace_summary.json")))


def pick(d, prefix):
    c = [k for k in d if k.startswith(prefix) and not k.rstrip(">").endswith(", 1")]
    c.sort(key=lambda k: -d[k].get("SQ_INSTS_VALU", d[k].get("steady_mean_ms", 0)))
    return (c[0], d[c[0]]) if c else (None, None)


fw_name, fw = pick(pmc, "k_forward_grid")
gr_name, gr = pick(pmc, "k_gradient")
_, fw_t = pick(trace, "k_forward_grid")
_, gr_t = pick(trace, "k_gradient")
# which build these counters belong to: bench.py matches `source_sha256_16` and reports the set as stale otherwise.
# `git` is the HEAD of the build container at profile time when tools/profile_round.sh was handed one (NLOS_GIT_HEAD;
# the GPU box has no .git), `clock_ghz` the shader clock of the issue model (tools/clock_probe: s_memtime ticks of the
# forward kernel / its HIP-event time) if the profile directory holds one.
stamp = _lib.source_stamp()
stamp["git"] = os.environ.get("NLOS_GIT_HEAD") or None
clock = None
if os.path.exists(os.path.join(root, "clock.json")):
    clock = json.load(open(os.path.join(root, "clock.json")))
out = {"kernel": "k_forward", "L": L, "F": F, "stamp": stamp,
       "hbm_bytes_per_launch": 1024.0 * (2 * fw["FETCH_SIZE"] + fw["WRITE_SIZE"]),
       "fetch_size_kb": fw["FETCH_SIZE"], "write_size_kb": fw["WRITE_SIZE"],
       "clock": clock,
       "note": "steady-state means of the largest launches, rocprofv3 --pmc in separate passes (tools/profile_round.sh); "
               "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md"}


def model(name, ms):
    """tools/issue_model.py on this profile's counters; the ISA text for the static split is optional"""
    isa = os.environ.get("NLOS_ISA", os.path.join(root, "forward_grid.s"))
    cmd = [sys.executable, os.path.join(here, "issue_model.py"), os.path.join(here, "..", "profiles", "r03_issue_rates.json"),
           os.path.join(root, "pmc_summary_all.json"), name, "%.6f" % ms] + ([isa] if os.path.exists(isa) else [])
    env = dict(os.environ)
    try:      # the CU count of the profiled device (an MI355X partition exposes fewer than 256)
        import torch
        if torch.cuda.is_available():
            env.setdefault("NLOS_COMPUTE_UNITS", str(torch.cuda.get_device_properties(0).multi_processor_count))
    except Exception:
        pass
    if clock and clock.get("clock_ghz"):
        env["NLOS_CLOCK_GHZ"] = "%.5f" % clock["clock_ghz"]
    return json.loads(subprocess.check_output(cmd, env=env))


dur_ms = fw_t["steady_mean_ms"] if fw_t else None
if dur_ms:
    out["kernel_ms_under_trace"] = dur_ms
    m = model(fw_name, dur_ms)
    out["issue"] = {
        "valu_busy": m["valu_busy"],
        "mean_cycles_per_valu_inst": m["mean_cycles_per_valu_inst"],
        "salu_busy": m["salu_busy"],
        "active_lane_frac": fw["SQ_THREAD_CYCLES_VALU"] / (64.0 * fw["SQ_ACTIVE_INST_VALU"]),
        "lds_conflict_frac": fw["SQ_LDS_BANK_CONFLICT"] / fw["SQ_LDS_IDX_ACTIVE"],
        "wave_time_shares": {"parked_at_waitcnt_or_barrier": fw["SQ_WAIT_ANY"] / fw["SQ_WAVE_CYCLES"],
                             "waiting_to_issue": fw.get("SQ_WAIT_INST_ANY", 0.0) / fw["SQ_WAVE_CYCLES"],
                             "issuing": fw["SQ_ACTIVE_INST_ANY"] / fw["SQ_WAVE_CYCLES"]},
        "valu_insts_per_launch": fw["SQ_INSTS_VALU"], "salu_insts_per_launch": fw["SQ_INSTS_SALU"],
        "lds_insts_per_launch": fw["SQ_INSTS_LDS"], "vmem_insts_per_launch": fw["SQ_INSTS_VMEM"],
        # round 6: the busy figure re-priced with the instruction costs measured INSIDE this kernel (tools/pad_test.sh), and the
        # marginal cost itself -- what removing one instruction gives back
        "valu_busy_in_situ": m.get("valu_busy_in_situ"), "in_situ_costs": m.get("in_situ_costs"),
        "classes": m["classes"], "mixed_full_rate_fraction": m["mixed_full_rate_fraction"],
        "basis": m["basis"] + "; kernel " + fw_name}
if gr:
    out["k_gradient"] = {"hbm_bytes_per_launch": 1024.0 * (2 * gr["FETCH_SIZE"] + gr["WRITE_SIZE"]),
                         "valu_insts_per_launch": gr["SQ_INSTS_VALU"],
                         "active_lane_frac": gr["SQ_THREAD_CYCLES_VALU"] / (64.0 * gr["SQ_ACTIVE_INST_VALU"])}
    if gr_t:
        mg = model(gr_name, gr_t["steady_mean_ms"])
        out["k_gradient"]["valu_busy"] = mg["valu_busy"]
        out["k_gradient"]["kernel_ms_under_trace"] = gr_t["steady_mean_ms"]
print(json.dumps(out, indent=1))
