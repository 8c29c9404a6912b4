#!/usr/bin/env python3
"""profiles/pmc_summary.json (what bench.py loads into `roofline.traffic` / `roofline.issue`) from a profile
directory written by tools/profile_round.sh:  python tools/round_summary.py gpurun_out/profile_<tag> <L> <F>

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB units; FETCH doubled: the gfx950 correction of
MI355X_MICROARCH.md, calibrated on wide streaming reads -- an upper bound for this gathering kernel).
issue block: valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch duration x 2.4 GHz)  (a wave64 VALU
instruction occupies its 16-lane SIMD for 4 cycles; 2.4 GHz = peak engine clock, so this is a lower bound),
active_lane_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), lds_conflict_frac = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE."""
import json
import os
import sys

root, L, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pmc = json.load(open(os.path.join(root, "pmc_summary_all.json")))
trace = json.load(open(os.path.join(root, "kernel_trace_summary.json")))


def pick(d, prefix):
    c = [k for k in d if k.startswith(prefix) and not k.rstrip(">").endswith(", 1")]
    c.sort(key=lambda k: -d[k].get("SQ_INSTS_VALU", d[k].get("steady_mean_ms", 0)))
    return d[c[0]] if c else None


fw, gr = pick(pmc, "k_forward_grid"), pick(pmc, "k_gradient")
fw_t = pick(trace, "k_forward_grid")
out = {"kernel": "k_forward", "L": L, "F": F,
       "hbm_bytes_per_launch": 1024.0 * (2 * fw["FETCH_SIZE"] + fw["WRITE_SIZE"]),
       "fetch_size_kb": fw["FETCH_SIZE"], "write_size_kb": fw["WRITE_SIZE"],
       "note": "steady-state means of the largest launches, rocprofv3 --pmc in separate passes (tools/profile_round.sh); "
               "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md"}
dur_ms = fw_t["steady_mean_ms"] if fw_t else None
if dur_ms:
    out["kernel_ms_under_trace"] = dur_ms
    out["issue"] = {
        "valu_busy": fw["SQ_INSTS_VALU"] * 4.0 / (1024.0 * dur_ms * 1e-3 * 2.4e9),
        "active_lane_frac": fw["SQ_THREAD_CYCLES_VALU"] / (64.0 * fw["SQ_ACTIVE_INST_VALU"]),
        "lds_conflict_frac": fw["SQ_LDS_BANK_CONFLICT"] / fw["SQ_LDS_IDX_ACTIVE"],
        "valu_insts_per_launch": fw["SQ_INSTS_VALU"], "salu_insts_per_launch": fw["SQ_INSTS_SALU"],
        "lds_insts_per_launch": fw["SQ_INSTS_LDS"], "vmem_insts_per_launch": fw["SQ_INSTS_VMEM"],
        "basis": "SQ_* counters of k_forward_grid<0, 0, false, 0>, 1024 SIMDs, 4 cycles per wave64 VALU instruction, 2.4 GHz"}
if gr:
    out["k_gradient"] = {"hbm_bytes_per_launch": 1024.0 * (2 * gr["FETCH_SIZE"] + gr["WRITE_SIZE"]),
                         "valu_insts_per_launch": gr["SQ_INSTS_VALU"],
                         "active_lane_frac": gr["SQ_THREAD_CYCLES_VALU"] / (64.0 * gr["SQ_ACTIVE_INST_VALU"])}
print(json.dumps(out, indent=1))
