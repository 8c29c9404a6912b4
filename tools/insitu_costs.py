#!/usr/bin/env python3
"""In-situ issue cost of a VALU instruction in k_forward_grid (round 6) from the pad A/B logs (tools/pad_test.sh):

    python tools/insitu_costs.py profiles/r06_pad_test_a.log profiles/r06_pad_stamps_a.log profiles/r06_pad_test_b.log [clock_ghz]

Every variant adds N copies of one instruction (+ 2 v_mov) at one site of the kernel; the stamped build gives the number of
times the site is executed per launch (wave level).  cost = (k_forward(variant) - k_forward(baseline of the same run)) x SIMDs x
clock / (executions x instructions), in shader cycles per wave64 instruction per SIMD -- what the launch pays for one more
instruction THERE, which is also what removing one gives back.  Back-to-back costs (tools/issue_rate.hip, >= 2 waves per SIMD)
beside them: 2.3 full rate, 4.1 "half rate".  Output: JSON."""
import json
import re
import sys

SIMDS = 1024
OPS = {0: "v_add_u32", 1: "v_lshlrev_b32", 2: "v_cmp_lt_u32", 3: "v_min_u32", 4: "v_mul_lo_u32", 5: "v_fma_f32"}
BACK_TO_BACK = {"v_add_u32": 2.3, "v_fma_f32": 2.3, "v_lshlrev_b32": 4.1, "v_cmp_lt_u32": 4.1, "v_min_u32": 4.1, "v_mul_lo_u32": 4.1}


def parse_runs(path):
    rows = {}
    for line in open(path):
        m = re.match(r"variant (\d+) \[(.*?)\] round \d+: .*?'k_forward': ([0-9.]+)", line)
        if m:
            rows.setdefault(m.group(2), []).append(float(m.group(3)))
    return {k: sum(v) / len(v) for k, v in rows.items()}


def parse_trips(path):
    t = {}
    for line in open(path):
        m = re.search(r"\[fwd trips\] items (\d+) walk_trips (\d+) push_slots (\d+) exact_rounds (\d+) count_face_waves (\d+)", line)
        if m:
            t = {"GEN": int(m.group(1)), "WALK": int(m.group(2)), "COUNT": int(m.group(5)), "push_slots": int(m.group(3)), "exact_rounds": int(m.group(4))}
    return t


def main():
    clock = float(sys.argv[4]) if len(sys.argv) > 4 else 2.29
    trips = parse_trips(sys.argv[2])
    out = {"clock_ghz": clock, "simds": SIMDS, "executions_per_launch": trips, "sites": {
        "WALK": "body of the lockstep cell-list walk (once per trip of a wave)", "GEN": "sample generation, once per 64-ray item",
        "COUNT": "counting pass of the build, once per 64-triangle wave iteration"}, "costs": []}
    for path in (sys.argv[1], sys.argv[3]):
        runs = parse_runs(path)
        base = runs[""]
        for flags, ms in sorted(runs.items()):
            m = re.search(r"NLOS_DIAG_PAD_(WALK|GEN|COUNT)=(\d+)", flags)
            if not m:
                continue
            site, n = m.group(1), int(m.group(2))
            op = OPS[int(re.search(r"PAD_OP=(\d+)", flags).group(1)) if "PAD_OP" in flags else 0]
            cycles = (ms - base) * 1e-3 * SIMDS * clock * 1e9 / trips[site]          # per execution of the site, per SIMD
            per_inst = (cycles - 2 * 2.0) / n                                       # the two v_mov at about the add's cost
            out["costs"].append({"site": site, "op": op, "n": n, "baseline_ms": round(base, 4), "variant_ms": round(ms, 4),
                                 "cycles_per_instruction": round(per_inst, 2), "back_to_back": BACK_TO_BACK[op],
                                 "ratio_to_back_to_back": round(per_inst / BACK_TO_BACK[op], 2), "log": path})
    full = [c["cycles_per_instruction"] for c in out["costs"] if c["back_to_back"] == 2.3]
    half = [c["cycles_per_instruction"] for c in out["costs"] if c["back_to_back"] == 4.1]
    out["summary"] = {"full_rate_ops_in_situ": [min(full), max(full)], "half_rate_ops_in_situ": [min(half), max(half)],
                      "reading": "an instruction in the trace costs the launch 0.8 - 1.0 of a full-rate issue slot (2.3 cycles) when it is a full-rate "
                                 "one and only 0.5 - 0.7 of its back-to-back cost (4.1) when it is a 'half-rate' one: half-rate operations of one "
                                 "wave overlap with other waves' full-rate ones; in the build phases both cost about a third less (the issue "
                                 "slots are not saturated there)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
