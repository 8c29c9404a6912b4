#!/usr/bin/env python3
"""Weighted VALU-issue model of a kernel from measured per-class issue costs and per-class instruction counters.

    python tools/issue_model.py profiles/r03_issue_rates.json <classes.json> <kernel prefix> <kernel ms> [isa.s]

* issue costs: tools/issue_rate.hip (cycles per wave64 instruction per SIMD, >= 2 waves resident):
  full rate 2.3 (v_add/mul/fma_f32, v_add_u32, v_and/xor, v_mov), half rate 4.1 (compares, selects, min/max, shifts,
  integer multiplies, conversions, fp64, lane ops), quarter rate 8.1 (v_sqrt/rcp/rsq_f32).
* counters: SQ_INSTS_VALU and its classes (tools/pmc_classes.sh).  ADD/MUL/FMA_F32 are full rate, TRANS_F32 quarter,
  *_F64 / CVT / INT64 half.  INT32 and the unclassified rest (moves, compares, selects, min/max, lane ops) mix full- and
  half-rate instructions: bounded by [all full, all half]; the central figure splits them by the STATIC instruction
  histogram of the kernel's ISA when one is given (else 50/50).
Output: JSON with valu_busy {lo, central, hi} = issue cycles / (1024 SIMDs x kernel time x clock), the scalar unit's
share, and the basis."""
import json
import os
import re
import sys

FULL, HALF, QUARTER, SALU = 2.3, 4.1, 8.1, 4.15
# four SIMDs per compute unit; the CU count of the profiled device comes from tools/round_summary.py / profile_round.sh
# (NLOS_COMPUTE_UNITS = torch's multi_processor_count there), 256 = an unpartitioned MI355X otherwise
N_SIMD = 4 * int(os.environ.get("NLOS_COMPUTE_UNITS", "256"))
# engine clock: measured on the forward kernel itself (tools/build_stamps.sh -> clock.json -> NLOS_CLOCK_GHZ, set by
# tools/round_summary.py) when available, else the 2.4 GHz peak (then `busy` is a lower bound)
CLOCK = float(os.environ.get("NLOS_CLOCK_GHZ", "2.4")) * 1e9

FULL_OPS = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac)_f32|^v_(add|sub|subrev)_u32|^v_(and|or|xor|not)_b32|^v_mov_b32|^v_bitop3|^v_add_nc")
SKIP = re.compile(r"^v_(sqrt|rcp|rsq|exp|log|sin|cos)_f32|_f64|^v_cvt|^v_pk_|^v_mad_u64|^v_lshl_add_u64|^v_lshlrev_b64|^v_lshrrev_b64|^v_ashrrev_i64|^v_mfma")


def static_split(isa_path, kernel_prefix):
    """fraction of full-rate instructions among the VALU instructions the counters do not classify, from the ISA text"""
    txt = open(isa_path).read()
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(kernel_prefix), txt, re.M)
    if not m:
        return None
    end = txt.find(".Lfunc_end", m.end())
    full = half = 0
    for line in txt[m.end():end if end > 0 else len(txt)].splitlines():
        t = line.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        op = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", t[0])
        if SKIP.search(op) or re.match(r"^v_(add|mul|fma|fmac|sub|subrev|mac)_f32", op):
            continue                     # classified by the counters
        if FULL_OPS.search(op):
            full += 1
        else:
            half += 1
    return full / max(full + half, 1)


def main():
    rates_path, classes_path, prefix, ms = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    isa = sys.argv[5] if len(sys.argv) > 5 else None
    d = json.load(open(classes_path))
    key = max((k for k in d if k.startswith(prefix) and d[k].get("SQ_INSTS_VALU", 0) > 0), key=lambda k: d[k]["SQ_INSTS_VALU"])
    c = d[key]
    g = lambda n: float(c.get(n, 0.0))
    valu = g("SQ_INSTS_VALU")
    f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
    trans = g("SQ_INSTS_VALU_TRANS_F32")
    half_known = (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") +
                  g("SQ_INSTS_VALU_CVT") + g("SQ_INSTS_VALU_INT64"))
    mixed = valu - f32 - trans - half_known          # INT32 + moves / compares / selects / min-max / lane ops
    sym = {"k_forward_grid": "k_forward_gridILi0ELi0ELb0ELi0E", "k_gradient": "k_gradientILi0ELi0ELb0ELi512E"}
    sym = next((v for k, v in sym.items() if prefix.startswith(k)), prefix)
    share = static_split(isa, sym) if isa else None
    frac_full = 0.5 if share is None else share
    cyc = lambda ff: f32 * FULL + trans * QUARTER + half_known * HALF + mixed * (ff * FULL + (1 - ff) * HALF)
    denom = N_SIMD * ms * 1e-3 * CLOCK
    out = {
        "kernel": key, "kernel_ms": ms,
        "valu_busy": {"lo": cyc(1.0) / denom, "central": cyc(frac_full) / denom, "hi": cyc(0.0) / denom},
        "mean_cycles_per_valu_inst": {"lo": cyc(1.0) / valu, "central": cyc(frac_full) / valu, "hi": cyc(0.0) / valu},
        "salu_busy": g("SQ_INSTS_SALU") * SALU / denom,
        "valu_insts": valu, "salu_insts": g("SQ_INSTS_SALU"), "branch_insts": g("SQ_INSTS_BRANCH"),
        "lds_insts": g("SQ_INSTS_LDS"), "vmem_insts": g("SQ_INSTS_VMEM"),
        "classes": {"f32_add_mul_fma (2.3 cycles)": f32, "trans_f32 (8.1)": trans, "f64 + cvt + int64 (4.1)": half_known,
                    "int32 + unclassified (2.3 ... 4.1)": mixed},
        "mixed_full_rate_fraction": frac_full, "mixed_fraction_basis": "static ISA histogram" if share is not None else "50/50",
        "basis": ("issue costs: profiles/r03_issue_rates.json (tools/issue_rate.hip, >= 2 waves per SIMD); counters: SQ_INSTS_VALU* "
                  "(tools/pmc_classes.sh, separate --pmc passes); %d SIMDs, engine clock %.3f GHz (%s); the scalar unit serves one "
                  "SIMD every 4.15 cycles and overlaps with vector issue of other waves"
                  % (N_SIMD, CLOCK / 1e9, "measured on the forward kernel: s_memtime / s_memrealtime, tools/build_stamps.sh"
                     if "NLOS_CLOCK_GHZ" in os.environ else "peak: a lower bound on busy")),
    }
    # Round 6: the same accounting with IN-SITU costs (tools/insitu_costs.py, profiles/r06_insitu_costs.json): what one more
    # instruction of a class costs this very kernel at three of its sites.  Back to back a "half-rate" instruction holds the SIMD
    # for 4.1 cycles; inside the kernel's mix it costs 2.1 - 2.9 (it overlaps with other waves' full-rate instructions), a
    # full-rate one 1.3 - 2.3.  central: class means of the trace sites, the build phases' share of the instructions (NLOS_BUILD_SHARE,
    # default 0.38) at the counting pass's discount; lo / hi: every class at its cheapest / dearest measured cost and the
    # unclassified instructions all full / all half rate.
    insitu = os.environ.get("NLOS_INSITU_JSON", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r06_insitu_costs.json"))
    if os.path.exists(insitu) and prefix.startswith("k_forward_grid"):
        t = json.load(open(insitu))
        trace = [c for c in t["costs"] if c["site"] in ("WALK", "GEN")]
        build = [c for c in t["costs"] if c["site"] == "COUNT"]
        fu = [c["cycles_per_instruction"] for c in trace if c["back_to_back"] == 2.3]
        ha = [c["cycles_per_instruction"] for c in trace if c["back_to_back"] == 4.1]
        mean = lambda v: sum(v) / len(v)
        disc = mean([c["cycles_per_instruction"] / (mean(fu) if c["back_to_back"] == 2.3 else mean(ha)) for c in build])
        share = float(os.environ.get("NLOS_BUILD_SHARE", "0.38"))
        scale = (1.0 - share) + share * disc
        tr_lo, tr_hi = QUARTER * min(min(ha) / HALF, 1.0), QUARTER        # transcendentals were not padded: scaled like the half-rate class
        cyc2 = lambda cf, ch, ct, ff: f32 * cf + trans * ct + half_known * ch + mixed * (ff * cf + (1 - ff) * ch)
        out["valu_busy_in_situ"] = {"lo": cyc2(min(fu), min(ha), tr_lo, 1.0) * scale / denom,
                                    "central": cyc2(mean(fu), mean(ha), 0.5 * (tr_lo + tr_hi), frac_full) * scale / denom,
                                    "hi": cyc2(max(fu), max(ha), tr_hi, 0.0) / denom}
        out["in_situ_costs"] = {"full_rate_trace": [min(fu), mean(fu), max(fu)], "half_rate_trace": [min(ha), mean(ha), max(ha)],
                                "build_discount": disc, "build_share_of_instructions": share, "source": os.path.basename(insitu)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
