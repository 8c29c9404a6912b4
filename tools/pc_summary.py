#!/usr/bin/env python3
"""Per-instruction summary of a rocprofv3 PC-sampling CSV:  python tools/pc_summary.py <pc_sampling.csv> [kernel_trace.csv]
Groups samples by kernel (through the dispatch / correlation id when the kernel trace is given) and instruction, and
counts samples, issued samples and stall reasons.  Output: JSON on stdout."""
import csv
import json
import sys
from collections import Counter, defaultdict

csv.field_size_limit(1 << 30)
path = sys.argv[1]
kname = {}
if len(sys.argv) > 2:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            for key in ("Dispatch_Id", "Correlation_Id"):
                if key in r:
                    kname[(key, r[key])] = r.get("Kernel_Name", "?")
    except Exception as e:  # noqa: BLE001
        print("kernel trace unreadable: %r" % (e,), file=sys.stderr)
rows = csv.DictReader(open(path))
cols = rows.fieldnames
per = defaultdict(lambda: {"n": 0, "issued": 0, "stall": Counter(), "type": Counter()})
tot = Counter()
for r in rows:
    k = kname.get(("Dispatch_Id", r.get("Dispatch_Id", ""))) or kname.get(("Correlation_Id", r.get("Correlation_Id", ""))) or "?"
    k = k[:60]
    ins = r.get("Instruction", "?")
    com = r.get("Instruction_Comment", "")
    key = (k, ins, com)
    d = per[key]
    d["n"] += 1
    tot[k] += 1
    wi = r.get("Wave_Issued_Instruction", r.get("Wave_Issued_Inst", ""))
    if str(wi) in ("1", "True", "true"):
        d["issued"] += 1
    if "Stall_Reason" in r:
        d["stall"][r["Stall_Reason"]] += 1
    if "Instruction_Type" in r:
        d["type"][r["Instruction_Type"]] += 1
out = {"columns": cols, "samples_per_kernel": dict(tot), "kernels": {}}
for (k, ins, com), d in per.items():
    out["kernels"].setdefault(k, []).append({"inst": ins, "at": com, "n": d["n"], "issued": d["issued"],
                                             "stall": dict(d["stall"]), "type": dict(d["type"])})
for k in out["kernels"]:
    out["kernels"][k].sort(key=lambda e: -e["n"])
    out["kernels"][k] = out["kernels"][k][:4000]
print(json.dumps(out))
