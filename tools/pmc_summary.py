#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel (short name), STEADY-STATE mean counter value per dispatch.

The first dispatch of every kernel in a bench run is the synthetic-data render (another mesh, another
seed: a different amount of work), so it is left out of the means whenever a kernel has three or more
dispatches in a pass; it is reported separately as `first_dispatch`."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "nlos" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # a bench run also renders small blocks (the parity gate): keep the launches of the largest grid of each kernel
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
    rows = [r for r in rows if int(r["Grid_Size"]) == biggest[r["Kernel_Name"]]]
    for r in rows:
        k = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", k)
        short = (m.group(1) + (m.group(2) or "")) if m else k
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    o = {}
    first = {}
    for c, v in sorted(d.items()):
        steady = v[1:] if len(v) >= 3 else v
        o[c] = sum(steady) / len(steady)
        if len(v) >= 3:
            first[c] = v[0]
    o["dispatches"] = max(len(v) for v in d.values())
    if first:
        o["first_dispatch"] = first
    out[k] = o
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
