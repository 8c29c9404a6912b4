#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel (short name), mean counter value per dispatch."""
import csv, glob, os, re, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nlos" not in k:
            continue
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", k)
        short = (m.group(1) + (m.group(2) or "")) if m else k
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    out[k]["dispatches"] = max(len(v) for v in d.values())
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
