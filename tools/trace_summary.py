#!/usr/bin/env python3
"""Per-kernel duration summary of a rocprofv3 --kernel-trace CSV: the first dispatch of the bench
(the synthetic-data render: cold clocks, first touch of every buffer) is reported separately, so the
steady-state mean can be compared with the HIP-event figure bench.py prints."""
import csv
import json
import re
import sys
from collections import defaultdict


def main(path):
    rows = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
            rows[m.group(1) + (m.group(2) or "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                     int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]), grid))
    out = {}
    for k, v in rows.items():
        # a bench run also renders small blocks (the parity gate): keep the launches of the largest grid
        biggest = max(x[5] for x in v)
        v = [x for x in v if x[5] == biggest]
        v.sort()
        d = [x[1] / 1e6 for x in v]
        rest = d[1:] if len(d) > 1 else d
        s = sorted(rest)
        out[k] = {"dispatches": len(d), "first_ms": round(d[0], 4), "steady_mean_ms": round(sum(rest) / len(rest), 4),
                  "steady_median_ms": round(s[len(s) // 2], 4), "steady_min_ms": round(s[0], 4),
                  "steady_max_ms": round(s[-1], 4), "vgprs": v[-1][2], "lds_bytes": v[-1][3], "scratch_bytes": v[-1][4]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
