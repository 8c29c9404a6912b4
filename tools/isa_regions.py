#!/usr/bin/env python3
"""Static instruction histogram of a kernel by SOURCE REGION (hipcc -S -gline-tables-only): every instruction is
attributed to the last `.loc` of the main source file seen before it (inlined helpers count for their call site's
neighbourhood), lines are bucketed into named regions, and each region reports VALU instructions by issue class
(costs: profiles/r06_issue_rates.json / r03_issue_rates.json).

    hipcc <FLAGS> --offload-device-only -S -gline-tables-only -DNLOS_ONLY_FEAT0 forward_grid.hip -o fg_g.s
    python tools/isa_regions.py fg_g.s k_forward_gridILi0ELi0ELb0ELi0E forward_grid.hip regions.json [rates.json]
regions.json: {"name": [first_line, last_line], ...} (lines of the main file); output: JSON."""
import collections
import json
import re
import sys


def strip(op):
    return re.sub(r"_e(32|64)$|_dpp$|_sdwa$|_e64_dpp$", "", op)


def main():
    txt = open(sys.argv[1]).read()
    kern, mainfile = sys.argv[2], sys.argv[3]
    regions = json.load(open(sys.argv[4]))
    rates = None
    if len(sys.argv) > 5:
        r = json.load(open(sys.argv[5]))["rates"]
        rates = {}
        for name, v in r.items():
            op = name.split()[0].split("+")[0]
            if "+" in name or " only" in name:
                continue
            w = v.get("w6") or v.get("w4")
            rates.setdefault(strip(op), w["cycles_per_inst_per_simd"])
    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt, re.M):
        files[int(m.group(1))] = (m.group(3) or m.group(2))
    main_ids = {k for k, v in files.items() if v.endswith(mainfile)}
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(kern), txt, re.M)
    end = txt.find(".Lfunc_end", m.end())
    cur = 0
    per_line = collections.defaultdict(collections.Counter)
    for line in txt[m.end():end].splitlines():
        t = line.strip()
        mm = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if mm:
            if int(mm.group(1)) in main_ids and int(mm.group(2)) > 0:
                cur = int(mm.group(2))
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        per_line[cur][strip(t.split()[0])] += 1
    out = {}
    for name, (a, b) in regions.items():
        c = collections.Counter()
        for ln, cc in per_line.items():
            if a <= ln <= b:
                c.update(cc)
        valu = {k: v for k, v in c.items() if k.startswith("v_")}
        rec = {"valu": sum(valu.values()), "salu": sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop"))),
               "lds": sum(v for k, v in c.items() if k.startswith("ds_")), "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "scratch_", "buffer_", "flat_")))}
        if rates:
            cyc, unknown = 0.0, collections.Counter()
            for k, v in valu.items():
                if k in rates:
                    cyc += v * rates[k]
                else:
                    unknown[k] += v
            rec["valu_cycles_known"] = round(cyc, 1)
            rec["valu_unpriced"] = dict(unknown.most_common())
        rec["top"] = dict(collections.Counter(valu).most_common(14))
        out[name] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
