#!/usr/bin/env python3
"""Loops of a kernel in hipcc's ISA text, with what they hold: instruction counts by unit, spill traffic
(scratch_*, v_readlane / v_writelane), and the instructions that identify a stage.
    hipcc ... --offload-device-only -S forward_grid.hip -o fg.s ;  python tools/isa_loops.py fg.s <mangled-name prefix> [min_len]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*%s\S*):" % re.escape(sys.argv[2]), txt, re.M)
if not m:
    sys.exit("kernel not found")
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 100
end = txt.find(".Lfunc_end", m.end())
labels, ins = {}, []
for line in txt[m.end():end].splitlines():
    t = line.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):", t)
    if mm:
        labels[mm.group(1)] = len(ins)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    ins.append(t.split(";")[0].strip())
loops = {}
for i, t in enumerate(ins):
    mm = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", t)
    if mm:
        lab = mm.group(1) or mm.group(2)
        if lab in labels and labels[lab] <= i:
            loops[lab] = max(loops.get(lab, 0), i)
print("%s: %d instructions, %d loops (shown: >= %d instructions)" % (m.group(1), len(ins), len(loops), min_len))
for lab, b in sorted(loops.items(), key=lambda kv: labels[kv[0]]):
    a = labels[lab]
    n = b - a + 1
    if n < min_len:
        continue
    c = collections.Counter(x.split()[0] for x in ins[a:b + 1])
    tags = ["%s:%d" % (k, c[k]) for k in ("ds_bpermute_b32", "v_sqrt_f32", "v_rcp_f32", "v_div_scale_f32", "s_barrier", "global_load_dwordx4",
                                           "ds_add_f64", "v_mul_lo_u32") if c.get(k)]
    lanes = c.get("v_readlane_b32", 0) + c.get("v_writelane_b32", 0)
    scratch = sum(v for k, v in c.items() if k.startswith("scratch_"))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop")))
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    print("@%5d %-10s len %5d | VALU %4d SALU %4d LDS %3d | v_readlane/writelane %3d scratch %2d | %s" % (a, lab, n, valu, salu, lds, lanes, scratch, " ".join(tags)))
