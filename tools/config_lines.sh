#!/bin/bash
# One bench line per BASELINE.json configuration on SURVEY 8(d)'s inputs + the reference's experiment shape (round 6):
#   gpurun -- bash tools/config_lines.sh     -> gpurun_out/config_lines.json (commit as profiles/rNN_config_lines.json)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for cfg in 2 3 4 4pairs 5 exp; do
  timeout 600 python3 bench.py --config $cfg --steps ${CFG_STEPS:-20} --warmup 3 --sustain-seconds 2 --no-cpu-baseline --share-steps 0 --dropin-steps 0 \
      > gpurun_out/cfg_$cfg.json 2> gpurun_out/cfg_$cfg.err || { echo "config $cfg: rc $?"; tail -5 gpurun_out/cfg_$cfg.err; }
done
python3 - <<'PY'
import json, os
out = {}
for cfg in ["2", "3", "4", "4pairs", "5", "exp"]:
    p = "gpurun_out/cfg_%s.json" % cfg
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        out[cfg] = {"error": str(e)}
        continue
    out[cfg] = d
    r = d.get("roofline", {})
    print("config %-6s %8.3f ms/step  %7.2f G samples/s  kernel_ms %s  frac %.3f  gate %s | %s" % (
        cfg, d["ms_per_step"], d["value"] / 1e9, {k: round(v, 4) for k, v in r.get("kernel_ms", {}).items()}, r.get("frac", float("nan")),
        {k: d["parity"].get(k) for k in ("rows", "transient_rel_l2", "gradient_rel_l2", "pass")}, d["config"]["path"].get("backend") if "path" in d["config"] else ""))
json.dump(out, open("gpurun_out/config_lines.json", "w"), indent=1)
PY
