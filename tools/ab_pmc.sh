#!/bin/bash
# A/B of build variants with instruction counters: tools/ab_pmc.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# each variant is built into /tmp/nlos_abp_<i> (never the shipped library); prints bench kernel times + VALU/SALU/LDS counts
cd "$GRAFT_REPO_ROOT"
i=0
for flags in "$@"; do
  d=/tmp/nlos_abp_$i; rm -rf $d; mkdir -p $d; cp -r nlos_surface_optimization_amd include tests oracle bench.py tools profiles $d/ 2>/dev/null
  make -s -C $d/nlos_surface_optimization_amd/csrc clean >/dev/null 2>&1
  make -s -C $d/nlos_surface_optimization_amd/csrc -j8 EXTRA="$flags" 2>&1 | grep -E "error" | head
  i=$((i+1))
done
i=0
for flags in "$@"; do
  d=/tmp/nlos_abp_$i
  echo "== variant $i [$flags]"
  (cd $d && export GRAFT_REPO_ROOT=$d && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --diagnostic-no-gate ${AB_ARGS:-} 2>&1 | grep -E "DIAGNOSTIC|^\{" | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read()
try:
    d=json.loads(l); print('   ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})
except Exception: print('  ', l.strip()[-260:])
"; PMC_BENCH_ARGS=--diagnostic-no-gate bash tools/pmc_quick.sh v$i 2>&1 | grep k_forward | tail -1)
  i=$((i+1))
done
