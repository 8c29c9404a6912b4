#!/bin/bash
# GPU-box half of tools/ab_prebuild.sh with the SQ counters: per variant under build/ab/<i>/ the bench's kernel times and one
# tools/pmc_quick.sh pass (VALU / SALU / LDS / VMEM wave-instructions, active lanes, LDS bank-conflict cycles) -- with the
# -DNLOS_DIAG_NO_* switches of forward_grid.hip (a stage compiled out) it attributes counters to stages.
#   AB_ARGS: extra bench.py flags; AB_KERNEL: grep pattern of the kernel lines to show (default k_forward)
cd "$GRAFT_REPO_ROOT"
n=$(ls -d build/ab/[0-9]* | wc -l)
for i in $(seq 0 $((n-1))); do
  d=/tmp/nlos_abp_$i; rm -rf $d; mkdir -p $d
  cp -r nlos_surface_optimization_amd include tests oracle bench.py tools profiles $d/ 2>/dev/null
  cp build/ab/$i/libnlos_hip.so $d/nlos_surface_optimization_amd/libnlos_hip.so
  flags=$(cat build/ab/$i/flags.txt)
  echo "== variant $i [$flags]"
  (cd $d && export GRAFT_REPO_ROOT=$d && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-seconds 0 --share-steps 0 --diagnostic-no-gate ${AB_ARGS:-} 2>&1 | grep -E "DIAGNOSTIC|^\{" | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read()
try:
    d=json.loads(l); print('   ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})
except Exception: print('  ', l.strip()[-260:])
"; PMC_BENCH_ARGS="--diagnostic-no-gate --share-steps 0 ${AB_ARGS:-}" bash tools/pmc_quick.sh v$i 2>&1 | grep -E "${AB_KERNEL:-k_forward}" | tail -2)
done
