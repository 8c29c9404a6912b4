#!/bin/bash
# A/B of build variants over several workloads: tools/ab_sweep.sh "<EXTRA A>" "<EXTRA B>" ...   (workloads: env AB_WORKLOADS, ';'-separated bench flags)
cd "$GRAFT_REPO_ROOT"
IFS=';' read -ra WL <<< "${AB_WORKLOADS:- ;--subdivide 1 --grid 32;--non-confocal;--mesh mannequin --bins 1024;--forward-only --grid 32}"
i=0
for flags in "$@"; do
  d=/tmp/nlos_abs_$i; rm -rf $d; mkdir -p $d; cp -r nlos_surface_optimization_amd include tests oracle bench.py tools profiles $d/ 2>/dev/null
  make -s -C $d/nlos_surface_optimization_amd/csrc clean >/dev/null 2>&1
  make -s -C $d/nlos_surface_optimization_amd/csrc -j8 EXTRA="$flags" 2>&1 | grep -E "error" | head
  i=$((i+1))
done
for w in "${WL[@]}"; do
  echo "== workload [$w]"
  for round in 1 2; do
    i=0
    for flags in "$@"; do
      d=/tmp/nlos_abs_$i
      (cd $d && python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --sustain-seconds 0.5 $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   variant $i [$flags]: sustained %.3f ms' % d['sustained_ms_per_step'], {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})")
      i=$((i+1))
    done
  done
done
