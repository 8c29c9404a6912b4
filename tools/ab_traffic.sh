#!/bin/bash
# A/B of build variants with the HBM traffic counters: tools/ab_traffic.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# each variant is built into /tmp/nlos_abt_<i> (never the shipped library); prints FETCH_SIZE / WRITE_SIZE per kernel
cd "$GRAFT_REPO_ROOT"
i=0
for flags in "$@"; do
  d=/tmp/nlos_abt_$i; rm -rf $d; mkdir -p $d; cp -r nlos_surface_optimization_amd include tests oracle bench.py tools profiles $d/ 2>/dev/null
  make -s -C $d/nlos_surface_optimization_amd/csrc clean >/dev/null 2>&1
  make -s -C $d/nlos_surface_optimization_amd/csrc -j8 EXTRA="$flags" 2>&1 | grep -E "error" | head
  i=$((i+1))
done
i=0
for flags in "$@"; do
  d=/tmp/nlos_abt_$i
  echo "== variant $i [$flags]"
  (cd $d && export GRAFT_REPO_ROOT=$d && bash tools/traffic_quick.sh v$i --diagnostic-no-gate ${AB_ARGS:-} 2>&1 | grep -E "k_forward|k_gradient" | tail -2)
  i=$((i+1))
done
