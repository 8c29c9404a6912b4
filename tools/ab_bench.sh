#!/bin/bash
# A/B bench of build variants: tools/ab_bench.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# builds each variant into /tmp/nlos_ab_<i> (never the shipped library) and runs bench.py against it, interleaved.
cd "$GRAFT_REPO_ROOT"
i=0
for flags in "$@"; do
  d=/tmp/nlos_ab_$i; rm -rf $d; mkdir -p $d; cp -r nlos_surface_optimization_amd include tests oracle bench.py $d/ 2>/dev/null
  make -s -C $d/nlos_surface_optimization_amd/csrc clean >/dev/null 2>&1
  make -s -C $d/nlos_surface_optimization_amd/csrc -j4 EXTRA="$flags" 2>&1 | grep -E "error" | head
  i=$((i+1))
done
for round in 1 2; do
  i=0
  for flags in "$@"; do
    d=/tmp/nlos_ab_$i
    (cd $d && python3 bench.py --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline --share-steps 0 ${AB_ARGS:-} 2>$d/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $i [$flags] round $round: %.2f Gs/s  %.3f ms' % (d['value']/1e9, d['ms_per_step']), {k: round(v,3) for k,v in d['roofline']['kernel_ms'].items()})" || tail -5 $d/err.log)
    i=$((i+1))
  done
done
