#!/bin/bash
# GPU-box half of tools/ab_prebuild.sh: benches the libraries under build/ab/<i>/ interleaved, two rounds.
#   AB_STEPS (10), AB_ARGS (extra bench.py flags, e.g. --diagnostic-no-gate), AB_ROUNDS (2)
cd "$GRAFT_REPO_ROOT"
n=$(ls -d build/ab/[0-9]* | wc -l)
for i in $(seq 0 $((n-1))); do
  d=/tmp/nlos_ab_$i; rm -rf $d; mkdir -p $d
  cp -r nlos_surface_optimization_amd include tests oracle bench.py profiles $d/ 2>/dev/null
  cp build/ab/$i/libnlos_hip.so $d/nlos_surface_optimization_amd/libnlos_hip.so
done
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  for i in $(seq 0 $((n-1))); do
    d=/tmp/nlos_ab_$i; flags=$(cat build/ab/$i/flags.txt)
    (cd $d && python3 bench.py --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline --share-steps 0 ${AB_ARGS:-} 2>$d/err.log | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('variant $i [$flags] round $round: %.2f Gs/s  %.3f ms' % (d['value']/1e9, d['ms_per_step']), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" 2>/dev/null || echo "variant $i [$flags] round $round: $(grep -E 'DIAGNOSTIC|FAILED|Error' $d/err.log | tail -2 | cut -c1-400)")
  done
done
