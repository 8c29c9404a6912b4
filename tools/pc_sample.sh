#!/bin/bash
# PC sampling of the benchmark's kernels (rocprofv3 beta feature): where do the waves of k_forward_grid spend their
# cycles, instruction by instruction?    gpurun -- bash tools/pc_sample.sh [tag] [method] [interval]
#   -> gpurun_out/pcs_<tag>/summary_*.json (tools/pc_summary.py), raw CSVs stay on the box (too large)
TAG=${1:-a}
METHOD=${2:-stochastic}
INTERVAL=${3:-65536}
UNIT=cycles
[ "$METHOD" = host_trap ] && UNIT=time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pcs_$TAG
mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INTERVAL \
    --output-format csv -d /tmp/pcs_$TAG -- python3 bench.py --steps ${PCS_STEPS:-20} --warmup 2 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 ${PCS_BENCH_ARGS:-} > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
find /tmp/pcs_$TAG -type f | head -20 >> $OUT/run.log
for f in $(find /tmp/pcs_$TAG -name "*pc_sampling*.csv"); do
  ls -la $f >> $OUT/run.log
  head -3 $f >> $OUT/run.log
  python3 tools/pc_summary.py $f $(find /tmp/pcs_$TAG -name "*kernel_trace.csv" | head -1) > $OUT/summary_$(basename $f .csv).json 2>> $OUT/run.log
done
tail -30 $OUT/run.log
