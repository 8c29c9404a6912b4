#!/bin/bash
# Round profile: kernel-trace stats + HBM traffic counters (separate --pmc passes), summaries -> gpurun_out/profile_<tag>/
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 3 --no-cpu-baseline --sustain-seconds 0 > $OUT/trace.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 > $OUT/write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 > $OUT/sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 > $OUT/sq2.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary_all.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary.json
python3 tools/round_summary.py $OUT 4096 4967 > $OUT/pmc_summary.json
tail -1 $OUT/trace.log > $OUT/bench_line_under_profiler.json
cat $OUT/kernel_stats.csv | cut -c1-200
