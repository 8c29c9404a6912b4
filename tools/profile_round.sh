#!/bin/bash
# Round profile: kernel-trace stats + HBM traffic counters (separate --pmc passes), summaries -> gpurun_out/profile_<tag>/
# NLOS_GIT_HEAD=<hash> (the GPU box has no .git) is recorded in pmc_summary.json's stamp beside the hash of the kernel
# sources, which is what bench.py matches.  Usage: gpurun -- "NLOS_GIT_HEAD=$(git rev-parse --short=12 HEAD) bash tools/profile_round.sh r04"
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 3 --no-cpu-baseline --sustain-seconds 0 --share-steps 0 > $OUT/trace.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/sq2.log 2>&1
# VALU instruction classes for the weighted issue model (tools/issue_model.py; issue costs: profiles/r03_issue_rates.json)
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/cls1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/cls1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE --output-format csv -d $OUT/cls2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 --share-steps 0 > $OUT/cls2.log 2>&1
# the shader clock the forward kernel runs at, on the metric workload (stamped copy of the library in /tmp): -> clock.json
# (a stamped library prebuilt in the build container -- AB_FILES="forward_grid nlos_api bvh_build" tools/ab_prebuild.sh
# "-DNLOS_BUILD_STAMPS -DNLOS_FWD_STAMPS" -> build/ab/0 -- saves the hipcc minutes on the GPU box)
if grep -q NLOS_FWD_STAMPS build/ab/0/flags.txt 2>/dev/null; then
  NLOS_STAMP_GRID=64 NLOS_STAMP_NS=20000 NLOS_CLOCK_JSON=$GRAFT_REPO_ROOT/$OUT/clock.json timeout 600 bash tools/run_stamps_prebuilt.sh > $OUT/stamps.log 2>&1
  NLOS_STAMP_GRID=64 NLOS_STAMP_NS=20000 NLOS_STAMP_STRIDE=8 timeout 600 bash tools/run_stamps_prebuilt.sh > $OUT/stamps_rank_of_8.log 2>&1
else
  NLOS_STAMP_GRID=64 NLOS_STAMP_NS=20000 NLOS_CLOCK_JSON=$GRAFT_REPO_ROOT/$OUT/clock.json timeout 600 bash tools/build_stamps.sh > $OUT/stamps.log 2>&1
fi
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary_all.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 tools/trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary.json
# the kernel's ISA text, for the static full-rate / half-rate split of the instruction classes the counters lump together
# (build/isa/forward_grid.s, if the build container left one, saves two minutes of hipcc here)
if [ -f build/isa/forward_grid.s ]; then cp build/isa/forward_grid.s /tmp/nlos_forward_grid.s; else
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -fno-slp-vectorize -Wno-unused-function --offload-device-only -S nlos_surface_optimization_amd/csrc/forward_grid.hip -o /tmp/nlos_forward_grid.s 2>/dev/null
fi
NLOS_ISA=/tmp/nlos_forward_grid.s python3 tools/round_summary.py $OUT 4096 4902 > $OUT/pmc_summary.json
tail -1 $OUT/trace.log > $OUT/bench_line_under_profiler.json
cat $OUT/kernel_stats.csv | cut -c1-200
