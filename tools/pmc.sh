#!/bin/bash
# usage: tools/pmc.sh <tag> -- collects SQ + cache counters for the bench kernels (separate --pmc passes)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $OUT/p5 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p5.log 2>&1
python3 tools/pmc_summary.py $OUT
