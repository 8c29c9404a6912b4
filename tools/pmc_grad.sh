#!/bin/bash
# targeted counters for the instruction mix / stall picture of the render kernels (bounded passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_grad
mkdir -p $OUT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/d -- $B > $OUT/d.log 2>&1
python3 tools/pmc_summary.py $OUT
