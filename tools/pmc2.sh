#!/bin/bash
# memory-pipeline counters for the render kernels; every pass is bounded by `timeout`
# (an over-subscribed counter block aborts rocprofv3 and can leave it hanging in finalisation)
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc2_$TAG
mkdir -p $OUT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN" \
           "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed/timeout"
done
python3 tools/pmc_summary.py $OUT > /dev/null
python3 -c "
import json
d=json.load(open('$OUT/summary.json'))
for k in d:
    if 'forward' in k or 'gradient' in k:
        print(k)
        for c,v in d[k].items(): print('   %-44s %.4g'%(c,v))
"
