#!/bin/bash
# VALU instruction classes of the render kernels (for the weighted issue model, tools/issue_model.py):
#   gpurun -- bash tools/pmc_classes.sh <tag>     -> gpurun_out/pmc_<tag>/classes.json
TAG=${1:-cls}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 ${PMC_BENCH_ARGS:-}"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/c1 -- $B > $OUT/c1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS --output-format csv -d $OUT/c2 -- $B > $OUT/c2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/c3 -- $B > $OUT/c3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED --output-format csv -d $OUT/c4 -- $B > $OUT/c4.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/classes.json
tail -2 $OUT/c1.log $OUT/c2.log $OUT/c3.log $OUT/c4.log | cut -c1-300
python3 - "$OUT/classes.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if v.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(k[:50], {a: round(b / 1e6, 2) for a, b in v.items() if isinstance(b, (int, float))})
PY
