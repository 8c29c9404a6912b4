#!/bin/bash
# quick counter pass of the render kernels (instruction mix / lanes / LDS): tools/pmc_quick.sh <tag>
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 ${PMC_BENCH_ARGS:-} > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 ${PMC_BENCH_ARGS:-} > $OUT/sq2.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary_all.json
python3 - "$OUT/pmc_summary_all.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "VALU" not in str(v) or v.get("SQ_INSTS_VALU", 0) < 1e6:
        continue
    lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * max(v.get("SQ_ACTIVE_INST_VALU", 1), 1))
    print("%-40s VALU %.4g SALU %.4g LDS %.4g VMEM %.4g | lanes %.3f | LDS conflict %.3f (%.4g of %.4g LDS cycles) | wave-cycles %.4g" % (
        k[:40], v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM", 0), lanes,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_LDS_BANK_CONFLICT", 0),
        v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_WAVE_CYCLES", 0)))
PY
