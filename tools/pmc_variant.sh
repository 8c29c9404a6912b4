#!/bin/bash
# SQ instruction/cycle counters for a build variant: tools/pmc_variant.sh "<EXTRA flags>" <tag>
FLAGS="$1"; TAG=${2:-v}
cd "$GRAFT_REPO_ROOT"
d=/tmp/nlos_pv_$TAG; rm -rf $d; mkdir -p $d; cp -r nlos_surface_optimization_amd include tests oracle bench.py tools $d/
make -s -C $d/nlos_surface_optimization_amd/csrc clean >/dev/null 2>&1
make -s -C $d/nlos_surface_optimization_amd/csrc -j4 EXTRA="$FLAGS" 2>&1 | grep error | head
OUT=$GRAFT_REPO_ROOT/gpurun_out/pv_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $d
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
python3 tools/pmc_summary.py $OUT > /dev/null
python3 -c "
import json
d=json.load(open('$OUT/summary.json'))
for k in d:
    if 'forward' in k:
        print('$TAG', k, {c: float('%.4g' % v) for c, v in d[k].items()})
"
