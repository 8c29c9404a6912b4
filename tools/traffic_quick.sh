#!/bin/bash
# HBM traffic counters of the forward kernel for given bench flags: tools/traffic_quick.sh <tag> [bench flags]
TAG=${1:-t}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic_$TAG
mkdir -p $OUT
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 "$@" > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --prewarm-seconds 0 "$@" > $OUT/write.log 2>&1
python3 tools/pmc_summary.py $OUT | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    if 'FETCH_SIZE' in v and v.get('WRITE_SIZE', 0) + v['FETCH_SIZE'] > 1000:
        print('%-44s fetch %.1f MB  write %.1f MB' % (k[:44], v['FETCH_SIZE'] / 1024, v.get('WRITE_SIZE', 0) / 1024))
"
