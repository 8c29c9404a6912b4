#!/bin/bash
# parity tests of the forward path + bench line + instruction counters of the render kernels: tools/quick_eval.sh <tag>
TAG=${1:-q}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --sustain-seconds 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f sustained %.3f' % (d['ms_per_step'], d['sustained_ms_per_step']), {k: round(v, 4) for k, v in d['roofline']['kernel_ms'].items()}, 'parity %.2e' % d['parity']['transient_rel_l2'])"; done
bash tools/pmc_quick.sh $TAG 2>&1 | tail -3
