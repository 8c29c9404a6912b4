#!/bin/bash
# which PMC counters does this box offer (names only)?   -> gpurun_out/counters_avail.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 120 rocprofv3 -L > gpurun_out/counters_avail_full.txt 2>&1
grep -oE "\b(SQ|SQC|TCP|TCC|TA|TD|GRBM|SPI|CPC|CPF)_[A-Z0-9_]+" gpurun_out/counters_avail_full.txt | sort -u > gpurun_out/counters_avail.txt
wc -l gpurun_out/counters_avail.txt
