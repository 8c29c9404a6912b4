#!/bin/bash
# VALU / LDS issue rates per instruction class and waves per SIMD (tools/issue_rate.hip), < 1 GPU-minute.
#   gpurun -- bash tools/issue_rate.sh     -> gpurun_out/issue_rates.json  (commit as profiles/rNN_issue_rates.json)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/issue_rate.hip -o /tmp/issue_rate
timeout 300 /tmp/issue_rate gpurun_out/issue_rates.json ${ISSUE_ITERS:-2048} | tee gpurun_out/issue_rates.log
