#!/bin/bash
# In-situ cost of a VALU instruction (round 6): benches the pad variants prebuilt HERE by
#   AB_FEAT0=1 AB_FILES=forward_grid tools/ab_prebuild.sh "" "-DNLOS_DIAG_PAD_WALK=16" ... (build/ab/0..8)
# interleaved, then runs the stamped library (build/ab/9, see tools/run_stamps_prebuilt.sh) for the trip counts the
# pads are multiplied by.   gpurun -- bash tools/pad_test.sh   -> gpurun_out/pad_test.log, gpurun_out/pad_stamps.log
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
mv build/ab/9 build/ab_stamps 2>/dev/null
AB_STEPS=${AB_STEPS:-30} AB_ROUNDS=${AB_ROUNDS:-3} AB_ARGS="--sustain-seconds 0" bash tools/ab_run_prebuilt.sh 2>&1 | tee gpurun_out/pad_test.log
if [ -d build/ab_stamps ]; then
  mkdir -p build/ab/9 && cp build/ab_stamps/libnlos_hip.so build/ab/9/
  bash tools/run_stamps_prebuilt.sh 9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pad_stamps.log
fi
