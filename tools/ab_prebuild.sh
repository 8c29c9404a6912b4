#!/bin/bash
# A/B of build variants WITHOUT spending GPU-box minutes on hipcc: builds every variant HERE (the build container
# cross-compiles gfx950) into build/ab/<i>/ -- which travels with the gpurun snapshot -- and tools/ab_run_prebuilt.sh
# benches them interleaved on the GPU box.
#   tools/ab_prebuild.sh "<EXTRA flags A>" "<EXTRA flags B>" ...      (AB_FEAT0=1: forward_grid.hip with one feature set, 10x faster)
# Only the translation units a flag can reach are rebuilt per variant: objects of the shipped build are reused for
# the rest (AB_FILES="forward_grid gradient ...", default: the four render files).
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
rm -rf build/ab; mkdir -p build/ab
FILES=${AB_FILES:-forward_grid gradient forward_bvh render_kernels}
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
[ -n "$AB_FEAT0" ] && BASE="$BASE -DNLOS_ONLY_FEAT0"
make -s -C nlos_surface_optimization_amd/csrc -j8 >/dev/null
i=0
for flags in "$@"; do
  d=build/ab/$i; mkdir -p $d/obj
  echo "$flags" > $d/flags.txt
  for f in bvh_build forward_grid forward_bvh gradient render_kernels regulariser optimiser nlos_api; do
    if echo " $FILES " | grep -q " $f " && { [ -n "$flags" ] || [ -n "$AB_FEAT0" ]; }; then
      ( /opt/rocm/bin/hipcc $BASE $flags -c nlos_surface_optimization_amd/csrc/$f.hip -o $d/obj/$f.o 2>&1 | grep -E "error" || true ) &
    else
      cp nlos_surface_optimization_amd/csrc/$f.o $d/obj/$f.o
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libnlos_hip.so $d/obj/*.o
  rm -rf $d/obj
  echo "variant $i [$flags]: $(stat -c %s $d/libnlos_hip.so) bytes"
  i=$((i+1))
done
