#!/bin/bash
# ISA metadata of the hot kernels (runs where hipcc is: the build container cross-compiles gfx950):
#   tools/isa_metadata.sh > profiles/rNN_isa_metadata.txt
cd "$(dirname "$0")/.." || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-fast-math -fno-slp-vectorize -Wno-unused-function --offload-device-only -S"
T=$(mktemp -d)
for f in forward_grid gradient bvh_build forward_bvh; do
  /opt/rocm/bin/hipcc $FLAGS nlos_surface_optimization_amd/csrc/$f.hip -o $T/$f.s 2>/dev/null
  echo "== $f.hip (hipcc $FLAGS)"
  grep -E "^\s+\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|max_flat_workgroup_size):" $T/$f.s \
    | sed 's/^\s*//' | paste -sd' ' | sed 's/ \.agpr_count/\n.agpr_count/g; s/ \.group_segment_fixed_size/\n.group_segment_fixed_size/g' \
    | grep -E "k_forward_gridILi0ELi0E|k_forward_gridILi0ELi2ELb0ELi0|k_forward_gridILi4ELi0ELb0ELi0|k_gradientILi0ELi0ELb0ELi512|k_gradient_fmILi0ELb0ELb0|k_build_bvh|k_forwardILi0ELi8" \
    | c++filt 2>/dev/null | sed 's/nlos::(anonymous namespace):://; s/nlos:://g'
done
rm -rf $T
