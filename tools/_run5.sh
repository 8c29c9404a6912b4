python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
AB_WORKLOADS=" ;--mesh mannequin --bins 1024;--non-confocal;--subdivide 1 --grid 32" bash tools/ab_sweep.sh "" "-DNLOS_VIS_FACE_ORDER" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl"
