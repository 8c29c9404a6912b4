bash tools/ab_traffic.sh "" "-DNLOS_VIS_WG_ATOMIC" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl"
AB_WORKLOADS=" ;--mesh mannequin --bins 1024" bash tools/ab_sweep.sh "" "-DNLOS_VIS_WG_ATOMIC" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl"
