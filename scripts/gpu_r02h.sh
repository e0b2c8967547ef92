#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
{
for rep in 1 2; do
TLOAM_HIP_LIB=$V/lib_nt2_d21.so timeout 200 python scripts/k3_stats.py 20 40 auto
TLOAM_HIP_LIB=$V/lib_nt2_d22.so timeout 200 python scripts/k3_stats.py 20 40 auto
TLOAM_HIP_LIB=$V/lib_nt2_w4d11.so timeout 200 python scripts/k3_stats.py 20 40 512
TLOAM_HIP_LIB=$V/lib_nt2_w3d11.so timeout 200 python scripts/k3_stats.py 20 40 512
TLOAM_HIP_LIB=$V/lib_nt2_w4d11.so timeout 200 python scripts/k3_stats.py 20 40 488
TLOAM_HIP_LIB=$V/lib_nt2_w4d11.so timeout 200 python scripts/k3_stats.py 20 40 576
timeout 200 python scripts/k3_stats.py 20 40 auto
done
} 2>&1 | grep -v "^$" | tee $O/k3_stats.txt
