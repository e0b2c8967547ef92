#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
{
for rep in 1 2 3; do
TLOAM_HIP_LIB=$V/lib_w2d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_w2d21nt2.so timeout 200 python scripts/k3_sweep.py 60 auto,512,384
TLOAM_HIP_LIB=$V/lib_nt2_d22.so timeout 200 python scripts/k3_sweep.py 60 auto,512
TLOAM_HIP_LIB=$V/lib_ntl_d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_nt2_w4d11.so timeout 200 python scripts/k3_sweep.py 60 512,768,1024
TLOAM_HIP_LIB=$V/lib_nt2_w3d11.so timeout 200 python scripts/k3_sweep.py 60 512,768
TLOAM_HIP_LIB=$V/lib_x_nt2_d21.so timeout 200 python scripts/k3_sweep.py 60 488,512
done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
