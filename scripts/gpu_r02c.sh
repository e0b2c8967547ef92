#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
U="1,1,1,1,1,1,1,1"; A="1.04,1.04,1.0,1.0,0.98,0.98,0.98,0.98"; B="1.08,1.08,1.0,1.0,0.96,0.96,0.96,0.96"
{
for W in $U $A $B; do
echo "== weights $W"
TLOAM_K3_XCDW=$W TLOAM_HIP_LIB=$V/lib_x_w2d21.so timeout 200 python scripts/k3_sweep.py 60 488,512
TLOAM_K3_XCDW=$W TLOAM_HIP_LIB=$V/lib_x_w4d11.so timeout 200 python scripts/k3_sweep.py 60 512,768,1024
done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
TLOAM_K3_BLOCKS=512 K3_PROFILE_REPS=24 TLOAM_K3_XCDW=$U K3_PROFILE_DUMP=$O/prof_x_u.npy TLOAM_HIP_LIB=$V/lib_x_w2d21prof.so timeout 200 python scripts/k3_profile.py 2>&1 | tee $O/tl_x_u.txt
