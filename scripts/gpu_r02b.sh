#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
U="1,1,1,1,1,1,1,1"; A="0.95,0.95,1.05,1.05,1.05,1.05,0.95,0.95"; B="0.9,0.9,1.1,1.1,1.1,1.1,0.9,0.9"; C2="0.93,0.93,1.07,1.07,1.07,1.07,0.9,0.9"
{
for W in $U $A $B; do
echo "== weights $W"
TLOAM_K3_XCDW=$W TLOAM_HIP_LIB=$V/lib_s2w4.so timeout 200 python scripts/k3_sweep.py 60 256,384,512,640,768
TLOAM_K3_XCDW=$W TLOAM_HIP_LIB=$V/lib_s2w8.so timeout 200 python scripts/k3_sweep.py 60 128,192,256,320,384
done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
K3_PROFILE_DUMP=$O/prof_default.npy TLOAM_HIP_LIB=$V/lib_k3prof.so timeout 200 python scripts/k3_profile.py > $O/tl_default.txt 2>&1
TLOAM_K3_XCDW=$U K3_PROFILE_DUMP=$O/prof_s2w4_u.npy TLOAM_HIP_LIB=$V/lib_s2w4prof.so timeout 200 python scripts/k3_profile.py 2>&1 | tee $O/tl_s2w4_u.txt
TLOAM_K3_XCDW=$A K3_PROFILE_DUMP=$O/prof_s2w4_a.npy TLOAM_HIP_LIB=$V/lib_s2w4prof.so timeout 200 python scripts/k3_profile.py 2>&1 | tee $O/tl_s2w4_a.txt
