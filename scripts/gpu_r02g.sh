#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
{
for rep in 1 2 3; do
TLOAM_HIP_LIB=$V/lib_nt2_d21.so timeout 200 python scripts/k3_sweep.py 60 auto,512,448,384
TLOAM_HIP_LIB=$V/lib_nt2_d22.so timeout 200 python scripts/k3_sweep.py 60 auto,512
TLOAM_HIP_LIB=$V/lib_ntl_d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_nts_d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_nt2_x_d21.so timeout 200 python scripts/k3_sweep.py 60 488,512
TLOAM_HIP_LIB=$V/lib_nt2_w4d11.so timeout 200 python scripts/k3_sweep.py 60 512,768,1024
TLOAM_HIP_LIB=$V/lib_nt2_w3d11.so timeout 200 python scripts/k3_sweep.py 60 512,768
done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
for L in nt2_d21 ntl_d21; do
TLOAM_HIP_LIB=$V/lib_$L.so timeout 300 python bench.py --no-cpu-baseline --no-kitti --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], '| m1 ms/frame', d['m1_frame']['ms_per_frame'], '| K3 working us', d['roofline']['working_sweeps']['avg_launch_us'], 'b2b', d['roofline']['back_to_back']['avg_launch_us'], 'prebuilt', d['roofline'].get('prebuilt_k3',{}).get('avg_launch_us'))"
done
