#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) > $O/pytest.txt; cat $O/pytest.txt
{
for rep in 1 2 3; do
timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_n_d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_p_d21.so timeout 200 python scripts/k3_sweep.py 60 auto,512
TLOAM_HIP_LIB=$V/lib_p_x_d21.so timeout 200 python scripts/k3_sweep.py 60 488,512
TLOAM_HIP_LIB=$V/lib_p_nt2_d21.so timeout 200 python scripts/k3_sweep.py 60 auto
TLOAM_HIP_LIB=$V/lib_p_w4d11.so timeout 200 python scripts/k3_sweep.py 60 512
done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
timeout 300 python bench.py --no-cpu-baseline --no-kitti --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], '| m1 ms/frame', d['m1_frame']['ms_per_frame'], '| K3 working us', d['roofline']['working_sweeps']['avg_launch_us'], 'prebuilt', d['roofline'].get('prebuilt_k3',{}).get('avg_launch_us'))"
