#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8) > $O/pytest.txt; cat $O/pytest.txt
for L in default heavy default heavy; do
if [ $L = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$V/lib_$L.so; fi
timeout 300 python bench.py --no-cpu-baseline --no-m1 --kitti-frames 600 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'seq', d['kitti_sequence']['ms_per_frame'], d['kitti_sequence']['pose_err_vs_truth_m'], 'loop', d['odometry_loop']['ms_per_frame'])"
done
