#!/bin/bash
# how many outer iterations to enqueue ahead of the device's verdicts: headline and odometry loop
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for a in 2 4 prev head; do
echo "== TLOAM_ENQUEUE_AHEAD=$a"
L=""; if [ $a = prev ]; then L=$R/tloam_amd/_variants/lib_prev.so; fi; if [ $a = head ]; then L=$R/tloam_amd/_variants/lib_head.so; fi
TLOAM_HIP_LIB=$L TLOAM_ENQUEUE_AHEAD=$a timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], 'host_wait', d['config']['host_wait_us_per_frame'], 'odometry', d.get('odometry_loop', {}).get('ms_per_frame'))"
done
done
