#!/bin/bash
# headline A/B: default lib vs variants, two alternating passes.  usage: gpu_headline.sh "<variant names>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for L in default $1; do
if [ $L = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_$L.so; fi
timeout 300 python bench.py --no-cpu-baseline --no-m1 --kitti-frames 200 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'seq', d['kitti_sequence']['ms_per_frame'], 'loop', d['odometry_loop']['ms_per_frame'])"
done; done
