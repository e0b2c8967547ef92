#!/bin/bash
# A/B of two builds on the same box: the shipped library against tloam_amd/_variants/lib_prev.so, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2 3; do
for lib in "" "$R/tloam_amd/_variants/lib_prev.so"; do
echo "== lib=${lib##*/}"
env TLOAM_HIP_LIB=$lib timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], 'host_wait', d['config']['host_wait_us_per_frame'])"
done
done
