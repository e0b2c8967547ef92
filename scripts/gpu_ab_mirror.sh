#!/bin/bash
# A/B on ONE box: host mirror (state + bounding boxes polled in pinned memory) on / off, alternating.
for rep in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then export TLOAM_NO_HOST_MIRROR=1; else unset TLOAM_NO_HOST_MIRROR; fi
timeout 200 python bench.py --no-cpu-baseline --no-kitti --m1-steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mirror', 'off' if $v else 'on ', 'kitti pair ms/frame', d['ms_per_step'], 'm1', d['m1_frame']['ms_per_frame'])"
done
done
