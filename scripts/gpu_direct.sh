#!/bin/bash
# the direct set (round 6): its tests, then the 1 M frame with and without it on ONE box, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -X faulthandler -m pytest tests/test_gpu_scale.py -m gpu -q -x --timeout 300 -k "direct" 2>&1 | tail -25
for rep in 1 2 3; do
for nd in "" 1; do
  env ${nd:+TLOAM_NO_DIRECT_SET=1} timeout 300 python bench.py --no-cpu-baseline --no-kitti --no-side --steps 50 --warmup 10 --m1-steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-8s rep $rep: 1M ms/frame %.4f  GN-iter us %s  K3 in-frame us %s frac %s  n_corr %s pose_err %.2e' % ('${nd:+compact}' or 'direct', d['m1_frame']['ms_per_frame'], r.get('gn_iteration_us'), r.get('in_frame_avg_launch_us'), r.get('in_frame_frac'), d['m1_frame']['n_corr'], d['m1_frame']['pose_err_vs_truth_m']))"
done
done
