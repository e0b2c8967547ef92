#!/bin/bash
# A/B of ONE environment knob on one box, interleaved, on the 1 M frame (and the headline):
#   gpu_env_ab.sh VAR [reps] [value ...]      (no values: unset against VAR=1; a value "-" means unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=${1:-TLOAM_NO_DIRECT_SET}; REPS=${2:-3}; shift 2 2>/dev/null
VALS=("$@"); [ ${#VALS[@]} = 0 ] && VALS=(- 1)
for rep in $(seq 1 $REPS); do
for v in "${VALS[@]}"; do
  env $([ "$v" != "-" ] && echo "$VAR=$v") timeout 300 python bench.py --no-cpu-baseline --no-kitti --no-side --steps 50 --warmup 10 --m1-steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-28s rep $rep: 1M ms/frame %.4f  GN-iter us %s  K3 in-frame us %s  n_corr %s pose_err %.2e | headline %.4f' % ('$VAR=' + '$v', d['m1_frame']['ms_per_frame'], r.get('gn_iteration_us'), r.get('in_frame_avg_launch_us'), d['m1_frame']['n_corr'], d['m1_frame']['pose_err_vs_truth_m'], d['ms_per_step']))"
done
done
