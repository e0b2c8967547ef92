#!/bin/bash
# round 4: the search grids built at target hand-over (tloam_set_target_frame) against inside scanMatching (TLOAM_NO_GRID_AHEAD) --
# the whole GPU suite, then the per-call split of the sequence block, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_tests.sh r4s
for rep in 1 2; do
for knob in TLOAM_X=1 TLOAM_NO_GRID_AHEAD=1; do
echo "== knob=[$knob]"
env $knob timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 50 --warmup 10 --loop-frames 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kitti_sequence']
print('ms/frame', d['ms_per_step'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], {a: b for a, b in k['per_call'].items() if a != 'note'}, 'all told', k['ms_per_frame_incl_pcie_upload'], 'seq', k['ms_per_frame'])"
done
done
