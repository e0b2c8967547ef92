#!/bin/bash
# final checks of round 4: the whole GPU suite (per-test time-out), then the profile set
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_tests.sh r4k_tests
bash scripts/gpu_profile_r04.sh r04 2>&1 | tail -45
for knob in "TLOAM_X=1" "TLOAM_NO_SCAN_1P=1" "TLOAM_X=1" "TLOAM_NO_SCAN_1P=1"; do
echo "== m1 knob=[$knob]"
env $knob timeout 300 python bench.py --workload m1 --steps 30 --warmup 3 --no-cpu-baseline --no-kitti --no-side 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done
