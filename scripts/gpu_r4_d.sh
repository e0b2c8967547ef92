#!/bin/bash
# round 4: exec-mask micro-benchmark, the full GPU suite, headline A/B k_solve_all | k_solve_small
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4d; mkdir -p $O; cd $R
timeout 60 scripts/ubench/_bin/execmask | tee $O/execmask.txt
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest.txt
grep -E "passed|failed|rror" $O/pytest.txt | tail -12
for rep in 1 2 3; do
for knob in "TLOAM_X=1" "TLOAM_SOLVE_V1=1"; do
echo "== headline knob=[$knob]"
env $knob timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], 'host_wait', d['config']['host_wait_us_per_frame'])"
done
done
