#!/bin/bash
# self-preparing / self-finishing Solve: parity subset, then the headline with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3i}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -30) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -6
for rep in 1 2; do
for knob in "" "TLOAM_NO_FINISH_IN_SOLVE=1" "TLOAM_NO_SELF_PREPARE=1"; do
echo "== $knob"
env $knob timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], 'host_wait', d['config']['host_wait_us_per_frame'])"
done
done
