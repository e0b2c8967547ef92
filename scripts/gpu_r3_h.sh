#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3h}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -6
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 100 python scripts/solve_profile.py 2>&1 | tail -6
for rep in 1 2; do
timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
timeout 120 python bench.py --steps 20 --warmup 5 --no-m1 --no-kitti --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style 20/5: value', d['value'], 'ms', d['ms_per_step'])"
done
