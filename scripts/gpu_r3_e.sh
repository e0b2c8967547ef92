#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3e}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 120 python scripts/step_profile_small.py > $O/stepprof.txt 2>&1; cat $O/stepprof.txt
for rep in 1 2; do
timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
timeout 120 python bench.py --steps 20 --warmup 5 --no-m1 --no-kitti --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style 20/5: value', d['value'], 'ms', d['ms_per_step'])"
done
TLOAM_DEBUG_ALLOC=1 timeout 120 python scripts/alloc_probe.py 40 2>&1 | tee $O/alloc_probe.txt | tail -60
