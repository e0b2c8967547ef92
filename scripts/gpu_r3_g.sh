#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3g}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -8
timeout 100 python scripts/k3_only.py 20 1; timeout 100 python scripts/k3_only.py 10 4
for rep in 1 2; do
timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
done
