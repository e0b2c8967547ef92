#!/bin/bash
# feature path: tests, then the bench's adjacent rows three times
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python -X faulthandler -m pytest -m gpu -q --timeout 200 tests/test_gpu_feature.py tests/test_gpu_replay.py 2>&1 | tail -3
for rep in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 20 --warmup 5 --kitti-frames 10 --loop-frames 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('feature_extract_ms', d['adjacent_rows']['feature_extract_ms'])"
done
