#!/bin/bash
# round-3 batch C: A/B of the hand-over (tagged rows vs ticket) and the single-sweep chunking (64 vs 128 for line/point kinds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3c}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
for rep in 1 2 3; do
for V in "default 0" "default 1" "chunk128 0" "chunk128 1"; do
set -- $V
if [ $1 = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_$1.so; fi
if [ $2 = 1 ]; then export TLOAM_NO_TAGGED_ROWS=1; else unset TLOAM_NO_TAGGED_ROWS; fi
timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib $1 no_tagged $2: ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done; done | tee $O/ab.txt
