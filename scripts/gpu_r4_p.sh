#!/bin/bash
# round 4: bucketed ranking of the feature candidates + one packed download -- feature tests, then the bench's adjacent rows
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
(timeout 600 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_feature.py tests/test_gpu_replay.py 2>&1 | tail -12) > $O/pytest.txt
tail -5 $O/pytest.txt
for rep in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 40 --warmup 10 --kitti-frames 40 --loop-frames 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['adjacent_rows'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-m1 --no-side --steps 10 --warmup 5 --kitti-frames 10 --loop-frames 10 > /dev/null 2> $O/trace.err
(cd $R && python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/stats.csv > /dev/null); rm -rf $O/trace
grep -i "rank\|feat\|pca" $O/stats.csv | cut -c1-60,200-260
grep -i "rank\|feat\|pca" $O/stats.csv | sed 's/.*)",//'
