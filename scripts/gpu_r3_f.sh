#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3f}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -5
for rep in 1 2; do
for V in 0 1; do
if [ $V = 1 ]; then export TLOAM_NO_PERSISTENT_SOLVE=1; else unset TLOAM_NO_PERSISTENT_SOLVE; fi
timeout 120 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_persistent=$V 200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
timeout 120 python bench.py --steps 20 --warmup 5 --no-m1 --no-kitti --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_persistent=$V driver-style 20/5: value', d['value'], 'ms', d['ms_per_step'])"
done; done
unset TLOAM_NO_PERSISTENT_SOLVE
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline.txt
rm -rf $O/trace
