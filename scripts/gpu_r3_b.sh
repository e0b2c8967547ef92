#!/bin/bash
# round-3 batch B: GPU tests, phase stamps, headline, timeline, latency ubench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3b}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 120 python scripts/step_profile_small.py > $O/stepprof.txt 2>&1; cat $O/stepprof.txt
timeout 300 python bench.py --no-cpu-baseline --no-m1 --kitti-frames 200 --steps 300 --warmup 30 2>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'seq', d['kitti_sequence']['ms_per_frame'], d['kitti_sequence']['ms_per_frame_p50'], d['kitti_sequence']['ms_per_frame_p99'], 'loop', d['odometry_loop']['ms_per_frame'])" | tee $O/headline.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline.txt
rm -rf $O/trace
[ -x scripts/ubench/_bin/lat ] && scripts/ubench/_bin/lat | tee $O/lat.txt
