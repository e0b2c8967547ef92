#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-tl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline.txt
rm -rf $O/trace
