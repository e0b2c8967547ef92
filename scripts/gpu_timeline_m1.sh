#!/bin/bash
# per-frame kernel / gap table of the 1 M frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-tlm1}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload m1 --steps 12 --warmup 2 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline.txt
rm -rf $O/trace
