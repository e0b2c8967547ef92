#!/bin/bash
# kernel-trace timeline of the 1 M frame (and, with KITTI=1, of the KITTI-density frame)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/tl; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/mtrace -o t -- python $R/bench.py --workload m1 --steps 12 --warmup 2 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/mtrace.err
(cd $R && python scripts/frame_timeline.py $(find $O/mtrace -name "*.db" | head -1) > $O/m1_frame_timeline.txt 2>&1); rm -rf $O/mtrace
cat $O/m1_frame_timeline.txt
if [ "${KITTI:-0}" = 1 ]; then
timeout 300 rocprofv3 --kernel-trace -d $O/ktrace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/ktrace.err
(cd $R && python scripts/frame_timeline.py $(find $O/ktrace -name "*.db" | head -1) > $O/kitti_frame_timeline.txt 2>&1); rm -rf $O/ktrace
cat $O/kitti_frame_timeline.txt
fi
