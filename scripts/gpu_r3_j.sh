#!/bin/bash
# frame timelines: default and with the finish outside the Solve launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3j}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for knob in "X=1" "TLOAM_NO_FINISH_IN_SOLVE=1"; do
echo "== $knob"
env $knob timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline_$knob.txt
rm -rf $O/trace; cd /tmp
done
