#!/bin/bash
# round-3 batch A: GPU tests, phase stamps of the fused GN iteration, headline, per-frame timeline, step instruction counts
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3a}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.txt; tail -3 $O/pytest.txt
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 120 python scripts/step_profile_small.py > $O/stepprof.txt 2>&1; cat $O/stepprof.txt
timeout 300 python bench.py --no-cpu-baseline --no-m1 --kitti-frames 200 --steps 300 --warmup 30 2>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'seq', d['kitti_sequence']['ms_per_frame'], d['kitti_sequence']['ms_per_frame_p50'], d['kitti_sequence']['ms_per_frame_p99'], 'loop', d['odometry_loop']['ms_per_frame'])" | tee $O/headline.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/frame_timeline.py $(find $O/trace -name "*.db" | head -1) | tee $O/timeline.txt
rm -rf $O/trace
cd /tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
T=$(echo $C | tr ' ' '_'); rm -rf $O/$T
timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$T -o p -- python $R/bench.py --workload kitti --no-m1 --no-kitti --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> $O/err_$T.txt
python $R/scripts/pmc_summary.py "k_sweep_step_small" $O/$T.json $(find $O/$T -name "*.db" | head -1) > /dev/null
python -c "
import json; d=json.load(open('$O/$T.json'))
for k,v in d['counters'].items(): print(k, 'launches', v['launches'], 'mean_working %.0f' % v['mean_working'], 'mean_all %.0f' % v['mean_all'])"
rm -rf $O/$T
done
