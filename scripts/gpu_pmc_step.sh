#!/bin/bash
# dynamic instruction counts of the stand-alone minimiser step (k_reduce_and_step; TLOAM_NO_FUSED_SMALL) on KITTI-size frames
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmcstep; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export TLOAM_NO_FUSED_SMALL=1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
T=$(echo $C | tr ' ' '_'); rm -rf $O/$T
timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$T -o p -- python $R/bench.py --workload kitti --no-m1 --no-kitti --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> $O/err_$T.txt
python $R/scripts/pmc_summary.py "k_reduce_and_step" $O/$T.json $(find $O/$T -name "*.db" | head -1) > /dev/null
python -c "
import json; d=json.load(open('$O/$T.json'))
for k,v in d['counters'].items(): print(k, 'launches', v['launches'], 'mean_working %.0f' % v['mean_working'], 'mean_all %.0f' % v['mean_all'])"
rm -rf $O/$T
done
