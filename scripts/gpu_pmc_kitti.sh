#!/bin/bash
# SQ counters of the KITTI-density frame's kernels (one PMC pass, --kernel-trace only): where the waves of the one-launch Solve
# and of the sixteen-lane search spend their cycles.  -> gpurun_out/<tag>/profiles/<tag>_pmc_sq_<kernel>.json
# usage: scripts/gpu_pmc_kitti.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; P=$O/profiles
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace -d $O/pmc_sq -o p -- python $R/bench.py --workload kitti --steps 60 --warmup 10 --no-cpu-baseline --no-kitti --no-m1 --no-side > /dev/null 2> $O/pmc_sq.err
cd $R
DB=$(find $O/pmc_sq -name "*.db" | head -1)
for K in k_solve_all k_build_sorted k_grid_count_all; do
  python scripts/pmc_summary.py "$K" $P/${TAG}_pmc_sq_$K.json $DB | python -c "
import sys, json
d = json.load(sys.stdin)
print('$K')
for k, v in d['counters'].items(): print('  %-24s launches %4d mean %.5g' % (k, v['launches'], v['mean_all']))"
done
rm -rf $O/pmc_sq
