#!/bin/bash
# One PMC pass over a short 1 M-frame bench; prints the per-launch means for one kernel.
# usage: scripts/gpu_pmc.sh <tag> <kernel-substring> <counter> [<counter> ...]
TAG=$1; KERN=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc -o p -- python $R/bench.py --workload m1 --steps 2 --warmup 1 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/pmc.err
cd $R && python scripts/pmc_summary.py "$KERN" $O/pmc.json $(find $O/pmc -name "*.db" | head -1) | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d['counters'].items(): print('%-28s launches %4d mean_working %.4g' % (k, v['launches'], v['mean_working']))"
rm -rf $O/pmc
