#!/bin/bash
# Kernel trace of the KITTI-density workload only (per-frame launch budget).  usage: scripts/gpu_kitti_trace.sh <tag>
TAG=${1:-kt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R && timeout 300 python bench.py --workload kitti --no-m1 --no-kitti --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti (same frame x200) ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 100 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/trace.err
cd $R && python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/stats.csv | head -${TOPN:-30}
rm -rf $O/trace
