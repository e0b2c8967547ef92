#!/bin/bash
# Quick GPU iteration loop: (optional) parity tests, then a short kernel trace of the 1 M-frame bench.
# usage: scripts/gpu_quick.sh <tag> [notest]
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ "${2:-}" != "notest" ]; then (cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > $O/pytest.txt); fi
cd $R && timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], '| m1 ms/frame', d['m1_frame']['ms_per_frame'], '| K3 working us', d['roofline']['working_sweeps']['avg_launch_us'], '| kitti seq ms/frame', d.get('kitti_sequence', {}).get('ms_per_frame'))"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload m1 --steps 3 --warmup 1 --no-cpu-baseline --no-kitti > /dev/null 2> $O/trace.err
cd $R && python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/stats.csv | head -${TOPN:-14}
rm -rf $O/trace; echo "pytest: $(tail -1 $O/pytest.txt 2>/dev/null)"
