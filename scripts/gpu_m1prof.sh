#!/bin/bash
# kernel stats of the 1 M frame only.  usage: gpu_m1prof.sh <tag> [pytest-k-expr]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-m1}; mkdir -p $O; cd $R
if [ -n "${2:-}" ]; then timeout 600 python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4; fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload m1 --steps 6 --warmup 2 --no-cpu-baseline --no-kitti --no-side > $O/bench.json 2> $O/trace.err
cd $R && python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/stats.csv | head -${TOPN:-16} | cut -c1-150
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('m1 ms/frame (under rocprof)', d['ms_per_step'])"
rm -rf $O/trace
