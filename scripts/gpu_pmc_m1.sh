#!/bin/bash
# counters of the 1 M frame's grid-build / query-sort kernels: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and where the
# waves spend their cycles (SQ pass); --kernel-trace only alongside.   usage: scripts/gpu_pmc_m1.sh <tag>
TAG=${1:-pm1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload m1 --steps 8 --warmup 2 --no-cpu-baseline --no-kitti --no-side"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o p -- $CMD > /dev/null 2> $O/f.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o p -- $CMD > /dev/null 2> $O/w.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
  --kernel-trace -d $O/s -o p -- $CMD > /dev/null 2> $O/s.err
cd $R
for K in k_grid_count_all k_grid_scan_finalize_1p k_grid_scatter_qbin k_scan_1p k_query_scatter "k_build_sorted<1>" k_build_finish_large k_finish_direct; do
  python scripts/pmc_summary.py "$K" $O/pmc_$(echo $K | tr -d '<>').json $(find $O/f -name "*.db" | head -1) $(find $O/w -name "*.db" | head -1) $(find $O/s -name "*.db" | head -1) | python -c "
import sys, json
d = json.load(sys.stdin)
print('$K', 'traffic MB per launch: fetch %.1f write %.1f' % (d.get('fetch_bytes_per_launch_all', 0) / 1e6, (d.get('traffic_bytes_per_launch_all', 0) - d.get('fetch_bytes_per_launch_all', 0)) / 1e6))
for k, v in d['counters'].items():
    if k.startswith('SQ'): print('  %-22s mean %.5g' % (k, v['mean_all']))"
done
rm -rf $O/f $O/w $O/s
