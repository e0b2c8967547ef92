#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/go; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for W in m1 kitti; do for ORD in random sorted; do
rm -rf $O/trace; timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/grid_order_probe.py $ORD $W 2> $O/err.txt | grep rc
python $R/scripts/rocpd_stats.py $(find $O/trace -name '*.db' | head -1) | grep -E "k_grid_count_all|k_query_bin|k_build_sorted|k_grid_scatter|k_grid_finalize_sc" | awk -F'",' '{print "   " substr($1,1,60) " | " $2}' | cut -c1-110
done; done; rm -rf $O/trace
