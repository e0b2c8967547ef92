#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
{
for rep in 1 2; do
for L in w2d21 w2d21nts w2d21nt2 x_w2d21nts; do
TLOAM_HIP_LIB=$V/lib_$L.so timeout 200 python scripts/k3_sweep.py 60 auto,512
done; done
} 2>&1 | grep -v "^$" | tee $O/k3_sweep.txt
cd /tmp && export TMPDIR=/tmp
for L in w2d21 w2d21nts; do
TLOAM_HIP_LIB=$V/lib_$L.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$L -o t -- python $R/scripts/k3_sweep.py 60 auto > /dev/null 2> $O/trace_$L.err
python $R/scripts/rocpd_stats.py $(find $O/trace_$L -name "*.db" | head -1) $O/stats_$L.csv | head -4
done
