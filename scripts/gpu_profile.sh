#!/bin/bash
# Round profile set on the GPU box: kernel trace of the default bench command + two PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).  Outputs under gpurun_out/<tag>/.
# usage: scripts/gpu_profile.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null
python scripts/pmc_summary.py "k3_accumulate<false>" $O/pmc_k3.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k_build_sorted" $O/pmc_k1.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > /dev/null
rm -rf $O/pmc_fetch/*.db.tmp
tail -1 $O/bench.json
head -12 $O/kernel_stats.csv
cat $O/pmc_k3.json | head -60
