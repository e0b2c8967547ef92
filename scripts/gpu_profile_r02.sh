#!/bin/bash
# Round-2 profile set: kernel trace of the default bench command, PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of
# the K3-only characterisation run (pre-built 74.88 MB set) and of the 1 M frames, K3 launch-time distribution.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python scripts/k3_stats.py 20 40 auto > $O/k3_stats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/scripts/k3_only.py 100 > $O/pmc_fetch.txt 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python $R/scripts/k3_only.py 100 > $O/pmc_write.txt 2> $O/pmc_write.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_fetch -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/pmcf_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcf_write -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/pmcf_write.err
cd $R
python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null
python scripts/pmc_summary.py "k3_accumulate<false" $O/pmc_k3_prebuilt.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k3_accumulate<false" $O/pmc_k3_bench_m1.json $(find $O/pmcf_fetch -name "*.db" | head -1) $(find $O/pmcf_write -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k_build_sorted" $O/pmc_k1_bench_m1.json $(find $O/pmcf_fetch -name "*.db" | head -1) $(find $O/pmcf_write -name "*.db" | head -1) > /dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmcf_fetch $O/pmcf_write
head -14 $O/kernel_stats.csv | cut -c1-160; cat $O/pmc_k3_prebuilt.json | head -30; cat $O/k3_stats.txt
