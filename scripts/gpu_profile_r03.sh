#!/bin/bash
# Round-3 profile set: the default bench line, kernel trace of the same command, PMC passes (FETCH_SIZE / WRITE_SIZE, separate
# runs) of the K3-only characterisation runs (74.88 MB Infinity-Cache-resident set and the 299.5 MB cold set) and of the 1 M
# frames, K3 launch-time distribution, KITTI frame timeline, timeline of the one-launch Solve.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
# (the PMC passes of the K3 characterisation sets come FIRST and their summaries are copied into profiles/ of this copy of the
#  repo, so that the bench line of the same run quotes counters collected on the kernels it timed)
cd /tmp && export TMPDIR=/tmp
for S in 1 4; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$S -o p -- python $R/scripts/k3_only.py 50 $S > $O/pmc_fetch_$S.txt 2> $O/pmc_fetch_$S.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$S -o p -- python $R/scripts/k3_only.py 50 $S > $O/pmc_write_$S.txt 2> $O/pmc_write_$S.err
done
cd $R
python scripts/pmc_summary.py "k3_accumulate<false" $O/pmc_k3_prebuilt.json $(find $O/pmc_fetch_1 -name "*.db" | head -1) $(find $O/pmc_write_1 -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k3_accumulate<false" $O/pmc_k3_prebuilt_cold.json $(find $O/pmc_fetch_4 -name "*.db" | head -1) $(find $O/pmc_write_4 -name "*.db" | head -1) > /dev/null
cp $O/pmc_k3_prebuilt.json profiles/${TAG}_pmc_k3_prebuilt.json; cp $O/pmc_k3_prebuilt_cold.json profiles/${TAG}_pmc_k3_prebuilt_cold.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python scripts/k3_stats.py 20 40 auto > $O/k3_stats.txt 2>&1
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 100 python scripts/solve_profile.py > $O/solve_timeline.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/trace.err
# instruction mix of the correspondence search at 1 M queries (two passes: issue counts, busy / wait cycles)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $O/pmck1_a -o p -- python $R/scripts/k1_time.py 4 > /dev/null 2> $O/pmck1_a.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/pmck1_b -o p -- python $R/scripts/k1_time.py 4 > /dev/null 2> $O/pmck1_b.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmcf_fetch -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/pmcf_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmcf_write -o p -- python $R/bench.py --workload m1 --steps 5 --warmup 1 --no-cpu-baseline --no-kitti --no-side > /dev/null 2> $O/pmcf_write.err
timeout 300 rocprofv3 --kernel-trace -d $O/ktrace -o t -- python $R/bench.py --workload kitti --no-m1 --no-kitti --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/ktrace.err
cd $R
python scripts/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats.csv > /dev/null
python scripts/pmc_summary.py "k_build_sorted<1>" $O/pmc_k1_instruction_mix.json $(find $O/pmck1_a -name "*.db" | head -1) $(find $O/pmck1_b -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k3_accumulate<false" $O/pmc_k3_bench_m1.json $(find $O/pmcf_fetch -name "*.db" | head -1) $(find $O/pmcf_write -name "*.db" | head -1) > /dev/null
python scripts/pmc_summary.py "k_build_sorted" $O/pmc_k1_bench_m1.json $(find $O/pmcf_fetch -name "*.db" | head -1) $(find $O/pmcf_write -name "*.db" | head -1) > /dev/null
python scripts/frame_timeline.py $(find $O/ktrace -name "*.db" | head -1) > $O/kitti_frame_timeline.txt 2>&1
rm -rf $O/pmck1_a $O/pmck1_b $O/trace $O/pmc_fetch_1 $O/pmc_write_1 $O/pmc_fetch_4 $O/pmc_write_4 $O/pmcf_fetch $O/pmcf_write $O/ktrace
head -16 $O/kernel_stats.csv | cut -c1-150; cat $O/k3_stats.txt; tail -22 $O/kitti_frame_timeline.txt
python - <<PY
import json
for n in ("pmc_k3_prebuilt", "pmc_k3_prebuilt_cold", "pmc_k3_bench_m1"):
    d = json.load(open("$O/%s.json" % n)); print(n, {k: d.get(k) for k in ("traffic_bytes_per_launch_all", "fetch_bytes_per_launch_all", "write_bytes_per_launch_all")})
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "cold", d["roofline"]["cold"]["frac"], "m1", d["m1_frame"]["ms_per_frame"])
PY
