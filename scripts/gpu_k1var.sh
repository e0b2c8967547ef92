#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-k1v}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in default $2; do
if [ $L = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_$L.so; fi
rm -rf $O/trace; timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --workload m1 --steps 4 --warmup 1 --no-cpu-baseline --no-kitti --no-side > $O/bench_$L.json 2> $O/trace.err
echo "== $L: $(python $R/scripts/rocpd_stats.py $(find $O/trace -name '*.db' | head -1) | grep 'build_sorted<1>' | sed 's/\"[^\"]*\",/K1,/')  frame $(python -c "import json; print(json.loads(open('$O/bench_$L.json').read().strip().splitlines()[-1])['ms_per_step'])")"
done
rm -rf $O/trace
