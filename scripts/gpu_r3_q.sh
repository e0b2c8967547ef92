#!/bin/bash
# the 1 M frame: shipped library against lib_prev.so on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for lib in "" "$R/tloam_amd/_variants/lib_prev.so"; do
TLOAM_HIP_LIB=$lib timeout 200 python bench.py --workload m1 --steps 10 --warmup 2 --no-cpu-baseline --no-kitti --no-side 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib ${lib##*/} m1 ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done
done
