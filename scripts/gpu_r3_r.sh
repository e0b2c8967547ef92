#!/bin/bash
# lanes per query of the KITTI-size search: sixteen (shipped) against four (lib_quad.so: -DTLOAM_K1_WIDE_LIMIT=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for lib in "" "$R/tloam_amd/_variants/lib_quad.so"; do
echo "== lib ${lib##*/}"
TLOAM_HIP_LIB=$lib timeout 100 python scripts/k1_time.py 20 kitti 2>&1 | tail -1
TLOAM_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
done
done
