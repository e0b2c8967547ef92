#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02k; mkdir -p $O; cd $R
V=$R/tloam_amd/_variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8) > $O/pytest.txt; cat $O/pytest.txt
TLOAM_HIP_LIB=$V/lib_stepprof_v1.so timeout 200 python scripts/step_profile.py 2>&1 | sed 's/^/v1 /' | tee $O/step_profile.txt
TLOAM_HIP_LIB=$V/lib_stepprof.so timeout 200 python scripts/step_profile.py 2>&1 | sed 's/^/v2 /' | tee -a $O/step_profile.txt
for L in default v1 default v1; do
if [ $L = default ]; then unset TLOAM_HIP_LIB; else export TLOAM_HIP_LIB=$V/lib_$L.so; fi
timeout 300 python bench.py --no-cpu-baseline --no-kitti --no-m1 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done
