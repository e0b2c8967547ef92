#!/bin/bash
# the 1 M frame: parity of the large path, then the frame with and without the riding finish on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_configs.py tests/test_gpu_stress.py -m gpu -q -x  2>&1 | tail -3
for rep in 1 2 3; do
for lib in "" "$R/tloam_amd/_variants/lib_prev.so"; do
knob="lib=${lib##*/}"
env TLOAM_HIP_LIB=$lib timeout 200 python bench.py --workload m1 --steps 10 --warmup 2 --no-cpu-baseline --no-kitti --no-side 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$knob m1 ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done
done
