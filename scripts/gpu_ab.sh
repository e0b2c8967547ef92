#!/bin/bash
# A/B of builds on ONE box, interleaved: gpu_ab.sh NAME [NAME ...] -- NAME = "base" (the shipped library) or a variant of
# tloam_amd/_variants/lib_NAME.so (python -m tloam_amd.build --variant NAME -- -DFLAG ...).  Per build and repetition: the headline
# frames (ms/frame, GN iteration by the device clock), the 1 M frame (ms/frame, GN iteration, in-frame K3).  TESTS=1: the parity /
# golden / sequence / SE(3) tests on every variant first.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/ab; mkdir -p $O
REPS=${REPS:-3}
for name in "$@"; do
  lib=""; [ "$name" != base ] && lib=$R/tloam_amd/_variants/lib_$name.so
  if [ "${TESTS:-0}" = 1 ]; then
    echo "== tests lib=$name"; env TLOAM_HIP_LIB=$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sequence.py tests/test_gpu_se3.py tests/test_third_party_pins.py tests/test_gpu_scale.py -m gpu -q -x --timeout 300 2>&1 | tail -3
  fi
done
for rep in $(seq 1 $REPS); do
for name in "$@"; do
  lib=""; [ "$name" != base ] && lib=$R/tloam_amd/_variants/lib_$name.so
  env TLOAM_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-kitti --no-side --steps 200 --warmup 20 --m1-steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-10s rep $rep: headline ms/frame %.4f  GN-iter us %s | 1M ms/frame %.4f  GN-iter us %s  K3 in-frame us %s  frac %s' % ('$name', d['ms_per_step'], d['config']['gn_iteration_us'], d['m1_frame']['ms_per_frame'], r.get('gn_iteration_us'), r.get('in_frame_avg_launch_us'), r.get('gn_iteration_frac')))"
done
done
