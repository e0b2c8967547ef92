#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4j; mkdir -p $O; cd $R
timeout 900 python -X faulthandler -m pytest -m gpu -q --timeout 300 -o faulthandler_timeout=200 tests/test_gpu_scale.py tests/test_gpu_parity.py 2>&1 | tail -6
for rep in 1 2; do
for knob in "TLOAM_X=1" "TLOAM_NO_SCAN_1P=1"; do
echo "== m1 knob=[$knob]"
env $knob timeout 300 python bench.py --workload m1 --steps 30 --warmup 3 --no-cpu-baseline --no-kitti --no-side 2>$O/err_$rep.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/frame', d['ms_per_step'], 'GN it/s', d['value'])"
done
done
for rep in 1 2 3; do
for lib in "" "$R/tloam_amd/_variants/lib_poll1.so"; do
echo "== headline lib=[${lib##*/}]"
env TLOAM_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'])"
done
done
bash scripts/gpu_timeline_m1.sh r4j_tl > /dev/null 2>&1; head -10 $R/gpurun_out/r4j_tl/timeline.txt
