#!/bin/bash
# round 4: the three-wave step of k_solve_all -- tests, then A/B: coop (default) | step on one wave, same binary | the kernel compiled without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
(timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sequence.py 2>&1 | tail -30) > $O/pytest.txt
grep -E "passed|failed|rror" $O/pytest.txt | tail -12
for rep in 1 2 3; do
for cfg in "TLOAM_X=1|" "TLOAM_NO_COOP_STEP=1|" "TLOAM_X=1|$R/tloam_amd/_variants/lib_nocoop.so" "TLOAM_SOLVE_V1=1|"; do
knob=${cfg%%|*}; lib=${cfg##*|}
echo "== headline knob=[$knob] lib=[${lib##*/}]"
env $knob TLOAM_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-kitti --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200/20: ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'repeated', d['config']['repeated_pair']['ms_per_frame'], 'host_wait', d['config']['host_wait_us_per_frame'])"
done
done
for knob in "TLOAM_X=1" "TLOAM_NO_COOP_STEP=1"; do
env $knob TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 120 python scripts/solve_profile2.py 2>&1 | tail -12 | tee $O/solve_timeline_$knob.txt
done
bash scripts/gpu_timeline.sh r4c_tl > /dev/null 2>&1; cat $R/gpurun_out/r4c_tl/timeline.txt 2>/dev/null | head -20
