#!/bin/bash
# round 4, first GPU call: the new tests, then the 1 M frame with / without the one-launch GN iteration, then its timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -q -x \
  tests/test_gpu_scale.py::test_one_launch_gn_iteration_of_large_sets_is_exact \
  tests/test_gpu_parity.py::test_staged_frames_equal_direct_hand_over \
  tests/test_gpu_parity.py::test_hand_over_time_out_falls_back_to_one_launch_per_iteration \
  "tests/test_gpu_parity.py::test_device_driven_loop_equals_host_driven_loop" \
  tests/test_gpu_parity.py::test_solve_launch_variants_are_exact \
  tests/test_gpu_configs.py tests/test_gpu_multirank.py tests/test_gpu_scale.py 2>&1 | tail -40) > $O/pytest.txt
grep -E "passed|failed|rror" $O/pytest.txt | tail -12
for rep in 1 2; do
for knob in "" "TLOAM_NO_FUSED_LARGE=1"; do
echo "== m1 knob=[$knob]"
env $knob timeout 300 python bench.py --workload m1 --steps 30 --warmup 3 --no-cpu-baseline --no-kitti --no-side 2>$O/err_$rep.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('ms/frame', d['ms_per_step'], 'GN it/s', d['value'], 'k3 avg', r.get('avg_launch_us'), 'work', (r.get('working_sweeps') or {}).get('avg_launch_us'), 'b2b', r['back_to_back']['avg_launch_us'], 'stream', r.get('stream_phase'))"
done
done
bash scripts/gpu_timeline_m1.sh r4a_tl > /dev/null 2>&1; cat $R/gpurun_out/r4a_tl/timeline.txt | head -70
