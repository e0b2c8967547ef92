#!/bin/bash
# round 4: the new scan's clouds read in the pinned staging (submap update) / brought up by a copy kernel (set_source) --
# tests, then A/B against the copy command on the same box: submap_update host to host, the per-call split of the sequence block
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
(timeout 900 python -X faulthandler -m pytest -m gpu -q --timeout 300 -o faulthandler_timeout=240 tests/test_gpu_submap.py tests/test_gpu_odometry_loop.py tests/test_gpu_replay.py "tests/test_gpu_parity.py::test_solve_launch_variants_are_exact" tests/test_gpu_sequence.py 2>&1 | tail -15) > $O/pytest.txt
tail -4 $O/pytest.txt
for rep in 1 2 3; do
for knob in TLOAM_X=1 TLOAM_SUBMAP_COPY=1 "TLOAM_SUBMAP_COPY=1 TLOAM_STAGE_MEMCPY=1"; do
echo "== submap knob=[$knob]"
env $knob timeout 120 python scripts/submap_time.py 2>&1 | tail -1
done
done
for rep in 1 2; do
for knob in TLOAM_X=1 TLOAM_STAGE_MEMCPY=1; do
echo "== per-call knob=[$knob]"
env $knob timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kitti_sequence']
print('ms/frame', d['ms_per_step'], 'per_call', {a: b for a, b in k['per_call'].items() if a != 'note'}, 'odometry', d.get('odometry_loop', {}).get('ms_per_frame'), 'submap', d.get('adjacent_rows', {}).get('submap_update_ms'))"
done
done
