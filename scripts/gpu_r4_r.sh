#!/bin/bash
# round 4: tloam_set_target_frame converting the four clouds and taking their bounds in ONE launch -- tests, per-call split
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
(timeout 900 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sequence.py tests/test_gpu_configs.py 2>&1 | tail -12) > $O/pytest.txt
tail -3 $O/pytest.txt
for rep in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 50 --warmup 10 --loop-frames 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kitti_sequence']
print('ms/frame', d['ms_per_step'], {a: b for a, b in k['per_call'].items() if a != 'note'}, 'all told', k['ms_per_frame_incl_pcie_upload'])"
done
