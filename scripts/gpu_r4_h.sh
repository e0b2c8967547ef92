#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4h; mkdir -p $O; cd $R
timeout 600 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_multirank.py::test_callback_exchange_reports_a_peer_that_dies_inside_it tests/test_gpu_submap.py tests/test_gpu_odometry_loop.py tests/test_gpu_replay.py tests/test_gpu_configs.py 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline --no-m1 --steps 100 --warmup 10 2>$O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/frame', d['ms_per_step'], 'per_call', {k: d['config'].get(k) for k in ('set_source_ms', 'set_target_ms', 'scan_match_ms', 'set_source_plus_scan_match_ms', 'ms_per_frame_incl_pcie_upload')})
print('odometry', d['odometry_loop']['ms_per_frame'], d['odometry_loop']['ms_per_frame_p50'], 'adjacent', d['adjacent_rows'])
print('multi_stream', d['multi_stream'])"
