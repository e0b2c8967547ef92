#!/bin/bash
# round-3 batch D: GPU tests, stamps of the fused iteration, the full default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3d}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
TLOAM_HIP_LIB=$R/tloam_amd/_variants/lib_stepprof.so timeout 120 python scripts/step_profile_small.py > $O/stepprof.txt 2>&1; cat $O/stepprof.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "distinct", d["config"]["distinct_frames"], "repeated", d["config"].get("repeated_pair"))
r = d["roofline"]; print("roofline l3", r["avg_launch_us"], r["frac"], "cold", r["cold"]["avg_launch_us"], r["cold"]["frac"], "copy", r["measured_copy_GBps"])
print("in_frame", r["in_frame"]["working_sweeps"], r["in_frame"]["back_to_back"]["avg_launch_us"])
print("k1", r["roofline_k1"]["avg_launch_us"], "m1", d["m1_frame"]["ms_per_frame"])
print("seq", d["kitti_sequence"]["ms_per_frame"], d["kitti_sequence"]["ms_per_frame_p50"], d["kitti_sequence"]["ms_per_frame_p99"], d["kitti_sequence"]["ms_per_frame_incl_pcie_upload"])
print("loop", d["odometry_loop"]["ms_per_frame"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
timeout 120 python bench.py --steps 20 --warmup 5 --no-m1 --no-kitti --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style 20/5: value', d['value'], 'ms', d['ms_per_step'])"
timeout 200 python scripts/seq_probe.py 300 | tee $O/seq_probe.txt
