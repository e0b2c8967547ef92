#!/bin/bash
# the default bench command, as the driver runs it.  usage: gpu_benchfull.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-bf}; mkdir -p $O; cd $R
T0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$? wall=$(( $(date +%s) - T0 ))s"; tail -3 $O/bench.err
python - <<P
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "pcie", d["config"].get("ms_per_frame_incl_pcie_upload"))
print("roofline", r["kernel"], r["avg_launch_us"], r["frac"], "traffic", r["traffic"], "| in_frame", r.get("in_frame",{}).get("working_sweeps",{}).get("avg_launch_us"), r.get("in_frame",{}).get("frac"))
print("k1", r.get("roofline_k1"))
c = d.get("cpu_baseline", {})
print("cpu", c.get("value"), c.get("cores"), c.get("ms_per_frame"), "1thr", c.get("single_thread"), "ref shape", c.get("reference_thread_shape"), "m1", c.get("m1_frame"), "seq", c.get("kitti_sequence"))
print("sweep", c.get("thread_sweep"))
print("seq", d.get("kitti_sequence",{}).get("ms_per_frame"), "loop", d.get("odometry_loop",{}).get("ms_per_frame"), "m1", d.get("m1_frame",{}).get("ms_per_frame"))
P
