#!/bin/bash
# tests + short bench summary (+ K3 stats) on the current build.  usage: gpu_quick2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-q}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5) > $O/pytest.txt; cat $O/pytest.txt
timeout 200 python scripts/k3_stats.py 20 40 auto | tee $O/k3_stats.txt
timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>$O/bench.err > $O/bench.json; python - <<'P'
import json,sys,os
d=json.loads(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out',sys.argv[1] if len(sys.argv)>1 else 'q','bench.json')).read().strip().splitlines()[-1]) if False else None
P
python -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('kitti pair ms/frame', d['ms_per_step'], 'GN it/s', d['value'], '| m1 ms/frame', d['m1_frame']['ms_per_frame'], '| K3 events us', r['working_sweeps']['avg_launch_us'], 'b2b', r['back_to_back']['avg_launch_us'], 'prebuilt', r.get('prebuilt_k3',{}).get('avg_launch_us'), r.get('prebuilt_k3',{}).get('frac'), '| seq', d.get('kitti_sequence',{}).get('ms_per_frame'), 'loop', d.get('odometry_loop',{}).get('ms_per_frame'))"
