#!/bin/bash
# confidence runs beyond the suite (usage: gpu_extra.sh [tag, default r06]): the whole synthetic KITTI-00 sequence
# (-> profiles/<tag>_kitti_sequence_4540.json), the
# long soak of the device-driven loop, the randomised HIP-vs-oracle sweep, the hypothesis search test with ten times the examples
TAG=${1:-r06}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${TAG}x; mkdir -p $O; cd $R
(timeout 600 python tests/tools/soak_device_loop.py 11 500 2>&1 | tail -3) > $O/soak.txt; cat $O/soak.txt
(timeout 600 python tests/tools/churn_contexts.py 8000 2>&1 | tail -2) > $O/churn.txt; cat $O/churn.txt
(timeout 600 python tests/tools/stress_dirty.py 500 2>&1 | tail -2) > $O/dirty.txt; cat $O/dirty.txt
(timeout 600 python tests/tools/stress_rows.py 400 2>&1 | tail -2) > $O/rows.txt; cat $O/rows.txt
(timeout 600 python tests/tools/stress_parity.py 160 2>&1 | tail -3) > $O/stress.txt; cat $O/stress.txt
(HYPOTHESIS_MAX_EXAMPLES=400 timeout 600 python -m pytest tests/test_third_party_pins.py -m gpu -q 2>&1 | tail -3) > $O/pins.txt; cat $O/pins.txt
timeout 900 python bench.py --kitti-frames 4540 --no-m1 --no-cpu-baseline > $O/${TAG}_kitti_sequence_4540.json 2> $O/seq.err
tail -c 400 $O/${TAG}_kitti_sequence_4540.json
