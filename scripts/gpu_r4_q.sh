#!/bin/bash
# round 4: the PCA pass of the feature extraction -- (a) points in the grid's order against index order (-DTLOAM_PCA_INDEX_ORDER),
# (b) the clipped four-per-trip walk of the registration path (tl_walk.hpp) against the plain one (-DTLOAM_PCA_PLAIN_WALK:
# lib_pcaplain.so).  Feature / parity tests, then the A/B named by $1 (default pcaplain), interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
V=${1:-pcaplain}
(timeout 900 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_feature.py tests/test_gpu_replay.py tests/test_gpu_parity.py tests/test_gpu_golden.py 2>&1 | tail -12) > $O/pytest.txt
tail -3 $O/pytest.txt
for rep in 1 2 3; do
for lib in "" $V; do
L=""; [ -n "$lib" ] && L=$R/tloam_amd/_variants/lib_$lib.so
echo "== lib=[$lib]"
env TLOAM_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 20 --warmup 5 --kitti-frames 10 --loop-frames 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('feature_extract_ms', d['adjacent_rows']['feature_extract_ms'], 'ms/frame', d['ms_per_step'])"
done
done
