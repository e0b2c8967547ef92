#!/bin/bash
# round 4: k_pca_info over the points in the grid's order against index order (lib_pcaidx.so) -- feature tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
(timeout 600 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_feature.py tests/test_gpu_replay.py 2>&1 | tail -12) > $O/pytest.txt
tail -3 $O/pytest.txt
for rep in 1 2 3; do
for lib in "" pcaidx; do
L=""; [ -n "$lib" ] && L=$R/tloam_amd/_variants/lib_$lib.so
echo "== lib=[$lib]"
env TLOAM_HIP_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-m1 --no-side --steps 20 --warmup 5 --kitti-frames 10 --loop-frames 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('feature_extract_ms', d['adjacent_rows']['feature_extract_ms'])"
done
done
