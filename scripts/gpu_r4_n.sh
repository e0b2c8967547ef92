#!/bin/bash
# round 4: k_vox_emit with the blocks' places from blockIdx against the start ticket (lib_voxticket.so) -- submap tests, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
(timeout 600 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_submap.py tests/test_gpu_odometry_loop.py 2>&1 | tail -8) > $O/pytest.txt
tail -3 $O/pytest.txt
for rep in 1 2 3; do
for lib in "" "$R/tloam_amd/_variants/lib_voxticket.so"; do
echo "== lib=[${lib##*/}]"
env TLOAM_HIP_LIB=$lib timeout 120 python scripts/submap_time.py 2>&1 | tail -1
done
done
