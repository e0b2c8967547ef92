#!/bin/bash
# parity subset, then the headline: shipped library against tloam_amd/_variants/lib_prev.so on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r3i}; mkdir -p $O; cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequence.py tests/test_gpu_golden.py tests/test_gpu_configs.py tests/test_gpu_odometry_loop.py -m gpu -q -x 2>&1 | tail -30) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -6
bash scripts/gpu_r3_k.sh
