#!/bin/bash
# K1 at 1 M queries: shipped library against lib_prev.so on the same box, + the bit-exactness tests of the search
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "knn or builder or 1m or scale" 2>&1 | tail -2
for rep in 1 2 3; do
for lib in "" "$R/tloam_amd/_variants/lib_prev.so"; do
TLOAM_HIP_LIB=$lib timeout 100 python scripts/k1_time.py 20 2>&1 | tail -1
done
done
