#!/bin/bash
# debug: which call hangs in test_nan_targets_are_never_neighbours[n_src1-n_tgt1] (hung the full suite in the previous call)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
for t in "tests/test_gpu_parity.py::test_nan_targets_are_never_neighbours" "tests/test_gpu_parity.py::test_non_finite_points_are_never_matched"; do
echo "=== $t"
timeout 100 python -X faulthandler -m pytest -m gpu -q -x "$t" -o faulthandler_timeout=25 2>&1 | tail -45
done
