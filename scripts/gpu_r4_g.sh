#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 400 python -X faulthandler -m pytest -m gpu -q --timeout 200 -o faulthandler_timeout=150 tests/test_gpu_multirank.py::test_callback_exchange_reports_a_peer_that_dies_inside_it tests/test_gpu_parity.py::test_non_finite_points_are_never_matched 2>&1 | tail -15
bash scripts/gpu_profile_r04.sh r04 2>&1 | tail -60
