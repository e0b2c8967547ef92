#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -X faulthandler -m pytest -m gpu -q --timeout 300 tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sequence.py 2>&1 | tail -4
bash scripts/gpu_ab_libs.sh
