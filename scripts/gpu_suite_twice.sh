#!/bin/bash
# the GPU suite twice in a row on one box (flakiness check before the round ends), then the smoke entry
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_tests.sh suite1
bash scripts/gpu_tests.sh suite2
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3
