#!/bin/bash
# the GPU suite with a per-test time-out (a test that hangs costs 5 minutes, not the call's limit) and a traceback of where
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-tests}; mkdir -p $O; cd $R
(timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -o faulthandler_timeout=240 ${2:-} 2>&1 | tail -80) > $O/pytest.txt
grep -E "passed|failed|rror|Timeout" $O/pytest.txt | tail -12
