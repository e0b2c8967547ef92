#!/bin/bash
# round 5: the GPU suite (with the new third-party pins) + the default bench line, one call
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5a}; mkdir -p $O; cd $R
(timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -o faulthandler_timeout=240 2>&1 | tail -60) > $O/pytest.txt
grep -E "passed|failed|rror|Timeout" $O/pytest.txt | tail -12
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err); tail -c 1500 $O/bench.json
