#!/bin/bash
# the GPU suite twice in a row on one box (flake hunt), slowest tests listed
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-suite2x}; mkdir -p $O; cd $R
for i in 1 2; do
  (timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --durations=8 -o faulthandler_timeout=240 2>&1 | tail -70) > $O/pytest_$i.txt
  grep -E "passed|failed|rror|Timeout" $O/pytest_$i.txt | tail -5
done
