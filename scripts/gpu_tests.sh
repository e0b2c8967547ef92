#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-tests}; mkdir -p $O; cd $R
(timeout 1500 python -m pytest tests -m gpu -q ${2:-} 2>&1 | tail -60) > $O/pytest.txt; grep -E "passed|failed|rror" $O/pytest.txt | tail -12
