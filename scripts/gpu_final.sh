#!/bin/bash
# end of a round: the GPU suite, the smoke entry, then the profile set of the tree as it is
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_tests.sh final
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
bash scripts/gpu_profile_r04.sh r04 2>&1 | tail -25
