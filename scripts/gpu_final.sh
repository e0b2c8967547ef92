#!/bin/bash
# end of a round: the GPU suite, the smoke entry, the profile set of the tree as it is, the whole synthetic KITTI-00 sequence
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash scripts/gpu_tests.sh final
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
bash scripts/gpu_profile_r04.sh r04 2>&1 | tail -25
O=$R/gpurun_out/seq4540; mkdir -p $O
timeout 900 python bench.py --kitti-frames 4540 --no-m1 --no-cpu-baseline > $O/kitti_sequence_4540.json 2> $O/kitti_sequence_4540.err
tail -c 300 $O/kitti_sequence_4540.json
