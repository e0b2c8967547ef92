#!/bin/bash
# the whole synthetic KITTI-00 sequence -> gpurun_out/seq4540/kitti_sequence_4540.json, after the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/seq4540; mkdir -p $O; cd $R
bash scripts/gpu_tests.sh seq4540
timeout 900 python bench.py --kitti-frames 4540 --no-m1 --no-cpu-baseline > $O/kitti_sequence_4540.json 2> $O/kitti_sequence_4540.err
tail -c 600 $O/kitti_sequence_4540.json
