#!/bin/bash
# tests/tools/fuzz_call_order.py over a range of seeds, one process per seed (the failures it hunts end the process); a failing
# seed leaves its call log under gpurun_out/fuzz/ -- shrink it with tests/tools/fuzz_reduce.py <steps> <seed>
# usage: scripts/gpu_fuzz.sh [first_seed=0] [last_seed=15] [steps=3000]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/fuzz
for s in $(seq ${1:-0} ${2:-15}); do
  FUZZ_LOG=1 FUZZ_SYNC=1 timeout 300 python -X faulthandler tests/tools/fuzz_call_order.py ${3:-3000} $s > gpurun_out/fuzz/out_$s.txt 2> gpurun_out/fuzz/err_$s.txt
  rc=$?
  if [ $rc -ne 0 ]; then echo "seed $s rc $rc"; grep "fault\|Error\|assert" gpurun_out/fuzz/err_$s.txt | head -3; grep "^call" gpurun_out/fuzz/err_$s.txt | tail -3
  else echo "seed $s ok"; rm gpurun_out/fuzz/err_$s.txt; fi
done
