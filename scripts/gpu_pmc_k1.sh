#!/bin/bash
# Instruction-mix counters of K1 on the 1 M frame (separate --pmc passes, kernel trace only).
# usage: scripts/gpu_pmc_k1.sh <tag>
TAG=${1:-k1pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $O/p$i -o p -- python $R/bench.py --workload m1 --steps 2 --warmup 1 --no-cpu-baseline --no-kitti > /dev/null 2> $O/p$i.err
done
cd $R
python scripts/pmc_summary.py "k_build_sorted" $O/k1.json $(find $O -name "*.db") | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in sorted(d['counters'].items()): print(k, round(v['mean_working'], 1), v['working_launches'])"
tail -3 $O/p*.err | cut -c1-300
