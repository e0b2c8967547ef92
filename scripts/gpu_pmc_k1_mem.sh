#!/bin/bash
# memory-side counters of the 1 M-query correspondence search (separate passes; kernel-trace only alongside)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-k1mem}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
# (two further sets -- TA_* busy / stall cycles and TCP_TAGRAM*_REQ -- hung rocprofv3 on this image until the time-out: left out)
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
i=$((i+1))
timeout 100 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p -- python $R/scripts/k1_time.py 4 > /dev/null 2> $O/p$i.err
done
cd $R
python scripts/pmc_summary.py "k_build_sorted<1>" $O/pmc_k1_memory_side.json $(for j in 1 2; do find $O/p$j -name "*.db" | head -1; done) > /dev/null
rm -rf $O/p1 $O/p2
python - <<PY
import json
d = json.load(open("$O/pmc_k1_memory_side.json"))
for k, v in d["counters"].items(): print("%-44s %14.0f  (%d launches)" % (k, v["mean_working"], v["launches"]))
PY
