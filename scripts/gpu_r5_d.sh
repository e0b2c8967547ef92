#!/bin/bash
# round 5: GPU suite twice (flake hunt) + default bench (cpu_baseline with the evaluator's persistent pool)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r5d}; mkdir -p $O; cd $R
for i in 1 2; do
  (timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 --durations=5 -o faulthandler_timeout=240 2>&1 | tail -70) > $O/pytest_$i.txt
  grep -E "passed|failed|rror|Timeout" $O/pytest_$i.txt | tail -5
done
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err); python - <<'PY'
import json,sys
j=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r5d/bench.json").read().strip().splitlines()[-1])
cb=j["cpu_baseline"]
print("headline", j["value"], j["ms_per_step"], "roofline", j["roofline"]["frac"], "m1", j["m1_frame"]["ms_per_frame"])
print("cpu", cb["value"], cb["cores"], cb["thread_sweep"], cb.get("reference_thread_shape"))
PY
