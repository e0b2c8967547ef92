#!/bin/bash
# round 6, first GPU call: the new tests first (fast verdict), the whole suite, the default bench line, the N > 1 line on one device
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
(timeout 900 python -X faulthandler -m pytest tests/test_gpu_cu_mask.py tests/test_gpu_multirank.py tests/test_adapter_compile.py -m gpu -q --timeout 600 -x 2>&1 | tail -40) > $O/pytest_new.txt
tail -15 $O/pytest_new.txt
(timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -o faulthandler_timeout=240 --deselect tests/test_gpu_cu_mask.py --deselect tests/test_gpu_multirank.py 2>&1 | tail -30) > $O/pytest_rest.txt
tail -8 $O/pytest_rest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
TLOAM_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --m1-steps 4 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; tail -3 $O/bench_gpus2.err
python - <<PY
import json
for f in ("bench_default", "bench_gpus2"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    flat = {k: v for k, v in d.items() if not isinstance(v, (dict, list))}
    print(f, json.dumps(flat))
    r = d.get("roofline") or {}
    print(f, "roofline", json.dumps({k: v for k, v in r.items() if not isinstance(v, (dict, list)) and k not in ("note", "workload", "working_set", "launch_timing", "read_stream_note", "gn_iteration_note")}))
    if "shard_size_iterations" in d: print(json.dumps(d["shard_size_iterations"]))
    if "cpu_baseline" in d: print({k: d["cpu_baseline"].get(k) for k in ("value", "cores", "ms_per_frame", "gpu_ms_per_frame", "sweeps_executed_per_sec", "evaluations_equal_gpu", "error")})
    if "m1_frame" in d: print(d["m1_frame"])
PY
