#!/bin/bash
# bench.py's N > 1 path on a ONE-GPU box (TLOAM_BENCH_ONE_DEVICE=1: every rank on device 0, launcher collectives over gloo),
# launched exactly as the driver launches it.  NOT a scaling measurement (the ranks share one GPU): it shows the replica
# headline aggregating over the ranks and the sharded 1 M frame (BASELINE.json configs[3]) running end to end with the mailbox.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r05}; P=$R/gpurun_out/$TAG/profiles; mkdir -p $P; cd $R
for n in 2 4 8; do
  TLOAM_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
    --master-port $((29500 + n)) bench.py --gpus $n --steps 20 --warmup 5 --m1-steps 4 2> $P/one_device_$n.err | grep '^{' | tail -1 > $P/${TAG}_bench_gpus${n}_one_device.json
  python - <<PY
import json
d = json.loads(open("$P/${TAG}_bench_gpus${n}_one_device.json").read())
sh = d["sharded_1m"]
print("n", d["n_gpus"], "value", d["value"], "ms_per_step", d["ms_per_step"], "| sharded_1m mailbox ms/frame", sh["mailbox"]["ms_per_frame"],
      "per_sweep_us", sh["mailbox"]["per_sweep_us"], "rccl:", sh["rccl"].get("error", sh["rccl"]), "pose err", sh["pose_err_vs_truth_m"])
PY
done
