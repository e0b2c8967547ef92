"""-m gpu: the line `python bench.py` prints at N = 1, as the driver reads it: stdout is that line and nothing else, it carries the
contract's keys, and the quantities the review is on are FLAT scalars of `roofline` and of the top level (the driver's parser
drops nested objects)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_is_the_only_thing_on_stdout_and_carries_the_flat_scalars(hip_module):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--m1-steps", "3", "--kitti-frames", "12",
           "--loop-frames", "12", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    # (the loop-back RCCL set-up of the shard-size figures makes RCCL print its version banner through C stdio -- at EXIT, i.e.
    #  behind the line, until bench.py kept descriptor 1 for the line alone)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-1500:]
    d = json.loads(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):     # (cpu_baseline: skipped here, tests/test_gpu_configs.py runs the port)
        assert key in d, key
    assert d["metric"] == "gauss_newton_iters_per_sec" and d["n_gpus"] == 1 and d["steps"] == 20 and d["dtype"] == "f64"
    assert d["value"] > 0 and abs(d["value"] - d["config"]["gn_iters_per_frame"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["avg_launch_us"] * 1e-3) / rf["achieved"] < 1e-3
    for key in ("config3_frac", "config3_avg_launch_us", "read_stream_GBps", "frac_of_read_stream", "in_frame_frac",
                "in_frame_avg_launch_us", "gn_iteration_us", "gn_iteration_frac"):
        assert isinstance(rf[key], float) and rf[key] > 0, key
    assert 0.3 < rf["frac"] < 1.0 and 0.3 < rf["config3_frac"] < 1.0 and rf["frac_of_read_stream"] < 1.05
    # one GN iteration of the 1 M frame holds a sweep: its fraction cannot exceed the sweep's own
    assert rf["gn_iteration_frac"] < rf["in_frame_frac"] < 1.0
    shard = [k for k in d if k.startswith("shard_iteration_us")]
    assert shard and all(not isinstance(d[k], (dict, list)) for k in shard), shard
