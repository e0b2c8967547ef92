"""bench.py's deadlines (CPU): a side figure that never returns -- the CPU port's thread pool on a many-core host, a collective
of the sharded 1 M frame on the driver's first multi-GPU run -- must not cost the run its JSON line."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_within_reports_a_call_that_does_not_return():
    import bench
    t0 = time.time()
    assert bench.within(lambda: time.sleep(30.0), 0.2) is False
    assert time.time() - t0 < 5.0
    assert bench.within(lambda: None, 5.0) is True
    with pytest.raises(ZeroDivisionError):
        bench.within(lambda: 1 / 0, 5.0)


def test_sharded_summary_reports_what_had_finished():
    import bench
    mailbox = {"exchange": "mailbox", "frames": 4, "ms_per_frame": 0.9, "gn_iters_per_sec": 1.0, "gn_iters_per_frame": 14.0,
               "n_corr": [1, 2, 3, 4], "pose_err_vs_truth_m": 1e-4, "per_sweep_us": {"sweep_alone": 3.0}}
    # the deadline passed while the second exchange was running
    cut = bench.sharded_summary({"workload": "w", "n_gpus": 8, "modes": ["mailbox", "rccl"], "mailbox": mailbox}, False, 300.0)
    assert cut["fastest_exchange"] == "mailbox" and cut["ms_per_frame"] == 0.9
    assert "did not finish" in cut["rccl"]["error"] and "cut off" in cut["note"]
    # ... before anything had finished
    none = bench.sharded_summary({}, False, 300.0)
    assert "ms_per_frame" not in none and "mailbox: did not finish" in none["error"] and "rccl: did not finish" in none["error"]
    # both ran: the faster one on top, nothing else added
    both = bench.sharded_summary({"modes": ["mailbox", "rccl"], "mailbox": mailbox, "rccl": dict(mailbox, exchange="rccl", ms_per_frame=1.2)},
                                 True, 300.0)
    assert both["fastest_exchange"] == "mailbox" and "note" not in both and "error" not in both
    json.dumps(both)


def test_guarded_side_figure_prints_the_line_and_leaves():
    """`guarded` around a call that hangs: the line goes out with the reason in place of the figure, exit code 0."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 1.0}\n"
            "out['cpu_baseline'] = bench.guarded(lambda: time.sleep(60.0), 0.3, 'cpu_baseline', out, 0)\n"
            "print('not reached')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and "did not finish" in d["cpu_baseline"]["error"]
