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
               "n_corr": [1, 2, 3, 4], "pose_err_vs_truth_m": 1e-4, "per_sweep_us": {"sweep_alone": 3.0, "exchange_adds": 2.5},
               "verified": True, "pose_delta_vs_one_rank": {"dt_m": 1e-15, "dR_rad": 2e-16}, "ranks_bit_identical": True,
               "counters_equal_one_rank": True, "gn_iteration_us": 9.5}
    # the deadline passed while the second exchange was running
    cut = bench.sharded_summary({"workload": "w", "n_gpus": 8, "modes": ["mailbox", "rccl"], "mailbox": mailbox}, False, 300.0)
    assert cut["fastest_exchange"] == "mailbox" and cut["ms_per_frame"] == 0.9
    assert "did not finish" in cut["rccl"]["error"] and "cut off" in cut["note"]
    # ... before anything had finished
    none = bench.sharded_summary({}, False, 300.0)
    assert "ms_per_frame" not in none and "mailbox: did not finish" in none["error"] and "rccl: did not finish" in none["error"]
    flat = bench.sharded_flat(none, 8)
    assert flat["sharded_1m_verified"] is False and flat["sharded_1m_ms_per_frame"] is None and flat["sharded_1m_predicted_speedup"] == 1.54
    # both ran: the faster one on top, nothing else added
    both = bench.sharded_summary({"modes": ["mailbox", "rccl"], "mailbox": mailbox, "rccl": dict(mailbox, exchange="rccl", ms_per_frame=1.2)},
                                 True, 300.0)
    assert both["fastest_exchange"] == "mailbox" and "note" not in both and "error" not in both
    json.dumps(both)
    # a form that FAILED its self-check (pose delta against the one-rank solve, ranks not bit-identical ...) never becomes the
    # figure of the line, however fast it was
    wrong = dict(mailbox, ms_per_frame=0.1, verified=False, error="verification failed: pose delta 3.0e-04 m")
    mixed = bench.sharded_summary({"modes": ["mailbox", "rccl"], "mailbox": wrong, "rccl": dict(mailbox, exchange="rccl", ms_per_frame=1.2,
                                                                                              rccl_comm_count=8)}, True, 300.0)
    assert mixed["fastest_exchange"] == "rccl" and mixed["ms_per_frame"] == 1.2
    mixed["speedup_vs_one_gpu_frame"] = 1.1
    flat = bench.sharded_flat(mixed, 8)
    assert flat["sharded_1m_ms_per_frame"] == 1.2 and flat["sharded_1m_speedup"] == 1.1 and flat["rccl_nranks"] == 8
    assert flat["sharded_1m_verified"] is True and flat["sharded_1m_pose_delta"] == 1e-15 and flat["sharded_1m_exchange_adds_us"] == 2.5
    assert flat["sharded_1m_mailbox_verified"] is False and flat["sharded_1m_rccl_ms_per_frame"] == 1.2
    assert all(not isinstance(v, (dict, list)) for v in flat.values())      # scalars only: what the driver's parser keeps
    none_ok = bench.sharded_summary({"modes": ["mailbox"], "mailbox": wrong}, True, 300.0)
    assert "ms_per_frame" not in none_ok and "verification failed" in none_ok["error"]


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
