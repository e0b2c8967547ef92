"""-m gpu: the C ABI under random call orders and bad arguments (tests/tools/fuzz_call_order.py), a short standing run: a status
code for every call, never a crash (the tool runs in a process of its own: the failures it hunts end the process), a hang or a
context that no longer solves a clean frame like the CPU restatement does.  Round 5 found five defects this way (tests/tools/README.md)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_call_orders_end_in_status_codes(hip_module, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_call_order.py"), "2000", str(seed)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, FUZZ_SYNC="1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "fuzz ok: 2000 calls" in r.stdout
    # every entry point the tool drives answered at least once with something other than OK, and scan_match also with OK
    lines = {l.split()[0]: l for l in r.stdout.splitlines() if l.startswith("  ")}
    assert "TLOAM_E_BAD_POSE" in lines["scan_match"] and "'OK'" in lines["scan_match"]
    assert "TLOAM_E_BAD_POSE" in lines["submap_update"]


@pytest.mark.parametrize("seed", [4, 5])
def test_dirty_frames_agree_with_the_restatement(hip_module, seed):
    """A short standing run of tests/tools/stress_dirty.py (lattice-quantised clouds, duplicates, NaN / infinite points, ten-point
    clouds, caps of 0 / 1, lifted thresholds): every frame's status, pose, counters, index lists and weights against the CPU
    restatement.  A frame may part only where its scaled normal equations are singular (eigenvalue ratio below 1e-6: the tool's
    bar) and is then listed, not failed; anything else ends the tool with a non-zero code."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "stress_dirty.py"), "250", str(seed)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "dirty sweep ok: 250 frames" in r.stdout
    assert "FAILED" not in r.stdout
    parted = sum(1 for l in r.stdout.splitlines() if "decisions parted" in l)
    assert parted <= 3, r.stdout[-3000:]      # (12 000 frames over four seeds: nine, all rank-deficient)


def test_random_submap_sequences_and_feature_clouds_bit_for_bit(hip_module):
    """A short standing run of tests/tools/stress_rows.py: the device submap and the PCA features (next-2)
    over sizes, contents and configurations the fixed cases do not enumerate (0 / 1 / thousands of points, duplicates, points on
    voxel and crop boundaries, non-finite points, every K / radius / voxel size) -- bit for bit against the restatement, call by
    call, status by status."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "stress_rows.py"), "200", "11"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "rows sweep ok: 200 submap sequences + 200 feature clouds" in r.stdout
