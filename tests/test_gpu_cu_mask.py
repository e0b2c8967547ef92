"""-m gpu: the context on a REDUCED chip, and what a context reports about itself (tloam_get_info).

A few launch forms spin on blocks of their own launch -- the one-launch Solve (k_solve_all: sixteen blocks, every one polling all
rows), the single-pass look-back scans of 1 M-class tables, the voxel down-sampling's look-back -- and are only chosen where the
device's CU count says that all their blocks are resident together; their waits are bounded (~1-2 s) all the same and a wait that
runs out moves the context to a form that waits for nothing.  Here the golden frame pairs and the 1 M-frame properties run in a
SUBPROCESS whose queues are confined to 32 of the 256 CUs (ROC_GLOBAL_CU_MASK / HSA_CU_MASK, set before the HIP runtime starts) and
the results are compared with the full chip's, bit for bit where the launch plan is the same.  The test also RECORDS what the
runtime tells a context about such a device (does hipDeviceAttributeMultiprocessorCount see the mask?) and asserts that no call went
through a bounded-wait time-out silently: `fallbacks_taken` / `fallback_events` of tloam_get_info stay 0, and no call took seconds."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tloam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIG = 1 << 30

CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from tloam_amd import registration as reg, synth
out = {"poses": {}, "stats": {}, "idx": {}, "seconds": {}}
def run(name, cfg, scene, frames=2):
    H = reg.HipRegistration(cfg)
    H.set_frames(scene.source, scene.target)
    t0 = time.perf_counter()
    for _ in range(frames):
        rc, T, st = H.scan_match(scene.T_pred)
    out["seconds"][name] = time.perf_counter() - t0
    out["poses"][name] = T.tobytes().hex()
    out["stats"][name] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items() if k != "host_wait_us"}
    out["stats"][name]["rc"] = rc
    out["idx"][name] = [H.get_correspondences(k)["idx"][:2000].tolist() for k in range(4)]
    info = H.info()
    out.setdefault("info", {})[name] = info
    H.close()
for seed in (11, 31, 47):     # KITTI-cap frame pairs: the one-launch Solve (16 blocks spinning on each other's rows)
    run("small%%d" %% seed, reg.default_config(), synth.make_scene(seed=seed))
run("kitti", reg.default_config(), synth.make_scene(seed=5, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT))
big = 1 << 30
run("m1", reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big),
    synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT), frames=2)
# the device submap (voxel down-sampling's look-back) on the reduced chip
from tloam_amd import synth_submap as ss
H = reg.HipRegistration(reg.default_config())
H.submap_init(*ss.frame_clouds(0, 0, n=(4000, 500, 7000, 30000), extent=60.0))
t0 = time.perf_counter()
for f in range(1, 6):
    H.submap_update(ss.frame_pose(f), *ss.frame_clouds(0, f, n=(4000, 500, 2000, 4000), extent=60.0))
out["seconds"]["submap"] = time.perf_counter() - t0
out["submap"] = [np.asarray(H.get_target(k)).tobytes().hex()[:4096] + str(len(H.get_target(k))) for k in range(4)]
out.setdefault("info", {})["submap"] = H.info()
H.close()
out["multiprocessor_count"] = out["info"]["submap"]["device_cus"]    # hipDeviceAttributeMultiprocessorCount as the context read it
print("RESULT " + json.dumps(out))
'''


def _run_child(mask_env):
    env = dict(os.environ)
    env.pop("ROC_GLOBAL_CU_MASK", None)
    env.pop("HSA_CU_MASK", None)
    env.update(mask_env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_context_reports_itself(hip_module):
    H = hip_module.HipRegistration()
    sc = synth.make_scene(seed=11)
    H.set_frames(sc.source, sc.target)
    rc, T, st = H.scan_match(sc.T_pred)
    info = H.info()
    assert rc == 0
    assert info["abi_version"] == 8 and info["device"] == 0 and info["device_cus"] >= 16
    assert info["comm_mode"] == 0 and info["nranks"] == 1 and info["rank"] == 0 and info["loopback"] == 0
    assert info["rccl_comm_count"] == -1 and info["fallbacks_taken"] == 0 and info["fallback_events"] == 0
    assert info["k3_single"] == 1 and 1 <= info["k3_grid"] <= 16 and info["one_launch_solve"] == 1
    # a bounded wait that runs out IS reported (the hand-raised fault word of the look-back scan: tloam_debug_raise_fault)
    hip_module.load_library().tloam_debug_raise_fault(H.h, 0)
    rc2, T2, st2 = H.scan_match(sc.T_pred)      # the frame is re-run by tloam_scan_match itself
    assert rc2 == 0 and T2.tobytes() == T.tobytes()
    info = H.info()
    assert info["fallbacks_taken"] & 1 and info["fallback_events"] == 1, info
    H.close()


def test_gn_iteration_timer_and_read_stream(hip_module):
    """tloam_gn_iter_timer: the period of a GN iteration by the device's clock, for every launch form; tloam_time_read_stream: the
    on-box ceiling of the roofline block."""
    H = hip_module.HipRegistration()
    sc = synth.make_scene(seed=5, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
    H.set_frames(sc.source, sc.target)
    rc, T0, st0 = H.scan_match(sc.T_pred)       # not armed yet: the kernels skip the stamps
    us, n = H.gn_iter_timer(reset=True)         # arms
    assert n == 0
    rc, T, st = H.scan_match(sc.T_pred)
    us, n = H.gn_iter_timer()
    assert rc == 0 and T.tobytes() == T0.tobytes()          # stamping changes nothing
    # periods = sweeps that follow another sweep of the same Solve: at most gn_sweeps - (Solves run), at least one
    assert 1 <= n <= st["gn_sweeps"] - 1, (n, st["gn_sweeps"])
    assert 2.0 < us / n < 60.0, (us, n)                     # one-launch Solve: ~7 us per GN iteration
    H.close()
    # large set, launch-per-iteration form
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=300_001, n_line=77_777, n_point=13_000)
    P = hip_module.HipRegistration()
    for rt in range(3):
        P.set_correspondences(rt, *sets[rt])
    P.gn_iter_timer(reset=True)
    x, ps = P.solve(x_eval)
    us, n = P.gn_iter_timer()
    assert n == ps["gn_sweeps"] - 1 and 5.0 < us / n < 200.0, (us, n, ps)
    hbm = P.time_read_stream(1_200_000_000, 5)
    l3 = P.time_read_stream(75_000_000, 20)
    assert 1500.0 < hbm < 8000.0 and 1500.0 < l3 < 20000.0, (hbm, l3)    # GB/s: below the 8 TB/s data sheet from HBM
    P.close()


@pytest.mark.parametrize("var", ["ROC_GLOBAL_CU_MASK", "HSA_CU_MASK"])
def test_reduced_chip_gives_the_full_chips_results_without_silent_fallbacks(hip_module, var):
    """32 of 256 CUs.  Both spellings of the mask: ROC_GLOBAL_CU_MASK (the HIP runtime applies it to every queue it creates) and
    HSA_CU_MASK (the ROCr layer, per device).  Whether either one is honoured at all for this process is the platform's business
    (an ordinary user may not be allowed to mask CUs); the test asserts what must hold EITHER way and records what it saw."""
    full = _run_child({})
    mask = "0xffffffff" if var == "ROC_GLOBAL_CU_MASK" else "0:0-31"
    red = _run_child({var: mask})
    seen = red["multiprocessor_count"]
    record = {"variable": var, "mask": mask, "multiprocessor_count_full": full["multiprocessor_count"], "multiprocessor_count_masked": seen,
              "attribute_sees_the_mask": seen != full["multiprocessor_count"],
              "seconds_full": full["seconds"], "seconds_masked": red["seconds"],
              "slowdown_m1": red["seconds"]["m1"] / full["seconds"]["m1"]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "cu_mask_%s.json" % var), "w") as f:
        json.dump(record, f, indent=1)
    print("CU-MASK RECORD", json.dumps(record))
    for name in full["poses"]:
        # no bounded wait ran out, on either chip: nothing fell back, no call took the seconds a time-out costs
        for side in (full, red):
            info = side["info"][name]
            assert info["fallbacks_taken"] == 0 and info["fallback_events"] == 0, (name, info)
            assert side["stats"][name]["rc"] == 0
            assert side["seconds"][name] < 30.0, (name, side["seconds"])
        a, b = full["stats"][name], red["stats"][name]
        for key in ("outer_iterations", "gn_evaluations", "gn_iterations", "accepted_steps", "gn_sweeps", "n_corr", "converged_early"):
            assert a[key] == b[key], (name, key)
        assert full["idx"][name] == red["idx"][name], name
        same_plan = full["info"][name]["k3_grid"] == red["info"][name]["k3_grid"]
        if same_plan:      # the same launch plan adds the same rows in the same order: the same bits
            assert full["poses"][name] == red["poses"][name], name
        else:              # a chip that REPORTS fewer CUs gets a smaller streaming grid: another summation order, the same pose
            Ta = np.frombuffer(bytes.fromhex(full["poses"][name])).reshape(4, 4)
            Tb = np.frombuffer(bytes.fromhex(red["poses"][name])).reshape(4, 4)
            assert np.abs(Ta - Tb).max() < 1e-9, name
    assert full["submap"] == red["submap"]
    assert red["info"]["submap"]["fallbacks_taken"] == 0 and red["seconds"]["submap"] < 30.0
