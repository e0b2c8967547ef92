"""Soak: the device-driven loop against the host-driven loop over N randomly drawn cases x 3 frames (not part of the suite).
usage: soak_device_loop.py [seed] [cases]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from tloam_amd import registration as reg, synth
import test_gpu_parity as tp
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 600
census = {"plateau": 0, "max_iter": 0, "frames": 0, "rc7": 0}
for case in range(N):
    over = dict(max_iterations=int(rng.integers(1, 8)), cost_threshold=float(rng.choice([1e-3, 2e-2, 0.5, 5.0, 50.0])),
                gnc_factor=float(rng.choice([0.5, 1.0, 1.4, 2.0])))
    if rng.random() < 0.4:
        over.update(planar_maxnum=int(rng.integers(20, 400)), ground_maxnum=int(rng.integers(20, 400)),
                    edge_maxnum=int(rng.integers(15, 200)), sphere_maxnum=int(rng.integers(5, 60)))
    scale = float(rng.choice([0.3, 1.0, 4.0, 12.0]))
    pred_err = tuple(scale * np.array((0.012, -0.008, 0.004, 0.0015, -0.001, 0.002)) * rng.normal(1.0, 0.3, 6))
    sc = synth.make_scene(seed=5000 + case, noise=float(rng.choice([0.0, 0.01, 0.02, 0.05])), pred_err=pred_err,
                          outlier_frac=float(rng.choice([0.0, 0.05, 0.2, 0.4])),
                          n_src=synth.SMALL_SRC if case % 4 else synth.KITTI_SRC, n_tgt=synth.SMALL_TGT if case % 4 else synth.KITTI_TGT)
    os.environ.pop("TLOAM_NO_DEVICE_LOOP", None)
    H1 = reg.HipRegistration(reg.default_config(**over)); H1.set_frames(sc.source, sc.target)
    os.environ["TLOAM_NO_DEVICE_LOOP"] = "1"
    H2 = reg.HipRegistration(reg.default_config(**over)); H2.set_frames(sc.source, sc.target)
    for frame in range(3):
        rc1, T1, st1 = H1.scan_match(sc.T_pred); rc2, T2, st2 = H2.scan_match(sc.T_pred)
        assert rc1 == rc2 and rc1 in (0, -7), (case, frame, rc1, rc2)
        tp._assert_same_frame(tp._frame_fingerprint(H1, T1, st1), tp._frame_fingerprint(H2, T2, st2))
        census["frames"] += 1; census["rc7"] += rc1 == -7
        census["plateau"] += int(st1["converged_early"]); census["max_iter"] += int(not st1["converged_early"])
    H1.close(); H2.close()
print("soak ok", census)
