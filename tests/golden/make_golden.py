#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the INDEPENDENT numpy/scipy
restatement (oracle/oracle_np.py).  Run in the build container:  python tests/golden/make_golden.py

Each case_<name>.npz holds inputs (8 feature clouds, predicted pose, config overrides) and expected
outputs (per outer GNC iteration: correspondence index lists, pose vector, mu, side-channel cost
sums, minimiser counters; final 4x4 pose; final weights).  The reference has no tests or vectors for
this path, so these pin the C oracle and the HIP path to a second, separately
written statement of the same algorithm -- "parity unpinned" with respect to real Ceres/Open3D.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp  # noqa: E402
from tloam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (scene kwargs, config overrides, omega perturbation or None)
    "street_seed0": (dict(seed=0), {}, None),
    "street_seed7_outliers": (dict(seed=7, outlier_frac=0.1), {}, None),
    "caps_bind": (dict(seed=3), dict(planar_maxnum=100, ground_maxnum=150, edge_maxnum=60, sphere_maxnum=30), None),
    "planar_ground_only": (dict(seed=4), dict(factor_num=2), None),
    "no_sphere": (dict(seed=5), dict(factor_num=3), None),
    "small_rotation_perturbed": (dict(seed=6, true_se3=(0.8, 0.05, 0.02, 0.002, 0.003, 0.004)), {}, (0.3, -0.2, 0.9)),
    "large_pred_error": (dict(seed=8, pred_err=(0.08, -0.05, 0.02, 0.01, -0.008, 0.012)), {}, None),
    # KITTI-density frames with the reference's own caps binding (2500 / 2000 / 1200 / 200): the size the metric is quoted on
    "kitti_caps": (dict(seed=24, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT), {}, None),
    "kitti_caps_no_sphere": (dict(seed=25, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT), dict(factor_num=3), None),
    # round 4 (review item 7b): factor_num = 2 (planar + ground builders only, registration.cpp:979-1016) at the KITTI caps, and a
    # KITTI-size frame whose prediction is off by the p99 of the reference's own KITTI-00 trajectory under the constant-velocity
    # model (4.9 cm / 11.5 mrad)
    "kitti_caps_planar_ground_only": (dict(seed=26, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT), dict(factor_num=2), None),
    "kitti_caps_p99_pred_error": (dict(seed=27, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT,
                                       pred_err=(0.035, -0.030, 0.016, 0.007, -0.006, 0.0068)), {}, None),
}
ONLY = [a for a in sys.argv[1:] if a in CASES]   # `make_golden.py kitti_caps ...`: regenerate only these


def run_case(name, scene_kw, cfg_over, omega):
    sc = synth.make_scene(**scene_kw)
    N = onp.NpRegistration(cfg_over)
    for k in range(4):
        N.set_source(k, sc.source.cloud(k))
        N.set_target(k, sc.target.cloud(k))
    trace = []
    T, st = N.scan_match(sc.T_pred, omega=omega, trace=trace)
    out = dict(T_pred=sc.T_pred, T_true=sc.T_true, T_result=T,
               cfg_json=np.array(json.dumps(cfg_over)),
               omega=np.zeros(0) if omega is None else np.asarray(omega, float),
               n_outer=np.int64(len(trace)), converged_early=np.int64(st["converged_early"]),
               final_se3=st["se3"], bad_weights=np.int64(st["bad_weights"]))
    big = sum(len(sc.source.cloud(k)) + len(sc.target.cloud(k)) for k in range(4)) > 20000
    if big:
        # KITTI-size cases: the eight clouds (2.2 MB) are not stored -- the committed generator reproduces them from the
        # scene arguments (numpy's PCG64 stream is stable across versions); their SHA-256 is, so a drifting generator is
        # noticed instead of silently testing something else
        import hashlib
        out["scene_json"] = np.array(json.dumps(scene_kw))
        h = hashlib.sha256()
        for k in range(4):
            h.update(np.ascontiguousarray(sc.source.cloud(k), np.float64).tobytes())
            h.update(np.ascontiguousarray(sc.target.cloud(k), np.float64).tobytes())
        out["clouds_sha256"] = np.array(h.hexdigest())
    for k in range(4):
        if not big:
            out[f"src{k}"] = sc.source.cloud(k)
            out[f"tgt{k}"] = sc.target.cloud(k)
        out[f"weights{k}"] = N.weights[k]
    for it, tr in enumerate(trace):
        out[f"it{it}_x"] = tr["x"]
        out[f"it{it}_mu_next"] = np.float64(tr["mu_next"])
        out[f"it{it}_kind_cost"] = np.asarray(tr["kind_cost"])
        out[f"it{it}_counters"] = np.array([tr["gn_evaluations"], tr["gn_iterations"], tr["accepted_steps"]], np.int64)
        for k in range(4):
            out[f"it{it}_idx{k}"] = tr["idx"][k].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), **out)
    return len(trace), st


def prebuilt_case():
    sets, x_true, x_eval = synth.make_prebuilt(seed=2, n_plane=1500, n_line=400, n_point=100)
    N = onp.NpRegistration()
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        N.set_correspondences(rt, p, a, b, d, w)
    H, g, cost = N.accumulate(x_eval)
    x, st = N.solve(x_eval)
    out = dict(x_eval=x_eval, x_true=x_true, H=H, g=g, cost=np.float64(cost), x_solved=x,
               counters=np.array([st["gn_evaluations"], st["gn_iterations"], st["accepted_steps"]], np.int64))
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        out[f"p{rt}"] = p; out[f"a{rt}"] = a; out[f"w{rt}"] = w
        if b is not None: out[f"b{rt}"] = b
        if d is not None: out[f"d{rt}"] = d
    np.savez_compressed(os.path.join(HERE, "prebuilt_small.npz"), **out)


if __name__ == "__main__":
    for name, (kw, cfg, om) in CASES.items():
        if ONLY and name not in ONLY:
            continue
        n, st = run_case(name, kw, cfg, om)
        print(f"{name}: outer={n} n_corr={st['n_corr']} gn=({st['gn_evaluations']},{st['gn_iterations']},{st['accepted_steps']})")
    if not ONLY:
        prebuilt_case()
    print("written to", HERE)
