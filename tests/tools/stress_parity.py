"""Randomised parity sweep (not part of the test suite): HIP vs the C oracle over many seeded scenes and
configurations -- final pose, counters, per-kind correspondence index lists and weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tloam_amd import registration as reg, synth
from oracle import binding as ob

def run(seed, n_src, n_tgt, over, scene_kw):
    sc = synth.make_scene(seed=seed, n_src=n_src, n_tgt=n_tgt, **scene_kw)
    H = reg.HipRegistration(reg.default_config(**over)); O = ob.Oracle(ob.make_config(**over))
    H.set_frames(sc.source, sc.target); O.set_frames(sc.source, sc.target)
    res = []
    for rep in range(2):                      # second pass: learned sweep budgets in effect
        rh, Th, sh = H.scan_match(sc.T_pred); ro, To, so = O.scan_match(sc.T_pred)
        D = np.linalg.inv(Th) @ To
        R = D[:3, :3]; w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        dt = np.linalg.norm(D[:3, 3]); dr = np.arctan2(np.linalg.norm(w), 0.5 * (np.trace(R) - 1.0))
        ok = rh == ro and dt < 1e-9 and dr < 1e-9
        for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "n_corr", "outer_iterations"):
            ok = ok and sh[k] == so[k]
        for kind in range(4):
            ih = H.get_correspondences(kind)["idx"]; io = O.get_correspondences(kind)["idx"]
            ok = ok and np.array_equal(ih, io)
            # w = sqrt(nb^2 mu (mu + 1) / c) - mu cancels catastrophically once the GNC mu is large (6 outer iterations:
            # 6e-9 seen on one slot for a 1e-15 difference of the cost) -- the reference's formula, not a defect
            ok = ok and np.allclose(H.get_weights(kind), O.get_weights(kind), rtol=0, atol=1e-7)
        res.append((ok, dt, dr))
    H.close()
    return res

bad = 0; n = 0; t0 = time.time()
cfgs = [dict(), dict(planar_maxnum=120, ground_maxnum=150, edge_maxnum=70, sphere_maxnum=25), dict(factor_num=3), dict(max_iterations=6)]
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    over = cfgs[seed % len(cfgs)]
    kw = dict(outlier_frac=0.1) if seed % 3 == 0 else {}
    if seed % 5 == 0:
        kw["pred_err"] = (0.05, -0.03, 0.02, 0.006, -0.004, 0.008)
    size = (synth.KITTI_SRC, synth.KITTI_TGT) if seed % 8 == 7 else (synth.SMALL_SRC, synth.SMALL_TGT)
    for ok, dt, dr in run(1000 + seed, size[0], size[1], over, kw):
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, over, kw, dt, dr)
print("parity sweep:", n, "runs,", bad, "mismatches, %.1f s" % (time.time() - t0))
