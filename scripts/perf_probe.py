"""Ad-hoc timing probe (not the bench contract): K3 on the 1 M pre-built set, full scan_match on
KITTI-density and 1 M frames."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth

def main():
    t0 = time.time()
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, weights="timing")
    print("gen prebuilt %.1fs" % (time.time() - t0), flush=True)
    H = reg.HipRegistration()
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        H.set_correspondences(rt, p, a, b, d, w)
    for _ in range(3):
        H.time_accumulate(x_eval, 10)
    us = [H.time_accumulate(x_eval, 100) for _ in range(5)]
    alg = 760000 * 72 + 200000 * 88 + 40000 * 64
    print("K3 1M: us/launch", us, "GB/s", alg / (np.median(us) * 1e-6) / 1e9, "frac of 8TB/s", alg / (np.median(us) * 1e-6) / 8e12, flush=True)
    t0 = time.time(); n = 20
    for _ in range(n):
        x, st = H.solve(x_eval)
    dt = (time.time() - t0) / n
    print("solve 1M: ms", dt * 1e3, st, flush=True)
    H.close()
    # KITTI-density frame
    sc = synth.make_scene(seed=0, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
    H = reg.HipRegistration()
    H.set_frames(sc.source, sc.target)
    for _ in range(3):
        rc, T, st = H.scan_match(sc.T_pred)
    t0 = time.time(); n = 50
    for _ in range(n):
        rc, T, st = H.scan_match(sc.T_pred)
    dt = (time.time() - t0) / n
    print("KITTI frame: ms/frame %.3f" % (dt * 1e3), st, flush=True)
    t0 = time.time()
    for _ in range(n):
        H.set_frames(sc.source, sc.target)
    print("KITTI set_frames ms %.3f" % ((time.time() - t0) / n * 1e3), flush=True)
    H.close()
    # 1 M frame
    t0 = time.time()
    sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
    print("gen 1M scene %.1fs" % (time.time() - t0), flush=True)
    cfg = reg.default_config(planar_maxnum=1 << 30, ground_maxnum=1 << 30, edge_maxnum=1 << 30, sphere_maxnum=1 << 30)
    H = reg.HipRegistration(cfg)
    t0 = time.time()
    H.set_frames(sc.source, sc.target)
    print("1M set_frames ms %.1f" % ((time.time() - t0) * 1e3), flush=True)
    for _ in range(2):
        rc, T, st = H.scan_match(sc.T_pred)
    t0 = time.time(); n = 10
    for _ in range(n):
        rc, T, st = H.scan_match(sc.T_pred)
    dt = (time.time() - t0) / n
    print("1M frame: ms/frame %.3f" % (dt * 1e3), st, flush=True)

main()
