"""K1 (+K2) launch time on the 1 M frame's state (tloam_time_build), for A/B runs with TLOAM_HIP_LIB variants.
usage: k1_time.py [launches] [n_src] [n_tgt] | k1_time.py [launches] kitti   (KITTI-density frame, the reference's caps)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kitti = len(sys.argv) > 2 and sys.argv[2] == "kitti"
n_src = synth.KITTI_SRC if kitti else (int(sys.argv[2]) if len(sys.argv) > 2 else synth.M1_SRC)
n_tgt = synth.KITTI_TGT if kitti else (int(sys.argv[3]) if len(sys.argv) > 3 else synth.M1_TGT)
big = 1 << 30
H = reg.HipRegistration(reg.default_config() if kitti else
                        reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big))
sc = synth.make_scene(seed=1, n_src=n_src, n_tgt=n_tgt)
H.set_frames(sc.source, sc.target)
rc, T, st = H.scan_match(sc.T_pred)
H.time_build(3)
us = [H.time_build(launches)[0] for _ in range(5)]
print("lib %-14s rc %d n_corr %s  K1 us/launch min %.1f median %.1f" % (
    os.path.basename(os.environ.get("TLOAM_HIP_LIB", "default")), rc, st["n_corr"], min(us), float(np.median(us))), flush=True)
