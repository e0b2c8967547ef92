"""Wall-clock timeline (10 ns units) of the one-launch Solve on a KITTI-cap pre-built set; needs a -DTLOAM_STEP_PROFILE build."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=4500, n_line=1200, n_point=200)
H = reg.HipRegistration()
for rt in range(3): H.set_correspondences(rt, *sets[rt])
for rep in range(3):
    x, st = H.solve(x_eval)
    pb = np.zeros(2048)
    g = H.L.tloam_debug_partials(H.h, pb.ctypes.data_as(C.POINTER(C.c_double)), 2048)
    u = pb.view(np.uint64)[1024:]
    t0 = int(u[0])
    print("rep", rep, "sweeps", st["gn_sweeps"], "evals", st["gn_evaluations"])
    for it in range(st["gn_sweeps"]):
        c = [int(v) - t0 for v in u[8 + it * 8: 8 + it * 8 + 4]]
        p = [int(v) - t0 for v in u[64 + it * 8: 64 + it * 8 + 3]]
        print("  it %d consumer: posted %6d rows_ok %6d step_done %6d published %6d | producer: pose_seen %6d eval_done %6d posted %6d   (x10 ns)" % (it, *c, *p))
