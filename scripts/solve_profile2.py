"""Wall-clock timeline (10 ns units) of the one-launch Solve on a KITTI-cap pre-built set; needs a -DTLOAM_STEP_PROFILE build
(TLOAM_HIP_LIB=tloam_amd/_variants/lib_stepprof.so).  k_solve_all (default) keeps its stamps at word 2048 of the row buffer,
Stamps of the lead block's stepper wave (c) and of one other wave (p)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
base = 2048
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=4500, n_line=1200, n_point=200)
H = reg.HipRegistration()
for rt in range(3): H.set_correspondences(rt, *sets[rt])
for rep in range(4):
    x, st = H.solve(x_eval)
    pb = np.zeros(4096)
    H.L.tloam_debug_partials(H.h, pb.ctypes.data_as(C.POINTER(C.c_double)), 4096)
    u = pb.view(np.uint64)[base:]
    t0 = int(u[0])
    print("rep", rep, "kernel", "k_solve_all", "sweeps", st["gn_sweeps"], "evals", st["gn_evaluations"])
    prev = None
    for it in range(st["gn_sweeps"]):
        c = [int(v) - t0 for v in u[8 + it * 8: 8 + it * 8 + 4]]
        p = [int(v) - t0 for v in u[64 + it * 8: 64 + it * 8 + 3]]
        per = "" if prev is None else "  (iteration: %d)" % (c[3] - prev)
        prev = c[3]
        print("  it %d stepper: row_posted %6d rows_ok %6d step_done %6d pose_out %6d | other wave: pose_seen %6d eval_done %6d posted %6d   (x10 ns)%s"
              % (it, *c, *p, per))
