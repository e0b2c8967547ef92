"""Cycle stamps of the minimiser step (needs a -DTLOAM_STEP_PROFILE build: TLOAM_HIP_LIB=tloam_amd/_variants/lib_stepprof.so).
GnState::dbg (the LAST twelve doubles of the state): [0] kernel / consumer entry, [1] gn_consume entry, [2] decision taken + (H, g)
of the accepted point stored, [3] dogleg data (scaling, Cholesky, solves), [4] step + model cost change, [5] the two Plus + candidate
pose in LDS, [6] state written back.  TLOAM_DEBUG_MAX_SWEEPS = 1: the stamps of the Solve's FIRST step (IterationZero), 2: of its
second (decide + accept + a full step).  KITTI-cap set with TLOAM_NO_PERSISTENT_SOLVE (one launch per iteration: k_sweep_step_small)
and the 1 M set (k_reduce_and_step)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["TLOAM_NO_PERSISTENT_SOLVE"] = "1"
from tloam_amd import registration as reg, synth
for (a, b, c) in [(4500, 1200, 200), (760000, 200000, 40000)]:
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=a, n_line=b, n_point=c)
    for ms in (1, 2, 3):
        os.environ["TLOAM_DEBUG_MAX_SWEEPS"] = str(ms)   # read once, when the context is created
        H = reg.HipRegistration()
        for rt in range(3):
            H.set_correspondences(rt, *sets[rt])
        for rep in range(2):
            x, st = H.solve(x_eval)
        buf = np.zeros(400)
        n = H.L.tloam_debug_state(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 400)
        dbg = buf[n - 12:n]
        d = np.diff(dbg[:7])
        print(a + b + c, "sweeps", ms, "cycles: entry->consume %d  decide+take %d  dogleg %d  step+mcc %d  plus %d  writeback %d  total %d" %
              (d[0], d[1], d[2], d[3], d[4], d[5], dbg[6] - dbg[0]),
              ("| k_reduce_and_step: state in %d, rows folded %d" % (dbg[7] - dbg[0], dbg[8] - dbg[7])) if a + b + c > 100000 else "", flush=True)
        H.close()
