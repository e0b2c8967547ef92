"""Phase stamps of the fused GN iteration (k_sweep_step_small) on a KITTI-cap pre-built set; needs a -DTLOAM_STEP_PROFILE build
(python -m tloam_amd.build --variant stepprof --units tl_gn.hip -- -DTLOAM_STEP_PROFILE; TLOAM_HIP_LIB=.../lib_stepprof.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=4500, n_line=1200, n_point=200)
for ms in (1, 2, 3):
    os.environ["TLOAM_DEBUG_MAX_SWEEPS"] = str(ms)   # read once, when the context is created
    H = reg.HipRegistration()
    for rt in range(3): H.set_correspondences(rt, *sets[rt])
    x, st = H.solve(x_eval)
    buf = np.zeros(400)
    n = H.L.tloam_debug_state(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 400)
    dbg = buf[n-12:n]
    d = np.diff(dbg[:7])
    print("sweeps", ms, "cycles (consumer block): entry->pose %d | evaluate %d | reduce %d | post %d | wait+fold %d decide %d dogleg %d model %d plus %d write-back %d | entry->end %d" % (
        dbg[9]-dbg[8], dbg[10]-dbg[9], dbg[7]-dbg[10], dbg[0]-dbg[7], d[0], d[1], d[2], d[3], d[4], d[5], dbg[6]-dbg[8]))
