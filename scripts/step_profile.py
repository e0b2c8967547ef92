import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
for (a,b,c) in [(4500,1200,200),(760000,200000,40000)]:
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=a, n_line=b, n_point=c)
    for ms in (1, 2):
        os.environ["TLOAM_DEBUG_MAX_SWEEPS"] = str(ms)   # read once, when the context is created
        H = reg.HipRegistration()
        for rt in range(3): H.set_correspondences(rt, *sets[rt])
        x, st = H.solve(x_eval)
        buf = np.zeros(400)
        n = H.L.tloam_debug_state(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 400)
        dbg = buf[n-8:n]
        d = np.diff(dbg[:7])
        print(a+b+c, "sweeps", ms, "cycles: reduce %d pre %d dogleg %d step %d plus %d writeback %d total %d" % (d[0], d[1], d[2], d[3], d[4], d[5], dbg[6]-dbg[0]))
