import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
from oracle import binding as ob
np.set_printoptions(linewidth=200, precision=8)
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=7600, n_line=2000, n_point=400)
O = ob.Oracle()
for rt in range(3):
    p, a, b, d, w = sets[rt]; O.set_correspondences(rt, p, a, b, d, w)
Ho, go, co = O.accumulate(x_eval)
print("oracle cost", co, "g", go)
names = [("x",6),("x_cand",6),("T_eval",7),("T_cur",7),("x_cost",1),("x_norm",1),("gmax",1),("mcc",1),("g",6),("H",36),("S",6),
         ("radius",1),("mu",1),("alpha",1),("step_norm",1),("D",6),("grad",6),("gn",6),("U",12),("sg",2),("sB",4)]
for ms in (1, 2, 3):
    os.environ["TLOAM_DEBUG_MAX_SWEEPS"] = str(ms)
    H = reg.HipRegistration()
    for rt in range(3):
        p, a, b, d, w = sets[rt]; H.set_correspondences(rt, p, a, b, d, w)
    x, st = H.solve(x_eval)
    buf = np.zeros(256)
    n = H.L.tloam_debug_state(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 256)
    print("== sweeps", ms, "stats", st["gn_evaluations"], st["gn_iterations"], st["accepted_steps"], "sizeof", n)
    off = 0
    for nm, k in names:
        if nm in ("H", "U"): off += k; continue
        print("  ", nm, buf[off:off+k]); off += k
    ints = buf[off:off+8].view(np.int32)
    print("   ints(reuse,sub1d,phase,iter,invalid,succ,done,evals,iters,acc)", ints[:10])
xo, so = O.solve(x_eval)
print("oracle solve", xo, so)
