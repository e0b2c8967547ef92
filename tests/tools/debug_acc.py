import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tloam_amd import registration as reg, synth
from oracle import binding as ob
np.set_printoptions(linewidth=200, precision=6)
for (npl, nl, npt) in [(1000, 0, 0), (0, 1000, 0), (0, 0, 1000), (7600, 2000, 400)]:
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=npl, n_line=nl, n_point=npt)
    H = reg.HipRegistration(); O = ob.Oracle()
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        H.set_correspondences(rt, p, a, b, d, w); O.set_correspondences(rt, p, a, b, d, w)
    Hh, gh, ch = H.accumulate(x_eval); Ho, go, co = O.accumulate(x_eval)
    print(npl, nl, npt, "cost", ch, co, "g", np.abs(gh-go).max()/np.abs(go).max(), "H", np.abs(Hh-Ho).max()/np.abs(Ho).max())
    if np.abs(gh-go).max()/np.abs(go).max() > 1e-9:
        print(gh); print(go)
