"""Cost of the fused hand-over (ticket + last-block fold) of the sweep: K3<*, true> against K3<*, false>, back to back,
on a KITTI-cap set and on the 1 M set (single rank: nothing is exchanged)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
for (a, b, c) in [(4500, 1200, 200), (760000, 200000, 40000)]:
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=a, n_line=b, n_point=c)
    H = reg.HipRegistration()
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    Hm, g, cost = H.accumulate(x_eval)
    for rep in range(2):
        plain = np.median([H.time_accumulate(x_eval, 50) for _ in range(5)])
        fused = np.median([H.time_sharded_sweep(x_eval, 50, False) for _ in range(5)])
        print(a + b + c, "K3 plain %.2f us, fused (ticket + fold) %.2f us" % (plain, fused), flush=True)
    H.close()
