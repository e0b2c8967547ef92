"""K3 launched back to back against K3 inside a Solve (sweep | fold + step | sweep ...), same pre-built set, same context:
what the alternation with the one-block step kernel costs the sweep.  usage: python scripts/k3_in_solve.py [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=760000 * scale, n_line=200000 * scale, n_point=40000 * scale)
H = reg.HipRegistration()
for rt in range(3):
    H.set_correspondences(rt, *sets[rt])
for rep in range(3):
    b2b = H.time_accumulate(x_eval, 50)
    H.k3_timer(reset=True)
    H.gn_iter_timer(reset=True)
    for _ in range(5):
        x, st = H.solve(x_eval)
    us, n, _ = H.k3_timer()
    ius, inn = H.gn_iter_timer()
    print("scale %d: K3 back to back %.2f us | in a Solve %.2f us (%d timed launches) | GN iteration %.2f us (%d periods), sweeps per Solve %d, wide %d"
          % (scale, b2b, us / max(n, 1), n, ius / max(inn, 1), inn, st["gn_sweeps"], H.info()["k3_wide"]))
H.close()
