"""K3-only run on the 1 M pre-built set: for rocprofv3 --pmc passes and grid sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sets, x_true, x_eval = synth.make_prebuilt(seed=1, weights="timing")
H = reg.HipRegistration()
for rt in range(3):
    p, a, b, d, w = sets[rt]
    H.set_correspondences(rt, p, a, b, d, w)
H.time_accumulate(x_eval, 10)
us = [H.time_accumulate(x_eval, launches) for _ in range(5)]
alg = 760000 * 72 + 200000 * 88 + 40000 * 64
print("blocks", os.environ.get("TLOAM_K3_BLOCKS", "auto"), "K3 us/launch median %.2f min %.2f" % (np.median(us), min(us)),
      "GB/s %.0f" % (alg / (np.median(us) * 1e-6) / 1e9), "frac %.3f" % (alg / (np.median(us) * 1e-6) / 8e12))
