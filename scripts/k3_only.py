"""K3-only run on the pre-built set of SURVEY 8(d) config 3 (scale 1: 74.88 MB per sweep, Infinity-Cache resident; scale 4:
299.5 MB per sweep, every byte from HBM): for rocprofv3 --pmc passes and grid sweeps.   usage: k3_only.py [launches] [scale]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = (760000 * scale, 200000 * scale, 40000 * scale)
sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=n[0], n_line=n[1], n_point=n[2], weights="timing")
H = reg.HipRegistration()
for rt in range(3):
    p, a, b, d, w = sets[rt]
    H.set_correspondences(rt, p, a, b, d, w)
H.time_accumulate(x_eval, 10)
us = [H.time_accumulate(x_eval, launches) for _ in range(5)]
alg = n[0] * 72 + n[1] * 88 + n[2] * 64
print("scale", scale, "blocks", os.environ.get("TLOAM_K3_BLOCKS", "auto"), "K3 us/launch median %.2f min %.2f" % (np.median(us), min(us)),
      "GB/s %.0f" % (alg / (np.median(us) * 1e-6) / 1e9), "frac %.3f" % (alg / (np.median(us) * 1e-6) / 8e12))
