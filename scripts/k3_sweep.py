"""K3 on the 1 M pre-built set over a list of grid sizes, for the library selected with
TLOAM_HIP_LIB (tuning builds under tloam_amd/_variants/).  usage: k3_sweep.py <launches> <blocks,blocks,...>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
blocks = sys.argv[2].split(",") if len(sys.argv) > 2 else ["auto"]
sets, x_true, x_eval = synth.make_prebuilt(seed=1, weights="timing")
alg = 760000 * 72 + 200000 * 88 + 40000 * 64
ref = None
for b in blocks:
    if b == "auto": os.environ.pop("TLOAM_K3_BLOCKS", None)
    else: os.environ["TLOAM_K3_BLOCKS"] = b
    H = reg.HipRegistration()
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    Hm, g, cost = H.accumulate(x_eval)
    if ref is None: ref = (Hm, g, cost)
    err = max(np.abs(Hm - ref[0]).max() / np.abs(ref[0]).max(), abs(cost - ref[2]) / abs(ref[2]))
    H.time_accumulate(x_eval, 10)
    us = [H.time_accumulate(x_eval, launches) for _ in range(5)]
    H.close()
    print("lib %-12s blocks %-5s K3 us/launch median %.2f min %.2f  frac %.3f  relerr_vs_first %.1e" % (
        os.path.basename(os.environ.get("TLOAM_HIP_LIB", "default")), b, np.median(us), min(us),
        alg / (np.median(us) * 1e-6) / 8e12, err), flush=True)
