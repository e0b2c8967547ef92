"""Distribution of the K3 launch time on the 1 M pre-built set: <batches> batches of <launches> back-to-back launches.
usage: k3_stats.py <launches> <batches> [blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
launches = int(sys.argv[1]); batches = int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] != "auto": os.environ["TLOAM_K3_BLOCKS"] = sys.argv[3]
sets, x_true, x_eval = synth.make_prebuilt(seed=1, weights="timing")
H = reg.HipRegistration()
for rt in range(3):
    H.set_correspondences(rt, *sets[rt])
H.time_accumulate(x_eval, 10)
us = np.array([H.time_accumulate(x_eval, launches) for _ in range(batches)])
alg = 760000 * 72 + 200000 * 88 + 40000 * 64
print("lib %-16s blocks %-5s us/launch mean %.2f p10 %.2f p50 %.2f p90 %.2f  frac(p50) %.3f" % (
    os.path.basename(os.environ.get("TLOAM_HIP_LIB", "default")), sys.argv[3] if len(sys.argv) > 3 else "auto",
    us.mean(), np.percentile(us, 10), np.median(us), np.percentile(us, 90), alg / (np.median(us) * 1e-6) / 8e12), flush=True)
