"""Timing probe: PCA feature extraction of a ~100 k-point scan-like cloud, device vs the C restatement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tloam_amd import registration as reg, synth_submap as ss
from oracle import binding as ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = ss.feature_cloud(0, n=n)
H = reg.HipRegistration()
for _ in range(2):
    H.extract_planar_sphere(p)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    L = H.extract_planar_sphere(p)
tg = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); Lo = ob.extract_planar_sphere(p); tc = time.perf_counter() - t0
print("points", len(p), "lists", [len(x) for x in L], "bit-exact vs oracle:", all(np.array_equal(a, b) for a, b in zip(L, Lo)))
print("GPU extract_planar_sphere ms %.3f (incl. upload + list download)   CPU restatement (1 thread) ms %.1f" % (tg * 1e3, tc * 1e3))
