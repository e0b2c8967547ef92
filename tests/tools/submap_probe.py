"""Timing probe: device submap update at KITTI-like sizes vs the CPU restatement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tloam_amd import registration as reg, synth_submap as ss
from oracle import binding as ob
N = (4000, 500, 2000, 4000)
H = reg.HipRegistration(); S = ob.OracleSubmap()
cl = ss.frame_clouds(0, 0, n=(4000, 500, 7000, 30000), extent=60.0)
H.submap_init(*cl); S.init(*cl)
tg, tc = [], []
for f in range(1, 60):
    cl = ss.frame_clouds(0, f, n=N, extent=60.0)
    T = ss.frame_pose(f)
    t0 = time.perf_counter(); H.submap_update(T, *cl); tg.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); S.update(T, *cl); tc.append(time.perf_counter() - t0)
sz = [len(H.get_target(k)) for k in range(4)]
ok = all(np.array_equal(H.get_target(k), S.get(k)) for k in range(4))
print("submap sizes", sz, "bit-exact vs oracle:", ok)
print("GPU update ms: mean %.3f p50 %.3f (last 30)" % (np.mean(tg[-30:]) * 1e3, np.median(tg[-30:]) * 1e3))
print("CPU update ms: mean %.3f" % (np.mean(tc[-30:]) * 1e3))
