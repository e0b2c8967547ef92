"""Per-frame time vs what the frame did, over the KITTI-density sequence: which frames are slow and why.
usage: seq_probe.py [frames]"""
import sys, os, time, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from tloam_amd import registration as reg, synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 400
H = reg.HipRegistration(reg.default_config())
rows = []
for f in range(nf):
    sc = bench.kitti_scene(synth, 0, f)
    H.set_frames(sc.source, sc.target); torch.cuda.synchronize()
    t1 = time.perf_counter(); rc, T, st = H.scan_match(sc.T_pred); t2 = time.perf_counter()
    rows.append(((t2 - t1) * 1e3, st["gn_sweeps"], st["gn_evaluations"], st["outer_iterations"], st["accepted_steps"], st["converged_early"], st["host_wait_us"]))
a = np.array(rows[5:], float)
print("frames %d  ms mean %.4f p50 %.4f p90 %.4f p99 %.4f" % (len(a), a[:,0].mean(), np.median(a[:,0]), np.percentile(a[:,0],90), np.percentile(a[:,0],99)))
for lo, hi in ((0, .24), (.24, .27), (.27, .31), (.31, .36), (.36, 9)):
    m = (a[:,0] >= lo) & (a[:,0] < hi)
    if m.any():
        print("ms in [%.2f,%.2f): %4d frames  sweeps %.2f evals %.2f outer %.2f accepted %.2f early %.2f host_wait_us %.0f" % (lo, hi, m.sum(), a[m,1].mean(), a[m,2].mean(), a[m,3].mean(), a[m,4].mean(), a[m,5].mean(), a[m,6].mean()))
for s in sorted(set(a[:,1].astype(int))):
    m = a[:,1] == s
    print("sweeps %2d: %4d frames  ms mean %.4f min %.4f max %.4f" % (s, m.sum(), a[m,0].mean(), a[m,0].min(), a[m,0].max()))
