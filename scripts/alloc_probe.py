"""Which frames of the KITTI-density sequence still (re)allocate inside scan_match, and what each of the first frames costs.
usage: TLOAM_DEBUG_ALLOC=1 python scripts/alloc_probe.py [frames]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import bench
from tloam_amd import registration as reg, synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H = reg.HipRegistration(reg.default_config())
for f in range(nf):
    sc = bench.kitti_scene(synth, 0, f)
    H.set_frames(sc.source, sc.target); H.frame_stash(f)
for f in range(nf):
    sc = bench.kitti_scene(synth, 0, f)
    H.frame_select(f)
    sys.stderr.flush()
    t1 = time.perf_counter(); rc, T, st = H.scan_match(sc.T_pred); t2 = time.perf_counter()
    print("frame %3d  %.4f ms  sweeps %2d evals %2d accepted %d wait_us %d" % (f, (t2 - t1) * 1e3, st["gn_sweeps"], st["gn_evaluations"], st["accepted_steps"], st["host_wait_us"]), flush=True)
