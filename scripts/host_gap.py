"""What the CALLER adds to a KITTI-density frame: the headline loop of bench.py (tloam_frame_select + tloam_scan_match through the
Python class: numpy views, a stats dict per frame) against the same calls made straight through ctypes with every argument object
made beforehand, on the same 200 resident frames, interleaved.  The difference is the harness, not the library.
usage: python scripts/host_gap.py [frames] [reps]"""
import ctypes as C
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from tloam_amd import registration as reg, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = reg.HipRegistration(reg.default_config())
scenes = [bench.kitti_scene(synth, 0, f) for f in range(frames)]
for f, sc in enumerate(scenes):
    H.set_frames(sc.source, sc.target)
    H.frame_stash(f)
L, h = H.L, H.h
preds = []
for sc in scenes:
    b = (C.c_double * 16)()
    np.frombuffer(b).reshape(4, 4).T[...] = sc.T_pred
    preds.append(b)
res = (C.c_double * 16)()
st = reg.Stats()
st_ref = C.byref(st)


def python_loop():
    sweeps = wait = 0
    t0 = time.perf_counter()
    for i in range(frames):
        H.frame_select(i)
        rc, T, s = H.scan_match(scenes[i].T_pred)
        sweeps += s["gn_sweeps"]
        wait += s["host_wait_us"]
    return (time.perf_counter() - t0) / frames * 1e6, sweeps, wait / frames


def ctypes_loop():
    sweeps = wait = 0
    sel, sm = L.tloam_frame_select, L.tloam_scan_match
    t0 = time.perf_counter()
    for i in range(frames):
        sel(h, i)
        sm(h, preds[i], None, res, None, 0, st_ref)
        sweeps += st.gn_sweeps
        wait += st.host_wait_us
    return (time.perf_counter() - t0) / frames * 1e6, sweeps, wait / frames


for _ in range(2):
    python_loop(), ctypes_loop()
for rep in range(reps):
    a, b = python_loop(), ctypes_loop()
    print("rep %d: python class %.2f us/frame (host wait %.1f)   bare ctypes %.2f us/frame (host wait %.1f)   sweeps %d / %d"
          % (rep, a[0], a[2], b[0], b[2], a[1], b[1]))
H.close()
