"""Aggregate throughput of N independent frame streams sharing ONE GPU from one process (one context + one host thread
per stream; ctypes releases the GIL during the C call).  usage: multi_stream.py [frames]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from tloam_amd import registration as reg, synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sc = bench.kitti_frame(synth, 0, 105)
for N in (1, 2, 3, 4, 6, 8):
    Hs = [reg.HipRegistration(reg.default_config()) for _ in range(N)]
    for H in Hs:
        H.set_frames(sc.source, sc.target)
        for _ in range(10): H.scan_match(sc.T_pred)
    its = [0] * N
    start = threading.Barrier(N + 1)
    def run(i):
        start.wait()
        for _ in range(frames):
            rc, T, st = Hs[i].scan_match(sc.T_pred)
            its[i] += st["gn_sweeps"]
    th = [threading.Thread(target=run, args=(i,)) for i in range(N)]
    for t in th: t.start()
    start.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("streams %d: %.1f frames/s aggregate, %.0f GN it/s, %.3f ms per frame per stream" % (N, N * frames / dt, sum(its) / dt, dt / frames * 1e3), flush=True)
    for H in Hs: H.close()
