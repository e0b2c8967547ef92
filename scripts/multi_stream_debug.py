"""N frame streams on one GPU; prints the status and message of every failing call.  usage: multi_stream_debug.py [streams] [frames]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tloam_amd import registration as reg, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sc = bench.kitti_scene(synth, 0, 105)
Hs = [reg.HipRegistration(reg.default_config()) for _ in range(N)]
for H in Hs:
    H.set_frames(sc.source, sc.target)
    for _ in range(10): H.scan_match(sc.T_pred)
its, bad = [0] * N, [[] for _ in range(N)]
start = threading.Barrier(N + 1)
def run(i):
    start.wait()
    for f in range(frames):
        t = time.perf_counter()
        rc, T, st = Hs[i].scan_match(sc.T_pred)
        dt = time.perf_counter() - t
        its[i] += st["gn_sweeps"]
        if rc != 0:
            bad[i].append((f, rc, Hs[i].L.tloam_last_error(Hs[i].h).decode(), round(dt * 1e3, 2), st["outer_iterations"], st["gn_sweeps"]))
th = [threading.Thread(target=run, args=(i,)) for i in range(N)]
for t in th: t.start()
start.wait(); t0 = time.perf_counter()
for t in th: t.join()
dt = time.perf_counter() - t0
print("knobs", {k: v for k, v in os.environ.items() if k.startswith("TLOAM_")})
print("streams %d: %.1f frames/s aggregate, %.0f GN it/s, %.3f ms per frame per stream" % (N, N * frames / dt, sum(its) / dt, dt / frames * 1e3))
for i in range(N):
    print(" stream", i, "failures", len(bad[i]), bad[i][:4])
