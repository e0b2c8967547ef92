"""Bitwise comparison of the three Solve paths of a KITTI-size set (one-launch Solve, one launch per GN iteration, sweep and
step as two launches) after 1, 2, 3 ... sweeps: where do they first differ?"""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import ctypes as C
    import numpy as np
    from tloam_amd import registration as reg, synth
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=4500, n_line=1200, n_point=200)
    out = {}
    for ms in (1, 2, 3, 5):
        os.environ["TLOAM_DEBUG_MAX_SWEEPS"] = str(ms)
        H = reg.HipRegistration()
        for rt in range(3):
            H.set_correspondences(rt, *sets[rt])
        x, st = H.solve(x_eval)
        buf = np.zeros(400)
        n = H.L.tloam_debug_state(H.h, buf.ctypes.data_as(C.POINTER(C.c_double)), 400)
        out[ms] = [float.hex(float(v)) for v in buf[:n - 12]]
        if ms == 1:
            pb = np.zeros(32 * 32)
            g = H.L.tloam_debug_partials(H.h, pb.ctypes.data_as(C.POINTER(C.c_double)), 32 * 32)
            out["rows"] = [[float.hex(float(v)) for v in pb[r * 32:(r + 1) * 32]] for r in range(g)]
        H.close()
    print(json.dumps(out))
    sys.exit(0)
res = {}
for name, env in (("solve1", {}), ("periter", {"TLOAM_NO_PERSISTENT_SOLVE": "1"}), ("unfused", {"TLOAM_NO_FUSED_SMALL": "1"})):
    e = dict(os.environ); e.update(env)
    o = subprocess.check_output([sys.executable, __file__, "child"], env=e, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
    res[name] = json.loads(o)
ra, rb = res["solve1"]["rows"], res["periter"]["rows"]
for r in range(len(ra)):
    d = [c for c in range(32) if ra[r][c] != rb[r][c]]
    print("row", r, "differing words", d)
for ms in ("1",):
    for a, b in (("solve1", "periter"), ("periter", "unfused")):
        A, B = res[a][ms], res[b][ms]
        diff = [i for i in range(len(A)) if A[i] != B[i]]
        print("sweeps", ms, a, "vs", b, "differing words:", diff[:24], "of", len(A))
        for i in diff[:6]:
            print("   word", i, A[i], B[i])
