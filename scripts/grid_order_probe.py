"""How much does the ORDER of the clouds matter to the atomic-based passes (grid count, query bin)?
usage: grid_order_probe.py random|sorted [kitti|m1]   (run under rocprofv3 --kernel-trace, see gpu_gridorder.sh)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
order = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "m1"
big = 1 << 30
if wl == "m1":
    cfg = reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big)
    sc = synth.make_scene(seed=1, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
else:
    cfg = reg.default_config()
    sc = synth.make_scene(seed=1, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
def reorder(a):
    if order == "random":
        return a
    key = np.floor(a / 0.5).astype(np.int64)     # scan-like coherence: consecutive points fall into the same cells
    o = np.lexsort((key[:, 0], key[:, 1], key[:, 2]))
    return np.ascontiguousarray(a[o])
src = synth.Frame(*[reorder(sc.source.cloud(k)) for k in range(4)])
tgt = synth.Frame(*[reorder(sc.target.cloud(k)) for k in range(4)])
H = reg.HipRegistration(cfg)
H.set_frames(src, tgt)
for _ in range(4):
    rc, T, st = H.scan_match(sc.T_pred)
print(order, wl, "rc", rc, "n_corr", st["n_corr"])
