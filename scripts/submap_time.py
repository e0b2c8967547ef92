"""Host-to-host time of tloam_submap_update at KITTI-like sizes (median of 200 calls), for A/B runs with TLOAM_HIP_LIB."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth_submap as ss
H = reg.HipRegistration(reg.default_config())
H.submap_init(*ss.frame_clouds(0, 0, n=(4000, 500, 7000, 30000), extent=60.0))
clouds = [ss.frame_clouds(0, f, n=(4000, 500, 2000, 4000), extent=60.0) for f in range(1, 41)]
ts = []
for rep in range(6):
    for f, cl in enumerate(clouds):
        T = ss.frame_pose(f + 1)
        t0 = time.perf_counter(); H.submap_update(T, *cl); ts.append(time.perf_counter() - t0)
ts = np.array(ts[40:]) * 1e3
print("lib %-12s submap_update ms: median %.4f mean %.4f p10 %.4f" % (os.path.basename(os.environ.get("TLOAM_HIP_LIB", "default")), np.median(ts), ts.mean(), np.percentile(ts, 10)), flush=True)
