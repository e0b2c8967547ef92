"""Host-to-host time of the cloud hand-over calls at KITTI-like sizes (median of 300), for A/B runs with TLOAM_HIP_LIB."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
sc = synth.make_scene(seed=1, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
H = reg.HipRegistration(reg.default_config())
H.set_frames(sc.source, sc.target)
def med(f, n=300):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts[20:]) * 1e3
print("lib %-12s set_input_source %.4f ms  set_input_target %.4f ms  scan_match %.4f ms" % (
    os.path.basename(os.environ.get("TLOAM_HIP_LIB", "default")), med(lambda: H.set_input_source(sc.source)),
    med(lambda: H.set_input_target(sc.target)), med(lambda: H.scan_match(sc.T_pred))), flush=True)
