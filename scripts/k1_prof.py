"""Wall-clock phases (10 ns units) of four sampled waves of the sixteen-lane correspondence search on a KITTI-density frame;
needs a -DTLOAM_K1_PROF build (python -m tloam_amd.build --variant k1prof --units tl_nn.hip -- -DTLOAM_K1_PROF=1)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tloam_amd import registration as reg, synth
H = reg.HipRegistration(reg.default_config())
sc = synth.make_scene(seed=1, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
H.set_frames(sc.source, sc.target)
rc, T, st = H.scan_match(sc.T_pred)
for rep in range(3):
    H.time_build(1)
    out = (C.c_ulonglong * 64)()
    H.L.tloam_debug_k1_prof(out)
    v = np.array(list(out), dtype=np.int64)
    t0 = v[15]
    print("rep", rep)
    for w in range(4):
        s = v[w * 16: w * 16 + 8] - t0
        print("  block %4d: entry %5d rows-start %5d rows-resolved %5d walked %5d merged %5d unpacked %5d fitted %5d stored %5d" %
              ((0, 800, 1600, 2300)[w], *s))
