"""-m gpu: the whole odometry inner loop on the device -- setInputSource, scanMatching, updateSubmap -- over a
consistent synthetic street (tloam_amd/synth_world.py), frame by frame against the same loop driven through the
CPU restatements.  The targets never leave the GPU between frames; poses and submaps must stay identical."""
import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from tloam_amd import synth_world as sw

pytestmark = pytest.mark.gpu


def test_device_loop_tracks_the_oracle_loop(hip_module):
    W = sw.make_world(seed=1)
    Ts = sw.trajectory(14)
    H = hip_module.HipRegistration()
    O = ob.Oracle()
    S = ob.OracleSubmap()
    s0 = sw.scan(W, Ts[0], 1, 0)
    H.submap_init(s0[0], s0[3], s0[2], s0[1])
    S.init(s0[0], s0[3], s0[2], s0[1])
    est_h, est_o = [np.eye(4)], [np.eye(4)]
    for f in range(1, len(Ts)):
        sc = sw.scan(W, Ts[f], 1, f)
        pred_h = est_h[-1] @ (np.linalg.inv(est_h[-2]) @ est_h[-1] if len(est_h) > 1 else np.eye(4))   # front_end.cpp:329-330
        pred_o = est_o[-1] @ (np.linalg.inv(est_o[-2]) @ est_o[-1] if len(est_o) > 1 else np.eye(4))
        for k in range(4):
            H.set_source(k, sc[k])
            O.set_source(k, sc[k])
            O.set_target(k, S.get(k))
        rh, Th, sh = H.scan_match(pred_h)
        ro, To, so = O.scan_match(pred_o)
        assert rh == ro == 0
        assert sh["n_corr"] == so["n_corr"], f
        dt, dr = pose_delta(Th, To)
        assert dt < 1e-8 and dr < 1e-8, (f, dt, dr)
        est_h.append(Th); est_o.append(To)
        H.submap_update(Th, sc[0], sc[3], sc[2], sc[1])
        S.update(To, sc[0], sc[3], sc[2], sc[1])
        for k in range(4):
            a, b = H.get_target(k), S.get(k)
            assert a.shape == b.shape, (f, k)
            assert np.abs(a - b).max() < 1e-8, (f, k)
        # the generator's pose is tracked to a centimetre
        assert pose_delta(Th, Ts[f])[0] < 0.05, f
    H.close()
