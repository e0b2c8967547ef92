"""-m gpu: randomised parity sweep -- HIP vs the C oracle over seeded scenes and configurations (caps binding,
factor_num, more outer iterations, outliers, large prediction errors); final pose, minimiser counters, per-kind
correspondence index lists and weights.  tests/tools/stress_parity.py runs the same over hundreds of seeds."""
import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from tloam_amd import synth

pytestmark = pytest.mark.gpu

CFGS = [dict(), dict(planar_maxnum=120, ground_maxnum=150, edge_maxnum=70, sphere_maxnum=25), dict(factor_num=3),
        dict(max_iterations=6)]


@pytest.mark.parametrize("seed", range(24))
def test_randomised_scene(hip_module, seed):
    over = CFGS[seed % len(CFGS)]
    kw = dict(outlier_frac=0.1) if seed % 3 == 0 else {}
    if seed % 5 == 0:
        kw["pred_err"] = (0.05, -0.03, 0.02, 0.006, -0.004, 0.008)
    size = (synth.KITTI_SRC, synth.KITTI_TGT) if seed % 8 == 7 else (synth.SMALL_SRC, synth.SMALL_TGT)
    sc = synth.make_scene(seed=2000 + seed, n_src=size[0], n_tgt=size[1], **kw)
    H = hip_module.HipRegistration(hip_module.default_config(**over))
    O = ob.Oracle(ob.make_config(**over))
    H.set_frames(sc.source, sc.target)
    O.set_frames(sc.source, sc.target)
    for rep in range(2):                                  # second pass: learned sweep budgets in effect
        rh, Th, sh = H.scan_match(sc.T_pred)
        ro, To, so = O.scan_match(sc.T_pred)
        assert rh == ro == 0
        dt, dr = pose_delta(Th, To)
        assert dt < 1e-9 and dr < 1e-9, (dt, dr)          # north-star tolerance is 1e-6
        for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "n_corr", "outer_iterations"):
            assert sh[k] == so[k], k
        assert sh["gn_sweeps"] <= sh["gn_evaluations"]
        for kind in range(4):
            assert np.array_equal(H.get_correspondences(kind)["idx"], O.get_correspondences(kind)["idx"])
            np.testing.assert_allclose(H.get_weights(kind), O.get_weights(kind), rtol=0, atol=1e-9)
    H.close()
