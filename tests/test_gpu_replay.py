"""-m gpu: the replay path of BASELINE.json configs[0]/[1] on a tiny synthetic `.bin` sequence: raw scans written with
write_velodyne_bin (KITTI wire format, float32 x y z intensity), read back with the reference's reader semantics,
labelled by the stand-in labeller, PCA feature extraction + scan matching + submap update on the device, trajectory
written in the reference's savePose format and read back.  (tloam_amd/replay.py; bench.py --kitti-dir runs the same
code on a real sequence directory.)"""
import os

import numpy as np
import pytest

from conftest import pose_delta
from tloam_amd import kitti_io, replay, synth_world as sw

pytestmark = pytest.mark.gpu


def _raw_scan(W, T, rng, max_range=30.0):
    Ti = np.linalg.inv(T)
    pts = np.concatenate([W.ground, W.planar, W.edge, W.sphere])
    d = pts[:, :2] - T[:2, 3]
    pts = pts[np.einsum("ij,ij->i", d, d) < max_range ** 2]
    p = pts @ Ti[:3, :3].T + Ti[:3, 3]
    p = p + rng.normal(0, 0.01, p.shape)
    return p[rng.permutation(len(p))]


def test_replay_of_a_synthetic_bin_sequence(hip_module, tmp_path):
    rng = np.random.default_rng(5)
    W = sw.make_world(seed=2, length=60.0, density=8.0)
    Ts = sw.trajectory(9)
    seq = tmp_path / "velodyne"
    seq.mkdir()
    for f, T in enumerate(Ts):
        kitti_io.write_velodyne_bin(str(seq / f"{f:06d}.bin"), _raw_scan(W, T, rng))
    files = replay.list_scans(str(tmp_path))
    assert len(files) == len(Ts)
    H = hip_module.HipRegistration()
    out = str(tmp_path / "traj.txt")
    fc = hip_module.default_feature_config(radius=0.3)
    poses, rep = replay.replay(H, files, out_poses=out, feature_cfg=fc)
    H.close()
    assert rep["frames"] == len(Ts) and rep["gn_iters_per_frame"] > 0
    back = kitti_io.read_poses(out)                       # the file is the KITTI pose format (front_end.cpp:169-179)
    assert back.shape == (len(Ts), 4, 4)
    assert np.abs(back[-1][:3, 3] - poses[-1][:3, 3]).max() < 1e-3 * max(1.0, np.abs(poses[-1][:3, 3]).max())   # %g, 6 digits
    # the vehicle moved ~0.8 m per frame along x and the estimate follows it (stand-in labeller: decimetres, not mm)
    for f in range(1, len(Ts)):
        dt, dr = pose_delta(poses[f], Ts[f])
        assert dt < 0.35 and dr < 0.05, (f, dt, dr)
    assert np.linalg.norm(poses[-1][:3, 3]) > 4.0


def test_replay_skips_cleanly_without_data(tmp_path):
    assert replay.list_scans(str(tmp_path / "no_such_sequence")) == []
