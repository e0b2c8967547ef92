"""KITTI wire formats: readVelodyneToO3d (read_file.hpp:307-327), savePose
(front_end.cpp:169-179) and the doc/tloam_XX.txt trajectory format."""
import numpy as np
import pytest

from tloam_amd import kitti_io as kio


def test_velodyne_round_trip_nan_and_eof_quirk(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.uniform(-50, 50, (100, 3)).astype(np.float32)
    inten = rng.uniform(0, 1, 100).astype(np.float32)
    xyz[7, 1] = np.nan; inten[20] = np.nan
    p = str(tmp_path / "000000.bin")
    kio.write_velodyne_bin(p, xyz, inten)
    got, gi = kio.read_velodyne_bin(p)
    keep = np.ones(100, bool); keep[[7, 20]] = False
    assert got.dtype == np.float64 and got.shape == (99, 3)          # 98 valid + the reference's EOF point
    assert np.array_equal(got[:-1], xyz[keep].astype(np.float64))     # float32 widened exactly
    assert np.array_equal(got[-1], [0, 0, 0]) and gi[-1] == 1.0       # Point4f() (read_file.hpp:89)
    got2, _ = kio.read_velodyne_bin(p, reference_eof_quirk=False)
    assert np.array_equal(got2, xyz[keep].astype(np.float64))


def test_partial_trailing_record_is_ignored(tmp_path):
    p = str(tmp_path / "x.bin")
    np.arange(10, dtype="<f4").tofile(p)                              # 2 records + 2 stray floats
    got, _ = kio.read_velodyne_bin(p, reference_eof_quirk=False)
    assert np.array_equal(got, [[0, 1, 2], [4, 5, 6]])


def test_save_pose_format(tmp_path):
    T = np.eye(4); T[0, 3] = 1234567.0; T[1, 3] = -0.000012345678; T[2, 3] = 0.5; T[0, 1] = 1.0 / 3.0
    line = kio.format_pose_line(T)
    assert line.endswith("\n") and len(line.split()) == 12
    assert line.split()[1] == "0.333333" and line.split()[3] == "1.23457e+06" and line.split()[7] == "-1.23457e-05"
    p = str(tmp_path / "poses.txt")
    poses = [np.eye(4), T]
    kio.write_poses(p, poses)
    back = kio.read_poses(p)
    assert back.shape == (2, 4, 4)
    assert np.allclose(back[1][:3], T[:3], rtol=1e-5)                 # 6 significant digits, like the reference
    rel = kio.relative_poses(back)
    assert rel.shape == (1, 4, 4)


def test_read_poses_rejects_other_shapes(tmp_path):
    p = str(tmp_path / "bad.txt")
    open(p, "w").write("1 2 3\n")
    with pytest.raises(ValueError):
        kio.read_poses(p)


def test_reference_trajectory_fixture(tmp_path):
    """tests/golden/tloam_00_ego_motion.npz is derived from the reference's own published KITTI-00 trajectory
    (doc/tloam_00.txt; generator: tests/golden/make_tloam00_ego_motion.py).  The first pose lines of that file,
    kept verbatim as numbers, pin the pose reader on real reference data; the ego-motion drives bench.py's
    KITTI-density sequence."""
    import os
    from tloam_amd import synth
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tloam_00_ego_motion.npz"))
    se3, lines = z["se3"], z["first_pose_lines"]
    assert se3.shape == (4540, 6) and lines.shape == (3, 12)
    p = tmp_path / "poses.txt"
    np.savetxt(p, lines, fmt="%.18e")
    poses = kio.read_poses(str(p))
    assert poses.shape == (3, 4, 4) and np.allclose(poses[0], np.eye(4), atol=1e-15)
    for T in poses:                                   # rigid transforms (to the precision the reference wrote them with)
        assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-4) and abs(np.linalg.det(T[:3, :3]) - 1) < 1e-4
    # camera -> velodyne axes: the car drives along +x (camera +z) at KITTI speeds, yaw about z
    rel_cam = kio.relative_poses(poses)
    assert np.allclose(se3[0][:3], [rel_cam[0][2, 3], -rel_cam[0][0, 3], -rel_cam[0][1, 3]], atol=2e-3)   # (upsilon ~ t for mrad rotations)
    step = np.linalg.norm(se3[:, :3], axis=1)
    assert 0.7 < step.mean() < 0.9 and step.max() < 1.5 and (se3[:, 0] > -0.05).all()
    assert np.abs(se3[:, 5]).mean() > 3 * np.abs(se3[:, 3]).mean()
    # exp/log used to build the constant-velocity prediction error
    for a in se3[::500]:
        assert np.allclose(synth.se3_log_np(synth.se3_exp_np(a)), a, atol=1e-12)
