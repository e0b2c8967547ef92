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
    got, gi = kio.read_velodyne_bin(p)                                # the reference keeps what the short read did get
    assert np.array_equal(got, [[0, 1, 2], [4, 5, 6], [8, 9, 0]]) and np.array_equal(gi, [3, 7, 1])


def _adversarial_files(tmp_path):
    """what a reader can meet: whole records, NaN records, every length of trailing partial record (1 .. 15 bytes), an empty
    file, a file shorter than one record, partial bytes that make a NaN, denormals, infinities"""
    rng = np.random.default_rng(3)
    base = rng.uniform(-80, 80, (37, 4)).astype("<f4")
    base[5, 2] = np.nan; base[11, 3] = np.nan; base[20, 0] = np.inf; base[21, 1] = -np.inf; base[22, 0] = 1e-42   # denormal
    blob = base.tobytes()
    files = {}
    for extra in range(16):
        files[f"tail{extra:02d}"] = blob + bytes(rng.integers(0, 256, extra, dtype=np.uint8))
    files["empty"] = b""
    files["short3"] = b"\x01\x02\x03"
    files["nan_tail"] = blob + np.array([1.0, np.nan], "<f4").tobytes()             # partial record whose y is NaN: dropped
    files["nan_high_bytes"] = blob + np.array([1.0, 2.0, 3.0], "<f4").tobytes() + b"\xff\xff"   # intensity bytes ff ff 80 3f: not a NaN
    files["all_nan"] = np.full((3, 4), np.nan, "<f4").tobytes()
    out = {}
    for name, data in files.items():
        p = tmp_path / (name + ".bin")
        p.write_bytes(data)
        out[name] = str(p)
    return out


def test_reader_equals_the_c_restatement_of_the_reference_byte_for_byte(tmp_path):
    from oracle import binding as ob
    for name, path in _adversarial_files(tmp_path).items():
        xyz, inten = kio.read_velodyne_bin(path)
        oxyz, ointen = ob.read_velodyne(path)
        assert xyz.shape == oxyz.shape, name
        assert xyz.tobytes() == oxyz.tobytes() and inten.tobytes() == ointen.tobytes(), name
    assert len(kio.read_velodyne_bin(str(tmp_path / "empty.bin"))[0]) == 1       # one default point, as the reference
    assert len(kio.read_velodyne_bin(str(tmp_path / "all_nan.bin"))[0]) == 1


def test_pose_writer_equals_the_c_restatement_byte_for_byte():
    from oracle import binding as ob
    rng = np.random.default_rng(4)
    cases = [np.eye(4), np.zeros((4, 4)), -np.zeros((4, 4))]
    for scale in (1e-300, 1e-12, 1e-5, 1e-4, 0.1, 1.0, 999999.5, 1e6, 1e7, 1e22, 1e300):
        cases.append(rng.normal(size=(4, 4)) * scale)
    T = np.eye(4); T[0, 0] = 0.1 + 0.2; T[0, 1] = 123456.5; T[0, 2] = 1234567.0; T[0, 3] = 0.0001; T[1, 0] = 0.00001
    T[1, 1] = -0.0; T[1, 2] = np.inf; T[1, 3] = -np.inf; T[2, 0] = np.nan; T[2, 1] = -np.nan; T[2, 2] = 2.5e-7; T[2, 3] = 100000.0
    cases.append(T)
    for T in cases:
        assert kio.format_pose_line(T).encode() == ob.format_pose(T), T



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
