"""Wire / disk formats either side of the path: the KITTI Velodyne `.bin` scan reader and
the KITTI odometry pose text, so that real KITTI-00 can be replayed when the data is present.

  read_velodyne_bin   readVelodyneToO3d            include/tloam/models/io/read_file.hpp:307-327
  format_pose_line    FrontEnd::savePose           src/front_end/front_end.cpp:169-179
  read_poses          the trajectory format of     doc/tloam_00.txt .. tloam_10.txt (12 numbers per line)

Host-side I/O only: nothing here touches the device or the oracle (oracle/io_oracle.c restates the same two reference
functions in C; tests/test_kitti_io.py holds this module against it byte for byte)."""
from __future__ import annotations

import numpy as np


def read_velodyne_bin(path: str, reference_eof_quirk: bool = True):
    """float32 x, y, z, intensity records -> (xyz float64 (n,3), intensity float64 (n,)).

    Follows read_file.hpp:314-324: records containing a NaN are dropped (`!point.HasNaNs()`); x/y/z are
    widened to double, intensity too.  Quirk kept on request: the reference's loop condition
    (`readFile.good() && !readFile.eof()`, :315) is tested BEFORE the two reads of an iteration, every iteration starts
    from a fresh `Point4f()` = (0, 0, 0, intensity 1) (read_file.hpp:89) and appends it unless it holds a NaN -- also the
    iteration in which a read comes up short.  So a file that ends on a record boundary yields ONE extra point
    (0, 0, 0, 1); a trailing partial record yields a point made of the bytes that were there (istream::read stores what it
    got) over the defaults -- the intensity is only read when x, y, z were complete -- and no extra point after it.
    reference_eof_quirk=False: whole, NaN-free records only."""
    blob = np.fromfile(path, dtype=np.uint8)
    n = blob.size // 16
    rec = blob[: n * 16].view("<f4").reshape(n, 4)
    if reference_eof_quirk:
        last = np.array([0.0, 0.0, 0.0, 1.0], "<f4")          # Point4f()
        tail = blob[n * 16:]                                   # 0 .. 15 bytes of a partial record
        lb = last.view(np.uint8).copy()
        lb[: tail.size] = tail                                 # (fewer than 12 bytes: only x, y, z are touched -- same bytes)
        rec = np.vstack([rec, lb.view("<f4").reshape(1, 4)])
    keep = ~np.isnan(rec).any(axis=1)
    rec = rec[keep]
    xyz = rec[:, :3].astype(np.float64)
    inten = rec[:, 3].astype(np.float64)
    return np.ascontiguousarray(xyz), inten


def write_velodyne_bin(path: str, xyz, intensity=None):
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    it = np.zeros(len(xyz), np.float32) if intensity is None else np.asarray(intensity, np.float32)
    np.column_stack([xyz, it]).astype("<f4").tofile(path)


def _fmt(v: float) -> str:
    """`ofs << double` with the stream defaults: precision 6, %g style (libstdc++ formats through vsnprintf("%.*g"):
    a NaN keeps its sign bit, "-nan", which Python's own formatting drops)."""
    v = float(v)
    if v != v:
        import math
        return "-nan" if math.copysign(1.0, v) < 0 else "nan"
    return "%g" % v


def format_pose_line(T) -> str:
    """savePose (front_end.cpp:169-179): the top 3x4 block, row-major, space separated, newline after (2,3)."""
    T = np.asarray(T, float)
    return " ".join(_fmt(T[i, j]) for i in range(3) for j in range(4)) + "\n"


def write_poses(path: str, poses) -> None:
    with open(path, "w") as f:
        for T in poses:
            f.write(format_pose_line(T))


def read_poses(path: str):
    """KITTI odometry format: 12 numbers per line = 3x4 row-major; returns (n,4,4)."""
    rows = np.loadtxt(path, ndmin=2)
    if rows.shape[1] != 12:
        raise ValueError(f"{path}: expected 12 numbers per line, got {rows.shape[1]}")
    out = np.tile(np.eye(4), (len(rows), 1, 1))
    out[:, :3, :] = rows.reshape(-1, 3, 4)
    return out


def relative_poses(poses):
    """T_{k-1}^{-1} T_k for k >= 1: the per-frame ego-motion SURVEY 8(d) uses to drive the synthetic sequence."""
    poses = np.asarray(poses, float)
    return np.array([np.linalg.inv(poses[k - 1]) @ poses[k] for k in range(1, len(poses))])
