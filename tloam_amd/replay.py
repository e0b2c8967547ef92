"""Sequence replay: KITTI Velodyne `.bin` scans in, a KITTI-format trajectory out -- the path either side of
scanMatching that BASELINE.json configs[0]/[1] name literally, ready for real KITTI-00 when the
data is present:

    read_velodyne_bin                 readVelodyneToO3d                      read_file.hpp:307-327
      -> label_scan                   (stand-in, see below)
      -> tloam_extract_planar_sphere  featureExtract::extractPlanarSphere    feature_extract.cpp:133-197   [device]
      -> tloam_set_source_frame / tloam_scan_match                           front_end.cpp:314-322          [device]
      -> tloam_submap_update          FrontEnd::updateSubmap                 front_end.cpp:201-275          [device]
      -> format_pose_line             FrontEnd::savePose                     front_end.cpp:169-179

`label_scan` is NOT the reference's segmentation (DCVC clustering + ground fitting, src/models/segmentation -- out of
scope): it is a small geometric labeller -- ground by height over the lowest returns, edge = tall,
isolated columns of a 2-D occupancy grid, the rest handed to the PCA feature extraction -- that only exists so that
the stages around it can be driven end to end.  Trajectories produced with it are therefore not comparable with the
reference's doc/tloam_XX.txt beyond plausibility; swapping in a real segmentation changes nothing downstream.

Host-side glue only (numpy + the C ABI through tloam_amd.registration); no oracle, no CPU fallback of any device stage."""
from __future__ import annotations

import glob
import os
import time

import numpy as np

from . import kitti_io
from .synth import Frame


def _voxel_first(xyz: np.ndarray, size: float) -> np.ndarray:
    """one point (the first in input order) per occupied voxel -- a cheap stand-in for the per-scan VoxelDownSample of
    the segmentation stage (edge 0.1 m, ground 0.3 m: lidar_odometry.yaml:6,8)"""
    if len(xyz) == 0:
        return xyz
    key = np.floor(xyz / size).astype(np.int64)
    key = (key[:, 0] * 73856093) ^ (key[:, 1] * 19349663) ^ (key[:, 2] * 83492791)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return np.ascontiguousarray(xyz[first])


def label_scan(xyz: np.ndarray, sensor_height: float = 1.73, max_range: float = 60.0):
    """-> (ground, edge, other): see the module docstring.  sensor_height: segmentation.yaml:4."""
    r2 = xyz[:, 0] ** 2 + xyz[:, 1] ** 2
    xyz = xyz[(r2 > 1.0) & (r2 < max_range ** 2)]
    z0 = -sensor_height
    ground_m = xyz[:, 2] < z0 + 0.25
    rest = xyz[~ground_m]
    # 2-D occupancy columns of 0.4 m: vertical extent and number of occupied neighbours
    cell = 0.4
    ij = np.floor(rest[:, :2] / cell).astype(np.int64)
    off = ij.min(axis=0) if len(ij) else np.zeros(2, np.int64)
    ij -= off
    w = int(ij[:, 0].max()) + 3 if len(ij) else 3
    h = int(ij[:, 1].max()) + 3 if len(ij) else 3
    lin = (ij[:, 0] + 1) * h + (ij[:, 1] + 1)
    zmin = np.full(w * h, np.inf); zmax = np.full(w * h, -np.inf)
    np.minimum.at(zmin, lin, rest[:, 2]); np.maximum.at(zmax, lin, rest[:, 2])
    occ = np.isfinite(zmin).reshape(w, h)
    tall = ((zmax - zmin) > 1.2).reshape(w, h)
    nb = np.zeros((w, h), np.int32)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dx or dy:
                nb += np.roll(np.roll(occ, dx, 0), dy, 1)
    edge_col = tall & (nb <= 2)            # poles, trunks, corners: tall and isolated
    is_edge = edge_col.reshape(-1)[lin]
    return (_voxel_first(xyz[ground_m], 0.3), _voxel_first(rest[is_edge], 0.1), np.ascontiguousarray(rest[~is_edge]))


def features_of_scan(H, xyz: np.ndarray, feature_cfg=None, sensor_height: float = 1.73):
    """One raw scan -> the clouds FrontEnd hands on (front_end.cpp:183-198): scan-side Frame (planar_scan, ground,
    edge, sphere_scan) and the submap-side selections (planar_submap, sphere_submap).  PCA lists on the device."""
    ground, edge, other = label_scan(xyz, sensor_height)
    ps, pm, ss, sm = H.extract_planar_sphere(other, feature_cfg)
    sel = lambda idx: np.ascontiguousarray(other[np.asarray(idx, np.int64) % max(len(other), 1)]) if len(other) else other  # noqa: E731
    planar_scan, planar_submap, sphere_scan, sphere_submap = sel(ps), sel(pm), sel(ss), sel(sm)

    def at_least(a, pool, n=10):   # the path needs >= 10 points per cloud (registration.cpp:928-929)
        return a if len(a) >= n or len(pool) < n else np.ascontiguousarray(pool[:: max(len(pool) // 64, 1)][:max(n, 64)])
    planar_scan, planar_submap = at_least(planar_scan, other), at_least(planar_submap, other)
    sphere_scan, sphere_submap = at_least(sphere_scan, other), at_least(sphere_submap, other)
    return Frame(planar_scan, ground, edge, sphere_scan), planar_submap, sphere_submap


def list_scans(path: str):
    """<path>/velodyne/*.bin (KITTI odometry layout), or <path>/*.bin"""
    for d in (os.path.join(path, "velodyne"), path):
        files = sorted(glob.glob(os.path.join(d, "*.bin")))
        if files:
            return files
    return []


def replay(H, scan_files, out_poses: str | None = None, feature_cfg=None, sensor_height: float = 1.73, max_frames=None,
           sync=None):
    """FrontEnd::updateLidarOdometry (front_end.cpp:278-337) over a list of `.bin` scans on the device.  Returns the
    poses (map <- sensor, 4x4) and per-stage host-to-host timings in ms."""
    poses, t_feat, t_match, t_submap, iters = [], [], [], [], 0
    files = scan_files[: max_frames] if max_frames else scan_files
    out = open(out_poses, "w") if out_poses else None
    try:
        for f, path in enumerate(files):
            xyz, _ = kitti_io.read_velodyne_bin(path)
            t0 = time.perf_counter()
            frame, planar_submap, sphere_submap = features_of_scan(H, xyz, feature_cfg, sensor_height)
            t1 = time.perf_counter()
            if f == 0:                                   # first frame: the submap IS the scan (front_end.cpp:283-304)
                H.submap_init(planar_submap, sphere_submap, frame.edge, frame.ground)
                T = np.eye(4)
                t2 = t1
            else:
                pred = poses[-1] @ (np.linalg.inv(poses[-2]) @ poses[-1] if len(poses) > 1 else np.eye(4))  # :329-330
                H.set_input_source(frame)
                rc, T, st = H.scan_match(pred)
                if rc not in (0, -7):
                    raise RuntimeError(f"frame {f} ({os.path.basename(path)}): scan_match status {rc}")
                iters += st["gn_sweeps"]
                t2 = time.perf_counter()
                H.submap_update(T, planar_submap, sphere_submap, frame.edge, frame.ground)
            if sync:
                sync()
            t3 = time.perf_counter()
            poses.append(T)
            if out:
                out.write(kitti_io.format_pose_line(T))
            if f > 0:
                t_feat.append((t1 - t0) * 1e3); t_match.append((t2 - t1) * 1e3); t_submap.append((t3 - t2) * 1e3)
    finally:
        if out:
            out.close()
    m = lambda v: round(float(np.mean(v)), 4) if v else None  # noqa: E731
    return poses, {"frames": len(poses), "ms_features_incl_labeller": m(t_feat), "ms_set_source_plus_scan_match": m(t_match),
                   "ms_submap_update": m(t_submap), "gn_iters_per_frame": round(iters / max(len(poses) - 1, 1), 2)}
