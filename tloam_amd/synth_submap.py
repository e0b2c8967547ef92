"""Synthetic feature-cloud sequences for the submap path: per frame the four clouds
FrontEnd::updateSubmap consumes (planar/sphere submap selections, edge and ground scan features, sensor
frame) and the odometry pose.  Coordinates are rounded to float32 like the ROS wire (open3d_to_ros.cpp)."""
from __future__ import annotations

import numpy as np


def _se3_exp(a):
    v, w = np.asarray(a[:3], float), np.asarray(a[3:], float)
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        R, V = np.eye(3) + W, np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def frame_pose(f, step=0.8, yaw_rate=0.01):
    """Map <- sensor pose of frame f: ~step metres forward per frame along a gentle arc."""
    T = np.eye(4)
    for _ in range(f):
        T = T @ _se3_exp((step, 0.0, 0.0, 0.0, 0.001, yaw_rate))
    return T


def frame_clouds(seed, f, n=(600, 120, 1500, 2500), extent=40.0):
    """(planar, sphere, edge, ground) of frame f in the SENSOR frame."""
    rng = np.random.default_rng(7919 * seed + f)
    n_planar, n_sphere, n_edge, n_ground = n
    planar = np.column_stack([rng.uniform(-extent, extent, n_planar), rng.choice([-8.0, 8.0], n_planar) +
                              rng.normal(0, 0.02, n_planar), rng.uniform(-1.5, 3.0, n_planar)])
    sphere = rng.uniform(-extent, extent, (n_sphere, 3)) * [1, 1, 0.05]
    poles = rng.uniform(-extent, extent, (max(n_edge // 25, 1), 2))
    pid = rng.integers(0, len(poles), n_edge)
    edge = np.column_stack([poles[pid] + rng.normal(0, 0.03, (n_edge, 2)), rng.uniform(-1.7, 2.5, n_edge)])
    r = extent * np.sqrt(rng.uniform(0.0004, 1.0, n_ground))       # denser near the sensor, like a LiDAR
    a = rng.uniform(0, 2 * np.pi, n_ground)
    ground = np.column_stack([r * np.cos(a), r * np.sin(a), -1.73 + rng.normal(0, 0.02, n_ground)])
    return tuple(np.ascontiguousarray(c.astype(np.float32).astype(np.float64)) for c in (planar, sphere, edge, ground))


def feature_cloud(seed, n=3000):
    """A small scan-like cloud for the PCA feature path: a wall and a ground patch dense enough for the
    r = 0.2 m / K = 20 search (planar candidates), compact blobs (high-curvature / sphere candidates) and sparse
    clutter (too few neighbours -> skipped), float32-rounded."""
    r = np.random.default_rng(4243 * seed + 17)
    nw, ng = n // 2, n // 3
    s = np.sqrt(nw / 750.0)                       # ~300 points per square metre on the wall
    wall = np.column_stack([r.uniform(-s, s, nw), 2.0 + r.normal(0, 0.004, nw), r.uniform(-0.5 * s, 0.75 * s, nw)])
    g = np.sqrt(ng / 1200.0)
    ground = np.column_stack([r.uniform(-g, g, ng), r.uniform(-g, g, ng), -1.7 + r.normal(0, 0.004, ng)])
    nb = max((n - nw - ng) // 50, 1)
    blobs = np.concatenate([c + r.normal(0, 0.04, (40, 3)) for c in r.uniform(-2, 2, (nb, 3)) + [0, -3.0, 0]])
    clutter = r.uniform(-6, 6, (max(n - nw - ng - 40 * nb, 1), 3))
    return np.ascontiguousarray(np.concatenate([wall, ground, blobs, clutter]).astype(np.float32).astype(np.float64))
