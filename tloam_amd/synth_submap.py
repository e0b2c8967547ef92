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
