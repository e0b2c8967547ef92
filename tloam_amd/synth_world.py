"""A consistent synthetic street for the whole odometry inner loop: one static world of
feature points per kind (ground plane, two building fronts, poles, small clutter blobs), a vehicle trajectory
through it, and per frame the four feature clouds a LiDAR at that pose would deliver (points within range,
subsampled, expressed in the sensor frame, range noise, rounded to float32 like the ROS wire).  Unlike
synth.make_scene -- independent frame pairs -- consecutive frames here see the SAME world, so a submap built from
earlier frames (tloam_submap_update) registers the next scan and drift can be measured against the generator."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .synth_submap import _se3_exp


@dataclass
class World:
    planar: np.ndarray
    ground: np.ndarray
    edge: np.ndarray
    sphere: np.ndarray

    def cloud(self, kind: int) -> np.ndarray:
        return (self.planar, self.ground, self.edge, self.sphere)[kind]


def make_world(seed=0, length=260.0, density=1.0) -> World:
    r = np.random.default_rng(900 + seed)
    x0, x1 = -40.0, length
    n_g = int(5.0 * density * (x1 - x0) * 60.0)
    ground = np.column_stack([r.uniform(x0, x1, n_g), r.uniform(-30, 30, n_g), -1.73 + r.normal(0, 0.01, n_g)])
    walls = []
    for side in (-1.0, 1.0):
        n_w = int(8.0 * density * (x1 - x0) * 5.0)
        xs = r.uniform(x0, x1, n_w)
        keep = (np.floor(xs / 17.0) % 4 != 3)                      # gaps between the building fronts
        ys = side * (11.0 + 2.0 * (np.floor(xs / 23.0) % 2)) + r.normal(0, 0.01, n_w)   # staggered fronts
        walls.append(np.column_stack([xs, ys, r.uniform(-1.7, 3.3, n_w)])[keep])
    planar = np.concatenate(walls)
    poles = []
    for px in np.arange(x0, x1, 6.0):
        for side in (-1.0, 1.0):
            n_p = int(70 * density)
            c = np.array([px + r.uniform(-1, 1), side * (7.0 + r.uniform(-0.5, 0.5))])
            poles.append(np.column_stack([c + r.normal(0, 0.015, (n_p, 2)), r.uniform(-1.7, 4.0, n_p)]))
    edge = np.concatenate(poles)
    nb = int((x1 - x0) / 3.0)
    centres = np.column_stack([r.uniform(x0, x1, nb), r.uniform(-9, 9, nb), r.uniform(-1.5, 0.5, nb)])
    sphere = np.concatenate([c + r.normal(0, 0.04, (int(30 * density), 3)) for c in centres])
    return World(planar, ground, edge, sphere)


def trajectory(n_frames, step=0.8, yaw_rate=0.004):
    """Map <- sensor poses: ~8 m/s at 10 Hz along a gentle S curve."""
    T = [np.eye(4)]
    for f in range(1, n_frames):
        w = yaw_rate * np.cos(2 * np.pi * f / 120.0)
        T.append(T[-1] @ _se3_exp((step, 0.0, 0.0, 0.0, 0.0003 * np.sin(f / 9.0), w)))
    return T


def scan(world: World, T, seed, f, n=(3000, 4000, 2000, 400), max_range=45.0, noise=0.012):
    """The four feature clouds seen from pose T (map <- sensor), in the SENSOR frame."""
    r = np.random.default_rng(100003 * seed + f)
    Ti = np.linalg.inv(T)
    out = []
    for k in range(4):
        pts = world.cloud(k)
        d = pts[:, :2] - T[:2, 3]
        idx = np.nonzero(np.einsum("ij,ij->i", d, d) < max_range ** 2)[0]
        if len(idx) > n[k]:
            idx = r.choice(idx, n[k], replace=False)
        idx.sort()
        p = pts[idx] @ Ti[:3, :3].T + Ti[:3, 3]
        p = p + r.normal(0, noise, p.shape)
        out.append(np.ascontiguousarray(p.astype(np.float32).astype(np.float64)))
    return tuple(out)   # (planar, ground, edge, sphere)
