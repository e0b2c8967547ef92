"""Synthetic LiDAR feature frames for the parity tests and bench.py.

KITTI data cannot be fetched here, so the inputs of the hot path (the eight feature clouds a
`FrontEnd` would hand to `RegistrationInterface::setInputSource/Target`, front_end.cpp:267,
:314) are produced by a seeded procedural street scene:

  ground   z = -1.73 (sensorHeight, config/mapping/segmentation.yaml:4), sigma 2 cm
  planar   vertical facade patches on both street sides
  edge     vertical poles (line features; along-line spacing >> lateral noise so that the
           reference's eigenvalue gate registration.cpp:481 passes)
  sphere   isolated blob points

Targets (the "submap") live in the map frame; sources (the current scan) are fresh samples of
the same surfaces expressed in the sensor frame  p_s = T_true^-1 (p_w + noise), rounded to
float32 like the ROS wire format does (open3d_to_ros.cpp:179-182).
This is host-side input generation only -- none of it is on the product's compute path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

KINDS = ("planar", "ground", "edge", "sphere")


def _hat(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def se3_exp_np(a):
    """Plain closed-form SE(3) exponential (input generation only)."""
    ups, om = np.asarray(a[:3], float), np.asarray(a[3:], float)
    th = float(np.linalg.norm(om))
    Om = _hat(om)
    if th < 1e-10:
        R = np.eye(3) + Om
        V = np.eye(3) + 0.5 * Om
    else:
        R = np.eye(3) + math.sin(th) / th * Om + (1 - math.cos(th)) / th**2 * (Om @ Om)
        V = np.eye(3) + (1 - math.cos(th)) / th**2 * Om + (th - math.sin(th)) / th**3 * (Om @ Om)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def se3_log_np(T):
    """Inverse of se3_exp_np (input generation only): (upsilon, omega) of a rigid transform, |omega| < pi."""
    R, t = np.asarray(T, float)[:3, :3], np.asarray(T, float)[:3, 3]
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = 0.5 * float(np.linalg.norm(w)), 0.5 * (float(np.trace(R)) - 1.0)
    th = math.atan2(s, c)
    om = 0.5 * w if th < 1e-10 else th / (2.0 * s) * w
    Om = _hat(om)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * Om
    else:
        V = np.eye(3) + (1 - math.cos(th)) / th**2 * Om + (th - math.sin(th)) / th**3 * (Om @ Om)
    return np.concatenate([np.linalg.solve(V, t), om])


@dataclass
class Frame:
    """The four feature clouds of tloam::Frame (registration_interface.hpp:19-38), (n,3) f64."""
    planar: np.ndarray
    ground: np.ndarray
    edge: np.ndarray
    sphere: np.ndarray

    def cloud(self, kind: int) -> np.ndarray:
        return (self.planar, self.ground, self.edge, self.sphere)[kind]


@dataclass
class Scene:
    source: Frame          # current scan, sensor frame
    target: Frame          # submap, map frame
    T_true: np.ndarray     # map <- sensor
    T_pred: np.ndarray     # constant-velocity style prediction handed to scanMatching
    seed: int


# sizes per kind: (planar, ground, edge, sphere)
KITTI_SRC = (3000, 4000, 2000, 400)
KITTI_TGT = (12000, 40000, 30000, 1500)
SMALL_SRC = (400, 500, 300, 80)            # parity-test size (oracle runs in milliseconds)
SMALL_TGT = (1600, 4000, 3000, 300)
M1_SRC = (500_000, 260_000, 200_000, 40_000)   # the 1 M-correspondence frame (760k plane : 200k line : 40k point)
M1_TGT = (500_000, 260_000, 200_000, 40_000)


def _sample_surfaces(rng, n_planar, n_ground, n_edge, n_sphere, geom):
    """Sample map-frame points on the scene geometry."""
    # ---- ground: disc around the origin
    r = geom["ground_r"] * np.sqrt(rng.random(n_ground))
    phi = rng.random(n_ground) * 2 * np.pi
    ground = np.stack([r * np.cos(phi), r * np.sin(phi), np.full(n_ground, -1.73)], axis=1)
    # ---- planar: facade patches (origin o, in-plane axes u (horizontal), z up)
    walls = geom["walls"]
    wi = rng.integers(0, len(walls), n_planar)
    s = rng.random(n_planar)
    h = rng.random(n_planar)
    planar = np.empty((n_planar, 3))
    for j, (o, u, length, height) in enumerate(walls):
        m = wi == j
        planar[m] = o + np.outer(s[m] * length, u) + np.outer(h[m] * height, [0, 0, 1.0])
    # ---- edge: poles (base b, height)
    poles = geom["poles"]
    pi_ = rng.integers(0, len(poles), n_edge)
    hh = rng.random(n_edge)
    edge = poles[pi_, :3].copy()
    edge[:, 2] += hh * poles[pi_, 3]
    # ---- sphere: blobs
    sphere = geom["blobs"][rng.integers(0, len(geom["blobs"]), n_sphere)].copy()
    return planar, ground, edge, sphere


def make_geometry(rng, n_tgt, density=10.0, pole_spacing=0.15):
    """Scene geometry sized so the TARGET clouds reach `density` points per m^2 on surfaces."""
    n_planar, n_ground, n_edge, n_sphere = n_tgt
    ground_r = math.sqrt(max(n_ground, 1) / density / math.pi)
    ground_r = max(ground_r, 8.0)
    # facades: 20 m x 6 m patches along a street of half-width 6..15 m
    wall_area = max(n_planar, 1) / density
    n_walls = max(4, int(math.ceil(wall_area / (20.0 * 6.0))))
    wall_len = max(2.0, wall_area / n_walls / 6.0)   # keeps `density` when few points are asked for
    extent = max(ground_r, 15.0 * math.sqrt(n_walls / 10.0))
    walls = []
    for j in range(n_walls):
        side = 1.0 if j % 2 == 0 else -1.0
        yaw = rng.normal(0.0, 0.15) + (math.pi / 2 if rng.random() < 0.2 else 0.0)
        u = np.array([math.cos(yaw), math.sin(yaw), 0.0])
        o = np.array([rng.uniform(-extent, extent), side * rng.uniform(6.0, 6.0 + extent * 0.6), -1.73])
        walls.append((o, u, wall_len, 6.0))
    pole_h = 8.0
    per_pole = pole_h / pole_spacing
    n_poles = max(8, int(math.ceil(n_edge / per_pole)))
    poles = np.stack([rng.uniform(-extent, extent, n_poles), rng.uniform(-extent, extent, n_poles),
                      np.full(n_poles, -1.73), np.full(n_poles, pole_h)], axis=1)
    blobs = np.stack([rng.uniform(-extent, extent, n_sphere), rng.uniform(-extent, extent, n_sphere),
                      rng.uniform(-1.5, 4.0, n_sphere)], axis=1)
    # keep isolated blob points >= 1.2 m apart is not needed: 1-NN only
    return dict(ground_r=ground_r, walls=walls, poles=poles, blobs=blobs, extent=extent)


def make_scene(seed=0, n_src=SMALL_SRC, n_tgt=SMALL_TGT, noise=0.02, density=10.0,
               true_se3=(0.8, 0.05, 0.02, 0.002, 0.003, 0.3),
               pred_err=(0.012, -0.008, 0.004, 0.0015, -0.001, 0.002),
               outlier_frac=0.0, float32_wire=True) -> Scene:
    """One frame pair.  `true_se3` is the map<-sensor pose (rotation 0.3 rad about z by default
    so that |omega| >= 1e-2 and the reference's random perturbation :884-886 is not triggered);
    `pred_err` ~ the constant-velocity prediction error of SURVEY section 6 (1.6 cm / 3 mrad)."""
    rng = np.random.default_rng(seed)
    geom = make_geometry(rng, n_tgt, density=density)
    tp, tg, te, ts = _sample_surfaces(rng, *n_tgt, geom)
    # target noise: planar along normal ~ isotropic small; edge lateral 1 cm; ground z sigma
    tp += rng.normal(0, noise * 0.5, tp.shape)
    tg[:, 2] += rng.normal(0, noise, len(tg))
    te[:, :2] += rng.normal(0, 0.01, (len(te), 2))
    # sources: fresh samples of the same surfaces
    sp, sg, se, _ = _sample_surfaces(rng, n_src[0], n_src[1], n_src[2], 1, geom)
    sp += rng.normal(0, noise * 0.5, sp.shape)
    sg[:, 2] += rng.normal(0, noise, len(sg))
    se[:, :2] += rng.normal(0, 0.01, (len(se), 2))
    pick = rng.integers(0, len(ts), n_src[3])
    ss = ts[pick] + rng.normal(0, noise, (n_src[3], 3))
    if outlier_frac > 0:
        for arr in (sp, sg, se, ss):
            m = rng.random(len(arr)) < outlier_frac
            arr[m] += rng.normal(0, 0.25, (int(m.sum()), 3))
    T_true = se3_exp_np(true_se3)
    Tinv = np.linalg.inv(T_true)

    def to_sensor(p):
        q = p @ Tinv[:3, :3].T + Tinv[:3, 3]
        if float32_wire:
            q = q.astype(np.float32).astype(np.float64)
        return np.ascontiguousarray(q)

    source = Frame(to_sensor(sp), to_sensor(sg), to_sensor(se), to_sensor(ss))
    target = Frame(*(np.ascontiguousarray(a) for a in (tp, tg, te, ts)))
    T_pred = T_true @ se3_exp_np(pred_err)
    return Scene(source, target, T_true, T_pred, seed)


def make_prebuilt(seed=1, n_plane=760_000, n_line=200_000, n_point=40_000, weights="ones", noise_scale=1.0):
    """Pre-built correspondence sets of SURVEY 8(d) config 3 (the K3 roofline run).
    Returns dict(res_type -> (p, a, b, d, w)), the true pose vector and an evaluation point."""
    rng = np.random.default_rng(seed)
    x_true = np.array([0.8, 0.05, 0.02, 0.002, 0.003, 0.013])
    T = se3_exp_np(x_true)
    R, t = T[:3, :3], T[:3, 3]

    def src(n):
        return np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-5, 5, n)], axis=1)

    def unit(n):
        v = rng.normal(size=(n, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    out = {}
    p = src(n_plane); pw = p @ R.T + t
    nrm = unit(n_plane)
    d = -(nrm * pw).sum(axis=1) + noise_scale * rng.normal(0, 0.05, n_plane)
    out[0] = [p, nrm, None, d]
    p = src(n_line); pw = p @ R.T + t
    dirs = unit(n_line)
    off = noise_scale * rng.normal(0, 0.05, (n_line, 3))
    mu = pw + off + dirs * rng.uniform(-0.3, 0.3, (n_line, 1))
    out[1] = [p, mu + 0.1 * dirs, mu - 0.1 * dirs, None]
    p = src(n_point); pw = p @ R.T + t
    out[2] = [p, pw + noise_scale * rng.normal(0, 0.02, (n_point, 3)), None, None]
    for k, n in ((0, n_plane), (1, n_line), (2, n_point)):
        if weights == "ones":
            w = np.ones(n)
        else:  # timing variant: U(0,1) with 10 % exact zeros
            w = rng.random(n)
            w[rng.random(n) < 0.1] = 0.0
        out[k].append(w)
        out[k] = tuple(None if a is None else np.ascontiguousarray(a) for a in out[k])
    x_eval = x_true + np.array([0.012, -0.008, 0.004, 0.0015, -0.001, 0.002])
    return out, x_true, x_eval
