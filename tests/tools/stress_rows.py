#!/usr/bin/env python3
"""TEST TOOL (GPU box): randomised HIP-vs-restatement sweep over the two widening rows that own their data formats -- the
device-resident submap and the PCA feature extraction (next-2) -- with the sizes, configurations and
contents the suite's fixed cases do not enumerate: clouds of 0 / 1 / a few / thousands of points, duplicates, points on exact
voxel and crop-box boundaries, NaN / infinite points, poses with large rotations, voxel sizes from 2 cm to 5 m, crop boxes smaller
than the scan, one to five buffered frames; feature clouds with every K, min_neigh and radius the asserts admit.  Bit for bit,
call by call; a status the restatement returns must be the status the device returns.

    python tests/tools/stress_rows.py [trials=150] [seed=0]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from tloam_amd import registration as reg, synth_submap as ss  # noqa: E402


def f32(a):
    return np.ascontiguousarray(np.asarray(a, float).astype(np.float32).astype(np.float64))


def rand_cloud(rng, n, extent, voxel):
    kind = rng.integers(0, 6)
    if n == 0:
        return np.zeros((0, 3))
    if kind == 0:      # uniform
        c = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.1]
    elif kind == 1:    # on the voxel lattice: points exactly on voxel boundaries
        c = np.round(rng.uniform(-extent, extent, (n, 3)) / voxel) * voxel
    elif kind == 2:    # many duplicates
        base = rng.uniform(-extent, extent, (max(n // 20, 1), 3))
        c = base[rng.integers(0, len(base), n)]
    elif kind == 3:    # dense blob: hundreds of members per voxel
        c = rng.normal(0, voxel * 0.4, (n, 3)) + rng.uniform(-extent / 2, extent / 2, 3)
    elif kind == 4:    # scan-like ground
        r = extent * np.sqrt(rng.uniform(0.0004, 1.0, n)); a = rng.uniform(0, 2 * np.pi, n)
        c = np.column_stack([r * np.cos(a), r * np.sin(a), -1.7 + rng.normal(0, 0.02, n)])
    else:              # exactly on the crop box faces / far outside
        c = rng.uniform(-extent, extent, (n, 3))
        c[rng.integers(0, n, max(n // 10, 1)), rng.integers(0, 3)] = rng.choice([-extent, extent])
    c = f32(c)
    if n >= 8 and rng.integers(0, 5) == 0:      # a few non-finite points
        j = rng.choice(n, 3, replace=False)
        c[j[0], rng.integers(0, 3)] = np.nan
        c[j[1], rng.integers(0, 3)] = rng.choice([np.inf, -np.inf])
        c[j[2]] = np.nan
    return c


def status(fn, *args):
    try:
        rc = fn(*args)
        return 0 if rc is None else int(rc)
    except reg.TloamHipError as e:
        return -1 if "INVALID" in str(e) else -99


def submap_trial(rng, t):
    extent = float(rng.choice([5.0, 20.0, 60.0]))
    vox = float(rng.choice([0.02, 0.1, 0.3, 0.45, 1.0, 5.0]))
    cfg = dict(edge_crop_box_length=float(rng.choice([extent * 0.3, extent, extent * 3])),
               ground_crop_box_length=float(rng.choice([extent * 0.3, extent, extent * 3])),
               planar_frame_size=int(rng.integers(1, 6)), sphere_frame_size=int(rng.integers(1, 6)),
               edge_down_sample_submap=vox, ground_down_sample_submap=float(rng.choice([vox, vox * 1.5])), ground_down_sample=vox)
    H = reg.HipRegistration()
    hcfg = reg.default_submap_config(**cfg)
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    sizes = [0, 1, 2, 7, 60, 700, 6000]
    T = np.eye(4)
    inited = False
    for f in range(int(rng.integers(3, 9))):
        finite_only = not inited
        cl = []
        for k in range(4):
            c = rand_cloud(rng, int(rng.choice(sizes)), extent, vox)
            if k < 2 or finite_only:              # planar / sphere selections (a kd-tree is built over them) and the un-cropped first frame
                c = c[np.isfinite(c).all(1)]
            cl.append(np.ascontiguousarray(c))
        if not inited:
            ra, rb = status(lambda: H.submap_init(*cl, cfg=hcfg)), status(B.init, *cl)
            assert (ra != 0) == (rb != 0), ("init", t, f, ra, rb, cfg)
            inited = ra == 0
        else:
            T = T @ ss._se3_exp((rng.uniform(0, 2), rng.normal(0, 0.2), rng.normal(0, 0.05), rng.normal(0, 0.02), rng.normal(0, 0.02),
                                 rng.normal(0, 0.3)))
            ra, rb = status(H.submap_update, T, *cl), status(B.update, T, *cl)
            assert (ra != 0) == (rb != 0), ("update", t, f, ra, rb, cfg)
            if ra != 0:
                break                             # (a refused update ends the reference's run as well)
        if inited:
            for k in range(4):
                a, b = H.get_target(k), B.get(k)
                assert a.shape == b.shape, ("submap", t, f, k, a.shape, b.shape, cfg)
                assert np.array_equal(a, b, equal_nan=True), ("submap", t, f, k, cfg)
    H.close()


def feature_trial(rng, t):
    n = int(rng.choice([0, 1, 2, 19, 20, 21, 300, 3000, 12000]))
    if n == 0:
        p = np.zeros((0, 3))
    else:
        base = ss.feature_cloud(int(rng.integers(0, 1000)), n=max(n, 60))[:n]
        mode = rng.integers(0, 4)
        if mode == 1:
            base = base[rng.integers(0, n, n)]                     # duplicates
        elif mode == 2:
            base = np.round(base / 0.05) * 0.05                    # a lattice: exact distance ties everywhere
        elif mode == 3 and n >= 10:
            base = base.copy(); base[rng.choice(n, 3, replace=False)] = np.nan
        p = f32(base)
    K = int(rng.choice([3, 5, 10, 19, 20]))
    over = dict(radius=float(rng.choice([0.0, 0.05, 0.2, 0.5, 3.0])), K=K, min_neigh=int(rng.integers(0, K + 1)),
                planar_num=int(rng.choice([0, 1, 50, 500, 100000])), sphere_num=int(rng.choice([0, 1, 30, 300, 100000])),
                cvr_scan=float(rng.uniform(0, 0.5)), cvr_submap=float(rng.uniform(0, 0.5)),
                planar_scan_thres=float(rng.uniform(0.3, 0.9)), planar_submap_thres=float(rng.uniform(0.3, 0.9)),
                planar_vertic_thres=float(rng.uniform(0.0, 0.5)))
    H = reg.HipRegistration()
    hc, oc = reg.default_feature_config(**over), ob.make_feature_config(**over)
    g, o = H.pca_info(p, hc), ob.pca_info(p, oc)
    for k in ("num_sum", "neigh", "flatness", "cvr", "sphericity", "normal"):
        assert np.array_equal(g[k], o[k], equal_nan=True), ("pca", t, k, n, over)
    for i, (a, b) in enumerate(zip(H.extract_planar_sphere(p, hc), ob.extract_planar_sphere(p, oc))):
        assert np.array_equal(a, b), ("lists", t, i, n, over, len(a), len(b))
    H.close()


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    for t in range(trials):
        submap_trial(np.random.default_rng([seed, t, 1]), t)
        feature_trial(np.random.default_rng([seed, t, 2]), t)
    print("rows sweep ok: %d submap sequences + %d feature clouds in %.1f s" % (trials, trials, time.time() - t0))


if __name__ == "__main__":
    main()
