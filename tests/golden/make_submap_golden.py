#!/usr/bin/env python3
"""Golden vectors for the submap path (tests/golden/submap_seq.npz) from the INDEPENDENT numpy restatement
(oracle/oracle_np.py: NpSubmap).  Run in the build container:  python tests/golden/make_submap_golden.py

Inputs: a 6-frame sequence of the four feature clouds + odometry poses, a config with a SMALL crop box so
that cropping is exercised.  Expected outputs: the four submap clouds after the first-frame branch and after
every update.  The reference holds no test or vector for this path: "parity unpinned" with respect to real
Open3D (voxel output order is unordered_map's there; first occurrence here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp  # noqa: E402
from tloam_amd import synth_submap as ss  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(planar_frame_size=3, sphere_frame_size=2, edge_crop_box_length=30.0, ground_crop_box_length=25.0,
           edge_down_sample_submap=0.3, ground_down_sample_submap=0.45, ground_down_sample=0.3)
N = (150, 40, 400, 700)
FRAMES = 6


def main():
    S = onp.NpSubmap(**CFG)
    out = {"cfg_keys": np.array(list(CFG.keys())), "cfg_vals": np.array([float(v) for v in CFG.values()]),
           "frames": np.int64(FRAMES)}
    for f in range(FRAMES):
        cl = ss.frame_clouds(11, f, n=N)
        T = ss.frame_pose(f, step=6.0, yaw_rate=0.05)      # big steps: points leave the crop box
        for name, c in zip(("planar", "sphere", "edge", "ground"), cl):
            out[f"f{f}_{name}"] = c
        out[f"f{f}_pose"] = T
        if f == 0:
            S.init(*cl)
        else:
            S.update(T, *cl)
        for k in range(4):
            out[f"f{f}_submap{k}"] = S.get(k)
    np.savez_compressed(os.path.join(HERE, "submap_seq.npz"), **out)
    print("submap_seq.npz:", {k: [len(out[f"f{f}_submap{k}"]) for f in range(FRAMES)] for k in range(4)})


if __name__ == "__main__":
    main()
