"""Generates tests/golden/tloam_00_ego_motion.npz from the reference's published KITTI-00 trajectory
(/root/reference/doc/tloam_00.txt, KITTI odometry format, camera axes): the per-frame ego-motion
T_{k-1}^{-1} T_k converted to velodyne axes (x forward, y left, z up), as se(3) vectors in the reference's
(translation, rotation) order, float64.  SURVEY 8(d) configs 1/2 drive the synthetic sequence with exactly these
relative poses and the constant-velocity prediction built from them (front_end.cpp:329-330).

Run in the build container only (the reference is not present on the GPU box):
    python tests/golden/make_tloam00_ego_motion.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tloam_amd import kitti_io as kio          # noqa: E402
from oracle import oracle_np as onp            # noqa: E402  (se3 log of the independent numpy restatement)

SRC = "/root/reference/doc/tloam_00.txt"


def main():
    poses = kio.read_poses(SRC)                      # (n, 4, 4), camera axes: x right, y down, z forward
    rel_cam = kio.relative_poses(poses)              # (n - 1, 4, 4)
    # velodyne <- camera axes (pure axis permutation; the cm-level lever arm of the calibration does not matter for
    # a synthetic replay): x_cam = -y_velo, y_cam = -z_velo, z_cam = x_velo
    P = np.array([[0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    rel = np.einsum("ij,njk,kl->nil", np.linalg.inv(P), rel_cam, P)
    se3 = np.array([onp.se3_log(T) for T in rel])
    out = os.path.join(HERE, "tloam_00_ego_motion.npz")
    np.savez_compressed(out, se3=se3, first_pose_lines=np.loadtxt(SRC, max_rows=3))
    step = np.linalg.norm(se3[:, :3], axis=1)
    print(out, se3.shape, "step m: mean %.3f max %.3f" % (step.mean(), step.max()),
          "yaw rate mrad/frame: mean |.| %.2f" % (1e3 * np.abs(se3[:, 5]).mean()))


if __name__ == "__main__":
    main()
