#!/usr/bin/env python3
"""Golden vectors for the PCA feature path (tests/golden/feature_small.npz) from the INDEPENDENT numpy
restatement (oracle/oracle_np.py: brute-force hybrid search + LAPACK eigh).  Run in the build container:
    python tests/golden/make_feature_golden.py
Inputs: a 1200-point scan-like cloud + config overrides.  Expected: per-point num_sum / neighbour lists (exact),
flatness / cvr / sphericity (to rounding: the C oracle and the device use a cyclic Jacobi), the four index
lists.  "parity unpinned": the reference holds no vector for this path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp  # noqa: E402
from tloam_amd import synth_submap as ss  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(planar_num=40, sphere_num=3)


def main():
    p = ss.feature_cloud(21, n=1200)
    (ps, pm, s_scan, s_sub), info = onp.extract_planar_sphere(p, **CFG)
    np.savez_compressed(os.path.join(HERE, "feature_small.npz"), cloud=p, cfg_keys=np.array(list(CFG.keys())),
                        cfg_vals=np.array([float(v) for v in CFG.values()]), flatness=info["flatness"],
                        cvr=info["cvr"], sphericity=info["sphericity"], num_sum=info["num_sum"], neigh=info["neigh"],
                        planar_scan=ps, planar_submap=pm, sphere_scan=s_scan, sphere_submap=s_sub)
    print("feature_small.npz:", len(p), "points;", [len(x) for x in (ps, pm, s_scan, s_sub)])


if __name__ == "__main__":
    main()
