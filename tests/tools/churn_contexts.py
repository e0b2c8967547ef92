#!/usr/bin/env python3
"""TEST TOOL (GPU box): contexts created and destroyed in quick succession, each handed the device memory of a dead one.

The round-5 bug this hunts for relatives of: a new context finding valid-looking hand-over rows of a dead context in its recycled
row buffer (tests/test_gpu_parity.py::test_contexts_following_each_other_never_see_each_others_rows runs twenty successions; this
runs as many as asked for, over scenes of different sizes and through BOTH drivers of the outer loop, and also keeps a second,
long-lived context solving between the successions so that allocations interleave).  Every result must reproduce the first pass
over its scene bit for bit (one-call driver) / to 1e-12 (stepwise driver: different finish kernel, same sums); no call may fail or
take anywhere near the second a timed-out hand-over costs.

    python tests/tools/churn_contexts.py [successions=400]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tloam_amd import registration as reg, synth  # noqa: E402


def pose_delta(A, B):
    D = np.linalg.inv(A) @ B
    R = D[:3, :3]
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * 0.5
    return float(np.linalg.norm(D[:3, 3])), float(np.arctan2(np.linalg.norm(w), (np.trace(R) - 1.0) * 0.5))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    scenes = [synth.make_scene(seed=41), synth.make_scene(seed=33),
              synth.make_scene(seed=7, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT),
              synth.make_scene(seed=8, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT, pred_err=(0.05, -0.02, 0.01, 0.004, -0.002, 0.006)),
              synth.make_scene(seed=9, n_src=(40_000, 50_000, 35_000, 8_000), n_tgt=(30_000, 30_000, 20_000, 5_000))]
    keys = ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations", "n_corr", "converged_early")

    def counters(st):
        return tuple(int(x) for k in keys for x in np.atleast_1d(st[k]))

    want = []
    for sc in scenes:
        H = reg.HipRegistration()
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0, rc
        want.append((T.copy(), counters(st)))
        H.close()
    keeper = reg.HipRegistration()
    keeper.set_frames(scenes[2].source, scenes[2].target)
    worst = 0.0
    t_start = time.time()

    def footprint():
        """(free device memory, resident host memory of this process) in MB"""
        import resource
        import torch
        free, _ = torch.cuda.mem_get_info(0)
        return free / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0

    base = None
    for i in range(n):
        if i == min(200, n // 4):     # (allocator pools, the HIP runtime's own caches and the largest scene have been seen by now)
            base = footprint()
        k = (i * 7 + i // 5) % len(scenes)
        sc, (T_want, c_want) = scenes[k], want[k]
        H = reg.HipRegistration()
        H.set_frames(sc.source, sc.target)
        reps = 1 + i % 3
        for r in range(reps):
            t0 = time.perf_counter()
            if (i + r) % 2 == 0:
                rc, T, st = H.scan_match(sc.T_pred)
                assert rc == 0, (i, r, rc)
                assert T.tobytes() == T_want.tobytes() and counters(st) == c_want, ("one-call", i, r, k)
            else:
                assert H.sm_begin(sc.T_pred) == 0
                done = False
                while not done:
                    rc, done, st = H.sm_outer()
                    assert rc == 0, (i, r, rc)
                rc, T, st = H.sm_end()
                assert rc == 0 and counters(st) == c_want, ("stepwise", i, r, k)
                dt, dr = pose_delta(T, T_want)
                assert dt < 1e-12 and dr < 1e-12, ("stepwise", i, r, k, dt, dr)
            worst = max(worst, time.perf_counter() - t0)
        if i % 4 == 0:   # the long-lived context between two short-lived ones
            rc, T, st = keeper.scan_match(scenes[2].T_pred)
            assert rc == 0 and T.tobytes() == want[2][0].tobytes() and counters(st) == want[2][1], ("keeper", i)
        H.close()
    keeper.close()
    if base is not None:
        end = footprint()
        print("footprint after %d successions: device free %.0f -> %.0f MB, host peak RSS %.0f -> %.0f MB" % (n, base[0], end[0], base[1], end[1]))
        assert base[0] - end[0] < 256.0, "device memory is leaking: %.0f MB in %d successions" % (base[0] - end[0], n - min(200, n // 4))
        assert end[1] - base[1] < 256.0, "host memory is leaking: %.0f MB" % (end[1] - base[1])
    print("churn ok: %d successions over %d scenes in %.1f s, slowest call %.1f ms" % (n, len(scenes), time.time() - t_start, worst * 1e3))
    assert worst < 0.5, worst


if __name__ == "__main__":
    main()
