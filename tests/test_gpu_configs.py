"""-m gpu: BASELINE.json configs[3] and configs[4] at their stated workloads, on the ONE GPU a test box has.

configs[3]  "Synthetic 1 M-correspondence frame sharded across 2/4/8 GPUs" -- here two ranks (two processes, two
            contexts) on cuda:0, the 1 M frame (synth.M1_SRC / M1_TGT, caps lifted): contiguous source blocks,
            replicated targets, the all-reduce of the 48-double normal-equation buffer carried by gloo through
            tloam_comm_init_callback, or exchanged by the library's own peer mailbox (tloam_comm_init_mailbox).  Pose, minimiser counters and the merged correspondence index lists must
            equal the single-rank solve of the same frame.
configs[4]  "8 independent KITTI scan-pairs, one per GPU" -- here >= 4 live contexts in one process on cuda:0, each
            with its own KITTI-density pair, their outer iterations interleaved and their scan_match calls
            issued from concurrent threads: every result bit-equal to the same pair run alone.
"""
import os
import socket
import threading

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pose_delta
from tloam_amd import synth

pytestmark = pytest.mark.gpu
BIG = 1 << 30


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _m1_cfg(reg):
    return reg.default_config(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)


def _worker_1m(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_gpu_multirank import init_comm
        from tloam_amd import registration as reg
        sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
        H = reg.HipRegistration(_m1_cfg(reg))
        init_comm(H, mode, rank, world)
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0, rc
        # this rank's block of every index list (contiguous source blocks -> the lists concatenate in rank order)
        idx = [H.get_correspondences(k, capacity=max(len(sc.source.cloud(k)), 1))["idx"] for k in range(4)]
        gathered = [None] * world
        dist.all_gather_object(gathered, [a.tolist() for a in idx])
        if rank == 0:
            q.put(dict(T=T, st={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items()}, idx=gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["callback", "mailbox"])
def test_config3_two_ranks_1m_frame_on_one_gpu(hip_module, mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_1m, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs: p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
    S = hip_module.HipRegistration(_m1_cfg(hip_module)); S.set_frames(sc.source, sc.target)
    rc, T1, st1 = S.scan_match(sc.T_pred)
    assert rc == 0
    assert sum(st1["n_corr"]) > 800_000, st1["n_corr"]         # the workload really is the 1 M-correspondence frame
    assert res["st"]["n_corr"] == st1["n_corr"]
    dt, dr = pose_delta(res["T"], T1)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)                    # different summation tree over the two halves only
    for key in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations"):
        assert res["st"][key] == st1[key], key
    for k in range(4):
        single = S.get_correspondences(k, capacity=len(sc.source.cloud(k)))["idx"]
        merged = np.concatenate([np.asarray(res["idx"][r][k], np.int32) for r in range(world)])
        assert np.array_equal(merged, single), k
    assert pose_delta(T1, sc.T_true)[0] < 5e-3


def _alone(reg, scene):
    H = reg.HipRegistration(reg.default_config())
    H.set_frames(scene.source, scene.target)
    rc, T, st = H.scan_match(scene.T_pred)
    assert rc == 0
    idx = [H.get_correspondences(k)["idx"].copy() for k in range(4)]
    w = [H.get_weights(k).copy() for k in range(4)]
    H.close()
    return T, st, idx, w


def _same(st_a, st_b):
    for k in ("outer_iterations", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "n_corr",
              "converged_early", "kind_cost", "mu", "solver_cost"):
        assert st_a[k] == st_b[k], k
    assert np.array_equal(st_a["se3"], st_b["se3"])


@pytest.mark.parametrize("n_ctx", [4, 8])
def test_config4_live_contexts_are_independent(hip_module, n_ctx):
    """n_ctx contexts alive on one device, each its own KITTI-density pair (seeds 0..n_ctx-1)."""
    reg = hip_module
    scenes = [synth.make_scene(seed=s, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT) for s in range(n_ctx)]
    ref = [_alone(reg, sc) for sc in scenes]
    ctxs = [reg.HipRegistration(reg.default_config()) for _ in scenes]
    for H, sc in zip(ctxs, scenes):
        H.set_frames(sc.source, sc.target)
    # (i) outer GNC iterations of the contexts interleaved on the calling thread
    for H, sc in zip(ctxs, scenes):
        assert H.sm_begin(sc.T_pred) == 0
    live = list(range(n_ctx))
    while live:
        for i in list(live):
            rc, done, st = ctxs[i].sm_outer()
            assert rc == 0
            if done:
                live.remove(i)
    for i, H in enumerate(ctxs):
        rc, T, st = H.sm_end()
        assert rc == 0 and np.array_equal(T, ref[i][0])
        _same(st, ref[i][1])
        for k in range(4):
            assert np.array_equal(H.get_correspondences(k)["idx"], ref[i][2][k])
            assert np.array_equal(H.get_weights(k), ref[i][3][k])
    # (ii) one thread per context, three frames each, all in flight together (ctypes releases the GIL)
    out, errs = [None] * n_ctx, []

    def run(i):
        try:
            for _ in range(3):
                rc, T, st = ctxs[i].scan_match(scenes[i].T_pred)
                assert rc == 0
            out[i] = (T, st)
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(n_ctx)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for i in range(n_ctx):
        assert np.array_equal(out[i][0], ref[i][0])
        _same(out[i][1], ref[i][1])
    for H in ctxs:
        H.close()
