"""-m gpu: the sharded (multi-rank) path on ONE GPU.  Two processes each own a context on cuda:0 with a
contiguous block of the source points; the sum all-reduce of the 48-double normal-equation buffer (and of
the cap / cost-sum side buffers) is supplied through tloam_comm_init_callback and carried by gloo, or done by
the library's own one-shot peer exchange (tloam_comm_init_mailbox: buffers shared through HIP IPC, every rank
stores its 48 doubles into every rank's buffer, the step kernel adds them in rank order).  The
pose, the per-iteration bookkeeping and the correspondence counts must equal the single-rank solve.
(RCCL refuses two ranks on one device, so the native ncclAllReduce path is exercised with nranks = 1.)"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pose_delta
from tloam_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_allreduce():
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def allreduce(dev_ptr, count, stream):
        assert hip.hipStreamSynchronize(stream) == 0
        host = np.zeros(count)
        assert hip.hipMemcpy(host.ctypes.data, dev_ptr, 8 * count, 2) == 0    # D2H
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert hip.hipMemcpy(dev_ptr, host.ctypes.data, 8 * count, 1) == 0    # H2D
        return 0
    return allreduce


def init_comm(H, mode, rank, world):
    """callback: sum all-reduce through host memory + gloo.  mailbox: the one-shot peer exchange -- every rank exports
    its buffer (HIP IPC), the handles are all-gathered (here through gloo), every rank maps its peers."""
    if mode in ("mailbox", "mailbox_fused"):
        handles = [None] * world
        dist.all_gather_object(handles, H.comm_mailbox_export())
        H.comm_init_mailbox(rank, world, handles)
    else:
        H.comm_init_callback(rank, world, _make_allreduce())


def _worker(rank, world, port, q, caps, mode="callback"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "mailbox_fused":   # sweep + exchange + step of a sharded GN iteration in ONE launch (k3_sweep_step); read at create
        os.environ["TLOAM_FUSED_LARGE"] = "1"
    try:
        from tloam_amd import registration as reg
        sc = synth.make_scene(seed=31)
        H = reg.HipRegistration(reg.default_config(**caps))
        init_comm(H, mode, rank, world)
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0
        idx = [H.get_correspondences(k)["idx"].tolist() for k in range(4)]
        # pre-built sets, sharded
        sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130)
        P = reg.HipRegistration(); init_comm(P, mode, rank, world)
        for rt in range(3):
            P.set_correspondences(rt, *sets[rt])
        Hm, g, cost = P.accumulate(x_eval)
        x, pst = P.solve(x_eval)
        gathered = [None] * world
        dist.all_gather_object(gathered, idx)
        poses = [None] * world      # every rank's result, bit for bit: the exchange is folded in rank order on every rank
        dist.all_gather_object(poses, (T.tobytes(), np.asarray(x).tobytes(), [int(st[k]) for k in ("gn_evaluations", "gn_iterations", "accepted_steps")]))
        if rank == 0:
            q.put(dict(T=T, st={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in st.items()},
                       idx=gathered, H=Hm, g=g, cost=cost, x=x, all_ranks_identical=all(p == poses[0] for p in poses)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["callback", "mailbox", "mailbox_fused"])
@pytest.mark.parametrize("caps", [{}, dict(planar_maxnum=90, ground_maxnum=130, edge_maxnum=70, sphere_maxnum=25)],
                         ids=["default_caps", "caps_bind_across_ranks"])
def test_two_ranks_on_one_gpu_match_single_rank(hip_module, caps, mode):
    _ranks_on_one_gpu_match_single_rank(hip_module, caps, mode, world=2)


@pytest.mark.parametrize("mode", ["callback", "mailbox"])
def test_eight_ranks_on_one_gpu_match_single_rank(hip_module, mode):
    """BASELINE.json configs[3] is eight ranks; nothing but this test has ever run more than two real-HIP ranks.  Eight
    processes on the one device of a test box, the Frame cut by tloam_shard_ranges_frame (a rank holds one or two kinds and builds
    only those kinds' search grids), caps binding on all four kinds (the cap prefix runs over seven lower ranks), the mailbox's
    two-parity protocol with seven peers / the caller's all-reduce: pose within 1e-9 of the single-rank solve, counters and the
    merged index lists identical, and EVERY rank returns the same bits (the 48 doubles are folded in rank order on every rank)."""
    caps = dict(planar_maxnum=90, ground_maxnum=130, edge_maxnum=70, sphere_maxnum=25)
    _ranks_on_one_gpu_match_single_rank(hip_module, caps, mode, world=8)


def _ranks_on_one_gpu_match_single_rank(hip_module, caps, mode, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, caps, mode)) for r in range(world)]
    for p in procs: p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    sc = synth.make_scene(seed=31)
    S = hip_module.HipRegistration(hip_module.default_config(**caps)); S.set_frames(sc.source, sc.target)
    rc, T1, st1 = S.scan_match(sc.T_pred)
    dt, dr = pose_delta(res["T"], T1)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)
    assert res["all_ranks_identical"]
    assert res["st"]["n_corr"] == st1["n_corr"]
    if caps:
        assert st1["n_corr"] == [caps["planar_maxnum"], caps["ground_maxnum"], caps["edge_maxnum"], caps["sphere_maxnum"]]   # every cap binds
    for key in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations"):
        assert res["st"][key] == st1[key], key
    for k in range(4):
        merged = sum((res["idx"][r][k] for r in range(world)), [])
        assert merged == S.get_correspondences(k)["idx"].tolist(), k
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130)
    P = hip_module.HipRegistration()
    for rt in range(3):
        P.set_correspondences(rt, *sets[rt])
    Hm, g, cost = P.accumulate(x_eval)
    np.testing.assert_allclose(res["H"], Hm, rtol=1e-12, atol=1e-12 * np.abs(Hm).max())
    np.testing.assert_allclose(res["g"], g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
    x, _ = P.solve(x_eval)
    np.testing.assert_allclose(res["x"], x, atol=1e-10)


def test_native_rccl_single_rank(hip_module):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce through the dlopen'ed librccl with nranks = 1:
    proves the library resolves and the call signatures (128-byte id by value, ncclFloat64, ncclSum) hold."""
    uid = hip_module.rccl_unique_id()
    assert len(uid) == 128
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130)
    A = hip_module.HipRegistration(); B = hip_module.HipRegistration()
    A.comm_init_rccl(0, 1, uid)
    for R in (A, B):
        for rt in range(3):
            R.set_correspondences(rt, *sets[rt])
    xa, _ = A.solve(x_eval); xb, _ = B.solve(x_eval)
    assert np.array_equal(xa, xb)


def _worker_second_setup(rank, world, port, q):
    """a SECOND tloam_comm_init_mailbox on a context.  The first session ends after ONE exchange
    (a single tloam_accumulate on a pre-built set = exchange id 1, parity 1 of the buffers), then every rank exports and
    sets up again and solves: the new session counts its exchange ids from 1, so the old session's row of id 1 -- another
    evaluation point's sums -- is exactly what a buffer that was not cleared would hand to new exchange 1 without waiting."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tloam_amd import registration as reg
        sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130)
        P = reg.HipRegistration()
        init_comm(P, "mailbox", rank, world)
        for rt in range(3):
            P.set_correspondences(rt, *sets[rt])
        far = np.asarray(x_eval) + np.array([0.5, -0.3, 0.2, 0.02, -0.01, 0.03])
        P.accumulate(far)                       # session 1: ONE exchange, at a point the second session never visits
        dist.barrier()
        init_comm(P, "mailbox", rank, world)    # session 2 on the same context and the same buffers
        assert P.info()["nranks"] == world and P.info()["comm_mode"] == 3
        Hm, g, cost = P.accumulate(x_eval)      # new exchange 1
        x, pst = P.solve(x_eval)
        # a set-up that FAILS half way leaves a single-rank context behind, not mailbox kernels over unmapped peers
        bad = [b"\0" * 64] * world
        try:
            P.comm_init_mailbox(rank, world, bad)
            failed = False
        except reg.TloamHipError:
            failed = True
        after = P.info()
        rows = [None] * world
        dist.all_gather_object(rows, (Hm.tobytes(), np.asarray(x).tobytes()))
        if rank == 0:
            q.put(dict(H=Hm, g=g, cost=cost, x=x, identical=all(r == rows[0] for r in rows), failed=failed, after=after))
    finally:
        dist.destroy_process_group()


def test_second_mailbox_setup_on_a_context_does_not_see_the_first_sessions_rows(hip_module):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_second_setup, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=3001, n_line=777, n_point=130)
    P = hip_module.HipRegistration()
    for rt in range(3):
        P.set_correspondences(rt, *sets[rt])
    Hm, g, cost = P.accumulate(x_eval)
    np.testing.assert_allclose(res["H"], Hm, rtol=1e-12, atol=1e-12 * np.abs(Hm).max())
    np.testing.assert_allclose(res["g"], g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
    x, _ = P.solve(x_eval)
    np.testing.assert_allclose(res["x"], x, atol=1e-10)
    assert res["identical"]
    assert res["failed"] and res["after"]["comm_mode"] == 0 and res["after"]["nranks"] == 1 and res["after"]["rank"] == 0, res["after"]


@pytest.mark.parametrize("form", ["mailbox", "mailbox_fused", "rccl"])
def test_one_rank_loopback_runs_the_sharded_forms_bit_for_bit(hip_module, form):
    """A mailbox / RCCL set-up with nranks = 1 exchanges with itself (include/tloam_hip.h): every launch of the sharded forms
    runs -- fused sweep + post, gather + step, the side exchanges of the caps and the cost sums; for RCCL the all-reduce is a
    real ncclAllReduce on a one-rank communicator, whose own ncclCommCount is reported -- and the result is the single-rank
    forms' bit for bit.  (What bench.py's shard_size_iterations times; until round 6 a one-rank RCCL context never reached
    ncclAllReduce at all.)"""
    caps = dict(planar_maxnum=90, ground_maxnum=130, edge_maxnum=70, sphere_maxnum=25)
    sc = synth.make_scene(seed=31)
    S = hip_module.HipRegistration(hip_module.default_config(**caps)); S.set_frames(sc.source, sc.target)
    rc1, T1, st1 = S.scan_match(sc.T_pred)
    if form == "mailbox_fused":
        os.environ["TLOAM_FUSED_LARGE"] = "1"
    try:
        H = hip_module.HipRegistration(hip_module.default_config(**caps))
    finally:
        os.environ.pop("TLOAM_FUSED_LARGE", None)
    if form == "rccl":
        H.comm_init_rccl(0, 1, hip_module.rccl_unique_id())
    else:
        H.comm_init_mailbox(0, 1, [H.comm_mailbox_export()])
    info = H.info()
    assert info["loopback"] == 1 and info["nranks"] == 1 and info["comm_mode"] == (2 if form == "rccl" else 3)
    if form == "rccl":
        assert info["rccl_comm_count"] == 1 and info["rccl_comm_rank"] == 0
    H.set_frames(sc.source, sc.target)
    H.gn_iter_timer(reset=True)
    rc, T, st = H.scan_match(sc.T_pred)
    assert rc == 0 and rc1 == 0
    assert T.tobytes() == T1.tobytes()
    for key in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations", "gn_sweeps", "n_corr"):
        assert st[key] == st1[key], key
    for k in range(4):
        assert H.get_correspondences(k)["idx"].tolist() == S.get_correspondences(k)["idx"].tolist()
    us, n = H.gn_iter_timer()
    assert n > 0 and 1.0 < us / n < 500.0, (us, n)   # the device clocked the GN iterations of the sharded forms
    # the 1 M-class streaming forms as well (pre-built set: fused sweep + last-block fold + post | gather + step)
    # (490 k factors: a full-chip grid, i.e. WIDE blocks -- the last block of eight waves folds with its first four)
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=400_001, n_line=77_777, n_point=13_000)
    P0 = hip_module.HipRegistration()
    if form == "mailbox_fused":
        os.environ["TLOAM_FUSED_LARGE"] = "1"
    try:
        P = hip_module.HipRegistration()
    finally:
        os.environ.pop("TLOAM_FUSED_LARGE", None)
    if form == "rccl":
        P.comm_init_rccl(0, 1, hip_module.rccl_unique_id())
    else:
        P.comm_init_mailbox(0, 1, [P.comm_mailbox_export()])
    for R in (P0, P):
        for rt in range(3):
            R.set_correspondences(rt, *sets[rt])
    assert P.info()["k3_single"] == 0 and P.info()["k3_wide"] == (1 if P.info()["device_cus"] <= 256 else 0)
    xa, sa = P0.solve(x_eval); xb, sb = P.solve(x_eval)
    assert np.array_equal(xa, xb) and sa["gn_evaluations"] == sb["gn_evaluations"]


def _worker_timeout(rank, world, port, q, frames_before_silence=0):
    """the last rank sets the mailbox up, solves `frames_before_silence` frames with the others and then never enters the next
    solve: the other ranks' exchanges must give up after their bounded wait (the peer's buffer stays mapped: a peer whose process is GONE would take
    its memory with it, and a store into it is a GPU page fault -- not something to provoke on a shared box)"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tloam_amd import registration as reg
        sc = synth.make_scene(seed=31)
        H = reg.HipRegistration()
        init_comm(H, "mailbox", rank, world)
        H.set_frames(sc.source, sc.target)
        for _ in range(frames_before_silence):
            rc, T, st = H.scan_match(sc.T_pred)
            assert rc == 0
        if rank != world - 1:      # the last rank goes silent; every other one must give up by itself
            t0 = time.perf_counter()
            rc, T, st = H.scan_match(sc.T_pred)
            dt = time.perf_counter() - t0
            msg = H.L.tloam_last_error(H.h).decode()
            q.put(dict(rank=rank, rc=rc, seconds=dt, msg=msg))
        dist.barrier()      # the silent rank stays alive (its buffer mapped) until the others have given up
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("frames_before_silence", [0, 2], ids=["from_the_start", "after_two_frames"])
def test_mailbox_exchange_times_out_when_a_peer_never_posts(hip_module, frames_before_silence):
    _mailbox_time_out(frames_before_silence, world=2)


def test_mailbox_exchange_times_out_with_seven_live_peers_and_one_silent(hip_module):
    _mailbox_time_out(frames_before_silence=1, world=8)


def _mailbox_time_out(frames_before_silence, world):
    """The bounded wait of the peer mailbox: a rank whose peer never posts -- from the start, or after
    two frames solved together (exchange counters and parities in mid-stream) -- does not hang: every exchange gives up after
    ~2 s, the Solve is stopped and scan_match returns TLOAM_E_RCCL with a message that says why -- on EVERY live rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_timeout, args=(r, world, port, q, frames_before_silence)) for r in range(world)]
    for p in procs: p.start()
    results = [q.get(timeout=180) for _ in range(world - 1)]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    assert sorted(r["rank"] for r in results) == list(range(world - 1))
    for res in results:
        assert res["rc"] == -5, res                                  # TLOAM_E_RCCL
        assert "timed out" in res["msg"] and "peer" in res["msg"], res
        assert 1.5 < res["seconds"] < 60.0, res                      # bounded: a few exchanges of ~2 s each, not forever


def _worker_peer_dies(rank, world, port, q, alive, entered):
    """callback contexts: rank 1's process EXITS at its 6th all-reduce (no clean-up, no goodbye).  What a collective layer does
    with a dead peer is its own business (gloo may abort the survivor, RCCL may hang): the caller's all-reduce here has the
    failure detector a production one needs -- a heartbeat (`alive`, `entered`: shared words) consulted before it commits to the
    collective -- and reports failure to the library, which hands TLOAM_E_RCCL to its caller instead of a result."""
    import time
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tloam_amd import registration as reg
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    calls = [0]

    def allreduce(dev_ptr, count, stream):
        calls[0] += 1
        if rank == 1 and calls[0] == 6:
            alive.value = 0
            os._exit(0)                                   # dies at the exchange
        entered[rank] = calls[0]
        t0 = time.perf_counter()
        while entered[1 - rank] < calls[0]:               # the peer has not arrived at this exchange yet
            if not alive.value or time.perf_counter() - t0 > 30.0:
                return 1                                  # ... and never will
            time.sleep(0.0005)
        try:
            assert hip.hipStreamSynchronize(stream) == 0
            host = np.zeros(count)
            assert hip.hipMemcpy(host.ctypes.data, dev_ptr, 8 * count, 2) == 0
            t = torch.from_numpy(host)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            assert hip.hipMemcpy(dev_ptr, host.ctypes.data, 8 * count, 1) == 0
            return 0
        except Exception:                                 # noqa: BLE001 -- report, never raise through the C ABI
            return 1

    sc = synth.make_scene(seed=31)
    H = reg.HipRegistration()
    H.comm_init_callback(rank, world, allreduce)
    H.set_frames(sc.source, sc.target)
    t0 = time.perf_counter()
    rc, T, st = H.scan_match(sc.T_pred)
    dt = time.perf_counter() - t0
    msg = H.L.tloam_last_error(H.h).decode()
    H.close()
    # the survivor is still in working order: a fresh single-rank context solves the frame
    S = reg.HipRegistration(); S.set_frames(sc.source, sc.target)
    rc2, T2, st2 = S.scan_match(sc.T_pred)
    S.close()
    q.put(dict(rc=rc, seconds=dt, msg=msg, calls=calls[0], rc_after=rc2))
    q.close(); q.join_thread()                            # (the queue's feeder thread has written the message out)
    os._exit(0)                                           # (the process group lost a member: no orderly shutdown to wait for)


def test_callback_exchange_reports_a_peer_that_dies_inside_it(hip_module):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    alive = ctx.Value("i", 1)
    entered = ctx.Array("i", [0, 0])
    port = _free_port()
    procs = [ctx.Process(target=_worker_peer_dies, args=(r, world, port, q, alive, entered)) for r in range(world)]
    for p in procs: p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
    assert res["rc"] == -5, res                                  # TLOAM_E_RCCL
    assert "allreduce callback failed" in res["msg"], res
    assert res["calls"] >= 6 and res["seconds"] < 60.0, res
    assert res["rc_after"] == 0, res


@pytest.mark.parametrize("n", [2, 8])
def test_bench_n_gpus_path_runs_on_one_device(hip_module, n):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process), with both ranks on
    the one GPU a test box has (TLOAM_BENCH_ONE_DEVICE=1: launcher collectives over gloo): the replica headline aggregates
    over the ranks, and the sharded 1 M frame reports BOTH exchanges -- the mailbox with its figures, RCCL with the reason it
    cannot run two ranks on one device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TLOAM_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "6", "--warmup", "3",
           "--m1-steps", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    # stdout of the whole job is the line and nothing else (round 6: RCCL's version banner sat in a C stdio buffer until exit and
    # came out BEHIND the line; bench.py now keeps descriptor 1 for the line alone, on every rank)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-1500:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == n and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_step"] == n
    sh = d["sharded_1m"]
    assert sh["n_gpus"] == n and "mailbox" in sh and "mailbox_fused" in sh and "rccl" in sh
    assert "one device" in sh["rccl"].get("error", ""), sh["rccl"]
    # every form that ran is SELF-VERIFIED against the one-rank solve of the same frame (rank 0, same run): pose within 1e-9,
    # every counter equal, every rank the same bits, and the context itself reports N ranks in the exchange
    for m in ("mailbox", "mailbox_fused"):
        f = sh[m]
        assert f["ms_per_frame"] > 0 and f["per_sweep_us"]["sweep_plus_exchange"] > 0, f
        assert f["verified"] and "error" not in f, f
        assert f["pose_delta_vs_one_rank"]["dt_m"] < 1e-9 and f["pose_delta_vs_one_rank"]["dR_rad"] < 1e-9, f
        assert f["ranks_bit_identical"] and f["counters_equal_one_rank"] and f["ranks_in_exchange"] == n, f
        assert f["context"]["comm_mode"] == 3 and f["context"]["nranks"] == n and f["gn_iteration_us"] > 0, f
    assert sh["fastest_exchange"] in ("mailbox", "mailbox_fused") and sh["ms_per_frame"] == sh[sh["fastest_exchange"]]["ms_per_frame"]
    assert sh["pose_err_vs_truth_m"] < 1e-2 and sh["one_rank"]["ms_per_frame"] > 0
    # ... and the verdict is there as TOP-LEVEL scalars, where the driver's parser keeps it
    assert d["sharded_1m_verified"] is True and d["sharded_1m_ms_per_frame"] == sh["ms_per_frame"]
    assert d["sharded_1m_speedup"] == sh["speedup_vs_one_gpu_frame"] > 0
    assert d["sharded_1m_pose_delta"] < 1e-9 and d["sharded_1m_ranks_bit_identical"] is True and d["sharded_1m_counters_equal"] is True
    assert d["sharded_1m_exchange_adds_us"] == sh["per_sweep_us"]["exchange_adds"]
    assert d["sharded_1m_predicted_speedup"] == {2: 1.13, 8: 1.54}[n] and d["rccl_nranks"] is None
    assert d["sharded_1m_mailbox_fused_ms_per_frame"] > 0 and d["sharded_1m_gn_iteration_us"] > 0


def test_bench_sharded_deadline_prints_the_line(hip_module):
    """The sharded 1 M frame is on a deadline (`--sharded-timeout`): with one that cannot be met, `bench.py --gpus 2` still prints
    its line -- the replica headline, and sharded_1m saying what was cut off -- and every rank leaves with exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TLOAM_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--m1-steps", "2", "--sharded-timeout", "0.05"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    sh = d["sharded_1m"]
    assert "cut off" in sh["note"] and "ms_per_frame" not in sh
    assert "did not finish" in sh["mailbox"]["error"] and "did not finish" in sh["rccl"]["error"]
    assert d["sharded_1m_verified"] is False and d["sharded_1m_ms_per_frame"] is None


def test_bench_line_survives_a_rank_whose_sharded_child_dies(hip_module):
    """The sharded 1 M frame runs in a CHILD process of every rank (bench.py sharded_in_children): its exchange forms have never met
    two devices, and a GPU fault ends the process it happens in.  Here rank 1's child aborts (TLOAM_BENCH_CHILD_CRASH, the way the
    runtime ends a process on a memory-access fault): the parents notice within a second, end the other child, and the line goes
    out -- replica headline intact, `sharded_1m` saying what happened -- with exit code 0, long before any deadline."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TLOAM_BENCH_ONE_DEVICE="1", TLOAM_BENCH_CHILD_CRASH="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
           "--m1-steps", "2"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-1500:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["frames_per_step"] == 2
    assert d["sharded_1m_verified"] is False and d["sharded_1m_ms_per_frame"] is None
    assert "ended abnormally" in d["sharded_1m"]["note"] and "ended abnormally" in (d["sharded_1m_error"] or "")
    assert took < 240.0, took      # (not the 300 s of --sharded-timeout, nor the parents' own limit)
