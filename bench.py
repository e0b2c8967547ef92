#!/usr/bin/env python3
"""bench.py -- T-LOAM pose-optimisation hot path on MI355X: Gauss-Newton iterations/s + ms/frame.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is ONE full `scan_match` (LocalRegistration::scanMatching, registration.cpp:879-1133 -- the
bracket the reference times at front_end.cpp:320-322: grid build, 4 outer GNC iterations of
[correspondence search + Ceres-configured solve + weight update]) over one synthetic frame pair
whose eight feature clouds are already resident in HBM.

Headline workload (BASELINE.json configs[1], the configuration the metric is quoted on): a synthetic frame pair
at KITTI-00 scan density -- ~10 k source / ~85 k target feature points, the reference's caps.  `value` =
EXECUTED GN iterations (residual+Jacobian sweep + 6x6 dogleg step + pose update) per second of wall time, whole
job; `ms_per_step` = ms/frame.  The minimiser's evaluations of a point bit-identical to the one just swept
(rejected steps retried in a halved trust region) are served from the totals in hand and are NOT counted in
`value`; they appear as `solver_evaluations_per_frame`.

`roofline` is the residual/Jacobian kernel K3 on the 1 M-correspondence frame (configs[2], "HBM-roofline
characterisation run" -- the configuration the north-star states its >= 70 % target on), timed in the same command
over its own region (`--m1-steps` frames); the same kernel family on the headline frames is launch-latency bound
(442 KB per launch) and is reported beside it as `roofline.headline_workload_k3`.

N > 1 (one process per GPU).  The reference runs ONE scanMatching per LiDAR frame and frames are independent
units, so the job is configs[4]: every rank registers its OWN frame pair (seed + rank) with no data-path collective
-- `"scaling": "weak"`, value = sum of the ranks' GN iterations / max-over-ranks time.  The path's one real exchange
step -- ONE 1 M frame's source points sharded in contiguous index blocks over the N GPUs, targets replicated, one
RCCL all-reduce of the 6x6/6x1 normal equations (48 doubles) over xGMI per GN sweep (configs[3]) -- is timed in
the same run and reported as "sharded_1m" (strong scaling of one frame; the dependent all-reduces bound it, see
DESIGN.md section 6).

At N = 1 the line also carries: "m1_frame" (the 1 M frame: ms/frame, GN iter/s), "kitti_sequence" (200 DISTINCT
KITTI-density frames: mean / p50 / p99, PCIe-inclusive figure), "adjacent_rows" (device submap update, PCA feature
extraction), "odometry_loop" (set source + scan matching + submap update over a consistent synthetic street),
"multi_stream" (three independent frame streams sharing the one GPU: aggregate frames/s; not part of `value`).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (the residual/Jacobian kernel K3, HIP events
around every K3 launch of the timed region, algorithmic bytes 72/88/64 B per plane/line/point
correspondence) and "cpu_baseline" (the C oracle -- a port, NOT Ceres -- on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory (the ROCm 7 default here; with =0 the KITTI-density frame goes from 0.56 to
# 0.69 ms): set before the HIP runtime initialises so a different default cannot silently change the numbers
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per K3 launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of
# THIS command, summarised by scripts/pmc_summary.py with the gfx950 x2 FETCH_SIZE correction).  PMC counters
# cannot be read from inside the process, so the line quotes the committed summary and says so.
def _first_existing(*names):
    paths = [os.path.join(ROOT, "profiles", n) for n in names]
    return next((p for p in paths if os.path.exists(p)), paths[0])


PMC_SUMMARY = _first_existing("r06_pmc_k3_prebuilt.json", "r05_pmc_k3_prebuilt.json", "r04_pmc_k3_prebuilt.json")
PMC_SUMMARY_COLD = _first_existing("r06_pmc_k3_prebuilt_cold.json", "r05_pmc_k3_prebuilt_cold.json", "r04_pmc_k3_prebuilt_cold.json")


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(world):
    if world != 1 or not os.path.exists(PMC_SUMMARY):
        return None, None
    try:
        d = json.load(open(PMC_SUMMARY))
        now = _sha16(os.path.join(ROOT, "tloam_amd", "csrc", "tl_gn.hip"))
        det = {
            "source": "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; a committed "
                      "summary quoted here, not a counter read by this run)" % os.path.basename(PMC_SUMMARY),
            "fetch_bytes_per_launch": d["fetch_bytes_per_launch_all"],
            "write_bytes_per_launch": d["write_bytes_per_launch_all"],
            "working_sweeps_traffic": d["traffic_bytes_per_launch_working"],
            # the summary records the hash of the kernel source it was measured on: a K3 edited since then makes it stale
            "kernel_source_sha16": {"measured_on": d.get("kernel_source_sha16"), "now": now},
            "stale": d.get("kernel_source_sha16") != now}
        return float(d["traffic_bytes_per_launch_all"]), det
    except Exception:
        return None, None


def read_stream_bandwidth(reg, device):
    """The on-box ceiling the roofline kernel is held against besides the 8 TB/s data-sheet figure: a READ stream
    with K3's access pattern and nothing else (tloam_time_read_stream: eight fp64 streams, 16-byte loads, persistent waves, two
    blocks per CU).  `hbm`: 1.2 GB per pass, every byte from HBM; `l3`: the sweep's own 75 MB, Infinity-Cache resident across the
    passes as the contract set is across the launches of the characterisation run.  (Rounds 1-5 quoted a device-to-device COPY
    here -- half its bytes are writes, and K3 measured 1.19-1.27x of it: no ceiling for a read stream.)"""
    H = reg.HipRegistration(device=device)
    try:
        H.time_read_stream(1_200_000_000, 3)
        hbm = sorted(H.time_read_stream(1_200_000_000, 10) for _ in range(3))[1]
        H.time_read_stream(75_000_000, 10)
        l3 = sorted(H.time_read_stream(75_000_000, 40) for _ in range(3))[1]
    finally:
        H.close()
    return hbm, l3


def prebuilt_k3(reg, synth, device, scale=1):
    """SURVEY 8(d) config 3, K3-only variant: the pre-built 760k:200k:40k set (74.88 MB algorithmic bytes per
    evaluation, timing weights U(0,1) with 10 % exact zeros), >= 100 launches after 10 warm-ups, median of the
    per-batch means (one HIP event pair around each batch of 20 consecutive launches).
    scale = 4: the same mix four times as large (3.04 M : 0.8 M : 0.16 M, 299.5 MB per sweep) -- the working set of
    back-to-back sweeps then exceeds the 256 MiB Infinity Cache, every byte of every launch comes from HBM ("cold")."""
    n = (760_000 * scale, 200_000 * scale, 40_000 * scale)
    sets, _, x_eval = synth.make_prebuilt(seed=1, n_plane=n[0], n_line=n[1], n_point=n[2], weights="timing")
    H = reg.HipRegistration(device=device)
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    del sets
    batch = 20 if scale == 1 else 10
    H.time_accumulate(x_eval, 10)
    us = sorted(H.time_accumulate(x_eval, batch) for _ in range(6))
    H.close()
    med = 0.5 * (us[2] + us[3])
    alg = n[0] * 72.0 + n[1] * 88.0 + n[2] * 64.0
    return {"workload": "pre-built set, plane:line:point = %dk:%dk:%dk (SURVEY 8(d) config 3%s, K3 only)" %
                        (n[0] // 1000, n[1] // 1000, n[2] // 1000, "" if scale == 1 else " x%d" % scale),
            "launches": 6 * batch, "avg_launch_us": round(med, 3), "algorithmic_bytes_per_launch": alg, "bound": "hbm",
            "achieved": round(alg / (med * 1e-6) / 1e9, 1), "frac": round(alg / (med * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "contract: >= 70 %% <=> <= %.1f us" % (alg / 0.7 / HBM_PEAK_GBS / 1e3)}


def k1_roofline(H, world):
    """roofline_k1: the correspondence-search kernel (K1 + K2) on the state the 1 M frame left behind, 20 back-to-back
    launches between one HIP event pair (tloam_time_build).  ALGORITHMIC bytes per query: the 32-byte
    query record in, the k nearest 32-byte target records kept (5, sphere kind 1), the 64-byte raw record + 8-byte flag out."""
    try:
        H.time_build(3)
        us, nq = H.time_build(20)
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:160]}
    per_kind = [H._n.get(("s", k), 0) for k in range(4)]
    alg = sum(per_kind[k] * (32.0 + 5 * 32.0 + 72.0) for k in (0, 1, 2)) + per_kind[3] * (32.0 + 32.0 + 72.0)
    return {"kernel": "k_build_sorted<1>", "bound": "hbm", "launches": 20, "avg_launch_us": round(us, 2), "queries_per_launch": int(nq),
            "algorithmic_bytes_per_launch": alg, "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "exact k-NN over the cells of the 27-cell neighbourhood the search ball can reach: ~25 candidate records per "
                    "query (the busiest lane of a wave: ~40), 32-byte gathers; knock-out builds (scripts/k1_time.py): ~28 us "
                    "set-up (query, nine cell-table rows, record out), ~82 us candidate walk, ~19 us per-query plane fit / "
                    "eigen solve -- bound by the walk's gather traffic and its lane imbalance, not by HBM bytes (DESIGN.md section 5)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="kitti", choices=["kitti", "m1"],
                    help="headline frame pair: kitti = KITTI-00 scan density (the metric's configuration), m1 = 1 M correspondences")
    ap.add_argument("--no-m1", action="store_true", help="skip the 1 M-correspondence roofline-characterisation block")
    ap.add_argument("--m1-steps", type=int, default=10, help="timed frames of the 1 M block")
    ap.add_argument("--sharded-timeout", type=float, default=300.0,
                    help="N > 1: seconds the sharded 1 M frame (both exchanges) may take before the line is printed without it")
    ap.add_argument("--sharded-child", action="store_true",
                    help="(internal) this process is a rank's CHILD that runs the sharded 1 M frame and nothing else, see sharded_in_children")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kitti", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side measurements of the roofline block (copy bandwidth, pre-built K3 set): the PMC passes "
                         "use it so that every K3 launch they count belongs to the frame workload")
    ap.add_argument("--kitti-frames", type=int, default=200, help="length of the KITTI-density sequence block")
    ap.add_argument("--loop-frames", type=int, default=150, help="length of the device odometry-loop block")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kitti-dir", default=None,
                    help="a KITTI odometry sequence directory (<dir>/velodyne/*.bin or <dir>/*.bin): replay it through "
                         "reader -> labeller -> device feature extraction -> scan matching -> device submap and write the "
                         "trajectory (--kitti-out); reported as \"kitti_replay\".  Skipped when the directory is absent.")
    ap.add_argument("--kitti-out", default=None, help="trajectory file of the replay (KITTI pose format; default <dir>/tloam_hip_poses.txt)")
    ap.add_argument("--kitti-max-frames", type=int, default=0)
    return ap.parse_args()


_LINE_FD = None


def claim_stdout():
    """From here on file descriptor 1 of this process carries ONE thing, the JSON line: the descriptor is kept aside for it and
    everything else that writes to stdout -- this script's own prints, and libraries that print through C stdio (RCCL's version
    banner sits in a stdio buffer until exit, i.e. it would come out BEHIND the line) -- goes to stderr."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    """the JSON line, whole, on the real stdout"""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)    # C stdio buffers of any library, to wherever descriptor 1 points now
    except OSError:
        pass
    sys.stdout.flush()
    data = (json.dumps(out) + "\n").encode()
    fd = _LINE_FD if _LINE_FD is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main():
    args = parse()
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # TLOAM_BENCH_FORCE_MULTI=1 (under torchrun, WORLD_SIZE=1): drive the N>1 code path -- process group,
    # replica aggregation and the RCCL-sharded block -- on a single-GPU box
    multi = world > 1 or os.environ.get("TLOAM_BENCH_FORCE_MULTI") == "1"

    import torch  # plumbing only: process group, barrier, device sync (imported first so that one HIP runtime is shared)
    from tloam_amd import registration as reg
    from tloam_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # TLOAM_BENCH_ONE_DEVICE=1 (development): all ranks on cuda:0, launcher collectives over gloo -- RCCL refuses two
    # ranks on one device; the library's own peer mailbox does not, so the N > 1 code path can be driven on a 1-GPU box
    one_dev = os.environ.get("TLOAM_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cdev = "cpu" if one_dev else "cuda"   # where the launcher's own collective tensors live
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    big = 1 << 30
    if args.sharded_child:
        # a rank's child (sharded_in_children): the sharded 1 M frame and nothing else; rank 0's child hands the result to its parent
        cfg = reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big)
        if os.environ.get("TLOAM_BENCH_CHILD_CRASH") == str(rank):   # test hook: this rank's child dies the way a GPU fault kills it
            os.abort()
        sharded = {}
        finished = within(lambda: sharded_frame(args, reg, synth, torch, dist, cfg, synth.M1_SRC, synth.M1_TGT, rank, world,
                                                local_rank, barrier, cdev, sharded), args.sharded_timeout)
        if rank == 0:
            emit({"sharded": sharded, "finished": bool(finished)})
        os._exit(0)    # (nothing to tear down in order: a stuck collective of a cut-off form may still hold a thread)

    # ---------------- workloads ----------------
    WL = {
        "kitti": dict(n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT, cfg=reg.default_config(),
                      name="synthetic KITTI-density sequence, every step a DISTINCT frame pair (9.4k src / 83.5k tgt pts per frame, "
                           "reference caps 2500/2000/1200/200; ego-motion + constant-velocity prediction from the reference's "
                           "KITTI-00 trajectory), all frames resident in HBM before the timed region -- BASELINE.json configs[1]"),
        "m1": dict(n_src=synth.M1_SRC, n_tgt=synth.M1_TGT,
                   cfg=reg.default_config(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big),
                   name="synthetic 1M-correspondence frame (1.0M src / 1.0M tgt pts, plane:line:point = 760k:200k:40k, "
                        "caps lifted) -- BASELINE.json configs[2]"),
    }

    def run_frames(wl, steps, warmup, seed):
        """One context, `warmup` untimed + `steps` timed scan_match calls between barriers, every input resident in HBM
        before the clock starts.  KITTI-density workload: every step is its OWN frame of the synthetic sequence (frames
        0 .. warmup + steps - 1, staged in the context's HBM frame store and activated by an O(1) tloam_frame_select) --
        the context sees a stream of distinct frames, as it does behind the reference's front end, so its learned sweep
        budgets are predictions from OTHER frames.  1 M workload: one frame pair (the characterisation frame).
        Returns the whole-job numbers (all-reduced) and rank 0's K3 event timings."""
        W = WL[wl]
        H = reg.HipRegistration(W["cfg"], device=local_rank)
        if wl == "kitti":
            scenes = [kitti_scene(synth, seed, f) for f in range(warmup + steps)]
            for f, sc in enumerate(scenes):
                H.set_frames(sc.source, sc.target)   # PCIe hand-over, outside the timed region
                H.frame_stash(f)
        else:
            scenes = [synth.make_scene(seed=seed, n_src=W["n_src"], n_tgt=W["n_tgt"])]
            H.set_frames(scenes[0].source, scenes[0].target)
        H.k3_timer(reset=True)                      # arm HIP event pairs around (every 3rd) K3 launch

        # The timed loop calls the C ABI itself -- tloam_frame_select + tloam_scan_match through ctypes, every argument object made
        # beforehand (the prediction of every frame as the 16 column-major doubles the entry point takes, one result buffer, one
        # tloam_stats) -- as a C++ host would: the Python class around them (numpy views, a dict of the stats per frame) costs ~3 us
        # of a 170 us frame (scripts/host_gap.py) and is the harness, not the library.
        import ctypes as C
        preds = []
        for sc in scenes:
            b = (C.c_double * 16)()
            np.frombuffer(b).reshape(4, 4).T[...] = sc.T_pred
            preds.append(b)
        res_buf = (C.c_double * 16)()
        c_stats = reg.Stats()
        c_stats_ref = C.byref(c_stats)
        c_select, c_scan_match, c_h = H.L.tloam_frame_select, H.L.tloam_scan_match, H.h
        select = wl == "kitti"

        def step(i):
            if select:
                c_select(c_h, i)
            rc = c_scan_match(c_h, preds[i % len(preds)], None, res_buf, None, 0, c_stats_ref)
            if rc != 0:
                raise SystemExit(f"scan_match failed: {reg.STATUS.get(rc, rc)} {H.L.tloam_last_error(c_h)}")

        H.gn_iter_timer(reset=True)
        for i in range(warmup):
            step(i)
        H.k3_timer(reset=True)
        H.k3_span(reset=True)
        H.gn_iter_timer(reset=True)                 # arm / reset the device-side period counter of the GN iterations
        barrier()
        t0 = time.perf_counter()
        gn_iters = 0      # executed GN iterations: residual/Jacobian sweep + 6x6 step (tloam_stats.gn_sweeps)
        gn_evals = 0      # solver evaluations served (some from the totals of the previous sweep, see gn_sweeps)
        host_wait = 0     # microseconds the calling thread spent waiting for the device (tloam_stats.host_wait_us)
        worst = 0.0
        for i in range(steps):
            step(warmup + i)
            gn_iters += c_stats.gn_sweeps
            gn_evals += c_stats.gn_evaluations
            host_wait += c_stats.host_wait_us
        barrier()
        elapsed = time.perf_counter() - t0
        # (the last timed frame, for the sanity figures below)
        scene, T, st = scenes[(warmup + steps - 1) % len(scenes)], np.frombuffer(res_buf).reshape(4, 4).T.copy(), c_stats.as_dict()
        gn_iters_job = float(gn_iters)
        if multi:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            gi = torch.tensor([float(gn_iters)], dtype=torch.float64, device=cdev)
            dist.all_reduce(gi, op=dist.ReduceOp.SUM)     # whole-job GN iterations
            gn_iters_job = float(gi.item())
        repeated = None
        if wl == "kitti" and rank == 0:
            # side figure (round 2's headline): ONE frame pair repeated -- every prediction of the context is perfect
            sc105 = kitti_scene(synth, seed, 105)
            H.frame_select(-1)
            H.set_frames(sc105.source, sc105.target)
            for _ in range(20):
                H.scan_match(sc105.T_pred)
            torch.cuda.synchronize()
            tr = time.perf_counter()
            itr = 0
            for _ in range(100):
                _, _, str_ = H.scan_match(sc105.T_pred)
                itr += str_["gn_sweeps"]
            dtr = time.perf_counter() - tr
            repeated = {"workload": "frame 105 of the sequence, the same pair 100 times (not part of value)",
                        "ms_per_frame": round(dtr / 100 * 1e3, 4), "gn_iters_per_sec": round(itr / dtr, 1),
                        "gn_iters_per_frame": itr / 100}
        iter_us, iter_n = H.gn_iter_timer()               # GN iterations of the timed frames as the device clocks them (pose to pose)
        span_us, span_n = H.k3_span()                    # one-launch GN iterations: streaming span of the working sweeps (device clock)
        k3_us, k3_n, _ = H.k3_timer()                    # working sweeps
        k3_all_us, k3_all_n = H.k3_timer_all()            # every sampled K3 launch incl. no-ops after a tolerance exit
        # cross-check without per-launch event overhead: ONE event pair around 60 consecutive launches of the same
        # kernel on the frame's last correspondence set (after the timed region, not part of `value`)
        bb_us = H.time_accumulate(np.asarray(st["se3"], float), 60)
        k1 = k1_roofline(H, world) if (wl == "m1" and rank == 0 and not args.no_side) else None
        H.close()
        D = np.linalg.inv(T) @ scene.T_true               # pose sanity (not timed)
        n_corr = st["n_corr"]
        alg = 72.0 * (n_corr[0] + n_corr[1]) + 88.0 * n_corr[2] + 64.0 * n_corr[3]   # SURVEY 8(d) bytes per sweep
        k3_avg_all_us = k3_all_us / max(k3_all_n, 1)
        k3_avg_work_us = k3_us / max(k3_n, 1)
        alg_avg = alg * k3_n / max(k3_all_n, 1)           # no-op launches move no algorithmic bytes
        ach = alg_avg / (k3_avg_all_us * 1e-6) / 1e9 if k3_all_n else 0.0
        ach_w = alg / (k3_avg_work_us * 1e-6) / 1e9 if k3_n else 0.0
        k3 = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(k3_avg_all_us, 3), "launches": int(k3_all_n),
              "launch_sampling": "HIP event pair (hipExtLaunchKernelGGL) on every 3rd K3 launch of the timed region",
              "algorithmic_bytes_per_launch": alg_avg,
              "working_sweeps": {"launches": int(k3_n), "avg_launch_us": round(k3_avg_work_us, 3),
                                 "algorithmic_bytes_per_launch": alg, "achieved": round(ach_w, 1),
                                 "frac": round(ach_w / HBM_PEAK_GBS, 4)},
              "back_to_back": {"launches": 60, "avg_launch_us": round(bb_us, 3), "algorithmic_bytes_per_launch": alg,
                               "achieved": round(alg / (bb_us * 1e-6) / 1e9, 1),
                               "frac": round(alg / (bb_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "one HIP event pair around 60 consecutive launches on the last correspondence "
                                       "set, after the timed region (no per-launch event overhead; launch gaps included)"}}
        if span_n:
            sp = span_us / span_n
            k3["stream_phase"] = {"launches": int(span_n), "avg_us": round(sp, 3), "algorithmic_bytes_per_launch": alg,
                                  "achieved": round(alg / (sp * 1e-6) / 1e9, 1), "frac": round(alg / (sp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                  "note": "a GN iteration of a large set is ONE launch (k3_sweep_step: sweep, row fold, minimiser step): "
                                          "the dispatch duration above includes the serial tail (~5 us of one wave); this is the "
                                          "STREAMING span of the same launches -- block 0's first instruction to the last block's row "
                                          "at the coherence point, device wall clock (100 MHz), every working sweep of the timed frames"}
        if k3_all_n == 0:   # fused sweep + step launches carry no event pair (tloam_k3_timer covers k3_accumulate dispatches only)
            k3.update({"sampled": False, "achieved": None, "frac": None, "avg_launch_us": None,
                       "launch_sampling": "not sampled: KITTI-size sets run sweep + minimiser step as ONE launch "
                                          "(k_sweep_step_small), which the per-dispatch K3 timer does not cover"})
            k3["working_sweeps"] = None
        return {"workload": W["name"], "repeated_pair": repeated, "distinct_frames": len(scenes) if wl == "kitti" else 1,
                "ms_per_frame": elapsed / steps * 1e3, "gn_iters_per_sec": gn_iters_job / elapsed,
                "gn_iters_per_frame": gn_iters / steps, "solver_evaluations_per_frame": gn_evals / steps, "n_corr": n_corr,
                "outer_iterations": st["outer_iterations"], "pose_err_vs_truth_m": float(np.linalg.norm(D[:3, 3])),
                "host_wait_us_per_frame": round(host_wait / steps, 1), "k3": k3, "k1": k1, "scene": scene, "cfg": W["cfg"],
                # SURVEY 8(d) "GN iteration": sweep + reduction + 6x6 step + pose update.  Period between the ends of two
                # consecutive minimiser steps of one Solve, device wall clock (tloam_gn_iter_timer), over the timed frames
                "gn_iteration_us": round(iter_us / iter_n, 3) if iter_n else None, "gn_iteration_periods": int(iter_n),
                "alg_bytes_per_sweep": alg,
                "last_frame": {"gn_sweeps": int(st["gn_sweeps"]), "gn_evaluations": int(st["gn_evaluations"])}}

    # ---- headline: the configuration the metric is quoted on (KITTI-00 scan density); every rank its own frame pair
    head = run_frames(args.workload, args.steps, args.warmup, args.seed + rank)
    # ---- the HBM-roofline characterisation workload (configs[2]) in the same command: its K3 is `roofline`
    other = "m1" if args.workload == "kitti" else "kitti"
    side = None
    if not args.no_m1 or args.workload == "m1":
        side = head if args.workload == "m1" else run_frames("m1", args.m1_steps, min(args.warmup, 3), args.seed + rank)

    out = None
    if rank == 0:
        traffic, traffic_detail = pmc_traffic(world)
        roofline = None
        if side is not None:
            in_frame = dict(side["k3"])
            in_frame.update({"workload": side["workload"],
                             "note": "the same kernel inside the timed 1 M frames (%d frames): every launch follows the one-block "
                                     "minimiser step, i.e. starts on an idle chip; per-launch HIP event pairs read 1.2-2.5 us "
                                     "more than the dispatch timestamps (DESIGN.md section 7).  With TLOAM_FUSED_LARGE (sweep + fold + "
                                     "step in one dispatch, measured slower, off by default) a stream_phase entry reports the sweep alone" %
                                     (args.steps if args.workload == "m1" else args.m1_steps)})
            roofline = None
            if not multi and not args.no_side:
                try:
                    # SURVEY 8(d) config 3, the roofline-characterisation run proper: K3 alone on the pre-built
                    # 760k:200k:40k set, >= 100 launches after 10 warm-ups, HIP events around batches of 20, median
                    pk = prebuilt_k3(reg, synth, local_rank)
                    roofline = {"bound": "hbm", "achieved": pk["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pk["frac"],
                                "traffic": traffic, "traffic_detail": traffic_detail, "kernel": "k3_accumulate<false, false>",
                                "avg_launch_us": pk["avg_launch_us"], "launches": pk["launches"],
                                "algorithmic_bytes_per_launch": pk["algorithmic_bytes_per_launch"], "workload": pk["workload"],
                                "working_set": "l3_resident: the 74.88 MB of the contract's configuration (SURVEY 8(d) config 3) stay in "
                                               "the 256 MiB Infinity Cache across back-to-back launches, as they do across the sweeps of "
                                               "a Solve; `cold` below is the same kernel with every byte from HBM",
                                "launch_timing": "one HIP event pair per batch of 20 consecutive launches on the context's stream, "
                                                 "median of the 6 batch means (120 launches after 10 warm-ups)",
                                "note": pk["note"]}
                    bw, bw_l3 = read_stream_bandwidth(reg, local_rank)
                    roofline["read_stream_l3_GBps"] = round(bw_l3, 1)
                    roofline["frac_of_read_stream_l3"] = round(roofline["achieved"] / bw_l3, 4)
                    # the same kernel on a working set beyond the Infinity Cache: 4 x the set (299.5 MB per sweep)
                    pc = prebuilt_k3(reg, synth, local_rank, scale=4)
                    cold = {k: pc[k] for k in ("workload", "launches", "avg_launch_us", "algorithmic_bytes_per_launch", "achieved", "frac")}
                    cold.update({"peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_read_stream": round(pc["achieved"] / bw, 4),
                                 "working_set": "299.5 MB per sweep > 256 MiB Infinity Cache: back-to-back sweeps evict each other, "
                                                "every launch streams from HBM",
                                 "launch_timing": "one HIP event pair per batch of 10 consecutive launches, median of 6 batch means"})
                    if os.path.exists(PMC_SUMMARY_COLD):
                        try:
                            dc = json.load(open(PMC_SUMMARY_COLD))
                            cold["traffic"] = float(dc["traffic_bytes_per_launch_all"])
                            cold["traffic_detail"] = {"source": "profiles/" + os.path.basename(PMC_SUMMARY_COLD) + " (committed PMC summary)",
                                                      "fetch_bytes_per_launch": dc["fetch_bytes_per_launch_all"],
                                                      "write_bytes_per_launch": dc["write_bytes_per_launch_all"],
                                                      "stale": dc.get("kernel_source_sha16") != _sha16(os.path.join(ROOT, "tloam_amd", "csrc", "tl_gn.hip"))}
                        except Exception:  # noqa: BLE001
                            pass
                    # the headline fields are the HBM figure: every byte of every launch from HBM (the 4x set); the contract's own
                    # 74.88 MB set, which lives in the Infinity Cache across launches, moves to `l3_resident`
                    l3 = {k: roofline[k] for k in ("achieved", "frac", "traffic", "traffic_detail", "avg_launch_us", "launches",
                                                   "algorithmic_bytes_per_launch", "workload", "working_set", "launch_timing", "note",
                                                   "read_stream_l3_GBps", "frac_of_read_stream_l3")}
                    roofline = {"bound": "hbm", "achieved": cold["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cold["frac"],
                                "traffic": cold.get("traffic"), "traffic_detail": cold.get("traffic_detail"),
                                "kernel": "k3_accumulate<false, false>", "avg_launch_us": cold["avg_launch_us"], "launches": cold["launches"],
                                "algorithmic_bytes_per_launch": cold["algorithmic_bytes_per_launch"], "workload": cold["workload"],
                                "working_set": cold["working_set"], "launch_timing": cold["launch_timing"],
                                # flat scalars (the driver's parser keeps scalars only): the contract's own configuration, the
                                # on-box read-stream ceiling, and -- filled below -- the kernel and the GN iteration inside the 1 M frames
                                "config3_frac": l3["frac"], "config3_avg_launch_us": l3["avg_launch_us"],
                                "config3_algorithmic_bytes": l3["algorithmic_bytes_per_launch"],
                                "read_stream_GBps": round(bw, 1), "frac_of_read_stream": cold["frac_of_read_stream"],
                                "read_stream_l3_GBps": round(bw_l3, 1), "config3_frac_of_read_stream_l3": l3["frac_of_read_stream_l3"],
                                "read_stream_note": "tloam_time_read_stream: a read stream with K3's access pattern (8 fp64 streams, 16-byte "
                                                    "loads, persistent waves): 1.2 GB per pass = HBM, 75 MB per pass = Infinity-Cache resident",
                                "note": "headline = the HBM figure (cold working set); the contract's 74.88 MB set (SURVEY 8(d) config 3: "
                                        ">= 70 % <=> <= 13.4 us) is Infinity-Cache resident across launches and is reported as l3_resident",
                                "l3_resident": l3}
                except Exception as e:  # noqa: BLE001  (side measurements never take the line down)
                    roofline = None
                    in_frame["side_measurements_error"] = repr(e)[:200]
            if roofline is None:   # N > 1 / --no-side: the in-frame figure stands in
                roofline = dict(in_frame)
                roofline.update({"kernel": "k3_accumulate<false, false>", "traffic": traffic, "traffic_detail": traffic_detail})
            else:
                roofline["in_frame"] = in_frame
            # the same kernel, and the GN iteration it lives in, INSIDE the 1 M frames -- flat, where the driver's parser sees them
            ws = in_frame.get("working_sweeps") or {}
            roofline["in_frame_frac"] = ws.get("frac")
            roofline["in_frame_avg_launch_us"] = ws.get("avg_launch_us")
            roofline["in_frame_back_to_back_frac"] = (in_frame.get("back_to_back") or {}).get("frac")
            if side.get("gn_iteration_us"):
                git = side["gn_iteration_us"]
                roofline["gn_iteration_us"] = git
                roofline["gn_iteration_frac"] = round(side["alg_bytes_per_sweep"] / (git * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                roofline["gn_iteration_note"] = ("one GN iteration as BASELINE defines it (sweep + reduction + 6x6 step + pose update) inside "
                                                 "the timed 1 M frames: period between the ends of two consecutive minimiser steps of a "
                                                 "Solve by the device's wall clock (%d periods); frac = the sweep's algorithmic bytes / "
                                                 "that period / 8 TB/s" % side["gn_iteration_periods"])
            if side.get("k1"):
                roofline["roofline_k1"] = side["k1"]
            if args.workload == "kitti":
                bb = head["k3"]["back_to_back"]
                roofline["headline_workload_k3"] = {
                    "kernel": "k_solve_all (one launch per run of outer iterations: compaction, every GN iteration -- sweep, rows, "
                              "6x6 step --, weights and loop decisions); the sweep alone = k3_accumulate<true>",
                    "sweep_alone_back_to_back": bb,
                    "note": "KITTI-cap sets (442 KB per sweep) are latency bound, not bandwidth bound: in the frames the sweep runs "
                            "inside the persistent Solve launch with its correspondences held in registers (no K3 launch to time; "
                            "5.3-5.5 us per GN iteration by the device clock, gn_iteration_us; profiles/r06_solve_all_timeline.txt); the figure here is the stand-alone "
                            "sweep kernel on the frame's last correspondence set"}
        out = {
            "metric": "gauss_newton_iters_per_sec", "value": round(head["gn_iters_per_sec"], 2), "unit": "GN iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_frame"], 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": head["workload"], "step": "one tloam_scan_match of the resident frame pair (ms_per_step = ms/frame), preceded by the O(1) tloam_frame_select that activates it; the C ABI called through ctypes with every argument object made beforehand",
                       "gn_iters_per_frame": head["gn_iters_per_frame"],
                       "gn_iteration_us": head["gn_iteration_us"],
                       "solver_evaluations_per_frame": head["solver_evaluations_per_frame"], "n_corr": head["n_corr"],
                       "outer_iterations": head["outer_iterations"], "frames_per_step": world,
                       "host_wait_us_per_frame": head["host_wait_us_per_frame"],
                       "parallelism": (f"{world} independent frame streams, one per GPU, no data-path collective (replicas, "
                                       "BASELINE.json configs[4]); the sharded single frame is reported under sharded_1m")
                       if multi else "1 GPU",
                       "distinct_frames": head["distinct_frames"],
                       "pose_err_vs_truth_m": head["pose_err_vs_truth_m"]},
            "roofline": roofline,
        }
        if head.get("repeated_pair"):
            out["config"]["repeated_pair"] = head["repeated_pair"]
        if side is not None and side is not head:
            out["m1_frame"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in side.items()
                               if k in ("workload", "ms_per_frame", "gn_iters_per_sec", "gn_iters_per_frame",
                                        "solver_evaluations_per_frame", "n_corr", "outer_iterations", "pose_err_vs_truth_m")}
    m1 = WL["m1"]

    # ---------------- N > 1: ONE frame sharded over the ranks (strong scaling, RCCL all-reduce per sweep) ----------------
    if multi and side is not None:
        # (on a deadline: RCCL with N > 1 ranks and the mailbox over xGMI run for the first time on the driver's 8-GPU node --
        #  a collective that never returns must not cost the run its line.  What has finished by then is reported.)
        # In a child process of every rank (a GPU fault there does not take the line with it) -- except when MORE THAN FOUR ranks
        # share ONE device (TLOAM_BENCH_ONE_DEVICE, a development set-up): sixteen processes oversubscribe the device's hardware
        # queues, the runlist is time-sliced in milliseconds and a gather that spins on a peer's post waits for the peer's queue to
        # be mapped again -- 22 ms per exchange instead of 29 us (profiles of round 6).  On one GPU per rank it is two processes.
        in_process = os.environ.get("TLOAM_BENCH_SHARDED_IN_PROCESS") == "1" or (one_dev and world > 4)
        stuck = False
        if in_process:
            sharded = {}
            finished = within(lambda: sharded_frame(args, reg, synth, torch, dist, m1["cfg"], m1["n_src"], m1["n_tgt"], rank, world,
                                                    local_rank, barrier, cdev, sharded), args.sharded_timeout)
            stuck = not finished   # a thread of this process is stuck inside a collective: nothing can be torn down in order
        else:
            sharded, finished = sharded_in_children(args, torch, dist, rank, cdev)
        if rank == 0:
            sharded = sharded_summary(sharded, finished, args.sharded_timeout)
            sharded["replica_ms_per_frame"] = round(side["ms_per_frame"], 4)
            if "ms_per_frame" in sharded:
                # against the SAME frame solved unsharded on rank 0's GPU in the same block (one_rank); the replica figure -- every
                # GPU busy with a 1 M frame of its own -- beside it
                one = (sharded.get("one_rank") or {}).get("ms_per_frame") or side["ms_per_frame"]
                sharded["speedup_vs_one_gpu_frame"] = round(one / sharded["ms_per_frame"], 3)
                sharded["speedup_vs_replica_frame"] = round(side["ms_per_frame"] / sharded["ms_per_frame"], 3)
            sharded["predicted_speedup"] = PREDICTED_SHARDED_SPEEDUP.get(world)
            sharded["ran_in"] = "the ranks' own processes" if in_process else "a child process of every rank"
            out["sharded_1m"] = sharded
            out.update(sharded_flat(sharded, world))
        if stuck:
            if rank == 0:
                emit(out)
            sys.stdout.flush()
            os._exit(0)

    # ---------------- sequence, adjacent rows, odometry loop, CPU baseline: rank 0, N = 1 only ----------------
    if rank == 0 and not multi:
        kitti_seq = None
        if not args.no_kitti:
            kitti_seq = kitti_sequence(args, reg, synth, torch, local_rank)
            out["kitti_sequence"] = kitti_seq["report"]
            out["adjacent_rows"] = adjacent_rows(args, reg, torch, local_rank)
            out["odometry_loop"] = odometry_loop(args, reg, torch, local_rank)
            out["multi_stream"] = multi_stream(args, reg, synth, local_rank)
            if side is not None and not args.no_side:
                try:
                    sh = shard_size_iterations(args, reg, synth, torch, local_rank, m1["cfg"], side["ms_per_frame"])
                    out.update(sh.pop("flat"))   # top-level scalars: shard_iteration_us_n{2,4,8}, shard_frame_ms_n*, shard_predicted_speedup_n*
                    out["shard_size_iterations"] = sh
                except Exception as e:  # noqa: BLE001
                    out["shard_size_iterations"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = guarded(lambda: cpu_baseline(head, side, args, kitti_seq), 300.0, "cpu_baseline", out, rank)
        if kitti_seq is not None:   # the honest per-frame cost of the plug-in as wired in INTEGRATION.md section 1
            out["config"]["ms_per_frame_incl_pcie_upload"] = kitti_seq["report"]["ms_per_frame_incl_pcie_upload"]
            for key in ("set_source_ms", "set_target_ms", "scan_match_ms", "set_source_plus_scan_match_ms", "set_target_plus_scan_match_ms",
                        "all_three_calls_ms"):
                out["config"][key] = kitti_seq["report"]["per_call"][key]
            out["config"]["pcie_note"] = ("value / ms_per_step time scan_match with the eight clouds resident in HBM (the reference's "
                                          "own bracket, front_end.cpp:320-322); handing the clouds over through "
                                          "setInputSource/setInputTarget every frame costs ms_per_frame_incl_pcie_upload, with "
                                          "the device-resident submap odometry_loop.ms_per_frame")
        if args.kitti_dir:
            out["kitti_replay"] = kitti_replay(args, reg, torch, local_rank)
    if rank == 0:
        emit(out)
    if multi:
        def leave():
            dist.barrier()
            dist.destroy_process_group()
        if not within(leave, 120.0):   # (a rank that gave up above is gone: the line is out, leave without it)
            sys.stdout.flush()
            os._exit(0)


def sharded_in_children(args, torch, dist, rank, cdev):
    """The sharded 1 M frame runs in a CHILD process of every rank (this script with --sharded-child; a process group of the
    children's own, one port up).  The exchange forms it times -- the peer mailbox over hipIpc handles, RCCL with N ranks -- have
    never met two devices before the driver's first multi-GPU run: a GPU fault there ends the process it happens in, and with the
    frame inside the ranks themselves that was the whole job and its JSON line.  Now it ends a child: the parents notice (they
    poll their children and agree on what they see through the launcher's group twice a second), end the other children, and the
    line goes out with the replica headline and the reason in `sharded_1m`.  Returns (what rank 0's child reported, finished)."""
    import subprocess
    import tempfile
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 211)
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"     # (the children's rank 0 hosts their store: the launcher's agent does not know them)
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--m1-steps", str(args.m1_steps), "--seed", str(args.seed),
           "--sharded-timeout", str(args.sharded_timeout)]
    res = tempfile.TemporaryFile()
    child = subprocess.Popen(cmd, env=env, stdout=res, stdin=subprocess.DEVNULL)
    limit = args.sharded_timeout + 120.0      # (the child's own deadline, plus its start: imports, process group, contexts)
    t0 = time.time()
    verdict = "finished"
    while True:
        rc = child.poll()
        flags = torch.tensor([1.0 if (rc is not None and rc != 0) else 0.0, 1.0 if rc is None else 0.0,
                              1.0 if time.time() - t0 > limit else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        failed, running, late = (float(v) > 0.0 for v in flags.tolist())
        if failed or late:
            verdict = "a rank's child process ended abnormally (exit code %s here)" % rc if failed else "no result after %.0f s" % limit
            if child.poll() is None:
                child.kill()          # (this rank's own child, by its process id)
            child.wait()
            break
        if not running:
            break
        time.sleep(0.5)
    out, finished = {}, False
    if rank == 0:
        res.seek(0)
        text = res.read().decode(errors="replace").strip()
        try:
            d = json.loads(text.splitlines()[-1]) if text else {}
            out, finished = d.get("sharded", {}), bool(d.get("finished"))
        except (ValueError, IndexError):
            pass
        if verdict != "finished":
            out = dict(out)
            out["child_error"] = verdict
            finished = False
    res.close()
    return out, finished


def within(fn, seconds):
    """fn() on a thread of its own; False if it has not returned after `seconds` (the library's calls release the GIL).  An
    exception of fn is re-raised here."""
    import threading
    box = {}

    def run():
        try:
            fn()
        except BaseException as e:   # noqa: BLE001
            box["error"] = e
        box["done"] = True

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if "error" in box:
        raise box["error"]
    return "done" in box


def guarded(fn, seconds, name, out, rank):
    """The CPU baseline is a reported side figure computed by test infrastructure (the oracle's port on the host's cores): it
    must never cost the run its line.  `fn` runs on a thread of its own (the port's calls release the GIL); if it has not
    returned after `seconds` -- round 5 met a hang of the port's thread pool on a 256-core host -- or raises, the JSON line is
    printed without it, with the reason in its place, and the process ends there."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:   # noqa: BLE001 -- reported, not raised
            box["error"] = repr(e)

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if "value" in box:
        return box["value"]
    if "error" in box:
        return {"error": box["error"], "kind": "port"}
    out[name] = {"error": "%s did not finish within %.0f s; every GPU figure of this line was measured before it started" % (name, seconds),
                 "kind": "port"}
    if rank == 0:
        emit(out)
    sys.stdout.flush()
    os._exit(0)


# the 8-GPU run is held against this table: speed-up of the sharded 1 M frame over the one-GPU frame,
# MEASURED on one MI355X at shard size -- one rank, the sharded launch forms, 1/N of the source points, full targets, loop-back
# exchange (`shard_size_iterations`, round 6: 0.99 / 0.82 / 0.72 ms against 1.11 ms).  A real rank builds one to three of the four
# search grids instead of four (-45 to -110 us) and pays ~13 exchanges over xGMI (+30 us by the mailbox's expected 2.5 us): the two
# roughly cancel.  What stays replicated -- the dependent chain of 6x6 steps with their launch boundaries, the grid build over the
# full targets, the small launches of the compact path -- bounds it by Amdahl far below the north star's 6x.  (Rounds 4-5 quoted
# 1.66 / 2.4 / 3.05 from a kernel-timeline estimate that shrank the per-iteration cost with N; the measurement says it does not.)
PREDICTED_SHARDED_SPEEDUP = {1: 1.0, 2: 1.13, 4: 1.34, 8: 1.54}
SHARDED_MODES = ("mailbox", "mailbox_fused", "rccl")
POSE_TOL = 1e-9   # |dt| (m) and |dR| (rad) of the sharded solve against the one-rank solve of the same frame (the one-device tests' bar)


def _pose_delta(Ta, Tb):
    D = np.linalg.inv(Ta) @ Tb
    dt = float(np.linalg.norm(D[:3, 3]))
    R = D[:3, :3]
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(np.linalg.norm(w), 0.5 * (np.trace(R) - 1.0)))


COUNTERS = ("outer_iterations", "gn_evaluations", "gn_iterations", "accepted_steps", "gn_sweeps", "converged_early")


def _counters(st):
    return {k: int(st[k]) for k in COUNTERS} | {"n_corr": [int(v) for v in st["n_corr"]]}


def sharded_frame(args, reg, synth, torch, dist, cfg, n_src, n_tgt, rank, world, local_rank, barrier, cdev="cuda", res=None):
    """BASELINE.json configs[3]: the SAME frame on every rank, source points sharded in contiguous index blocks,
    targets replicated, the 48 doubles of the normal equations exchanged once per GN sweep.  Three forms are timed in the same
    run, each in a context of its own: `mailbox` (the library's one-shot peer mailbox over xGMI: sweep + post | gather + step),
    `mailbox_fused` (the same exchange, sweep + post + gather + step as ONE launch) and `rccl` (one ncclAllReduce of 48 f64 on
    the compute stream -- the form the north star names).  Per form: the whole frame, the sweep alone and the sweep + exchange
    (SURVEY 8(d) config 4 (i)), the GN iteration as the device clocks it.

    SELF-VERIFYING: rank 0 first solves the UNSHARDED frame on its own GPU; every form then reports
    the pose delta of its result against that solve (bar: 1e-9 m / 1e-9 rad), whether every counter of the solve is equal,
    whether all ranks returned the same bits (all-gather of the 16 result doubles), and what the context itself says it ran on
    (tloam_get_info: exchange, ranks, and for RCCL the communicator's own ncclCommCount).  A form that fails a check is reported
    as failed and never becomes `fastest_exchange`.  A form that cannot be set up on every rank (RCCL refuses two ranks on one
    device: TLOAM_BENCH_ONE_DEVICE) is reported as such."""
    res = {} if res is None else res   # (filled as the modes finish: the caller reports what is there when its deadline passes)
    res.update({"workload": "one 1M-correspondence frame, source points sharded x%d, targets replicated, one exchange of "
                            "48 f64 per GN sweep" % world, "scaling": "strong", "n_gpus": world})
    one_dev = os.environ.get("TLOAM_BENCH_ONE_DEVICE") == "1"

    def agreed(err):
        """every rank learns whether ALL ranks succeeded before anyone enqueues a collective"""
        flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return float(flag.item()) == 0.0

    scene = synth.make_scene(seed=args.seed, n_src=n_src, n_tgt=n_tgt)
    warm = max(min(args.warmup, 3), 1)

    # ---- the reference: the unsharded frame on rank 0's GPU (the other ranks wait at the broadcast: on one shared device
    #      nothing competes with it)
    ref = [None]
    if rank == 0:
        try:
            H1 = reg.HipRegistration(cfg, device=local_rank)
            H1.set_frames(scene.source, scene.target)
            for _ in range(warm):
                rc1, T1, st1 = H1.scan_match(scene.T_pred)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.m1_steps):
                rc1, T1, st1 = H1.scan_match(scene.T_pred)
            ms1 = (time.perf_counter() - t0) / args.m1_steps * 1e3
            H1.close()
            ref = [{"rc": int(rc1), "T": T1.tobytes(), "counters": _counters(st1), "ms_per_frame": ms1}]
        except Exception as e:  # noqa: BLE001
            ref = [{"rc": -999, "error": f"{type(e).__name__}: {e}"}]
    dist.broadcast_object_list(ref, src=0)
    ref = ref[0]
    if ref.get("rc") != 0:
        res["error"] = "the one-rank reference solve failed: %s" % (ref.get("error") or reg.STATUS.get(ref.get("rc"), ref.get("rc")))
        res["modes"] = []
        return res
    T_ref = np.frombuffer(ref["T"], dtype=np.float64).reshape(4, 4)
    res["one_rank"] = {"ms_per_frame": round(ref["ms_per_frame"], 4), "counters": ref["counters"],
                       "note": "the same frame unsharded, on rank 0's GPU alone, before the sharded forms: the reference of every check below"}

    def create(mode):
        # (the fused form is chosen when the context is created: TLOAM_FUSED_LARGE)
        if mode == "mailbox_fused":
            os.environ["TLOAM_FUSED_LARGE"] = "1"
        try:
            return reg.HipRegistration(cfg, device=local_rank)
        finally:
            os.environ.pop("TLOAM_FUSED_LARGE", None)

    def init(H, mode):
        err = None
        if mode in ("mailbox", "mailbox_fused"):
            try:
                handles = [None] * world
                dist.all_gather_object(handles, H.comm_mailbox_export())
                H.comm_init_mailbox(rank, world, handles)
            except Exception as e:  # noqa: BLE001
                err = e
        else:
            uid = [None]
            if rank == 0:
                try:
                    uid = [reg.rccl_unique_id()]
                except Exception as e:  # noqa: BLE001
                    uid = [e]
            dist.broadcast_object_list(uid, src=0)
            try:
                if isinstance(uid[0], Exception):
                    raise uid[0]
                if one_dev and world > 1:
                    raise RuntimeError("RCCL refuses two ranks on one device (TLOAM_BENCH_ONE_DEVICE=1)")
                H.comm_init_rccl(rank, world, uid[0])
            except Exception as e:  # noqa: BLE001
                err = e
        return err

    def run_mode(mode):
        out = {"exchange": {"mailbox": "one-shot peer mailbox over xGMI (no collective library on the data path); a GN iteration is "
                                       "sweep + post | gather + step: two launches",
                            "mailbox_fused": "the same mailbox, sweep + post + gather + step as ONE launch (k3_sweep_step)",
                            "rccl": "ncclAllReduce (RCCL) of 48 f64 on the compute stream between sweep and step"}[mode]}
        H = create(mode)
        try:
            err = init(H, mode)
            if not agreed(err):
                out["error"] = repr(err)[:200] if err is not None else "initialisation failed on another rank"
                return out
            info = H.info()
            out["context"] = {k: info[k] for k in ("comm_mode", "rank", "nranks", "rccl_comm_count", "rccl_comm_rank", "loopback", "device_cus")}
            H.set_frames(scene.source, scene.target)
            ok = 1.0
            for _ in range(warm):
                rc, T, st = H.scan_match(scene.T_pred)
                ok = ok if rc == 0 else 0.0
            H.gn_iter_timer(reset=True)
            barrier()
            t0 = time.perf_counter()
            it = 0
            steps = args.m1_steps
            for _ in range(steps):
                rc, T, st = H.scan_match(scene.T_pred)
                ok = ok if rc == 0 else 0.0
                it += st["gn_sweeps"]
            barrier()
            dt = time.perf_counter() - t0
            iter_us, iter_n = H.gn_iter_timer()
            tt = torch.tensor([dt, -ok], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0].item())
            if float(tt[1].item()) != -1.0:
                out["error"] = "scan_match failed on at least one rank: %s" % reg.STATUS.get(rc, rc)
                return out
            # ---- the checks.  Every rank against the one-rank solve; all ranks against each other, bit for bit
            dpos, drot = _pose_delta(T, T_ref)
            mine = _counters(st)
            equal = mine == ref["counters"]
            every = [None] * world
            dist.all_gather_object(every, (T.tobytes(), dpos, drot, equal, info["nranks"], info["rccl_comm_count"]))
            bit_identical = all(e[0] == every[0][0] for e in every)
            dmax, rmax = max(e[1] for e in every), max(e[2] for e in every)
            counters_equal = all(e[3] for e in every)
            nranks_ok = all(e[4] == world for e in every) and (mode != "rccl" or all(e[5] == world for e in every))
            verified = bool(dmax < POSE_TOL and rmax < POSE_TOL and bit_identical and counters_equal and nranks_ok)
            D = np.linalg.inv(T) @ scene.T_true
            out.update({"frames": steps, "ms_per_frame": round(dt / steps * 1e3, 4), "gn_iters_per_sec": round(it / dt, 2),
                        "gn_iters_per_frame": it / steps, "n_corr": st["n_corr"],
                        "pose_err_vs_truth_m": float(np.linalg.norm(D[:3, 3])),
                        "pose_delta_vs_one_rank": {"dt_m": dmax, "dR_rad": rmax, "bar": POSE_TOL, "note": "max over the ranks"},
                        "ranks_bit_identical": bit_identical, "counters_equal_one_rank": counters_equal,
                        "counters": mine, "ranks_in_exchange": min(e[4] for e in every),
                        "rccl_comm_count": (min(e[5] for e in every) if mode == "rccl" else None), "verified": verified,
                        "gn_iteration_us": round(iter_us / iter_n, 3) if iter_n else None})
            if not verified:
                out["error"] = ("verification failed: pose delta %.3e m / %.3e rad (bar %.0e), ranks bit-identical %s, counters equal %s, "
                                "ranks in the exchange as the contexts report them %s" %
                                (dmax, rmax, POSE_TOL, bit_identical, counters_equal, [(e[4], e[5]) for e in every]))
            # the sweep alone / sweep + exchange on this frame's set (collective calls; max over ranks); the fused form has no
            # sweep-only launch of its own: the two-launch kernels are what is timed there as well
            x = np.asarray(st["se3"], float)
            H.time_sharded_sweep(x, 10, True)
            t_ex = sorted(H.time_sharded_sweep(x, 40, True) for _ in range(3))[1]
            t_no = sorted(H.time_sharded_sweep(x, 40, False) for _ in range(3))[1]
            tt = torch.tensor([t_ex, t_no], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out["per_sweep_us"] = {"sweep_plus_exchange": round(float(tt[0].item()), 3), "sweep_alone": round(float(tt[1].item()), 3),
                                   "exchange_adds": round(float(tt[0].item() - tt[1].item()), 3),
                                   "note": "40 back-to-back launches per HIP event pair, median of 3, max over ranks; the sweep "
                                           "covers this rank's 1/%d of the set, its last block folds the rows" % world}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline
            out["error"] = f"{type(e).__name__}: {e}"
        finally:
            H.close()
        return out

    modes = [m for m in SHARDED_MODES if not (m.startswith("mailbox") and os.environ.get("TLOAM_BENCH_NO_MAILBOX") == "1")]
    res["modes"] = modes
    for m in modes:
        res[m] = run_mode(m)
    return res


def sharded_summary(res, finished, seconds):
    """the top-level figures of sharded_1m = those of the fastest exchange that ran AND passed its checks; a mode the deadline cut
    off says so"""
    res = dict(res)
    modes = res.pop("modes", list(SHARDED_MODES))
    child_error = res.pop("child_error", None)   # (sharded_in_children: a rank's child ended abnormally / never answered)
    for m in modes:
        if m not in res:
            res[m] = {"error": ("not run: " + child_error) if child_error else "did not finish within the %.0f s of --sharded-timeout" % seconds}
    if child_error:
        res["note"] = "the sharded frame runs in child processes of the ranks: " + child_error + "; the replica headline of this line is unaffected"
    elif not finished:
        res["note"] = "cut off by --sharded-timeout: the figures are those of the exchanges that had finished"
    ran = [m for m in modes if "ms_per_frame" in res[m] and res[m].get("verified")]
    if ran:
        best = min(ran, key=lambda m: res[m]["ms_per_frame"])
        for k in ("exchange", "frames", "ms_per_frame", "gn_iters_per_sec", "gn_iters_per_frame", "n_corr", "pose_err_vs_truth_m",
                  "per_sweep_us", "pose_delta_vs_one_rank", "ranks_bit_identical", "counters_equal_one_rank", "verified",
                  "gn_iteration_us"):
            if k in res[best]:
                res[k] = res[best][k]
        res["fastest_exchange"] = best
    elif "error" not in res:
        res["error"] = "; ".join("%s: %s" % (m, res[m].get("error")) for m in modes)
    return res


def sharded_flat(sh, world):
    """The sharded frame's verdict as TOP-LEVEL SCALARS of the JSON line (the driver's parser keeps scalars and drops nested
    objects): north star's strong-scaling target is on THIS quantity, not on the replica headline."""
    flat = {"sharded_1m_predicted_speedup": PREDICTED_SHARDED_SPEEDUP.get(world),
            "sharded_1m_verified": bool(sh.get("verified")), "sharded_1m_fastest_exchange": sh.get("fastest_exchange"),
            "sharded_1m_ms_per_frame": sh.get("ms_per_frame"), "sharded_1m_speedup": sh.get("speedup_vs_one_gpu_frame"),
            "sharded_1m_one_rank_ms_per_frame": (sh.get("one_rank") or {}).get("ms_per_frame"),
            "sharded_1m_exchange_adds_us": (sh.get("per_sweep_us") or {}).get("exchange_adds"),
            "sharded_1m_gn_iteration_us": sh.get("gn_iteration_us"),
            "sharded_1m_pose_delta": (sh.get("pose_delta_vs_one_rank") or {}).get("dt_m"),
            "sharded_1m_pose_delta_rad": (sh.get("pose_delta_vs_one_rank") or {}).get("dR_rad"),
            "sharded_1m_ranks_bit_identical": sh.get("ranks_bit_identical"),
            "sharded_1m_counters_equal": sh.get("counters_equal_one_rank"),
            "rccl_nranks": (sh.get("rccl") or {}).get("rccl_comm_count"),
            "sharded_1m_error": sh.get("error")}
    for m in SHARDED_MODES:
        d = sh.get(m) or {}
        flat["sharded_1m_%s_ms_per_frame" % m] = d.get("ms_per_frame")
        flat["sharded_1m_%s_verified" % m] = d.get("verified") if "ms_per_frame" in d else None
    return flat


def shard_size_iterations(args, reg, synth, torch, device, cfg, one_rank_ms):
    """Strong-scaling headroom measured on ONE GPU at shard size.  At N ranks each rank sweeps 1/N of
    the correspondences and every rank takes every 6x6 step: what bounds the sharded 1 M frame is the part of a GN iteration
    that does not shrink with N.  Here one rank runs the SHARDED launch forms on 1/N of the 1 M frame's source points (all kinds
    cut alike) against the full targets, exchanging with itself (a mailbox / RCCL set-up with nranks = 1: every launch of the
    sharded forms runs, the exchange is a loop-back -- include/tloam_hip.h), and reports per N and per form the GN iteration as
    the device clocks it and the frame.  The results are those of the single-rank forms (checked: counters equal, pose within
    1e-12 -- bit for bit where both keep the compact factor set).  What one GPU
    cannot show is the xGMI latency of the exchange; a real rank also builds only the grids of the one or two kinds it holds
    (here: all four), so the frame figure is an upper bound and the predicted speed-up a lower one."""
    out = {"note": "one rank, sharded launch forms, 1/N of the source points, full targets, loop-back exchange"}
    forms = ("mailbox", "mailbox_fused", "rccl_one_rank")
    flat = {}
    for N in (2, 4, 8):
        n_src = tuple(max(n // N, 16) for n in synth.M1_SRC)
        scene = synth.make_scene(seed=args.seed, n_src=n_src, n_tgt=synth.M1_TGT)
        row = {"source_points": int(sum(n_src))}
        # the single-rank forms on the same shard: the reference of the bit-for-bit check
        H0 = reg.HipRegistration(cfg, device=device)
        H0.set_frames(scene.source, scene.target)
        for _ in range(2):
            rc0, T0, st0 = H0.scan_match(scene.T_pred)
        H0.close()
        for form in forms:
            r = {}
            try:
                if form == "mailbox_fused":
                    os.environ["TLOAM_FUSED_LARGE"] = "1"
                try:
                    H = reg.HipRegistration(cfg, device=device)
                finally:
                    os.environ.pop("TLOAM_FUSED_LARGE", None)
                if form == "rccl_one_rank":
                    H.comm_init_rccl(0, 1, reg.rccl_unique_id())
                else:
                    H.comm_init_mailbox(0, 1, [H.comm_mailbox_export()])
                H.set_frames(scene.source, scene.target)
                H.gn_iter_timer(reset=True)
                for _ in range(2):
                    rc, T, st = H.scan_match(scene.T_pred)
                H.gn_iter_timer(reset=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                frames = 5
                for _ in range(frames):
                    rc, T, st = H.scan_match(scene.T_pred)
                ms = (time.perf_counter() - t0) / frames * 1e3
                us, n = H.gn_iter_timer()
                info = H.info()
                H.close()
                dpos, drot = _pose_delta(T, T0)
                r = {"ms_per_frame": round(ms, 4), "gn_iteration_us": round(us / n, 3) if n else None, "periods": int(n),
                     "gn_sweeps": int(st["gn_sweeps"]), "rc": int(rc), "loopback": info["loopback"], "rccl_comm_count": info["rccl_comm_count"],
                     # (the sharded forms keep the COMPACT factor set -- the caps' prefix over the ranks needs the index order --, the
                     #  single-rank forms of a frame this size the DIRECT one: the same factors, the sums in another order)
                     "pose_delta_vs_single_rank_forms": {"dt_m": dpos, "dR_rad": drot},
                     "bit_identical_to_single_rank_forms": bool(T.tobytes() == T0.tobytes()),
                     "equal_to_single_rank_forms": bool(rc == 0 and rc0 == 0 and dpos < 1e-12 and drot < 1e-12
                                                        and _counters(st) == _counters(st0))}
            except Exception as e:  # noqa: BLE001 -- a side measurement never takes the line down
                r = {"error": f"{type(e).__name__}: {e}"[:200]}
            row[form] = r
        ok = [f for f in forms if row[f].get("gn_iteration_us") and row[f].get("equal_to_single_rank_forms")]
        if ok:
            bi = min(ok, key=lambda f: row[f]["gn_iteration_us"])
            bf = min(ok, key=lambda f: row[f]["ms_per_frame"])
            row["best_iteration"] = {"form": bi, "gn_iteration_us": row[bi]["gn_iteration_us"]}
            row["best_frame"] = {"form": bf, "ms_per_frame": row[bf]["ms_per_frame"]}
            flat["shard_iteration_us_n%d" % N] = row[bi]["gn_iteration_us"]
            flat["shard_frame_ms_n%d" % N] = row[bf]["ms_per_frame"]
            if one_rank_ms:
                flat["shard_predicted_speedup_n%d" % N] = round(one_rank_ms / row[bf]["ms_per_frame"], 3)
        out["n%d" % N] = row
    out["flat"] = flat
    return out


_EGO = None


def ego_motion():
    """Per-frame ego-motion of the reference's published KITTI-00 trajectory (doc/tloam_00.txt), velodyne axes, as
    se(3) vectors: the committed fixture tests/golden/tloam_00_ego_motion.npz (made by
    tests/golden/make_tloam00_ego_motion.py in the build container).  None if the fixture is missing."""
    global _EGO
    if _EGO is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "tloam_00_ego_motion.npz")
        _EGO = np.load(path)["se3"] if os.path.exists(path) else False
    return None if _EGO is False else _EGO


def kitti_frame(synth, seed, f):
    """Frame f of the synthetic KITTI-density sequence (BASELINE.json configs[1]): an
    independent procedural scan pair per frame (seed + f); the ego-motion of the frame is the relative pose k -> k+1
    of the reference's own KITTI-00 trajectory and the prediction is the constant-velocity one built from the previous
    relative pose (front_end.cpp:329-330) -- initial error ~1.6 cm / 3 mrad in the mean, more in the turns.  Without
    the fixture: a gentle synthetic arc with a random prediction error of that size."""
    ego = ego_motion()
    if ego is not None:
        k = 1 + f % (len(ego) - 1)
        true_se3 = tuple(ego[k])
        T_true, T_cv = synth.se3_exp_np(ego[k]), synth.se3_exp_np(ego[k - 1])
        pred_err = tuple(synth.se3_log_np(np.linalg.inv(T_true) @ T_cv))
    else:
        rng = np.random.default_rng(1000003 * seed + f)
        yaw = 0.3 + 0.01 * f
        true_se3 = (0.8 * np.cos(0.01 * f) * (1 + f * 0.0), 0.8 * np.sin(0.01 * f), 0.02, 0.002, 0.003, yaw % 3.0)
        pred_err = tuple(rng.normal(0.0, 0.016 / np.sqrt(3), 3)) + tuple(rng.normal(0.0, 0.003 / np.sqrt(3), 3))
    return synth.make_scene(seed=seed + f, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT, true_se3=true_se3,
                            pred_err=pred_err)


_SCENES = {}


def kitti_scene(synth, seed, f):
    """kitti_frame, cached: the headline, the sequence block and the CPU check run over the same frames"""
    key = (seed, f)
    if key not in _SCENES:
        _SCENES[key] = kitti_frame(synth, seed, f)
    return _SCENES[key]


def kitti_sequence(args, reg, synth, torch, device):
    """KITTI-density sequence on one GPU: per frame the eight clouds are handed over (PCIe), then ONE
    scan_match is timed with the clouds resident (the bracket of front_end.cpp:320-322)."""
    nf = args.kitti_frames
    H = reg.HipRegistration(reg.default_config(), device=device)
    warm = kitti_scene(synth, args.seed, 0)
    H.set_frames(warm.source, warm.target)
    for _ in range(5):
        H.scan_match(warm.T_pred)
    ms, ms_up, it, ev, terr, poses = [], [], 0, 0, [], {}
    n_corr = None
    for f in range(nf):
        sc = kitti_scene(synth, args.seed, f)
        t0 = time.perf_counter()
        H.set_frames(sc.source, sc.target)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rc, T, st = H.scan_match(sc.T_pred)
        t2 = time.perf_counter()
        if rc != 0:
            raise SystemExit(f"KITTI-density frame {f}: scan_match failed: {reg.STATUS.get(rc, rc)}")
        ms.append((t2 - t1) * 1e3)
        ms_up.append((t2 - t0) * 1e3)
        it += st["gn_sweeps"]
        ev += st["gn_evaluations"]
        D = np.linalg.inv(T) @ sc.T_true
        terr.append(float(np.linalg.norm(D[:3, 3])))
        n_corr = st["n_corr"]
        if f < 8:
            poses[f] = T
    # the same frames once more in the REFERENCE'S call order -- setInputTarget after the previous solve (front_end.cpp:267),
    # then setInputSource and scanMatching back to back (:314, :321) with no synchronisation of the caller's in between: what
    # a maintainer who follows INTEGRATION.md section 1 pays per frame, split by call.  Only set_source + scan_match sit
    # inside the per-frame latency (:314-:322); set_target is the tail of the previous frame.
    t_tgt, t_src, t_sm = [], [], []
    for f in range(nf):
        sc = kitti_scene(synth, args.seed, f)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        H.set_input_target(sc.target)
        tb = time.perf_counter()
        H.set_input_source(sc.source)
        tc = time.perf_counter()
        rc, T, st = H.scan_match(sc.T_pred)
        td = time.perf_counter()
        if rc != 0:
            raise SystemExit(f"KITTI-density frame {f} (call-order pass): scan_match failed: {reg.STATUS.get(rc, rc)}")
        t_tgt.append((tb - ta) * 1e3); t_src.append((tc - tb) * 1e3); t_sm.append((td - tc) * 1e3)
    H.close()
    ms = np.array(ms)
    split = {"set_target_ms": round(float(np.mean(t_tgt)), 4), "set_source_ms": round(float(np.mean(t_src)), 4),
             "scan_match_ms": round(float(np.mean(t_sm)), 4),
             "set_source_plus_scan_match_ms": round(float(np.mean(np.array(t_src) + np.array(t_sm))), 4),
             # (this block's scan_match does not contain the grid builds the headline's does -- they are enqueued by
             #  set_target -- so the figure to hold beside the headline is the sum of the two calls)
             "set_target_plus_scan_match_ms": round(float(np.mean(np.array(t_tgt) + np.array(t_sm))), 4),
             "all_three_calls_ms": round(float(np.mean(np.array(t_tgt) + np.array(t_src) + np.array(t_sm))), 4),
             "note": "reference call order, no caller-side synchronisation between the calls: set_target (4 submap clouds, 83.5k "
                     "points, H2D + bounds, synchronises; then ENQUEUES the four search-grid builds behind it without waiting -- the ~24 us of launches the "
                     "headline's scan_match contains are outside this block's scan_match; TLOAM_NO_GRID_AHEAD=1 puts them back) | set_source (4 scan clouds, 9.4k points: pinned staging + ONE "
                     "copy kernel reading it in place, returns without waiting for the device) | scan_match (waits for the copy through the "
                     "stream).  Only set_source + scan_match sit between front_end.cpp:314 and :322"}
    rep = {"workload": "synthetic KITTI-density sequence (9.4k src / 83.5k tgt pts per frame, reference caps 2500/2000/1200/200)",
           "grids": "built at target hand-over (tloam_set_target_frame), not inside scan_match -- unlike the headline, whose frames "
                    "are staged in HBM and selected, and whose scan_match builds its four grids itself",
           "frames": nf, "ms_per_frame": round(float(ms.mean()), 4), "ms_per_frame_p50": round(float(np.median(ms)), 4),
           "ms_per_frame_p99": round(float(np.percentile(ms, 99)), 4),
           "gn_iters_per_sec": round(it / (ms.sum() * 1e-3), 1), "gn_iters_per_frame": round(it / nf, 2),
           "solver_evaluations_per_frame": round(ev / nf, 2),
           "ms_per_frame_incl_pcie_upload": round(float(np.mean(ms_up)), 4),
           "per_call": split,
           "n_corr_last_frame": n_corr,
           "pose_err_vs_truth_m": {"mean": round(float(np.mean(terr)), 6), "max": round(float(np.max(terr)), 6)}}
    return {"report": rep, "poses": poses}


def adjacent_rows(args, reg, torch, device):
    """The rows either side of the path, timed on the GPU at KITTI-like sizes: the device-resident
    submap update (FrontEnd::updateSubmap) and the PCA feature extraction (extractPlanarSphere).  Reported
    beside the headline, never part of `value`."""
    from tloam_amd import synth_submap as ss
    H = reg.HipRegistration(reg.default_config(), device=device)
    H.submap_init(*ss.frame_clouds(args.seed, 0, n=(4000, 500, 7000, 30000), extent=60.0))
    ts = []
    # (the scans are generated BEFORE the timed calls: in the pipeline an update follows its scan_match at once -- a device left
    #  idle for the milliseconds numpy takes to make a cloud answers the next call from a lower power state)
    scans = [(ss.frame_pose(f), ss.frame_clouds(args.seed, f, n=(4000, 500, 2000, 4000), extent=60.0)) for f in range(1, 40)]
    for T, cl in scans:
        t0 = time.perf_counter()
        H.submap_update(T, *cl)
        ts.append(time.perf_counter() - t0)
    sizes = [len(H.get_target(k)) for k in range(4)]
    cloud = ss.feature_cloud(args.seed, n=100000)
    for _ in range(2):
        H.extract_planar_sphere(cloud)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lists = H.extract_planar_sphere(cloud)
    tf = (time.perf_counter() - t0) / 5
    H.close()
    return {"submap_update_ms": round(float(np.mean(ts[-20:])) * 1e3, 4),
            "submap_update_ms_p50": round(float(np.median(ts[-20:])) * 1e3, 4), "submap_points": sizes,
            "feature_extract_ms": round(tf * 1e3, 4), "feature_cloud_points": int(len(cloud)),
            "feature_lists": [int(len(x)) for x in lists],
            "note": "host call to host return, incl. the upload of the per-scan clouds / the cloud and the list download"}


def multi_stream(args, reg, synth, device, streams=3, frames=300):
    """A single frame stream keeps one MI355X busy with kernels of a few blocks each.  Frames of DIFFERENT streams are
    independent (BASELINE.json configs[4] puts one per GPU); this block runs `streams` of them on ONE GPU from one process
    -- a context, a HIP stream and a host thread each (the C call releases the GIL) -- and reports the aggregate.  Not part
    of `value`: a single vehicle's frames are sequential, the headline stays the latency of one stream.
    Three streams: the HIP runtime multiplexes a process's streams onto four hardware queues (GPU_MAX_HW_QUEUES; more
    queues measured worse) and torch's own default stream holds one of them in this process -- a fourth frame stream would
    share a queue with another and serialise (scripts/multi_stream.py, without torch: 1 / 2 / 3 / 4 streams = 43 / 79 / 100 /
    124 k GN iter/s, 8 streams 138 k)."""
    import threading
    sc = kitti_scene(synth, args.seed, 105)
    try:
        Hs = [reg.HipRegistration(reg.default_config(), device=device) for _ in range(streams)]
        for H in Hs:
            H.set_frames(sc.source, sc.target)
            for _ in range(10):
                H.scan_match(sc.T_pred)
        its, bad = [0] * streams, [0] * streams
        start = threading.Barrier(streams + 1)

        # (an OpenMP runtime loaded by torch may have narrowed the main thread's affinity mask, which new threads
        #  inherit: four polling threads on one core measure the scheduler, so each gets a core of its own)
        try:
            cores = sorted(os.sched_getaffinity(0))
            if len(cores) < streams + 1:
                os.sched_setaffinity(0, range(os.cpu_count() or 1))
                cores = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = []

        def run(i):
            if len(cores) > streams:
                try:
                    os.sched_setaffinity(0, {cores[1 + i]})
                except OSError:
                    pass
            start.wait()
            for _ in range(frames):
                rc, T, st = Hs[i].scan_match(sc.T_pred)
                its[i] += st["gn_sweeps"]
                bad[i] += rc != 0
        th = [threading.Thread(target=run, args=(i,)) for i in range(streams)]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        for H in Hs:
            H.close()
        if sum(bad):
            return {"error": "scan_match failed in %d calls" % sum(bad)}
        return {"workload": "%d independent streams of the headline frame pair on one GPU, one context + one host thread each" % streams,
                "streams": streams, "frames_per_stream": frames, "frames_per_sec": round(streams * frames / dt, 1),
                "gn_iters_per_sec": round(sum(its) / dt, 1), "ms_per_frame_per_stream": round(dt / frames * 1e3, 4),
                "host_cores_available": len(cores)}
    except Exception as e:  # noqa: BLE001 -- a side measurement never takes the line down
        return {"error": f"{type(e).__name__}: {e}"}


def odometry_loop(args, reg, torch, device):
    """SURVEY 8(d) config 2 in miniature: the whole inner loop of FrontEnd::updateLidarOdometry
    (front_end.cpp:278-337) on the device over a CONSISTENT synthetic street -- per frame setInputSource (the four
    scan clouds cross PCIe), scanMatching against the device-resident submap, updateSubmap with the pose just
    found.  The targets never leave HBM.  Reports ms per frame and the drift against the generator's trajectory."""
    from tloam_amd import synth, synth_world as sw
    nf = args.loop_frames
    W = sw.make_world(seed=args.seed)
    Ts = sw.trajectory(nf + 1)
    scans = [sw.scan(W, Ts[f], args.seed, f) for f in range(nf + 1)]      # generated outside the timed loop
    H = reg.HipRegistration(reg.default_config(), device=device)
    s0 = scans[0]
    H.submap_init(s0[0], s0[3], s0[2], s0[1])
    est = [np.eye(4)]
    ms, errs, its = [], [], 0
    for f in range(1, nf + 1):
        sc = scans[f]
        pred = est[-1] @ (np.linalg.inv(est[-2]) @ est[-1] if len(est) > 1 else np.eye(4))   # front_end.cpp:329-330
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H.set_input_source(synth.Frame(sc[0], sc[1], sc[2], sc[3]))      # setInputSource(const Frame&): one call
        rc, T, st = H.scan_match(pred)
        if rc != 0:
            raise SystemExit(f"odometry loop frame {f}: {reg.STATUS.get(rc, rc)}")
        H.submap_update(T, sc[0], sc[3], sc[2], sc[1])
        ms.append((time.perf_counter() - t0) * 1e3)
        est.append(T)
        its += st["gn_sweeps"]
        errs.append(float(np.linalg.norm((np.linalg.inv(T) @ Ts[f])[:3, 3])))
    sizes = [len(H.get_target(k)) for k in range(4)]
    H.close()
    ms = np.array(ms[5:])                                                  # first frames: buffers still growing
    return {"workload": "consistent synthetic street, %d frames, 0.8 m / frame; per frame: set_input_source (4 clouds) + scan_match "
                        "+ submap_update (device-resident submap)" % nf,
            "ms_per_frame": round(float(ms.mean()), 4), "ms_per_frame_p50": round(float(np.median(ms)), 4),
            "ms_per_frame_p99": round(float(np.percentile(ms, 99)), 4), "gn_iters_per_frame": round(its / nf, 2),
            "submap_points_end": sizes, "path_length_m": round(float(np.linalg.norm(Ts[nf][:3, 3])), 2),
            "final_position_error_m": round(errs[-1], 4), "max_position_error_m": round(max(errs), 4)}


def cpu_baseline(head, side, args, kitti_seq=None):
    """The CPU path timed beside the GPU's in the same run: the oracle -- a dependency-free port of the
    reference's Ceres-configured solve, NOT Ceres -- compiled for this host as the survey specifies for the timed path
    (-O3 -march=native, its own `fast` build; the -ffp-contract=off parity build is not what is timed), swept over the
    evaluator thread counts {1,2,4,8,16,32} with min(4, threads) builder threads (the reference runs its four builders as
    four async tasks, registration.cpp:976-1020, and Ceres on hardware_concurrency()/2 threads, :184,:1044).  `value` is
    the BEST of the sweep with its thread count in `cores`; the single-thread figure and the reference's own thread shape
    are reported beside it.  Bounded sample: the headline frame pair x10 per thread count, one 1 M frame, and the first 8
    frames of the KITTI-density sequence, where the pose the GPU returned is checked against the port's."""
    from oracle import binding as ob
    cores = os.cpu_count() or 1

    mismatch = []

    def port(block, reps, bt, et, grain=None):
        cfg = block["cfg"]
        oc = ob.make_config(**{f: getattr(cfg, f) for f, _ in cfg._fields_ if f != "reserved0"})
        O = ob.Oracle(oc, builder_threads=bt, eval_threads=et, fast=True, eval_grain=grain)
        O.set_frames(block["scene"].source, block["scene"].target)
        O.scan_match(block["scene"].T_pred)          # warm (thread pool, caches)
        t0 = time.perf_counter()
        ev = 0
        for _ in range(reps):
            rc, T, st = O.scan_match(block["scene"].T_pred)
            ev += st["gn_evaluations"]
        dt = time.perf_counter() - t0
        # the numerator of `value`: GN iterations at DISTINCT points -- what the GPU line counts (tloam_stats.gn_sweeps of this
        # very frame pair: the minimiser's evaluations of a point it has just evaluated are served from the totals in hand).  The
        # port has no such reuse and executes every evaluation as a sweep of its own: that count is `evals` (VERDICT round 5,
        # weak 7: the two lines used to count different things under one unit)
        it = reps * block["last_frame"]["gn_sweeps"]
        if ev != reps * block["last_frame"]["gn_evaluations"]:   # (the -O3 -march=native build of the port took another branch somewhere)
            mismatch.append((ev / reps, block["last_frame"]["gn_evaluations"]))
        return it / dt, dt / reps * 1e3, ev / dt

    small = head["ms_per_frame"] < 1.0
    sweep = {}
    for et in [t for t in (1, 2, 4, 8, 16, 32) if t <= cores]:
        v, ms, evs = port(head, 10 if small else 1, min(4, et), et)
        sweep[et] = {"value": round(v, 3), "ms_per_frame": round(ms, 3), "sweeps_executed_per_sec": round(evs, 3)}
    best = max(sweep, key=lambda t: sweep[t]["value"])
    res = {"value": sweep[best]["value"], "unit": "GN iter/s", "cores": best, "host_cores": cores, "kind": "port",
           "build": "oracle/tloam_oracle.c, gcc -O3 -march=native -fopenmp -pthread (the cpu_baseline build; parity uses -ffp-contract=off)",
           "threads": "builders: 4 OpenMP tasks (registration.cpp:976-1020 runs four std::async builders); evaluator: a persistent pool of "
                      "`cores` threads (Ceres keeps one for the problem's context), static slices, cache-line-padded accumulators; a sweep "
                      "uses min(cores, blocks / 256) of them (orc_set_eval_grain) -- Ceres has no such bound",
           "ms_per_frame": sweep[best]["ms_per_frame"], "workload": head["workload"],
           "gpu_ms_per_frame": round(head["ms_per_frame"], 4),
           "numerator": "GN iterations at distinct points per second of wall time -- the SAME count as the line's `value` "
                        "(tloam_stats.gn_sweeps of the frame pair the port is timed on: %d per frame); the port executes every one of the "
                        "minimiser's %d evaluations per frame as a sweep (`sweeps_executed_per_sec`)" %
                        (head["last_frame"]["gn_sweeps"], head["last_frame"]["gn_evaluations"]),
           "sweeps_executed_per_sec": sweep[best]["sweeps_executed_per_sec"],
           "evaluations_equal_gpu": not mismatch,
           "note": "value = the BEST of the thread sweep (its thread count in `cores`); the reference's own thread shape (4 builder tasks, "
                   "hardware_concurrency()/2 evaluator threads, registration.cpp:184,:1044) is `reference_thread_shape`, with the port's "
                   "one-thread-per-256-blocks bound and, under `every_thread_every_sweep`, without it (all 128 threads woken for the 46 "
                   "blocks each of a KITTI-cap frame: what an unbounded ParallelFor pays); labelled port, NOT Ceres.  A reported "
                   "baseline, not a target: the GPU/CPU ratio says nothing about kernel quality",
           "single_thread": sweep[1], "thread_sweep": {str(t): sweep[t] for t in sweep},
           "sample": "scan_match of the headline frame pair x10 per thread count (1..32 evaluator threads, min(4, t) builder "
                     "threads) + one 1M frame + the first 8 frames of the KITTI-density sequence"}
    ref_et = max(1, cores // 2)
    if ref_et not in sweep:   # the reference's own shape: 4 builder tasks, Ceres on hardware_concurrency()/2 threads
        v, ms, _ = port(head, 3 if small else 1, 4, ref_et)
        v0, ms0, _ = port(head, 3 if small else 1, 4, ref_et, grain=0)
        res["reference_thread_shape"] = {"value": round(v, 3), "ms_per_frame": round(ms, 3), "cores": ref_et,
                                         "every_thread_every_sweep": {"value": round(v0, 3), "ms_per_frame": round(ms0, 3)}}
    if side is not None and side is not head:
        v1, ms1, _ = port(side, 1, 4, min(32, max(1, cores // 2)))
        res["m1_frame"] = {"value": round(v1, 3), "ms_per_frame": round(ms1, 2), "cores": min(32, max(1, cores // 2))}
    if kitti_seq is not None:
        # the port on the first frames of the KITTI-density sequence: CPU ms/frame beside the GPU's, and the pose the
        # GPU returned for those frames checked against the PARITY build of the port (the oracle as the checker)
        from tloam_amd import registration as reg
        from tloam_amd import synth
        kc = reg.default_config()
        ko = ob.make_config(**{f: getattr(kc, f) for f, _ in kc._fields_ if f != "reserved0"})
        K = ob.Oracle(ko, builder_threads=min(4, best), eval_threads=best, fast=True)
        K1 = ob.Oracle(ko, builder_threads=1, eval_threads=1)   # parity build, single thread: the checker
        tms, tms1, dts, drs, its = [], [], [], [], 0
        for f, T_gpu in sorted(kitti_seq["poses"].items()):
            sc = kitti_scene(synth, args.seed, f)
            K1.set_frames(sc.source, sc.target)
            t0 = time.perf_counter()
            rc, T_cpu, stc = K1.scan_match(sc.T_pred)
            tms1.append((time.perf_counter() - t0) * 1e3)
            K.set_frames(sc.source, sc.target)
            t0 = time.perf_counter()
            K.scan_match(sc.T_pred)
            tms.append((time.perf_counter() - t0) * 1e3)
            its += stc["gn_evaluations"]
            D = np.linalg.inv(T_cpu) @ T_gpu
            dts.append(float(np.linalg.norm(D[:3, 3])))
            drs.append(float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1.0) / 2.0, -1.0, 1.0))))
        if tms:
            res["kitti_sequence"] = {"frames": len(tms), "ms_per_frame": round(float(np.mean(tms)), 3), "cores": best,
                                    "ms_per_frame_parity_build_1_thread": round(float(np.mean(tms1)), 3),
                                    "gn_iters_per_sec": round(its / (sum(tms) * 1e-3), 1),
                                    "gpu_vs_port_pose_delta": {"max_dt_m": max(dts), "max_dR_rad": max(drs)}}
    return res


def kitti_replay(args, reg, torch, device):
    """--kitti-dir: BASELINE.json configs[0]/[1] in their literal wording -- a KITTI sequence directory replayed through
    tloam_amd/replay.py (reader, stand-in labeller, device PCA feature extraction, scan matching, device submap) with the
    trajectory written in the reference's savePose format.  Absent directory: reported, nothing run."""
    from tloam_amd import replay
    files = replay.list_scans(args.kitti_dir)
    if not files:
        return {"kitti_dir": args.kitti_dir, "skipped": "no .bin scans found (KITTI data is not part of this image)"}
    outp = args.kitti_out or os.path.join(args.kitti_dir, "tloam_hip_poses.txt")
    H = reg.HipRegistration(reg.default_config(), device=device)
    try:
        poses, rep = replay.replay(H, files, out_poses=outp, max_frames=args.kitti_max_frames or None,
                                   sync=torch.cuda.synchronize)
    finally:
        H.close()
    rep.update({"kitti_dir": args.kitti_dir, "trajectory_file": outp,
                "path_length_m": round(float(sum(np.linalg.norm(poses[i][:3, 3] - poses[i - 1][:3, 3]) for i in range(1, len(poses)))), 2),
                "note": "labels from the stand-in labeller of tloam_amd/replay.py, not the reference's DCVC segmentation"})
    return rep


if __name__ == "__main__":
    main()
