#!/usr/bin/env python3
"""TEST TOOL (GPU box): the C ABI under random call orders and bad arguments.

A host may call the entry points in any order and with any sizes; the contract is "a status code, never a crash, a hang or a
corrupted context".  This drives one context through `steps` randomly chosen calls -- valid ones, calls out of order (the stepwise
API without its begin, a solve without clouds, a getter before any solve, a frame store slot that holds nothing), sizes of 0 / 9 /
10 points, kinds and slots out of range, null pointers where the header allows them and where it does not, predictions that are
not poses -- checks every status against the set the header documents, and every few hundred calls hands a clean frame over and
checks the solve against the CPU restatement (the context must still be good).

    python tests/tools/fuzz_call_order.py [steps=3000] [seed=0]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from tloam_amd import registration as reg, synth  # noqa: E402

KNOWN = set(reg.STATUS.keys()) | {0}


def pose_delta(A, B):
    D = np.linalg.inv(A) @ B
    R = D[:3, :3]
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * 0.5
    return float(np.linalg.norm(D[:3, 3])), float(np.arctan2(np.linalg.norm(w), (np.trace(R) - 1.0) * 0.5))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    scenes = [synth.make_scene(seed=70 + i) for i in range(3)]
    want = []
    for sc in (scenes if not os.environ.get("FUZZ_ONLY") else []):   # (a reducer trial makes no clean-frame check)
        O = ob.Oracle()
        O.set_frames(sc.source, sc.target)
        rc, T, st = O.scan_match(sc.T_pred)
        assert rc == 0
        want.append((T, st["n_corr"]))
    # two larger frames for the hand-over ops only (four-lane and thread-per-query search, the large-set sweep and finish kernels,
    # the query sort): what is checked on them is "a status code and the context stays good", not the pose
    big = [synth.make_scene(seed=80, n_src=(6000, 8000, 5000, 1000), n_tgt=(8000, 9000, 6000, 1500)),
           synth.make_scene(seed=81, n_src=(40_000, 50_000, 35_000, 8_000), n_tgt=(30_000, 30_000, 20_000, 5_000)),
           # above the thread-per-query limit: with the caps lifted (a context re-created by op 27 may have them so) the frame keeps
           # its factors as a DIRECT set (rows in the search's order, two weight streams, the finish riding on the next search)
           synth.make_scene(seed=82, n_src=(70_000, 40_000, 30_000, 8_000), n_tgt=(40_000, 30_000, 20_000, 5_000))]
    LIFTED = dict(planar_maxnum=1 << 30, ground_maxnum=1 << 30, edge_maxnum=1 << 30, sphere_maxnum=1 << 30)
    H = reg.HipRegistration()
    L, h = H.L, H.h
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))          # noqa: E731
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))           # noqa: E731
    null_d = C.POINTER(C.c_double)()
    counts = {}
    seen = {}

    def note(name, rc):
        if os.environ.get("FUZZ_LOG"):
            print("   -> %s: %s %s" % (name, reg.STATUS.get(rc, rc), (L.tloam_last_error(h) or b"").decode()[:100] if rc else ""), file=sys.stderr, flush=True)
        counts[name] = counts.get(name, 0) + 1
        assert rc in KNOWN, (name, rc)
        # no call of this tool may end in a failed HIP call: nothing it does exhausts the device, so TLOAM_E_HIP here is a defect of
        # the library (round 6: a getter that copied past the end of a buffer was "a documented status" for a round)
        assert reg.STATUS.get(rc) != "TLOAM_E_HIP", (name, (L.tloam_last_error(h) or b"").decode()[:300])
        seen.setdefault(name, set()).add(rc)

    def cloud(n):
        return np.ascontiguousarray(rng.normal(0, 10, (n, 3)))

    def some_pose():
        k = rng.integers(0, 5)
        P = scenes[rng.integers(0, 3)].T_pred.copy()
        if k == 0:
            P[:3, :3] *= 1.0 + rng.uniform(0.001, 0.5)
        elif k == 1:
            P[rng.integers(0, 3), 3] = rng.choice([np.nan, np.inf, -np.inf])
        elif k == 2:
            P[3, :] = rng.normal(0, 1, 4)
        return np.ascontiguousarray(P.T.ravel())      # column-major

    # FUZZ_ONLY="3,17,40": only these calls are made (every call draws from a generator of its own, seeded by (seed, call
    # number), so a call is the same call whatever else runs -- what tests/tools/fuzz_reduce.py needs to shrink a failure)
    only = os.environ.get("FUZZ_ONLY")
    only = None if not only else set(int(v) for v in only.split(","))

    def clean_check(i):
        sc, (T_want, n_want) = scenes[i % 3], want[i % 3]
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0, ("clean frame refused", rc, L.tloam_last_error(h))
        dt, dr = pose_delta(T, T_want)
        assert st["n_corr"] == n_want and dt < 1e-6 and dr < 1e-6, ("context corrupted", dt, dr, st["n_corr"], n_want)

    t0 = time.time()
    sync_buf = np.zeros(8)
    res = np.zeros(16)
    stats = reg.Stats()
    for i in range(steps):
        rng = np.random.default_rng([seed, i])
        if only is not None and i not in only:
            continue
        h = H.h
        op = int(rng.integers(0, 33))
        kind = int(rng.choice([-1, 0, 1, 2, 3, 4, 7]))
        n = int(rng.choice([0, 1, 9, 10, 11, 200, 3000]))
        if os.environ.get("FUZZ_LOG"):     # (a fault of the GPU ends the process: the last lines say which call it was)
            print("call %d: op %d kind %d n %d" % (i, op, kind, n), file=sys.stderr, flush=True)
        if op == 0:
            a = cloud(max(n, 1))
            note("set_source", L.tloam_set_source(h, kind, dp(a), n))
        elif op == 1:
            a = cloud(max(n, 1))
            note("set_target", L.tloam_set_target(h, kind, dp(a), n))
        elif op == 2:
            note("set_source(null)", L.tloam_set_source(h, kind, null_d, n))
        elif op == 3:
            j = int(rng.integers(0, 17))
            sc = big[j - 14] if j >= 14 else scenes[j % 3]
            H.set_frames(sc.source, sc.target)
        elif op == 4:
            note("sm_begin", L.tloam_sm_begin(h, dp(some_pose()), null_d))
        elif op == 5:
            done = C.c_int(0)
            note("sm_outer", L.tloam_sm_outer(h, C.byref(done), C.byref(stats)))
        elif op == 6:
            note("sm_end", L.tloam_sm_end(h, dp(res), C.byref(stats)))
        elif op == 7:
            note("scan_match", L.tloam_scan_match(h, dp(some_pose()), null_d, dp(res), null_d, 0, C.byref(stats)))
        elif op == 8:
            sc = scenes[rng.integers(0, 3)]
            scan = cloud(n) if n else None
            note("scan_match+cloud", L.tloam_scan_match(h, dp(np.ascontiguousarray(sc.T_pred.T.ravel())), null_d, dp(res),
                                                        dp(scan) if scan is not None else null_d, n, C.byref(stats)))
        elif op == 9:
            f, r = np.zeros(1), np.zeros(1)
            note("fitness", L.tloam_fitness(h, dp(f), dp(r)))
        elif op == 10:
            note("frame_stash", L.tloam_frame_stash(h, int(rng.choice([-3, -1, 0, 1, 2, 5, 1000, 1 << 20]))))
        elif op == 11:
            note("frame_select", L.tloam_frame_select(h, int(rng.choice([-3, -1, 0, 1, 2, 5, 1000, 1 << 20]))))
        elif op == 12:
            cap = int(rng.choice([0, 1, 100, 5000]))
            nn = C.c_size_t(0)
            idx = np.zeros(max(cap, 1), np.int32); a = np.zeros((max(cap, 1), 3)); b = np.zeros((max(cap, 1), 3)); d = np.zeros(max(cap, 1))
            w = np.zeros(max(cap, 1)); cs = np.zeros(max(cap, 1))
            if rng.integers(0, 2):
                rc = L.tloam_get_correspondences(h, kind, C.c_size_t(cap), C.byref(nn), ip(idx), dp(a), dp(b), dp(d), dp(w), dp(cs))
            else:   # "any output pointer may be NULL"
                rc = L.tloam_get_correspondences(h, kind, C.c_size_t(cap), C.byref(nn), ip(idx), null_d, null_d, null_d, dp(w), null_d)
            note("get_correspondences", rc)
            assert nn.value <= 10_000_000
        elif op == 13:
            cap = int(rng.choice([0, 1, 100, 5000]))
            nn = C.c_size_t(0)
            w = np.zeros(max(cap, 1))
            note("get_weights", L.tloam_get_weights(h, kind, C.c_size_t(cap), C.byref(nn), dp(w)))
        elif op == 14:
            q = cloud(max(n, 1))
            k = int(rng.choice([0, 1, 5, 8, 9, 64, -2]))
            radius = float(rng.choice([-1.0, 0.0, 0.5, 3.0, 1e9, np.nan]))
            kk = max(k, 1)
            idx = np.zeros((max(n, 1), kk), np.int32); d2 = np.zeros((max(n, 1), kk)); cnt = np.zeros(max(n, 1), np.int32)
            if os.environ.get("FUZZ_LOG"):
                print("   knn k %d radius %r" % (k, radius), file=sys.stderr, flush=True)
            note("knn", L.tloam_knn(h, kind, dp(q), n, radius, k, ip(idx), dp(d2), ip(cnt)))
        elif op == 15:
            x = np.ascontiguousarray(rng.normal(0, [1, 1, 1, 0.1, 0.1, 0.1]))
            Hm = np.zeros(36); g = np.zeros(6); c = C.c_double(0)
            note("accumulate", L.tloam_accumulate(h, dp(x), dp(Hm), dp(g), C.byref(c)))
        elif op == 16:
            x = np.ascontiguousarray(rng.normal(0, [1, 1, 1, 0.1, 0.1, 0.1]))
            note("solve", L.tloam_solve(h, dp(x), C.byref(stats)))
        elif op == 17:
            rt = int(rng.choice([-1, 0, 1, 2, 3]))
            m = max(n, 1)
            p = cloud(m); a = cloud(m); b = cloud(m); d = rng.normal(0, 1, m); w = rng.uniform(0, 1, m)
            a /= np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-9)
            note("set_correspondences", L.tloam_set_correspondences(h, rt, n, dp(p), dp(a), dp(b), dp(d), dp(w)))
        elif op == 18:
            cl = [cloud(max(int(rng.choice([0, 1, 50, 2000])), 1)) for _ in range(4)]
            ns = [int(rng.choice([0, 1, len(c)])) for c in cl]
            args = []
            for c, m in zip(cl, ns):
                args += [dp(c), m]
            if os.environ.get("FUZZ_LOG"):
                print("   submap sizes %r of %r" % (ns, [len(c) for c in cl]), file=sys.stderr, flush=True)
            if rng.integers(0, 2):
                note("submap_init", L.tloam_submap_init(h, None, *args))
            else:
                note("submap_update", L.tloam_submap_update(h, dp(some_pose()), *args))
        elif op == 19:
            cap = int(rng.choice([0, 10, 100000]))
            nn = C.c_size_t(0)
            out = np.zeros((max(cap, 1), 3))
            note("get_target", L.tloam_get_target(h, kind, C.c_size_t(cap), C.byref(nn), dp(out)))
        elif op == 20:
            Hm = np.zeros(36); g = np.zeros(6); c = C.c_double(0)
            note("get_normal_equations", L.tloam_get_normal_equations(h, dp(Hm), dp(g), C.byref(c)))
        elif op == 21:
            cap = int(rng.choice([0, 1, 100, 5000]))
            nn = C.c_size_t(0)
            c = np.zeros(max(cap, 1))
            note("get_costs", L.tloam_get_costs(h, int(rng.choice([-1, 0, 1, 2, 3])), C.c_size_t(cap), C.byref(nn), dp(c)))
        elif op == 22:
            # a whole Frame through the frame entry points: sizes of every kind, NULL where n == 0 (allowed) and where it is not
            which = L.tloam_set_source_frame if rng.integers(0, 2) else L.tloam_set_target_frame
            ns = [int(rng.choice([0, 1, 9, 10, 300, 4000])) for _ in range(4)]
            cl = [cloud(max(m, 1)) for m in ns]
            ptrs = (C.POINTER(C.c_double) * 4)(*[dp(cl[j]) if (ns[j] or rng.integers(0, 2)) else null_d for j in range(4)])
            if rng.integers(0, 8) == 0:
                ptrs[int(rng.integers(0, 4))] = null_d          # NULL with n > 0: refused
            note("set_*_frame", which(h, ptrs, (C.c_size_t * 4)(*ns)))
        elif op == 23:
            m = int(rng.choice([0, 1, 19, 20, 21, 500, 5000]))
            a = cloud(max(m, 1))
            if rng.integers(0, 4) == 0 and m:
                a[rng.integers(0, m)] = np.nan
            fc = reg.default_feature_config()
            if rng.integers(0, 3) == 0:
                fc.radius = float(rng.choice([-1.0, 0.0, 0.3, 5.0, 1e6, np.nan]))
            fl = np.zeros(max(m, 1)); cv = np.zeros(max(m, 1)); sp = np.zeros(max(m, 1)); nm = np.zeros((max(m, 1), 3))
            ns_ = np.zeros(max(m, 1), np.int32); ng = np.zeros((max(m, 1), max(int(fc.K), 1)), np.int32)
            if os.environ.get("FUZZ_LOG"):
                print("   pca_info m %d radius %r K %d nan-point %s" % (m, fc.radius, fc.K, bool(np.isnan(a).any())), file=sys.stderr, flush=True)
            note("pca_info", L.tloam_pca_info(h, C.byref(fc), dp(a), m, dp(fl), dp(cv), dp(sp), dp(nm), ip(ns_), ip(ng)))
        elif op == 24:
            m = int(rng.choice([0, 1, 19, 20, 21, 500, 5000]))
            a = cloud(max(m, 1))
            fc = reg.default_feature_config()
            lists = [np.zeros(max(m, 1), np.int32) for _ in range(4)]
            cnts = [C.c_size_t(0) for _ in range(4)]
            args = []
            for l_, k_ in zip(lists, cnts):
                args += [ip(l_), C.byref(k_)]
            note("extract_planar_sphere", L.tloam_extract_planar_sphere(h, C.byref(fc), dp(a), m, *args))
            assert all(k_.value <= max(m, 1) for k_ in cnts)
        elif op == 25:
            m = int(rng.choice([0, 1, 7, 300]))
            x = np.ascontiguousarray(rng.normal(0, 1, (max(m, 1), 6))); d = np.ascontiguousarray(rng.normal(0, 0.1, (max(m, 1), 6)))
            if m and rng.integers(0, 3) == 0:
                x[0, 3:] = (np.pi, 0, 0)
                d[0] = np.nan
            out = np.zeros((max(m, 1), 26))
            note("debug_se3", L.tloam_debug_se3(h, m, dp(x), dp(d), dp(out)))
        elif op == 26:
            # sharding set-up with arguments that cannot be right, then the context must still solve alone
            note("comm_init_mailbox(bad)", L.tloam_comm_init_mailbox(h, int(rng.choice([-1, 0, 3, 99])), int(rng.choice([-2, 0, 1, 17, 64])), None))
        elif op == 27:
            # the context destroyed in whatever state it is in (a stepwise frame open, clouds staged, a submap) and a new one in its
            # place: it is handed the dead one's device memory
            if rng.integers(0, 4) == 0:
                H.close()
                # (every third successor with the caps lifted: the clean-frame check holds either way -- the scenes' caps do not bind)
                H.__init__(reg.default_config(**LIFTED) if rng.integers(0, 3) == 0 else None)
                h = H.h
                note("destroy + create", 0)
        elif op == 28:
            # a second context working beside this one for a moment (allocations interleave; its blocks come back to the pool)
            sc = scenes[int(rng.integers(0, 3))]
            X = reg.HipRegistration()
            X.set_frames(sc.source, sc.target)
            rc2, T2, st2 = X.scan_match(sc.T_pred)
            note("second context", rc2)
            if rng.integers(0, 2):
                m = int(rng.choice([100, 5000, 60000]))
                X.set_correspondences(0, cloud(m), cloud(m), None, rng.normal(0, 1, m), None)
            X.close()
        elif op == 29:
            ci = reg.CtxInfo()
            note("get_info", L.tloam_get_info(h, C.byref(ci) if rng.integers(0, 4) else None))
            assert ci.fallbacks_taken == 0 and 0 <= ci.k3_wide <= 1 and ci.nranks >= 0
        elif op == 30:
            us = C.c_double(0); cnt = C.c_int64(0)
            note("gn_iter_timer", L.tloam_gn_iter_timer(h, int(rng.integers(0, 2)), C.byref(us), C.byref(cnt)))
            assert cnt.value >= 0 and us.value >= 0.0
        elif op == 31:
            gb = C.c_double(0)
            nb = int(rng.choice([0, 1, 4096, 10_000_000]))
            note("time_read_stream", L.tloam_time_read_stream(h, C.c_size_t(nb), int(rng.choice([0, 1, 3])), C.byref(gb)))
        elif op == 32:
            # a VALID one-rank mailbox set-up: from here on the context runs the sharded launch forms with a loop-back exchange
            # (same results bit for bit: the clean-frame check keeps holding) -- and a second set-up on a context that has one
            if rng.integers(0, 6) == 0:
                buf = C.create_string_buffer(64)
                rc = L.tloam_comm_mailbox_export(h, C.cast(buf, C.c_void_p))
                note("comm_mailbox_export", rc)
                if rc == 0:
                    note("comm_init_mailbox(loop-back)", L.tloam_comm_init_mailbox(h, 0, 1, C.cast(buf, C.c_void_p)))
        if os.environ.get("FUZZ_SYNC"):   # a blocking copy after every call: a GPU fault is then reported in the call that caused it
            L.tloam_debug_state(h, dp(sync_buf), 8)
        if i % 250 == 249 and only is None:
            # (a stepwise frame left open by the fuzz is closed first: sm_end is the documented way out)
            L.tloam_sm_end(h, dp(res), C.byref(stats))
            clean_check(i // 250)
    L.tloam_sm_end(h, dp(res), C.byref(stats))
    if only is None:
        clean_check(0)
    H.close()
    print("fuzz ok: %d calls in %.1f s; statuses seen per entry point:" % (steps, time.time() - t0))
    for k in sorted(seen):
        print("  %-22s x%-5d %s" % (k, counts[k], sorted(reg.STATUS.get(r, r) if r else "OK" for r in seen[k])))


if __name__ == "__main__":
    main()
