"""-m gpu: BASELINE.json's full sizes (1 M correspondences / 1 M-point frames) through size-independent
properties -- the oracle needs seconds per frame at this size, so it checks a bounded sample only."""
import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from tloam_amd import synth

pytestmark = pytest.mark.gpu
BIG = 1 << 30


@pytest.fixture(scope="module")
def prebuilt_1m():
    return synth.make_prebuilt(seed=1)


def test_k3_additivity_over_splits_1m(hip_module, prebuilt_1m):
    """H, g and cost are sums over correspondences: sweeping two halves separately must add up to the
    sweep over the whole 1 M set (the property the multi-GPU sharding relies on)."""
    sets, x_true, x_eval = prebuilt_1m
    H = hip_module.HipRegistration()
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    Hf, gf, cf = H.accumulate(x_eval)
    tot = [np.zeros((6, 6)), np.zeros(6), 0.0]
    for half in (0, 1):
        Hh = hip_module.HipRegistration()
        for rt in range(3):
            p, a, b, d, w = sets[rt]
            m = len(p) // 2
            sl = slice(0, m) if half == 0 else slice(m, None)
            Hh.set_correspondences(rt, p[sl], a[sl], None if b is None else b[sl], None if d is None else d[sl], w[sl])
        Hp, gp, cp = Hh.accumulate(x_eval)
        tot[0] += Hp; tot[1] += gp; tot[2] += cp
        Hh.close()
    np.testing.assert_allclose(tot[0], Hf, rtol=1e-11, atol=1e-11 * np.abs(Hf).max())
    np.testing.assert_allclose(tot[1], gf, rtol=1e-11, atol=1e-11 * np.abs(gf).max())
    assert abs(tot[2] - cf) < 1e-11 * cf
    assert np.allclose(Hf, Hf.T) and np.all(np.linalg.eigvalsh(Hf) > 0)


def test_k3_sample_against_oracle_1m(hip_module, prebuilt_1m):
    """Side-channel costs of the full 1 M sweep, checked against the oracle on a 20 k sample."""
    sets, x_true, x_eval = prebuilt_1m
    H = hip_module.HipRegistration()
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    H.accumulate(x_eval)
    rng = np.random.default_rng(0)
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        pick = np.sort(rng.choice(len(p), min(len(p), 7000), replace=False))
        O = ob.Oracle()
        O.set_correspondences(rt, p[pick], a[pick], None if b is None else b[pick], None if d is None else d[pick], w[pick])
        O.accumulate(x_eval)
        np.testing.assert_allclose(H.get_costs(rt)[pick], O.get_costs(rt), rtol=1e-9, atol=1e-18)


def test_solve_1m_known_answer(hip_module, prebuilt_1m):
    sets, x_true, x_eval = prebuilt_1m
    H = hip_module.HipRegistration()
    for rt in range(3):
        H.set_correspondences(rt, *sets[rt])
    x, st = H.solve(x_eval)
    assert np.linalg.norm(x[:3] - x_true[:3]) < 5e-4 and np.linalg.norm(x[3:] - x_true[3:]) < 2e-5
    assert st["accepted_steps"] >= 2
    x2, st2 = H.solve(x_eval)
    assert np.array_equal(x, x2)                        # bit-reproducible


def test_scan_match_1m_frame_properties(hip_module):
    sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
    cfg = hip_module.default_config(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
    H = hip_module.HipRegistration(cfg)
    H.set_frames(sc.source, sc.target)
    rc, T, st = H.scan_match(sc.T_pred)
    assert rc == 0 and sum(st["n_corr"]) > 800_000
    dt, dr = pose_delta(T, sc.T_true)
    assert dt < 1e-3 and dr < 1e-4 and dt < pose_delta(sc.T_pred, sc.T_true)[0] / 10
    for k in range(4):
        c = H.get_correspondences(k, capacity=len(sc.source.cloud(k)))
        assert len(c["idx"]) == st["n_corr"][k]
        assert np.all(np.diff(c["idx"]) > 0)            # index order, no duplicates
        w = H.get_weights(k)
        assert np.all((w >= 0) & (w <= 1)) and st["bad_weights"] == 0
        if k <= 1:
            assert np.allclose(np.linalg.norm(c["a"], axis=1), 1.0, atol=1e-12)   # unit plane normals
        if k == 2:
            assert np.allclose(np.linalg.norm(c["a"] - c["b"], axis=1), 0.2, atol=1e-12)
    rc, T2, st2 = H.scan_match(sc.T_pred)
    assert np.array_equal(T, T2)
    # oracle spot check of the builders on a 3 k-query sample of the 1 M-point planar cloud
    O = ob.Oracle(ob.make_config(planar_maxnum=BIG)); O.set_target(0, sc.target.planar)
    q = sc.source.planar[:3000] @ sc.T_pred[:3, :3].T + sc.T_pred[:3, 3]
    hi, hd, hc = H.knn(0, q, 0.5, 5); oi, od, oc = O.knn(0, q, 0.5, 5)
    assert np.array_equal(hc, oc) and np.array_equal(hi, oi) and np.array_equal(hd, od)
    assert np.all(np.diff(hd, axis=1)[hi[:, 1:] >= 0] >= 0)                     # ascending distances


@pytest.mark.parametrize("caps", ["lifted", "reference"])
def test_thread_per_query_builder_index_parity(hip_module, caps):
    """> 131072 source points switch K1 to the thread-per-query variant (flattened candidate stream, distance-only
    insertion with the exact-tie redo, pipelined record loads).  Full index-level parity against the oracle at a
    size it still handles in seconds: every correspondence list, the counters and the pose."""
    n_src = (70_000, 40_000, 30_000, 6_000)
    n_tgt = (80_000, 50_000, 40_000, 8_000)
    sc = synth.make_scene(seed=5, n_src=n_src, n_tgt=n_tgt, density=40.0)
    over = dict(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG) if caps == "lifted" else {}
    H = hip_module.HipRegistration(hip_module.default_config(**over))
    O = ob.Oracle(ob.make_config(**over), builder_threads=4, eval_threads=16)
    H.set_frames(sc.source, sc.target)
    O.set_frames(sc.source, sc.target)
    rh, Th, sh = H.scan_match(sc.T_pred)
    ro, To, so = O.scan_match(sc.T_pred)
    assert rh == ro == 0
    assert sh["n_corr"] == so["n_corr"] and sum(so["n_corr"]) > (50_000 if caps == "lifted" else 5_000)
    for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations"):
        assert sh[k] == so[k], k
    for kind in range(4):
        cap = len(sc.source.cloud(kind))
        assert np.array_equal(H.get_correspondences(kind, capacity=cap)["idx"], O.get_correspondences(kind, capacity=cap)["idx"])
    dt, dr = pose_delta(Th, To)
    assert dt < 1e-9 and dr < 1e-9
    # exact duplicates among the targets force bit-equal distances: the tie redo must reproduce the index order
    tgt = [np.concatenate([sc.target.cloud(k), sc.target.cloud(k)[::7]]) for k in range(4)]
    for k in range(4):
        H.set_target(k, tgt[k]); O.set_target(k, tgt[k])
    rh, Th, sh = H.scan_match(sc.T_pred)
    ro, To, so = O.scan_match(sc.T_pred)
    assert sh["n_corr"] == so["n_corr"]
    for kind in range(4):
        cap = len(sc.source.cloud(kind))
        assert np.array_equal(H.get_correspondences(kind, capacity=cap)["idx"], O.get_correspondences(kind, capacity=cap)["idx"])
    dt, dr = pose_delta(Th, To)
    assert dt < 1e-9 and dr < 1e-9
    H.close()


def test_quad_builder_mid_size_index_parity(hip_module):
    """Between 16 Ki and 128 Ki source points K1 runs four lanes per query (sixteen below, one above): index-level
    parity against the oracle at ~36 k queries."""
    n_src = (14_000, 10_000, 9_000, 3_000)
    n_tgt = (40_000, 30_000, 25_000, 5_000)
    sc = synth.make_scene(seed=9, n_src=n_src, n_tgt=n_tgt, density=25.0)
    over = dict(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
    H = hip_module.HipRegistration(hip_module.default_config(**over))
    O = ob.Oracle(ob.make_config(**over), builder_threads=4, eval_threads=16)
    H.set_frames(sc.source, sc.target)
    O.set_frames(sc.source, sc.target)
    rh, Th, sh = H.scan_match(sc.T_pred)
    ro, To, so = O.scan_match(sc.T_pred)
    assert rh == ro == 0 and sh["n_corr"] == so["n_corr"] and sum(so["n_corr"]) > 10_000
    for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations"):
        assert sh[k] == so[k], k
    for kind in range(4):
        cap = len(sc.source.cloud(kind))
        assert np.array_equal(H.get_correspondences(kind, capacity=cap)["idx"], O.get_correspondences(kind, capacity=cap)["idx"])
    dt, dr = pose_delta(Th, To)
    assert dt < 1e-9 and dr < 1e-9
    H.close()


def test_one_launch_gn_iteration_of_large_sets_is_exact(hip_module, prebuilt_1m, monkeypatch):
    """Round 4: a GN iteration of a LARGE set as one launch (TLOAM_FUSED_LARGE) -- the streaming sweep's last block folds the
    rows and advances the minimiser (k3_sweep_step) -- instead of k3_accumulate + k_reduce_and_step.  Same sweep, same
    fold tree, same step: the pre-built 1 M Solve and a 100 k-point frame (thread-per-query search, riding finish, learned
    sweep budgets over three frames) must come out bit for bit the same; the streaming span the fused launches report about
    themselves (tloam_k3_span) counts exactly the executed sweeps."""
    sets, x_true, x_eval = prebuilt_1m
    big = 1 << 30
    sc = synth.make_scene(seed=1, n_src=(50000, 26000, 20000, 4000), n_tgt=(50000, 26000, 20000, 4000))
    over = dict(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big)

    def run(H):
        for rt in range(3):
            H.set_correspondences(rt, *sets[rt])
        x, st = H.solve(x_eval)
        out = [(x.copy(), {k: st[k] for k in ("gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "solver_cost")},
                [H.get_costs(rt).copy() for rt in range(3)])]
        Hn, gn, cn = H.get_normal_equations()
        out.append((Hn.copy(), gn.copy(), cn))
        return out

    monkeypatch.setenv("TLOAM_FUSED_LARGE", "1")                           # read once, when the context is created
    H1 = hip_module.HipRegistration()
    a = run(H1)
    us, n = H1.k3_span(reset=True)
    assert n == a[0][1]["gn_sweeps"] and 5.0 < us / n < 200.0, (us, n)     # ~12 us per sweep of 74.88 MB
    monkeypatch.delenv("TLOAM_FUSED_LARGE")
    H2 = hip_module.HipRegistration()
    b = run(H2)
    assert H2.k3_span()[1] == 0
    assert np.array_equal(a[0][0], b[0][0]) and a[0][1] == b[0][1]
    for ca, cb in zip(a[0][2], b[0][2]):
        assert np.array_equal(ca, cb)
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1]) and a[1][2] == b[1][2]
    H1.close(); H2.close()
    monkeypatch.setenv("TLOAM_FUSED_LARGE", "1")
    F1 = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.delenv("TLOAM_FUSED_LARGE")
    F2 = hip_module.HipRegistration(hip_module.default_config(**over))
    for F in (F1, F2):
        F.set_frames(sc.source, sc.target)
    for frame in range(3):
        rc1, T1, st1 = F1.scan_match(sc.T_pred)
        rc2, T2, st2 = F2.scan_match(sc.T_pred)
        assert rc1 == rc2 == 0 and np.array_equal(T1, T2)
        for k in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "outer_iterations", "kind_cost", "se3"):
            assert np.array_equal(np.asarray(st1[k]), np.asarray(st2[k])), k
        for k in range(4):
            c1 = F1.get_correspondences(k, capacity=len(sc.source.cloud(k)))
            c2 = F2.get_correspondences(k, capacity=len(sc.source.cloud(k)))
            for f in ("idx", "w", "cost"):
                assert np.array_equal(c1[f], c2[f]), (k, f)
            assert np.array_equal(F1.get_weights(k), F2.get_weights(k))
    F1.close(); F2.close()


def test_single_pass_scans_and_their_time_out_fallback_are_exact(hip_module):
    """The two ~3 M-entry histograms of a 1 M-class frame (cells of the target grids, bins of the query sort) are scanned by
    ONE single-pass look-back launch each -- the grid's also writes cell_start and re-zeroes the histogram -- where the
    device's CU count says all blocks of the launch are resident at once; the look-back is bounded, and a wait that runs out
    (a partitioned / shared device) raises a fault word: tloam_scan_match then runs the SAME frame again with
    tile scan + scan of the totals + add (+ finalize) and the context stays there.  H2's fault word is raised by hand
    (tloam_debug_raise_fault) before its first frame: first frame = the re-run path, second = the multi-launch scans from the
    start.  Integer work: the frames must come out identical, twice in a row (epochs, re-armed counters), and a 1 M-point kNN
    as well."""
    sc = synth.make_scene(seed=0, n_src=synth.M1_SRC, n_tgt=synth.M1_TGT)
    cfg = hip_module.default_config(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
    H1 = hip_module.HipRegistration(cfg)
    H2 = hip_module.HipRegistration(cfg)
    for H in (H1, H2):
        H.set_frames(sc.source, sc.target)
    assert H2.L.tloam_debug_raise_fault(H2.h, 0) == 0
    for rep in range(2):
        rc1, T1, st1 = H1.scan_match(sc.T_pred)
        rc2, T2, st2 = H2.scan_match(sc.T_pred)
        assert rc1 == rc2 == 0 and np.array_equal(T1, T2)
        for k in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "outer_iterations", "kind_cost", "se3"):
            assert np.array_equal(np.asarray(st1[k]), np.asarray(st2[k])), k
        for k in range(4):
            n = len(sc.source.cloud(k))
            assert np.array_equal(H1.get_correspondences(k, capacity=n)["idx"], H2.get_correspondences(k, capacity=n)["idx"]), k
    q = sc.source.planar[:2000] @ sc.T_pred[:3, :3].T + sc.T_pred[:3, 3]
    a, b = H1.knn(0, q, 0.5, 5), H2.knn(0, q, 0.5, 5)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    H1.close(); H2.close()


# ---- the DIRECT factor set of large frames (tloam_amd/csrc/tl_common.hpp DirectSet) ------------------------------------------
def _direct_fingerprint(H, sc):
    out = {}
    for k in range(4):
        c = H.get_correspondences(k, capacity=len(sc.source.cloud(k)))
        out[k] = {f: np.asarray(c[f]).copy() for f in ("idx", "a", "b", "d", "w", "cost") if f in c and c[f] is not None}
        out[k]["weights"] = H.get_weights(k).copy()
    return out


def test_direct_set_against_compact_set_and_stepwise_api(hip_module, monkeypatch):
    """Frames searched one thread per query whose caps cannot bind keep their factors as a DIRECT set: the search writes every
    factor into the row of its query (tile-sorted order, fixed for the frame), queries without a factor leave holes that add exact
    zeros, the Solve reads one of two weight streams and the finish writes the other -- no flag scan, no compaction.  Held here
    against (i) the COMPACT set of the same frames (TLOAM_NO_DIRECT_SET): the same factors -- every index list, every captured
    weight and GNC weight, every counter --, the same geometry, and a pose within 1e-12 (the sums run in another order: the last
    bits differ); (ii) the stepwise API (TLOAM_NO_DEVICE_LOOP: the finish as a launch of its own instead of riding on the next
    search): bit for bit; (iii) the oracle: index lists and pose.  Three frames, so that the learned sweep budgets are in play."""
    n_src, n_tgt = (70_000, 40_000, 30_000, 6_000), (80_000, 50_000, 40_000, 8_000)
    sc = synth.make_scene(seed=5, n_src=n_src, n_tgt=n_tgt, density=40.0)
    over = dict(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
    Hd = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.setenv("TLOAM_NO_DEVICE_LOOP", "1")
    Hs = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.delenv("TLOAM_NO_DEVICE_LOOP")
    monkeypatch.setenv("TLOAM_NO_DIRECT_SET", "1")
    Hc = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.delenv("TLOAM_NO_DIRECT_SET")
    for H in (Hd, Hs, Hc):
        H.set_frames(sc.source, sc.target)
    for frame in range(3):
        rd, Td, sd = Hd.scan_match(sc.T_pred)
        rs, Ts, ss = Hs.scan_match(sc.T_pred)
        rcc, Tc, scs = Hc.scan_match(sc.T_pred)
        assert rd == rs == rcc == 0
        assert Hd.info()["direct_set"] == 1 and Hs.info()["direct_set"] == 1 and Hc.info()["direct_set"] == 0
        fd, fs, fc = _direct_fingerprint(Hd, sc), _direct_fingerprint(Hs, sc), _direct_fingerprint(Hc, sc)
        # (ii) device-driven loop == stepwise API, bit for bit
        assert np.array_equal(Td, Ts)
        for key in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "outer_iterations", "kind_cost", "se3", "converged_early"):
            assert np.array_equal(np.asarray(sd[key]), np.asarray(ss[key])), key
        for k in range(4):
            for f in fd[k]:
                assert np.array_equal(fd[k][f], fs[k][f], equal_nan=True), (k, f)
        # (i) direct == compact: the same factors, the same geometry, the pose to the last bits' rounding
        dt, dr = pose_delta(Td, Tc)
        assert dt < 1e-12 and dr < 1e-12, (dt, dr)
        for key in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "outer_iterations", "converged_early"):
            assert np.array_equal(np.asarray(sd[key]), np.asarray(scs[key])), key
        np.testing.assert_allclose(sd["kind_cost"], scs["kind_cost"], rtol=1e-9)
        for k in range(4):
            assert np.array_equal(fd[k]["idx"], fc[k]["idx"]), k
            assert len(fd[k]["idx"]) == sd["n_corr"][k] > 0
            for f in ("a", "b", "d"):
                if f in fd[k] and fd[k][f] is not None and len(np.atleast_1d(fd[k][f])):
                    np.testing.assert_allclose(fd[k][f], fc[k][f], rtol=0, atol=1e-9, err_msg=f"{k} {f}")
            np.testing.assert_allclose(fd[k]["w"], fc[k]["w"], rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(fd[k]["weights"], fc[k]["weights"], rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(fd[k]["cost"], fc[k]["cost"], rtol=1e-6, atol=1e-18)
    # (iii) the oracle
    O = ob.Oracle(ob.make_config(**over), builder_threads=4, eval_threads=16)
    O.set_frames(sc.source, sc.target)
    ro, To, so = O.scan_match(sc.T_pred)
    assert ro == 0 and sd["n_corr"] == so["n_corr"]
    for k in range(4):
        assert np.array_equal(fd[k]["idx"], O.get_correspondences(k, capacity=len(sc.source.cloud(k)))["idx"])
    dt, dr = pose_delta(Td, To)
    assert dt < 1e-9 and dr < 1e-9
    for H in (Hd, Hs, Hc):
        H.close()


def test_direct_set_is_rebuilt_when_the_loop_ends_beside_a_search_that_had_run(hip_module, monkeypatch):
    """The device-driven loop runs the search of outer iteration k in the launch that finishes iteration k - 1, on the Solve's
    own verdict ("ended at another pose than the set was built at").  If that finish then ends the loop (the plateau test of
    registration.cpp:1108) the search has written the rows of a set that will never be solved: the result slot says so
    (OS_SET_STALE) and the getters rebuild the rows at the pose of the SOLVED set before they read them.  Forced here: a large
    prediction error (the first Solve runs out of its four iterations still moving), a noise bound that keeps every weight at 1
    (the second Solve moves on), a plateau threshold that ends the loop right after it.  The compact path of the same frame
    (which never had the problem: a discarded search only leaves raw records behind) is the reference."""
    n_src, n_tgt = (70_000, 40_000, 30_000, 6_000), (80_000, 50_000, 40_000, 8_000)
    pred_err = tuple(12.0 * np.array((0.012, -0.008, 0.004, 0.0015, -0.001, 0.002)))
    sc = synth.make_scene(seed=6, n_src=n_src, n_tgt=n_tgt, density=40.0, pred_err=pred_err)
    over = dict(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG, noise_bound=1.0e4, cost_threshold=1.0e15)
    Hd = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.setenv("TLOAM_NO_DIRECT_SET", "1")
    Hc = hip_module.HipRegistration(hip_module.default_config(**over))
    monkeypatch.delenv("TLOAM_NO_DIRECT_SET")
    for H in (Hd, Hc):
        H.set_frames(sc.source, sc.target)
    rd, Td, sd = Hd.scan_match(sc.T_pred)
    rcc, Tc, scs = Hc.scan_match(sc.T_pred)
    assert rd == rcc == 0 and sd["converged_early"] == 1 and sd["outer_iterations"] == 2 == scs["outer_iterations"]
    info = Hd.info()
    assert info["direct_set"] == 1 and info["set_stale"] == 1, (info, sd)      # the case the test is about did happen
    fd = _direct_fingerprint(Hd, sc)
    assert Hd.info()["set_stale"] == 0                                          # ... and the getters have dealt with it
    fc = _direct_fingerprint(Hc, sc)
    dt, dr = pose_delta(Td, Tc)
    assert dt < 1e-12 and dr < 1e-12
    for k in range(4):
        assert np.array_equal(fd[k]["idx"], fc[k]["idx"]), k
        for f in ("a", "b", "d"):
            if f in fd[k] and fd[k][f] is not None and len(np.atleast_1d(fd[k][f])):
                np.testing.assert_allclose(fd[k][f], fc[k][f], rtol=0, atol=1e-9, err_msg=f"{k} {f}")
        np.testing.assert_allclose(fd[k]["cost"], fc[k]["cost"], rtol=1e-6, atol=1e-18)
    Hd.close(); Hc.close()


def test_one_context_through_direct_compact_small_and_prebuilt_frames(hip_module):
    """State must not leak between the forms a context's frames take.  ONE context is walked through: a large frame with lifted
    caps (direct set) | a KITTI-size frame (one-launch Solve) | the large frame with the reference's caps (compact: they bind) |
    a LARGER lifted frame (direct, buffers grow, other row counts) | a pre-built set (compact by definition) | the first frame
    again -- and after every step a FRESH context solves the same input: pose, counters, index lists, captured weights, GNC
    weights and side-channel costs must agree bit for bit (weight-stream parity, stale-set flag, the sort's riding first pass,
    the planar weight-stream bit of the sweep's flag word are all per frame)."""
    big = dict(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
    A = synth.make_scene(seed=21, n_src=(70_000, 40_000, 30_000, 6_000), n_tgt=(80_000, 50_000, 40_000, 8_000), density=40.0)
    B = synth.make_scene(seed=22, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
    C_ = synth.make_scene(seed=23, n_src=(90_000, 50_000, 40_000, 9_000), n_tgt=(90_000, 60_000, 50_000, 9_000), density=40.0)
    sets, x_true, x_eval = synth.make_prebuilt(seed=4, n_plane=30_001, n_line=7_777, n_point=1_300)
    steps = [("direct", A, big, 1), ("small", B, {}, 0), ("caps", A, {}, 0), ("direct_larger", C_, big, 1), ("prebuilt", None, None, 0), ("direct_again", A, big, 1)]
    ctx = {}   # one long-lived context per configuration (the caps are part of a context): each sees every frame of ITS configuration in turn

    def fingerprint(H, sc, T, st):
        out = [T.tobytes(), tuple(st[k] if not hasattr(st[k], "tolist") else tuple(np.asarray(st[k]).tolist())
                                  for k in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "outer_iterations", "converged_early"))]
        for k in range(4):
            c = H.get_correspondences(k, capacity=len(sc.source.cloud(k)))
            out += [c["idx"].tobytes(), c["w"].tobytes(), c["cost"].tobytes(), H.get_weights(k).tobytes()]
        return out

    # the long-lived contexts: `big` and default caps; every step runs on the one its caps ask for, so each of the two is taken
    # through direct -> (prebuilt) -> direct and small -> caps respectively, interleaved with the other's frames on the same GPU
    for name, sc, over, direct in steps:
        if name == "prebuilt":
            for key in list(ctx):
                H = ctx[key]
                for rt in range(3):
                    H.set_correspondences(rt, *sets[rt])
                x, ps = H.solve(x_eval)
                F = hip_module.HipRegistration()
                for rt in range(3):
                    F.set_correspondences(rt, *sets[rt])
                xf, pf = F.solve(x_eval)
                assert np.array_equal(x, xf) and ps["gn_evaluations"] == pf["gn_evaluations"], name
                assert H.info()["direct_set"] == 0
                F.close()
            continue
        key = tuple(sorted(over.items()))
        if key not in ctx:
            ctx[key] = hip_module.HipRegistration(hip_module.default_config(**over))
        H = ctx[key]
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0 and H.info()["direct_set"] == direct, (name, H.info())
        F = hip_module.HipRegistration(hip_module.default_config(**over))
        F.set_frames(sc.source, sc.target)
        rcf, Tf, stf = F.scan_match(sc.T_pred)
        assert rcf == 0
        for i, (a, b) in enumerate(zip(fingerprint(H, sc, T, st), fingerprint(F, sc, Tf, stf))):
            assert a == b, (name, i)
        # ... and the same frame a second time on the long-lived context (learned sweep budgets, grids built ahead or not)
        rc2, T2, st2 = H.scan_match(sc.T_pred)
        assert rc2 == 0 and T2.tobytes() == T.tobytes(), name
        F.close()
    # the SAME context across the cap change: a context cannot change its caps, so the direct <-> compact switch on one context is
    # driven by the frame's SIZE instead -- a frame below the thread-per-query limit compacts, the large one is direct again
    H = ctx[tuple(sorted(big.items()))]
    mid = synth.make_scene(seed=24, n_src=(14_000, 10_000, 9_000, 3_000), n_tgt=(40_000, 30_000, 25_000, 5_000), density=25.0)
    for sc, direct in ((mid, 0), (A, 1), (mid, 0)):
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        F = hip_module.HipRegistration(hip_module.default_config(**big))
        F.set_frames(sc.source, sc.target)
        rcf, Tf, stf = F.scan_match(sc.T_pred)
        assert rc == rcf == 0 and H.info()["direct_set"] == direct
        for i, (a, b) in enumerate(zip(fingerprint(H, sc, T, st), fingerprint(F, sc, Tf, stf))):
            assert a == b, ("size switch", direct, i)
        F.close()
    for H in ctx.values():
        H.close()


@pytest.mark.parametrize("direct", [False, True], ids=["compact", "direct"])
def test_getters_after_a_new_source_cloud_describe_the_frame_that_was_solved(hip_module, direct):
    """The weights and the correspondences a context hands out are those of the frame its last scanMatching began with, whatever
    source cloud has been registered since.  Until round 6 tloam_get_weights (and the direct set's getters) took their sizes from
    the REGISTERED cloud: a larger cloud handed over after a solve turned the getter into a copy past the end of the weights
    (TLOAM_E_HIP, found by tests/tools/fuzz_call_order.py once it stopped accepting that status)."""
    import ctypes as C
    if direct:
        n_src, n_tgt = (70_000, 40_000, 30_000, 6_000), (80_000, 50_000, 40_000, 8_000)
        cfg = hip_module.default_config(planar_maxnum=BIG, ground_maxnum=BIG, edge_maxnum=BIG, sphere_maxnum=BIG)
        sc = synth.make_scene(seed=5, n_src=n_src, n_tgt=n_tgt, density=40.0)
        other = synth.make_scene(seed=6, n_src=tuple(2 * v for v in n_src), n_tgt=n_tgt, density=40.0)
    else:
        cfg = hip_module.default_config()
        sc = synth.make_scene(seed=11)
        other = synth.make_scene(seed=12, n_src=tuple(3 * len(sc.source.cloud(k)) for k in range(4)))
    H = hip_module.HipRegistration(cfg)
    H.set_frames(sc.source, sc.target)
    rc, T, st = H.scan_match(sc.T_pred)
    assert rc == 0 and H.info()["direct_set"] == (1 if direct else 0)
    w0 = [H.get_weights(k).copy() for k in range(4)]
    c0 = [H.get_correspondences(k) for k in range(4)]
    sizes0 = [len(w) for w in w0]
    H.set_input_source(other.source)            # a LARGER cloud of every kind; nothing is solved with it yet
    for k in range(4):
        n = C.c_size_t(0)
        cap = 4 * max(sizes0) + 16
        w = np.zeros(cap)
        rc = H.L.tloam_get_weights(H.h, k, cap, C.byref(n), w.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0 and n.value == sizes0[k], (k, rc, n.value, sizes0[k])
        assert np.array_equal(w[:n.value], w0[k])
        c1 = H.get_correspondences(k, capacity=cap)
        assert c1["idx"].tolist() == c0[k]["idx"].tolist() and np.array_equal(c1["w"], c0[k]["w"]) and np.array_equal(c1["a"], c0[k]["a"])
    # ... and the next solve is the new frame's
    rc, T2, st2 = H.scan_match(sc.T_pred)
    assert rc == 0 and [len(H.get_weights(k)) for k in range(4)] == [2 * v if direct else 3 * v for v in sizes0]
    H.close()
