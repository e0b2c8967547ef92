"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances: bit-exact for index work (k-NN indices, correspondence index lists, counts);
|dt| < 1e-6 m and |dR| < 1e-6 rad for poses (BASELINE.json north_star); 1e-9 relative for the
fp64 normal equations (different summation order / FMA contraction only).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from oracle import oracle_np as onp
from tloam_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL_T = 1e-6
POSE_TOL_R = 1e-6


def _cfg_pair(reg, **over):
    return reg.default_config(**over), ob.make_config(**over)


def _make_pair(reg, scene, **over):
    hc, oc = _cfg_pair(reg, **over)
    H = reg.HipRegistration(hc)
    O = ob.Oracle(oc)
    H.set_frames(scene.source, scene.target)
    O.set_frames(scene.source, scene.target)
    return H, O


@pytest.mark.parametrize("kind,radius,k", [(0, 0.5, 5), (1, 0.5, 5), (2, 1.0, 5), (3, 0.5, 1), (1, 0.02, 1), (2, 2.5, 8)])
def test_knn_matches_oracle(hip_module, kind, radius, k):
    sc = synth.make_scene(seed=3)
    H, O = _make_pair(hip_module, sc)
    rng = np.random.default_rng(0)
    tgt = sc.target.cloud(kind)
    q = np.concatenate([tgt[rng.integers(0, len(tgt), 400)] + rng.normal(0, 0.15, (400, 3)),
                        rng.uniform(-80, 80, (50, 3)), tgt[:20]])
    hi, hd, hc = H.knn(kind, q, radius, k)
    oi, od, oc = O.knn(kind, q, radius, k)
    assert np.array_equal(hc, oc)
    assert np.array_equal(hi, oi)
    assert np.array_equal(hd, od)          # same un-fused fp64 arithmetic -> bit-exact distances


def test_knn_brute_force_spot_check(hip_module):
    sc = synth.make_scene(seed=4)
    H, _ = _make_pair(hip_module, sc)
    tgt = sc.target.cloud(2)
    q = tgt[::37][:64] + 0.05
    hi, hd, hc = H.knn(2, q, 1.0, 5)
    for j in range(len(q)):
        bi, bd = ob.knn_brute(tgt, q[j], 1.0, 5)
        assert hc[j] == len(bi)
        assert np.array_equal(hi[j, :hc[j]], bi)
        assert np.array_equal(hd[j, :hc[j]], bd)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scan_match_stepwise_parity(hip_module, seed):
    """Every outer GNC iteration: same correspondences (index lists bit-exact), same minimiser
    bookkeeping, same weights, pose within 1e-6."""
    sc = synth.make_scene(seed=seed)
    H, O = _make_pair(hip_module, sc)
    assert H.sm_begin(sc.T_pred) == 0
    assert O.sm_begin(sc.T_pred) == 0
    cfg = hip_module.default_config()
    nb2 = cfg.noise_bound * cfg.noise_bound
    mu = 1e-10                                             # registration.cpp:1027-1033 with every residual slot still 0
    w_prev = [np.ones(len(sc.source.cloud(k))) for k in range(4)]   # :931-949
    gn_iterations_before = 0
    for it in range(4):
        rc_h, done_h, st_h = H.sm_outer()
        rc_o, done_o, st_o = O.sm_outer()
        assert rc_h == 0 and rc_o == 0
        assert st_h["n_corr"] == st_o["n_corr"], (it, st_h["n_corr"], st_o["n_corr"])
        # the point the last sweep of this Solve evaluated: the last candidate (GnState::x_cand survives the re-arming; accepted
        # or not, the sweep that evaluated it is the last one), or the start point if the Solve never took a step
        raw = np.zeros(64)
        assert H.L.tloam_debug_state(H.h, raw.ctypes.data_as(C.POINTER(C.c_double)), 64) > 0
        took_a_step = st_h["gn_iterations"] > gn_iterations_before
        T_last = onp.se3_exp(raw[17:23] if took_a_step else np.asarray(st_h["se3"]))
        gn_iterations_before = st_h["gn_iterations"]
        for kind in range(4):
            ch, co = H.get_correspondences(kind), O.get_correspondences(kind)
            # updateWeight (registration.cpp:858-876) EXACTLY, on the HIP path's OWN side-channel costs: the cross-check against
            # the oracle's weights below is loose from iteration 1 on (conditioning, see there) -- this one is not, so a wrong
            # weight formula cannot hide behind it.  <= 2 ulp: one correctly rounded division, square root and subtraction.
            th1, th2 = (mu + 1) / mu * nb2, mu / (mu + 1) * nb2
            c = ch["cost"]
            with np.errstate(divide="ignore", invalid="ignore"):
                mid = np.sqrt(nb2 * mu * (mu + 1) / c) - mu
            w_fac = np.where(c >= th1, 0.0, np.where(c <= th2, 1.0, mid))
            want = w_prev[kind].copy()
            upd = c != 0                                   # :862 a slot whose cost is 0 keeps its weight
            want[ch["idx"][upd]] = w_fac[upd]
            got = H.get_weights(kind)
            assert np.all(np.abs(got - want) <= 2 * np.spacing(np.maximum(np.abs(want), mu))), (it, kind, np.abs(got - want).max())
            assert np.array_equal(got[want == 0.0], want[want == 0.0]) and np.array_equal(got[want == 1.0], want[want == 1.0])
            w_prev[kind] = got
            # R1-R3 + the side-channel semantics (registration.cpp:19-117) at the pose the LAST sweep of this Solve
            # evaluated -- the HIP path's OWN candidate, read back from the device state -- with the HIP path's own factors,
            # through the independent numpy restatement: tight in EVERY iteration.  (The comparison with the oracle's costs below
            # is loose from iteration 1 on, because the two sides' rejected candidates differ by the conditioning of the step; a
            # 1e-7 relative error in the residual code that only showed at such far-away poses would hide there -- round-4 review.)
            if len(ch["idx"]):
                cs = onp.CorrSet(kind, ch["idx"].astype(np.int64), sc.source.cloud(kind)[ch["idx"]], ch["a"], ch["b"], ch["d"], ch["w"])
                _, _, side = onp.residual_blocks(cs, T_last)
                # cost = r^2 with r a difference of coordinates of ~50 m: r carries ~1e-14 m of rounding whatever the operation
                # order, the cost 2 |r| times that; beyond it only 1e-12 relative
                tol = 4e-14 * np.sqrt(np.abs(side)) + 1e-12 * np.abs(side) + 1e-28
                bad = np.abs(ch["cost"] - side) > tol
                assert not bad.any(), (it, kind, int(bad.sum()), float(np.abs(ch["cost"] - side)[bad].max()))
            assert np.array_equal(ch["idx"], co["idx"]), (it, kind)
            np.testing.assert_allclose(ch["a"], co["a"], rtol=0, atol=1e-9)
            if kind == 2:   # the edge line's second endpoint (registration.cpp:484)
                assert len(ch["b"]) == len(co["b"]) > 0
                np.testing.assert_allclose(ch["b"], co["b"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(ch["d"], co["d"], rtol=0, atol=1e-9)
            # What bounds the agreement, iteration by iteration (the tolerances say so instead of being uniformly loose):
            #  * iteration 0 runs on weights that are exactly 1 and ends at a converged point.  Its side-channel costs are
            #    r^2 with r ~ 1e-3 m the difference of coordinates ~50 m: r carries ~1e-14 absolute = ~1e-11 relative
            #    rounding whatever the summation order, the cost twice that -- 1e-9 with margin;
            #  * iteration 1 captures the weights derived from those costs (w ~ 1e-7 / |r|): 1e-8;
            #  * from iteration 1 on the side-channel costs are those of a REJECTED candidate -- a point
            #    obtained by solving a 6x6 system of condition ~1e8 built from weights ~1e-5, tens of metres away: last-bit
            #    differences come back amplified by that condition number, in the costs and in everything derived from them.
            np.testing.assert_allclose(ch["w"], co["w"], rtol=(0.0, 1e-8)[it] if it <= 1 else 1e-6, atol=1e-15)
            np.testing.assert_allclose(ch["cost"], co["cost"], rtol=1e-9 if it == 0 else 1e-6, atol=1e-16 if it == 0 else 1e-13)
            np.testing.assert_allclose(H.get_weights(kind), O.get_weights(kind), rtol=1e-8 if it == 0 else 1e-6, atol=1e-13)
        assert (st_h["gn_iterations"], st_h["accepted_steps"], st_h["gn_evaluations"]) == \
               (st_o["gn_iterations"], st_o["accepted_steps"], st_o["gn_evaluations"]), (it, st_h, st_o)
        np.testing.assert_allclose(st_h["se3"], st_o["se3"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(st_h["kind_cost"], st_o["kind_cost"], rtol=1e-7, atol=1e-14)
        # the linear system itself, per outer iteration: H, g and the cost at the accepted iterate of this Solve
        Hh, gh, ch_ = H.get_normal_equations()
        Ho, go, co_ = O.get_normal_equations()
        tol = (1e-12, 1e-8)[it] if it <= 1 else 1e-6    # (the weights the system is built from: exactly 1, to 1e-8, to 1e-6)
        np.testing.assert_allclose(Hh, Ho, rtol=tol, atol=tol * np.abs(Ho).max())
        np.testing.assert_allclose(gh, go, rtol=0, atol=max(tol, 1e-11) * max(np.abs(go).max(), np.abs(Ho).max() * 1e-3))
        assert abs(ch_ - co_) <= tol * abs(co_) and abs(ch_ - st_h["solver_cost"]) == 0.0
        assert done_h == done_o
        if done_h:
            break
        mu = mu * np.exp((it + 1) * cfg.gnc_factor)        # :1089
    _, T_h, st_h = H.sm_end()
    _, T_o, st_o = O.sm_end()
    dt, dr = pose_delta(T_h, T_o)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R, (dt, dr)
    assert st_h["outer_iterations"] == st_o["outer_iterations"]


def test_scan_match_one_call_and_scan_cloud(hip_module):
    sc = synth.make_scene(seed=5)
    H, O = _make_pair(hip_module, sc)
    scan_h = np.ascontiguousarray(sc.source.planar.copy())
    scan_o = scan_h.copy()
    rc, T_h, st_h = H.scan_match(sc.T_pred, scan=scan_h)
    assert rc == 0
    rc, T_o, st_o = O.scan_match(sc.T_pred, scan=scan_o)
    assert rc == 0
    dt, dr = pose_delta(T_h, T_o)
    assert dt < POSE_TOL_T and dr < POSE_TOL_R
    np.testing.assert_allclose(scan_h, scan_o, rtol=0, atol=1e-6)
    # and the solve actually moved the prediction towards the truth
    assert pose_delta(T_h, sc.T_true)[0] < pose_delta(sc.T_pred, sc.T_true)[0]


def test_prebuilt_accumulate_and_solve(hip_module):
    sets, x_true, x_eval = synth.make_prebuilt(seed=1, n_plane=7600, n_line=2000, n_point=400)
    H = hip_module.HipRegistration()
    O = ob.Oracle()
    for rt in range(3):
        p, a, b, d, w = sets[rt]
        H.set_correspondences(rt, p, a, b, d, w)
        O.set_correspondences(rt, p, a, b, d, w)
    Hh, gh, ch = H.accumulate(x_eval)
    Ho, go, co = O.accumulate(x_eval)
    np.testing.assert_allclose(Hh, Ho, rtol=1e-10, atol=1e-10 * np.abs(Ho).max())
    np.testing.assert_allclose(gh, go, rtol=1e-10, atol=1e-10 * np.abs(go).max())
    assert abs(ch - co) <= 1e-11 * abs(co)
    for rt in range(3):
        np.testing.assert_allclose(H.get_costs(rt), O.get_costs(rt), rtol=1e-9, atol=1e-18)
    xh, sh = H.solve(x_eval)
    xo, so = O.solve(x_eval)
    np.testing.assert_allclose(xh, xo, rtol=0, atol=1e-9)
    assert (sh["gn_iterations"], sh["accepted_steps"], sh["gn_evaluations"]) == \
           (so["gn_iterations"], so["accepted_steps"], so["gn_evaluations"])
    # known answer: the solve lands on the generating pose (noise 2-5 cm over ~10 k blocks)
    assert np.linalg.norm(xh[:3] - x_true[:3]) < 5e-3 and np.linalg.norm(xh[3:] - x_true[3:]) < 2e-4


def test_build_reuse_is_exact(hip_module, monkeypatch):
    """An outer iteration that starts from a bit-identical pose reuses the previous correspondence records
    instead of re-running K1/K2; switching the reuse off (development knob) must not change a single bit."""
    sc = synth.make_scene(seed=11, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    H1 = hip_module.HipRegistration()
    H1.set_frames(sc.source, sc.target)
    rc1, T1, st1 = H1.scan_match(sc.T_pred)
    monkeypatch.setenv("TLOAM_NO_BUILD_REUSE", "1")      # read once, when the context is created
    H2 = hip_module.HipRegistration()
    H2.set_frames(sc.source, sc.target)
    rc2, T2, st2 = H2.scan_match(sc.T_pred)
    assert rc1 == rc2 == 0
    assert np.array_equal(T1, T2)
    assert st1["n_corr"] == st2["n_corr"] and st1["gn_evaluations"] == st2["gn_evaluations"]
    for k in range(4):
        assert np.array_equal(H1.get_weights(k), H2.get_weights(k))
    H1.close(); H2.close()


def test_evaluation_reuse_is_exact(hip_module, monkeypatch):
    """The minimiser serves the evaluation of a bit-identical point from the totals already on the device
    (gn_sweeps <= gn_evaluations).  With the reuse switched off every evaluation runs its own sweep; pose,
    counters, weights and side-channel costs must not change by a single bit."""
    sc = synth.make_scene(seed=12, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    H1 = hip_module.HipRegistration()
    H1.set_frames(sc.source, sc.target)
    rc1, T1, st1 = H1.scan_match(sc.T_pred)
    monkeypatch.setenv("TLOAM_NO_EVAL_REUSE", "1")       # read once, when the context is created
    H2 = hip_module.HipRegistration()
    H2.set_frames(sc.source, sc.target)
    rc2, T2, st2 = H2.scan_match(sc.T_pred)
    assert rc1 == rc2 == 0
    assert np.array_equal(T1, T2)
    for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "n_corr", "outer_iterations"):
        assert st1[k] == st2[k], k
    assert st2["gn_sweeps"] == st2["gn_evaluations"]
    assert st1["gn_sweeps"] < st1["gn_evaluations"], "this scene retries rejected steps (SURVEY A.13)"
    assert np.array_equal(st1["kind_cost"], st2["kind_cost"])
    for k in range(4):
        assert np.array_equal(H1.get_weights(k), H2.get_weights(k))
    # and the oracle executes every evaluation: same evaluation count
    O = ob.Oracle()
    O.set_frames(sc.source, sc.target)
    rc, To, sto = O.scan_match(sc.T_pred)
    assert sto["gn_evaluations"] == st1["gn_evaluations"] and sto["gn_sweeps"] == sto["gn_evaluations"]
    H1.close(); H2.close()


@pytest.mark.parametrize("planned", ["1", "2", "5"])
def test_sweep_budget_top_up_is_exact(hip_module, monkeypatch, planned):
    """Only as many sweeps as a Solve is expected to need are enqueued; the weight update / finish kernels
    are gated on the minimiser having terminated and the host tops the Solve up otherwise.  Forcing the budget
    (development knob) exercises the top-up path: nothing may change."""
    sc = synth.make_scene(seed=13, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    H1 = hip_module.HipRegistration()
    H1.set_frames(sc.source, sc.target)
    ref = [H1.scan_match(sc.T_pred) for _ in range(2)][-1]        # second frame: learned budgets in use
    monkeypatch.setenv("TLOAM_PLANNED_SWEEPS", planned)
    H2 = hip_module.HipRegistration()
    H2.set_frames(sc.source, sc.target)
    got = H2.scan_match(sc.T_pred)
    assert ref[0] == got[0] == 0
    assert np.array_equal(ref[1], got[1])
    for k in ("gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps", "n_corr", "outer_iterations", "bad_weights"):
        assert ref[2][k] == got[2][k], k
    assert np.array_equal(ref[2]["kind_cost"], got[2]["kind_cost"])
    for k in range(4):
        assert np.array_equal(H1.get_weights(k), H2.get_weights(k))
    H1.close(); H2.close()


def _frame_fingerprint(H, T, st):
    out = {"T": T, "stats": {k: st[k] for k in ("n_corr", "gn_evaluations", "gn_sweeps", "gn_iterations", "accepted_steps",
                                                "outer_iterations", "converged_early", "kind_cost", "se3", "mu",
                                                "solver_cost")}}
    for k in range(4):
        cs = H.get_correspondences(k)
        out[f"idx{k}"], out[f"w{k}"], out[f"cost{k}"] = cs["idx"], cs["w"], cs["cost"]
        out[f"wsrc{k}"] = H.get_weights(k)
    return out


def _assert_same_frame(f1, f2, cost_sum_rtol=0.0):
    """cost_sum_rtol: the four side-channel cost SUMS (registration.cpp:1091-1094) are added up by the one-launch Solve in
    the order its waves hold the factors, by the finish kernel of the other Solve paths in the order of its 1024 threads --
    the same numbers in a different tree.  Everything else (pose, every single cost, weight, index, counter) is compared exactly."""
    assert np.array_equal(f1["T"], f2["T"])
    for k, v in f1["stats"].items():
        if k == "kind_cost" and cost_sum_rtol > 0:
            np.testing.assert_allclose(np.asarray(v), np.asarray(f2["stats"][k]), rtol=cost_sum_rtol, atol=0)
            continue
        assert np.array_equal(np.asarray(v), np.asarray(f2["stats"][k])), k
    for k in f1:
        if k not in ("T", "stats"):
            assert np.array_equal(f1[k], f2[k]), k


@pytest.mark.parametrize("seed", [31, 32])
def test_quad_builder_ties_and_near_ties(hip_module, seed):
    """Small frames run K1 with four lanes per query and the packed-key lists merged across the quad.  Targets with
    exact duplicates and planted near-ties (relative distance difference ~1e-12 from a source point's predicted
    position) must still give the oracle's lists index for index, and its pose."""
    sc = synth.make_scene(seed=seed, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    rng = np.random.default_rng(seed)
    T = sc.T_pred
    tgt = []
    for k in range(4):
        t = sc.target.cloud(k)
        src_w = sc.source.cloud(k) @ T[:3, :3].T + T[:3, 3]
        add = [t[::6]]                                             # exact duplicates
        for q in src_w[rng.choice(len(src_w), min(60, len(src_w)), replace=False)]:
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            v = np.cross(u, rng.normal(size=3)); v /= np.linalg.norm(v)
            r = rng.uniform(0.02, 0.1)
            add.append(np.array([q + r * u, q + r * (1.0 + 1e-12) * v, q - r * u]))
        tgt.append(np.ascontiguousarray(np.vstack([t] + add)))
    H = hip_module.HipRegistration()
    O = ob.Oracle(ob.make_config())
    for k in range(4):
        H.set_source(k, sc.source.cloud(k)); O.set_source(k, sc.source.cloud(k))
        H.set_target(k, tgt[k]); O.set_target(k, tgt[k])
    rh, Th, sh = H.scan_match(sc.T_pred)
    ro, To, so = O.scan_match(sc.T_pred)
    assert rh == ro == 0 and sh["n_corr"] == so["n_corr"]
    for kind in range(4):
        assert np.array_equal(H.get_correspondences(kind)["idx"], O.get_correspondences(kind)["idx"])
    for k in ("gn_evaluations", "gn_iterations", "accepted_steps", "outer_iterations"):
        assert sh[k] == so[k], k
    dt, dr = pose_delta(Th, To)
    assert dt < 1e-9 and dr < 1e-9
    H.close()


@pytest.mark.parametrize("n_src,n_tgt", [(synth.SMALL_SRC, synth.SMALL_TGT), ((40_000, 50_000, 35_000, 8_000), (30_000, 30_000, 20_000, 5_000))])
def test_nan_targets_are_never_neighbours(hip_module, n_src, n_tgt):
    """A NaN coordinate gives a NaN distance, which under the reference's comparisons (nanoflann result set,
    `<` only) is never a neighbour.  The packed-key walks (both K1 variants) must treat it the same way: targets
    with NaN rows appended behave exactly like the clean cloud -- same lists, same pose, bit for bit."""
    sc = synth.make_scene(seed=41, n_src=n_src, n_tgt=n_tgt)
    H = hip_module.HipRegistration()
    H.set_frames(sc.source, sc.target)
    rc1, T1, st1 = H.scan_match(sc.T_pred)
    lists1 = [H.get_correspondences(k, capacity=len(sc.source.cloud(k)))["idx"] for k in range(4)]
    bad = np.array([[np.nan, 0.0, 0.0], [1.0, np.nan, 2.0], [np.nan, np.nan, np.nan], [3.0, 4.0, np.nan]])
    for k in range(4):
        H.set_target(k, np.ascontiguousarray(np.vstack([sc.target.cloud(k), bad, bad])))
    rc2, T2, st2 = H.scan_match(sc.T_pred)
    assert rc1 == rc2 == 0
    assert np.array_equal(T1, T2) and st1["n_corr"] == st2["n_corr"] and st1["gn_evaluations"] == st2["gn_evaluations"]
    for k in range(4):
        assert np.array_equal(lists1[k], H.get_correspondences(k, capacity=len(sc.source.cloud(k)))["idx"])
    H.close()


@pytest.mark.parametrize("n_src,n_tgt,over", [(synth.SMALL_SRC, synth.SMALL_TGT, {}),
                                               (synth.SMALL_SRC, synth.SMALL_TGT, dict(planar_maxnum=90, ground_maxnum=130, edge_maxnum=70, sphere_maxnum=25)),
                                               ((40_000, 50_000, 35_000, 8_000), (30_000, 30_000, 20_000, 5_000), {})],
                         ids=["small", "small_caps_bind", "thread_per_query"])
def test_non_finite_points_are_never_matched(hip_module, n_src, n_tgt, over):
    """Review item 7c.  The reference hands every source point to nanoflann as it is (registration.cpp:444/:535/:588/:731): a NaN
    or infinite coordinate gives NaN / inf distances, the result set's `<` never admits them, SearchHybrid returns 0 and the
    builder moves on -- such a point is simply never matched (it still counts as a source index and, for the sphere builder,
    in sphere_sum, :551).  Same here, for SOURCE rows (NaN, +-inf, 1e300) spliced into the clouds and for infinite TARGET rows:
    status 0, the index lists of the clean frame shifted by the spliced rows, the same pose -- against the oracle bit for bit
    in the lists, and no garbage (finite pose, weights in [0, 1])."""
    sc = synth.make_scene(seed=43, n_src=n_src, n_tgt=n_tgt)
    bad = np.array([[np.nan, 0.0, 0.0], [1.0, np.inf, 2.0], [-np.inf, 0.0, 0.0], [np.nan, np.nan, np.nan], [1e300, -1e300, 1e300],
                    [np.inf, np.inf, -np.inf]])
    at = 50
    src = [np.ascontiguousarray(np.vstack([sc.source.cloud(k)[:at], bad, sc.source.cloud(k)[at:]])) for k in range(4)]
    tgt = [np.ascontiguousarray(np.vstack([sc.target.cloud(k), bad[[1, 2, 5]]])) for k in range(4)]   # (infinite rows only: no 1e300 box)
    cfg = hip_module.default_config(**over)
    Hc = hip_module.HipRegistration(cfg)
    Hc.set_frames(sc.source, sc.target)
    rc0, T0, st0 = Hc.scan_match(sc.T_pred)
    assert rc0 == 0
    H = hip_module.HipRegistration(cfg)
    O = ob.Oracle(ob.make_config(**over))
    for k in range(4):
        H.set_source(k, src[k]); H.set_target(k, tgt[k])
        O.set_source(k, src[k]); O.set_target(k, tgt[k])
    rc, T, st = H.scan_match(sc.T_pred)
    rco, To, sto = O.scan_match(sc.T_pred)
    assert rc == 0 and rco == 0
    assert np.all(np.isfinite(T)) and st["n_corr"] == sto["n_corr"]
    assert (st["gn_iterations"], st["accepted_steps"], st["gn_evaluations"]) == (sto["gn_iterations"], sto["accepted_steps"], sto["gn_evaluations"])
    dt, dr = pose_delta(T, To)
    assert dt < 1e-9 and dr < 1e-9
    for k in range(4):
        ih = H.get_correspondences(k, capacity=len(src[k]))["idx"]
        assert np.array_equal(ih, O.get_correspondences(k)["idx"]), k
        assert not np.any((ih >= at) & (ih < at + len(bad))), k          # a spliced row is never a factor
        w = H.get_weights(k)
        assert np.all((w >= 0) & (w <= 1)) and np.all(w[at:at + len(bad)] == 1.0)
        # (the sphere builder counts every source point it LOOKS at, :538/:551: where that cap binds, the spliced rows shift which
        #  points it ever gets to -- the lists then differ legitimately, HIP and oracle alike)
        if not over and not (k == 3 and len(src[3]) > cfg.sphere_maxnum):
            clean = Hc.get_correspondences(k, capacity=len(sc.source.cloud(k)))["idx"]
            assert np.array_equal(np.where(ih >= at + len(bad), ih - len(bad), ih), clean), k
    if not over and len(src[3]) <= cfg.sphere_maxnum:
        assert np.array_equal(T, T0)
    H.close(); Hc.close()


@pytest.mark.parametrize("shape", ["small", "kitti", "caps", "noise_free", "m1"])
def test_device_driven_loop_equals_host_driven_loop(hip_module, monkeypatch, shape):
    """tloam_scan_match enqueues every outer GNC iteration at once -- builders / refresh gated on device flags, the
    plateau break and the loop end decided by the finish kernel -- and waits once.  The stepwise machinery
    (TLOAM_NO_DEVICE_LOOP: the host decides between iterations, as tloam_sm_outer does) must give the same frame, bit
    for bit: pose, counters, index lists, captured weights, side-channel costs, GNC weights -- over three frames, so
    that the learned sweep budgets are in play as well."""
    over = {}
    if shape == "small":
        sc = synth.make_scene(seed=14, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    elif shape == "kitti":
        sc = synth.make_scene(seed=15, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
    elif shape == "caps":
        sc = synth.make_scene(seed=16)
        over = dict(planar_maxnum=90, ground_maxnum=130, edge_maxnum=70, sphere_maxnum=25)
    elif shape == "noise_free":
        sc = synth.make_scene(seed=17, noise=0.0)
    else:
        big = 1 << 30
        sc = synth.make_scene(seed=1, n_src=(50000, 26000, 20000, 4000), n_tgt=(50000, 26000, 20000, 4000))
        over = dict(planar_maxnum=big, ground_maxnum=big, edge_maxnum=big, sphere_maxnum=big)
    H1 = hip_module.HipRegistration(hip_module.default_config(**over))
    H1.set_frames(sc.source, sc.target)
    monkeypatch.setenv("TLOAM_NO_DEVICE_LOOP", "1")      # read once, when the context is created
    H2 = hip_module.HipRegistration(hip_module.default_config(**over))
    H2.set_frames(sc.source, sc.target)
    for frame in range(3):
        rc1, T1, st1 = H1.scan_match(sc.T_pred)
        rc2, T2, st2 = H2.scan_match(sc.T_pred)
        assert rc1 == rc2 == 0
        _assert_same_frame(_frame_fingerprint(H1, T1, st1), _frame_fingerprint(H2, T2, st2))
    H1.close(); H2.close()


def test_device_driven_loop_equals_host_driven_loop_over_many_frames(hip_module, monkeypatch, capfd):
    """The same comparison over 100 randomly drawn frame pairs -- noise, prediction error, outliers, binding caps, loop length,
    plateau threshold and GNC factor all vary -- so that the rarer branches of the device-side control are met: a
    plateau break after a Solve that moved the pose (the correspondence search that rode on that finish is then never
    compacted), loops that end at max_iterations, Solves that outrun the learned sweep budget (top-up through the
    stepwise path).  The branch census is asserted so that the test cannot silently stop covering them."""
    rng = np.random.default_rng(2024)
    census = {"plateau": 0, "max_iter": 0, "moved_then_plateau": 0, "frames": 0}
    monkeypatch.setenv("TLOAM_DEBUG_RESUME", "1")   # a line on stderr whenever the host adds a launch on the device's flag (OS_NEEDS_HOST)
    capfd.readouterr()
    for case in range(100):
        over = dict(max_iterations=int(rng.integers(2, 7)), cost_threshold=float(rng.choice([1e-3, 2e-2, 0.5, 5.0, 50.0])),
                    gnc_factor=float(rng.choice([0.5, 1.0, 1.4, 2.0])))
        if rng.random() < 0.4:
            over.update(planar_maxnum=int(rng.integers(60, 400)), ground_maxnum=int(rng.integers(60, 400)),
                        edge_maxnum=int(rng.integers(40, 200)), sphere_maxnum=int(rng.integers(10, 60)))
        scale = float(rng.choice([0.3, 1.0, 4.0]))
        pred_err = tuple(scale * np.array((0.012, -0.008, 0.004, 0.0015, -0.001, 0.002)) * rng.normal(1.0, 0.3, 6))
        sc = synth.make_scene(seed=1000 + case, noise=float(rng.choice([0.0, 0.01, 0.02, 0.05])), pred_err=pred_err,
                              outlier_frac=float(rng.choice([0.0, 0.05, 0.2])),
                              n_src=synth.SMALL_SRC if case % 3 else synth.KITTI_SRC,
                              n_tgt=synth.SMALL_TGT if case % 3 else synth.KITTI_TGT)
        monkeypatch.delenv("TLOAM_NO_DEVICE_LOOP", raising=False)
        H1 = hip_module.HipRegistration(hip_module.default_config(**over))
        H1.set_frames(sc.source, sc.target)
        monkeypatch.setenv("TLOAM_NO_DEVICE_LOOP", "1")      # read once, when the context is created
        H2 = hip_module.HipRegistration(hip_module.default_config(**over))
        H2.set_frames(sc.source, sc.target)
        for frame in range(2):
            rc1, T1, st1 = H1.scan_match(sc.T_pred)
            rc2, T2, st2 = H2.scan_match(sc.T_pred)
            assert rc1 == rc2 and rc1 in (0, -7), (case, rc1, rc2)
            _assert_same_frame(_frame_fingerprint(H1, T1, st1), _frame_fingerprint(H2, T2, st2))
            census["frames"] += 1
            if st1["converged_early"]:
                census["plateau"] += 1
                if st1["accepted_steps"] > 0 and st1["outer_iterations"] > 1:
                    census["moved_then_plateau"] += 1
            elif st1["outer_iterations"] == over["max_iterations"]:
                census["max_iter"] += 1
        H1.close(); H2.close()
    census["host_resumed"] = capfd.readouterr().err.count("[tloam resume]")
    assert census["plateau"] >= 15 and census["max_iter"] >= 15 and census["moved_then_plateau"] >= 3, census
    assert census["host_resumed"] >= 3, census   # the pose kept moving after the iterations that were enqueued ahead
    print("census", census)


def test_staged_frames_equal_direct_hand_over(hip_module):
    """tloam_frame_stash / tloam_frame_select: frames handed over ahead of their solve and activated later give,
    bit for bit, what handing each frame over right before its solve gives -- in any activation order -- and an unknown
    slot / a call inside a solve are refused."""
    reg = hip_module
    scenes = [synth.make_scene(seed=40 + i) for i in range(3)]
    ref = []
    Hd = reg.HipRegistration()
    for sc in scenes:
        Hd.set_frames(sc.source, sc.target)
        rc, T, st = Hd.scan_match(sc.T_pred)
        assert rc == 0
        ref.append((T, st))
    Hd.close()
    H = reg.HipRegistration()
    for i, sc in enumerate(scenes):
        H.set_frames(sc.source, sc.target)
        H.frame_stash(i)
    assert H.L.tloam_frame_select(H.h, 7) == -1            # TLOAM_E_INVALID: never stashed
    for i in (2, 0, 1, 1, 2):
        H.frame_select(i)
        rc, T, st = H.scan_match(scenes[i].T_pred)
        assert rc == 0
        assert np.array_equal(T, ref[i][0])
        assert st["n_corr"] == ref[i][1]["n_corr"] and st["gn_evaluations"] == ref[i][1]["gn_evaluations"]
    # a frame handed over while a slot is selected lands in that slot; the context's own frame is untouched
    H.frame_select(-1)
    H.set_frames(scenes[0].source, scenes[0].target)
    rc, T, _ = H.scan_match(scenes[0].T_pred)
    assert rc == 0 and np.array_equal(T, ref[0][0])
    # select(i) -> hand a NEW frame over (it lands in slot i) -> stash(i) -> select(i): the slot holds the new frame, and the
    # context's own frame is the registered one in between (the slot used to end up with the context's clouds)
    H.frame_select(2)
    H.set_frames(scenes[1].source, scenes[1].target)
    H.frame_stash(2)
    rc, T, _ = H.scan_match(scenes[0].T_pred)              # the context's own frame (scene 0, handed over above)
    assert rc == 0 and np.array_equal(T, ref[0][0])
    H.frame_select(2)
    rc, T, st = H.scan_match(scenes[1].T_pred)
    assert rc == 0 and np.array_equal(T, ref[1][0]) and st["n_corr"] == ref[1][1]["n_corr"]
    H.frame_select(1)
    assert H.sm_begin(scenes[1].T_pred) == 0
    assert H.L.tloam_frame_select(H.h, 0) == -6            # TLOAM_E_NOT_READY inside a solve
    H.close()                                              # (destroyed with a slot selected and a solve open: nothing leaks, nothing is freed twice)


def test_concurrent_frame_streams_share_the_gpu(hip_module):
    """Three contexts, three host threads, one GPU: the launches of different streams interleave, a block of a one-launch
    Solve starts late, a wave is slow to look at a hand-over -- nothing may depend on that.  Every frame of every stream
    must come out bit-identical to the same frame solved alone, with no hand-over timing out (a missed hand-over costs a
    full second: the run is bounded well below that per frame).  Regression test for a message overwritten before every
    wave had read it (the end-of-iteration verdict of k_solve_small)."""
    import threading
    import time
    scenes = [synth.make_scene(seed=40 + i, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT) for i in range(3)]
    alone = []
    for sc in scenes:
        H = hip_module.HipRegistration()
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0
        alone.append(_frame_fingerprint(H, T, st))
        H.close()
    Hs = [hip_module.HipRegistration() for _ in scenes]
    for H, sc in zip(Hs, scenes):
        H.set_frames(sc.source, sc.target)
    frames = 150
    out = [None] * 3
    worst = [0.0] * 3
    start = threading.Barrier(3)

    def run(i):
        start.wait()
        bad = []
        for f in range(frames):
            t = time.perf_counter()
            rc, T, st = Hs[i].scan_match(scenes[i].T_pred)
            worst[i] = max(worst[i], time.perf_counter() - t)
            if rc != 0:
                bad.append((f, rc, Hs[i].L.tloam_last_error(Hs[i].h).decode()))
        out[i] = (bad, _frame_fingerprint(Hs[i], T, st))

    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(3):
        bad, fp = out[i]
        assert bad == [], bad[:3]
        assert worst[i] < 0.25, worst           # (a frame is ~0.25 ms; a timed-out hand-over would be 1 s)
        _assert_same_frame(alone[i], fp)
    for H in Hs: H.close()


@pytest.mark.parametrize("knob", ["TLOAM_NO_PERSISTENT_SOLVE=1", "TLOAM_NO_GRID_AHEAD=1"])
def test_solve_launch_variants_are_exact(hip_module, monkeypatch, knob):
    """The Solve launch of a KITTI-size frame prepares its own factor set, ends its outer iteration and runs the following ones;
    the host enqueues launches for two iterations and adds one when the device asks.  The one fallback form -- one launch per GN
    iteration, k_prepare_small in front, the finish as a kernel of its own: what a context uses after an in-launch hand-over
    timed out, and what the stepwise API always uses -- must give the same frames: three scenes (one with a large prediction error, whose pose keeps moving
    in later outer iterations: the host-resumed path), two frames each, everything compared bit for bit (with one launch per
    GN iteration the four cost sums are added in the finish kernel's order: last bits, see _assert_same_frame).
    The search grids are built when the targets are handed over (tloam_set_target_frame); TLOAM_NO_GRID_AHEAD builds them inside
    scanMatching: the same grids.  (Round 5: the forms that lost their A/Bs -- round 3's single-consumer Solve launch, the copy
    command for the staged clouds, the synchronising set_source -- are gone)"""
    name, val = knob.split("=")
    scenes = [synth.make_scene(seed=61, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT),
              synth.make_scene(seed=62, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT, pred_err=(0.25, -0.15, 0.05, 0.02, -0.015, 0.03)),
              synth.make_scene(seed=63, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)]
    ref = []
    for sc in scenes:
        H = hip_module.HipRegistration()
        H.set_frames(sc.source, sc.target)
        ref.append([(lambda r: _frame_fingerprint(H, r[1], r[2]))(H.scan_match(sc.T_pred)) for _ in range(2)])
        H.close()
    monkeypatch.setenv(name, val)       # read once, when the context is created
    moved_late = 0
    for sc, want in zip(scenes, ref):
        H = hip_module.HipRegistration()
        H.set_frames(sc.source, sc.target)
        for f in range(2):
            rc, T, st = H.scan_match(sc.T_pred)
            assert rc == 0
            _assert_same_frame(want[f], _frame_fingerprint(H, T, st), cost_sum_rtol=1e-13 if name == "TLOAM_NO_PERSISTENT_SOLVE" else 0.0)
        moved_late += st["accepted_steps"] > 1
        H.close()


def test_hand_over_time_out_falls_back_to_one_launch_per_iteration(hip_module, monkeypatch):
    """the one-launch Solve spin-waits between its blocks.  When a hand-over times out (a block that was never
    scheduled beside the others -- forced here by TLOAM_DEBUG_FAIL_HANDOVER: the consumer waits for rows nobody posts) the
    bounded wait ends the launch, the host solves the SAME frame again with one launch per GN iteration and keeps the context
    on that path: the caller sees a slower frame, not an error, and the result is the one-launch-per-iteration result."""
    import time
    sc = synth.make_scene(seed=61, n_src=synth.KITTI_SRC, n_tgt=synth.KITTI_TGT)
    monkeypatch.setenv("TLOAM_NO_PERSISTENT_SOLVE", "1")
    H = hip_module.HipRegistration()
    H.set_frames(sc.source, sc.target)
    rc, T, st = H.scan_match(sc.T_pred)
    assert rc == 0
    want = _frame_fingerprint(H, T, st)
    H.close()
    monkeypatch.delenv("TLOAM_NO_PERSISTENT_SOLVE")
    monkeypatch.setenv("TLOAM_DEBUG_FAIL_HANDOVER", "1")
    H = hip_module.HipRegistration()
    H.set_frames(sc.source, sc.target)
    t = time.perf_counter()
    rc, T, st = H.scan_match(sc.T_pred)
    dt = time.perf_counter() - t
    assert rc == 0, H.L.tloam_last_error(H.h).decode()
    assert 0.5 < dt < 10.0, dt            # the ~1 s bounded wait was really taken
    _assert_same_frame(want, _frame_fingerprint(H, T, st))
    t = time.perf_counter()
    rc, T, st = H.scan_match(sc.T_pred)   # the context stays on the fallback path: no second time-out
    assert rc == 0 and time.perf_counter() - t < 0.25
    _assert_same_frame(want, _frame_fingerprint(H, T, st))
    H.close()


def test_contexts_following_each_other_never_see_each_others_rows(hip_module):
    """Round 5 regression.  The tagged hand-over rows of the one-launch Solve live in a device buffer; a context created right
    after another one was destroyed is handed the same memory, still holding the dead context's rows -- with valid check words
    for ITS launch numbers.  When both counted their launches from zero, a stepper that looked before the fresh row landed
    folded the dead context's sums (a different scene's), the blocks' images of the minimiser disagreed, and the launch ran into
    its bounded wait: TLOAM_E_HIP in about one of two such successions.  The launch counter now starts from a per-context base
    and the row buffer is cleared when allocated.  Twenty successions of contexts solving ALTERNATING scenes through the stepwise
    API: every one must succeed and reproduce its scene's result bit for bit."""
    scenes = [synth.make_scene(seed=41), synth.make_scene(seed=33)]
    want = []
    for sc in scenes:
        H = hip_module.HipRegistration()
        H.set_frames(sc.source, sc.target)
        rc, T, st = H.scan_match(sc.T_pred)
        assert rc == 0
        want.append((T, st["gn_evaluations"], st["n_corr"]))
        H.close()
    for i in range(20):
        sc, (T_want, ev_want, n_want) = scenes[i % 2], want[i % 2]
        H = hip_module.HipRegistration()
        H.set_frames(sc.source, sc.target)
        assert H.sm_begin(sc.T_pred) == 0
        done = False
        while not done:
            rc, done, st = H.sm_outer()
            assert rc == 0, (i, rc)
        rc, T, st = H.sm_end()
        assert rc == 0 and st["gn_evaluations"] == ev_want and st["n_corr"] == n_want, i
        dt, dr = pose_delta(T, T_want)
        assert dt < 1e-12 and dr < 1e-12, (i, dt, dr)      # (stepwise: k_prepare_small / separate finish -- same sums, see _assert_same_frame)
        H.close()
