"""-m gpu: degenerate geometry through the C ABI -- the inputs a uniform search grid, a plane fit and a 3x3 eigen solve like
least: coincident points, exactly coplanar / collinear neighbourhoods, clouds with fewer points than a neighbourhood needs,
coordinates of a UTM-sized frame, two clusters kilometres apart in one cloud.  Every case is held against the C restatement of
the reference (same gates, registration.cpp:445 / :589 / :605-613 / :481) and, for the search, against brute force: nothing may
crash, hang, or differ."""
import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from tloam_amd import synth

pytestmark = pytest.mark.gpu


def _pair(hip_module, source, target, cfg=None):
    H = hip_module.HipRegistration(cfg) if cfg else hip_module.HipRegistration()
    O = ob.Oracle(cfg) if cfg else ob.Oracle()
    for R in (H, O):
        R.set_frames(source, target)
    return H, O


def _same_solve(H, O, T_pred, what):
    rc, T, st = H.scan_match(T_pred)
    rco, To, sto = O.scan_match(T_pred)
    assert rc == rco, (what, rc, rco)
    if rc != 0:
        return
    assert st["n_corr"] == sto["n_corr"], (what, st["n_corr"], sto["n_corr"])
    assert st["outer_iterations"] == sto["outer_iterations"] and st["gn_evaluations"] == sto["gn_evaluations"], what
    dt, dr = pose_delta(T, To)
    assert dt < 1e-6 and dr < 1e-6, (what, dt, dr)


def _knn_equals_brute_force(H, kind, tgt, q, radius, k):
    hi, hd, hc = H.knn(kind, q, radius, k)
    for j in range(len(q)):
        bi, bd = ob.knn_brute(tgt, q[j], radius, k)
        assert hc[j] == len(bi), (j, hc[j], len(bi))
        # coincident targets tie exactly: the reference's order among equal distances is the kd-tree's, unspecified; the
        # distances must agree bit for bit and the index sets wherever the distances are distinct
        assert np.array_equal(hd[j, :hc[j]], bd), j
        if len(np.unique(bd)) == len(bd):
            assert np.array_equal(hi[j, :hc[j]], bi), j
        else:
            assert set(hi[j, :hc[j]]) <= set(np.flatnonzero(np.isin(((tgt - q[j]) ** 2).sum(1), bd))), j


def test_coincident_target_points(hip_module):
    """2000 copies of ONE target point among ordinary ones: a single grid cell holds them all (the walk is long, not wrong);
    every neighbourhood that reaches the pile is five identical points -- a zero covariance for the edge builder (no direction:
    rejected at :481), a rank-deficient plane fit for the others."""
    sc = synth.make_scene(seed=51)
    rng = np.random.default_rng(1)
    tgt = [sc.target.cloud(k).copy() for k in range(4)]
    for k in range(4):
        pile = tgt[k][rng.integers(0, len(tgt[k]))]
        tgt[k] = np.concatenate([tgt[k], np.repeat(pile[None], 2000, 0)])
    target = synth.Frame(*tgt)
    H, O = _pair(hip_module, sc.source, target)
    for k in (0, 2, 3):
        q = np.concatenate([tgt[k][-1:] + 0.01, tgt[k][-1:], sc.source.cloud(k)[:40]])
        _knn_equals_brute_force(H, k, tgt[k], q, 1.0 if k == 2 else 0.5, 1 if k == 3 else 5)
    _same_solve(H, O, sc.T_pred, "coincident")


def test_exactly_coplanar_and_collinear_targets(hip_module):
    """Targets on an exact lattice of the plane z = 0 (planar / ground kinds: the fit's residuals are exact zeros, equal
    distances everywhere) and on the exact line y = z = 0 (edge kind: two zero eigenvalues)."""
    g = np.arange(-10, 10.01, 0.25)
    X, Y = np.meshgrid(g, g)
    plane = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], 1)
    line = np.stack([np.arange(-20, 20, 0.1), np.zeros(400), np.zeros(400)], 1)
    rng = np.random.default_rng(2)
    src_plane = np.stack([rng.uniform(-8, 8, 300), rng.uniform(-8, 8, 300), rng.normal(0, 0.02, 300)], 1)
    src_line = np.stack([rng.uniform(-15, 15, 100), rng.normal(0, 0.02, 100), rng.normal(0, 0.02, 100)], 1)
    sc = synth.make_scene(seed=52)
    Fr = type(sc.source)
    # Frame(planar, ground, edge, sphere) in the binding's kind order
    source = Fr(src_plane, src_plane[::2] + 0.001, src_line, src_plane[:50])
    target = Fr(plane, plane + np.array([0.125, 0.125, 0.0]), line, plane[::7])
    H, O = _pair(hip_module, source, target)
    for k, tg in ((0, plane), (2, line)):
        q = (src_plane if k == 0 else src_line)[:60]
        hi, hd, hc = H.knn(k, q, 1.0, 5)
        oi, od, oc = O.knn(k, q, 1.0, 5)
        assert np.array_equal(hc, oc) and np.array_equal(hd, od)
    T_pred = np.eye(4)
    T_pred[:3, 3] = (0.03, -0.02, 0.05)
    _same_solve(H, O, T_pred, "coplanar / collinear")


@pytest.mark.parametrize("n_tgt", [1, 3, 4, 5])
def test_fewer_targets_than_a_neighbourhood(hip_module, n_tgt):
    """Target clouds of 1 / 3 / 4 / 5 points: the search over a grid that has degenerated to a handful of cells returns the
    exact neighbours (fewer than k of them), and scanMatching refuses the frame with the status that stands for the reference's
    assert on clouds of fewer than ten points (registration.cpp:928-929), as the restatement does."""
    sc = synth.make_scene(seed=53)
    Fr = type(sc.source)
    tgt = [sc.target.cloud(k)[:n_tgt].copy() for k in range(4)]
    near = [np.repeat(tgt[k].mean(0)[None], 30, 0) + np.random.default_rng(3 + k).normal(0, 0.05, (30, 3)) for k in range(4)]
    H, O = _pair(hip_module, Fr(*near), Fr(*tgt))
    for k in range(4):
        _knn_equals_brute_force(H, k, tgt[k], near[k][:10], 0.5, 1 if k == 3 else 5)
    _same_solve(H, O, np.eye(4), f"{n_tgt} targets")


def test_utm_sized_coordinates(hip_module):
    """The whole scene shifted by (4.5e5, 5.4e6, 300) m: cell indices, the packed keys of the top-k lists (low mantissa bits
    replaced by the candidate's position) and the un-fused squared distances at coordinates where one ulp is 1e-9 m."""
    sc = synth.make_scene(seed=54)
    off = np.array([4.5e5, 5.4e6, 300.0])
    Fr = type(sc.source)
    source = Fr(*[sc.source.cloud(k) + off for k in range(4)])
    target = Fr(*[sc.target.cloud(k) + off for k in range(4)])
    T_pred = sc.T_pred.copy()
    T_pred[:3, 3] += off - T_pred[:3, :3] @ off      # the same prediction, about the shifted origin
    H, O = _pair(hip_module, source, target)
    for k in range(4):
        q = (T_pred[:3, :3] @ source.cloud(k)[:80].T).T + T_pred[:3, 3]
        hi, hd, hc = H.knn(k, q, 1.0 if k == 2 else 0.5, 1 if k == 3 else 5)
        oi, od, oc = O.knn(k, q, 1.0 if k == 2 else 0.5, 1 if k == 3 else 5)
        assert np.array_equal(hc, oc) and np.array_equal(hi, oi) and np.array_equal(hd, od), k
    _same_solve(H, O, T_pred, "UTM offset")


def test_two_clusters_kilometres_apart(hip_module):
    """One target cloud = the scene + a copy of it 7 km away: the bounding box is ~10^4 x its cell size per axis, the dense cell
    table cannot hold cells of one search radius (<= 4 M cells), so the cells grow and every cell holds hundreds of points --
    the search must still return the exact neighbours."""
    sc = synth.make_scene(seed=55)
    far = np.array([7000.0, -3000.0, 40.0])
    Fr = type(sc.source)
    tgt = [np.concatenate([sc.target.cloud(k), sc.target.cloud(k)[::3] + far]) for k in range(4)]
    H, O = _pair(hip_module, sc.source, Fr(*tgt))
    for k in range(4):
        q = np.concatenate([sc.source.cloud(k)[:40], sc.source.cloud(k)[:10] + far])
        hi, hd, hc = H.knn(k, q, 1.0 if k == 2 else 0.5, 1 if k == 3 else 5)
        oi, od, oc = O.knn(k, q, 1.0 if k == 2 else 0.5, 1 if k == 3 else 5)
        assert np.array_equal(hc, oc) and np.array_equal(hi, oi) and np.array_equal(hd, od), k
    _same_solve(H, O, sc.T_pred, "two clusters")


def test_no_correspondence_at_all(hip_module):
    """Every source point 2 km from every target: four empty residual sets, `ceres::Solve` on an empty problem in every outer
    iteration, the plateau break of :1108 on 0 - inf ... the restatement and the device agree on whatever that gives."""
    sc = synth.make_scene(seed=56)
    Fr = type(sc.source)
    source = Fr(*[sc.source.cloud(k) + np.array([2000.0, 0.0, 0.0]) for k in range(4)])
    H, O = _pair(hip_module, source, sc.target)
    rc, T, st = H.scan_match(sc.T_pred)
    rco, To, sto = O.scan_match(sc.T_pred)
    assert rc == rco and st["n_corr"] == sto["n_corr"] == [0, 0, 0, 0]
    assert st["outer_iterations"] == sto["outer_iterations"]
    dt, dr = pose_delta(T, To)
    assert dt < 1e-9 and dr < 1e-9
    dt, dr = pose_delta(T, sc.T_pred)
    assert dt < 1e-9 and dr < 1e-9          # nothing to minimise: the prediction comes back


def test_all_clouds_empty_and_non_finite_prediction(hip_module):
    """No point in any of the eight clouds; then a prediction holding a NaN / an Inf: a status code, never a hang or a crash
    (the reference would run into SOPHUS_ENSURE, se3.hpp:497-504)."""
    sc = synth.make_scene(seed=57)
    Fr = type(sc.source)
    empty = Fr(*[np.zeros((0, 3)) for _ in range(4)])
    H, O = _pair(hip_module, empty, empty)
    rc, T, st = H.scan_match(np.eye(4))
    rco, To, sto = O.scan_match(np.eye(4))
    assert rc == rco and st["n_corr"] == sto["n_corr"] == [0, 0, 0, 0]
    if rc == 0:
        assert np.array_equal(T, np.eye(4))
    H2 = hip_module.HipRegistration()
    H2.set_frames(sc.source, sc.target)
    for bad in (np.nan, np.inf):
        P = sc.T_pred.copy()
        P[0, 3] = bad
        rc, T, st = H2.scan_match(P)
        assert rc != 0, bad
        P = sc.T_pred.copy()
        P[1, 1] = bad
        rc, T, st = H2.scan_match(P)
        assert rc != 0, bad
    rc, T, st = H2.scan_match(sc.T_pred)       # the context is still good
    O2 = ob.Oracle()
    O2.set_frames(sc.source, sc.target)
    rco, To, sto = O2.scan_match(sc.T_pred)
    dt, dr = pose_delta(T, To)
    assert rc == 0 and dt < 1e-6 and dr < 1e-6


def test_refused_frame_leaves_the_registered_one_alone(hip_module):
    """`tloam_set_source_frame` / `tloam_set_target_frame` with a NULL cloud of non-zero size are refused AS A WHOLE: the frame
    registered before stays registered -- its sizes, its pointers -- so the fitness score and the next solve are those of the
    frame before (until round 5 the kinds in front of the offending one had already taken the new sizes, and getFitnessScore
    then read past the old clouds: a GPU memory fault found by tests/tools/fuzz_call_order.py)."""
    import ctypes as C
    sc = synth.make_scene(seed=58)
    H = hip_module.HipRegistration()
    O = ob.Oracle()
    for R in (H, O):
        R.set_frames(sc.source, sc.target)
    rc, T0, st0 = H.scan_match(sc.T_pred)
    assert rc == 0
    f0 = H.get_fitness_score()
    big = [np.ascontiguousarray(np.random.default_rng(k).normal(0, 20, (50_000, 3))) for k in range(4)]
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    for fn in (H.L.tloam_set_source_frame, H.L.tloam_set_target_frame):
        for bad in range(4):
            ptrs = (C.POINTER(C.c_double) * 4)(*[dp(b) for b in big])
            ptrs[bad] = C.POINTER(C.c_double)()
            assert fn(H.h, ptrs, (C.c_size_t * 4)(*[len(b) for b in big])) == -1      # TLOAM_E_INVALID
            assert H.get_fitness_score() == f0
    rc, T1, st1 = H.scan_match(sc.T_pred)
    assert rc == 0 and T1.tobytes() == T0.tobytes() and st1["n_corr"] == st0["n_corr"]
    rco, To, sto = O.scan_match(sc.T_pred)
    dt, dr = pose_delta(T1, To)
    assert dt < 1e-6 and dr < 1e-6


def test_large_frame_after_prebuilt_sets_on_a_used_context(hip_module):
    """The call sequence tests/tools/fuzz_call_order.py shrank a GPU memory fault to: a small frame, a pre-built correspondence
    set (its buffers fill and free blocks of device memory), then the context's FIRST frame large enough for the sorted query
    order.  The query-tile histogram was re-allocated after the frame's start had zeroed it; on a block that was not fresh the
    counting sort scattered the queries by garbage.  The solve must equal the restatement's -- and a second context that does
    the same with its allocations in another order."""
    small = synth.make_scene(seed=60)
    big = synth.make_scene(seed=80, n_src=(6000, 8000, 5000, 1000), n_tgt=(8000, 9000, 6000, 1500))
    O = ob.Oracle()
    O.set_frames(big.source, big.target)
    rco, To, sto = O.scan_match(big.T_pred)
    assert rco == 0
    rng = np.random.default_rng(5)
    for order in range(3):
        H = hip_module.HipRegistration()
        H.set_frames(small.source, small.target)
        if order != 1:
            assert H.scan_match(small.T_pred)[0] == 0
        for rt, m in ((0, 3000), (1, 700), (2, 300))[: 1 + order]:
            p = rng.normal(0, 10, (m, 3)); a = rng.normal(0, 1, (m, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
            H.set_correspondences(rt, p, a, rng.normal(0, 10, (m, 3)), rng.normal(0, 1, m), rng.uniform(0.1, 1, m) * 1e300)
        H.set_frames(big.source, big.target)
        rc, T, st = H.scan_match(big.T_pred)
        assert rc == 0 and st["n_corr"] == sto["n_corr"], (order, st["n_corr"], sto["n_corr"])
        dt, dr = pose_delta(T, To)
        assert dt < 1e-6 and dr < 1e-6, (order, dt, dr)
        H.close()


def test_sizes_beyond_the_slot_space_are_refused(hip_module):
    """A count of more than 2^28 points (slots, cells and ranks are 32-bit integers on the device) is TLOAM_E_INVALID at every entry
    point that takes one -- before the buffer it claims to describe is touched -- and the context goes on working."""
    import ctypes as C
    sc = synth.make_scene(seed=59)
    H = hip_module.HipRegistration()
    H.set_frames(sc.source, sc.target)
    L, h = H.L, H.h
    a = np.zeros((16, 3))
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))    # noqa: E731
    huge = (1 << 28) + 1
    assert L.tloam_set_source(h, 0, dp(a), huge) == -1
    assert L.tloam_set_target(h, 1, dp(a), huge) == -1
    ptrs = (C.POINTER(C.c_double) * 4)(*[dp(a)] * 4)
    assert L.tloam_set_source_frame(h, ptrs, (C.c_size_t * 4)(16, 16, huge, 16)) == -1
    assert L.tloam_set_target_frame(h, ptrs, (C.c_size_t * 4)(huge, 16, 16, 16)) == -1
    idx = np.zeros(64, np.int32); d2 = np.zeros(64); cnt = np.zeros(16, np.int32)
    assert L.tloam_knn(h, 0, dp(a), huge, 1.0, 5, ip(idx), dp(d2), ip(cnt)) == -1
    w = np.ones(16)
    assert L.tloam_set_correspondences(h, 0, huge, dp(a), dp(a), dp(a), dp(w), dp(w)) == -1
    st = hip_module.Stats(); res = np.zeros(16)
    assert L.tloam_scan_match(h, dp(np.ascontiguousarray(sc.T_pred.T.ravel())), None, dp(res), dp(a), huge, C.byref(st)) == -1
    fc = hip_module.default_feature_config()
    assert L.tloam_pca_info(h, C.byref(fc), dp(a), huge, None, None, None, None, None, None) == -1
    assert L.tloam_submap_init(h, None, dp(a), 16, dp(a), 16, dp(a), huge, dp(a), 16) == -1
    rc, T, s2 = H.scan_match(sc.T_pred)
    O = ob.Oracle(); O.set_frames(sc.source, sc.target)
    rco, To, so = O.scan_match(sc.T_pred)
    dt, dr = pose_delta(T, To)
    assert rc == 0 and dt < 1e-6 and dr < 1e-6
