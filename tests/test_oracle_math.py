"""CPU: the oracle's building blocks against analytic facts and the independent numpy restatement."""
import math

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from scipy.spatial import cKDTree

from oracle import binding as ob
from oracle import oracle_np as onp

RNG = np.random.default_rng(1234)


def rand_se3(scale_t=2.0, scale_r=1.0):
    return np.concatenate([RNG.normal(0, scale_t, 3), RNG.normal(0, scale_r, 3)])


@pytest.mark.parametrize("i", range(20))
def test_exp_log_roundtrip(i):
    x = rand_se3()
    if np.linalg.norm(x[3:]) > 3.0:       # keep away from the pi branch cut
        x[3:] *= 3.0 / np.linalg.norm(x[3:])
    T = ob.se3_exp(x)
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-14)
    assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-14
    np.testing.assert_allclose(ob.se3_log(T), x, atol=1e-12)
    np.testing.assert_allclose(T, onp.se3_exp(x), atol=1e-13)      # independent closed form
    np.testing.assert_allclose(onp.se3_log(T), x, atol=1e-11)


@pytest.mark.parametrize("omega_norm", [0.0, 1e-12, 1e-11, 0.9e-10, 1.1e-10, 1e-8, 1e-5])
def test_exp_small_angle_branches(omega_norm):
    """theta^2 < 1e-20 (Taylor quaternion) and theta < 1e-10 (V = R) branches, sophus so3.hpp:594-607,
    se3.hpp:772-774."""
    u = np.array([0.3, -0.5, 0.81])
    u /= np.linalg.norm(u)
    x = np.concatenate([[0.4, -0.2, 0.1], u * omega_norm])
    T = ob.se3_exp(x)
    np.testing.assert_allclose(T, onp.se3_exp(x), atol=1e-15)
    np.testing.assert_allclose(ob.se3_log(T), x, atol=1e-15)


def test_plus_identity_and_left_perturbation():
    x = rand_se3(1.0, 0.3)
    np.testing.assert_allclose(ob.plus(x, np.zeros(6)), x, atol=1e-14)
    d = rand_se3(0.01, 0.01)
    np.testing.assert_allclose(ob.se3_exp(ob.plus(x, d)), ob.se3_exp(d) @ ob.se3_exp(x), atol=1e-13)
    np.testing.assert_allclose(ob.plus(x, d), onp.plus(x, d), atol=1e-12)


def test_from_matrix_rejects_non_rigid():
    T = ob.se3_exp(rand_se3())
    bad = T.copy(); bad[:3, :3] *= 1.001
    with pytest.raises(ValueError):
        ob.se3_log(bad)
    bad = T.copy(); bad[3, 0] = 1e-3
    with pytest.raises(ValueError):
        ob.se3_log(bad)
    refl = T.copy(); refl[:3, 0] *= -1.0
    with pytest.raises(ValueError):
        ob.se3_log(refl)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_quaternion_from_matrix_negative_trace_branches(axis):
    """Rotations by ~pi-0.2 about each axis: trace <= 0 -> the three i/j/k branches of Eigen's conversion."""
    w = np.zeros(3); w[axis] = math.pi - 0.2
    x = np.concatenate([[0.1, 0.2, 0.3], w])
    T = onp.se3_exp(x)
    assert np.trace(T[:3, :3]) < 0
    np.testing.assert_allclose(ob.se3_exp(ob.se3_log(T)), T, atol=1e-13)


def test_fit_plane_exact_and_degenerate():
    n = np.array([0.3, -0.4, 0.866]); n /= np.linalg.norm(n)
    basis = np.linalg.svd(n[None])[2][1:]
    pts = RNG.normal(size=(5, 2)) @ basis + 2.5 * n
    pl = ob.fit_plane(pts)
    assert abs(abs(pl[:3] @ n) - 1.0) < 1e-12
    assert np.allclose(pts @ pl[:3] + pl[3], 0.0, atol=1e-12)
    np.testing.assert_allclose(pl, onp.fit_best_plane(pts), atol=1e-13)
    same = np.tile(pts[0], (5, 1))
    assert np.array_equal(ob.fit_plane(same), np.zeros(4))        # registration.cpp:361-364


@pytest.mark.parametrize("i", range(10))
def test_eig3_vs_lapack(i):
    A = RNG.normal(size=(3, 3)); cov = A @ A.T * 10.0 ** RNG.uniform(-6, 1)
    ev, V = ob.eig3(cov)
    ev_ref, V_ref = np.linalg.eigh(cov)
    np.testing.assert_allclose(ev, ev_ref, rtol=1e-12, atol=1e-14 * ev_ref[-1])
    assert np.allclose(V.T @ V, np.eye(3), atol=1e-13)
    assert abs(abs(V[:, 2] @ V_ref[:, 2]) - 1.0) < 1e-9
    assert np.allclose(cov @ V, V * ev, atol=1e-12 * ev_ref[-1])


@pytest.mark.parametrize("scale", [1e-160, 1e-100, 1.0, 1e100, 1e160])
def test_eig3_degenerate_scales(scale):
    """Equal diagonal entries with a tiny (or huge) off-diagonal one: d*d + b*b under-/overflows in the rotation
     -- the eigen pairs must still come out finite and right (numpy as the judge)."""
    for cov in (np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.5]]) * scale,
                np.array([[2.0, 1.0, 0.5], [1.0, 2.0, 0.25], [0.5, 0.25, 2.0]]) * scale):
        ev, V = ob.eig3(cov)
        ev_ref = np.linalg.eigvalsh(cov)
        assert np.all(np.isfinite(ev)) and np.all(np.isfinite(V))
        np.testing.assert_allclose(ev, ev_ref, rtol=1e-12, atol=1e-14 * abs(ev_ref).max())
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose((cov / scale) @ V, V * (ev / scale), atol=1e-12)


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(12, 400), k=st.sampled_from([1, 5]),
       radius=st.sampled_from([0.02, 0.5, 1.0, 3.0]))
def test_hybrid_search_exact_vs_ckdtree(seed, n, k, radius):
    """KDTreeFlann::SearchHybrid semantics: exact k-NN, ascending, cut at d^2 < r^2."""
    rng = np.random.default_rng(seed)
    tgt = rng.uniform(-3, 3, (n, 3))
    q = rng.uniform(-3.5, 3.5, (40, 3))
    O = ob.Oracle()
    O.set_target(0, tgt)
    idx, d2, cnt = O.knn(0, q, radius, k)
    tree = cKDTree(tgt)
    dd, ii = tree.query(q, k=k)
    dd = dd.reshape(len(q), k); ii = ii.reshape(len(q), k)
    for j in range(len(q)):
        keep = np.isfinite(dd[j]) & (dd[j] ** 2 < radius * radius)
        ref_d2 = ((tgt[ii[j][keep]] - q[j]) ** 2).sum(axis=1)
        if len(ref_d2) and abs(ref_d2[-1] - radius * radius) < 1e-12:
            continue                                    # knife edge on the radius cut
        assert cnt[j] == keep.sum()
        assert np.array_equal(idx[j, :cnt[j]], ii[j][keep])
        np.testing.assert_allclose(d2[j, :cnt[j]], ref_d2, rtol=1e-12, atol=1e-15)
        bi, bd = ob.knn_brute(tgt, q[j], radius, k)      # grid == brute force, bit for bit
        assert np.array_equal(bi, idx[j, :cnt[j]]) and np.array_equal(bd, d2[j, :cnt[j]])


def test_knn_tie_break_lower_index():
    tgt = np.array([[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [0, -1.0, 0], [0, 0, 1.0], [0, 0, -1.0], [5, 5, 5],
                    [6, 6, 6], [7, 7, 7], [8, 8, 8]])
    O = ob.Oracle(); O.set_target(0, tgt)
    idx, d2, cnt = O.knn(0, np.zeros((1, 3)), 2.0, 5)
    assert cnt[0] == 5 and list(idx[0]) == [0, 1, 2, 3, 4]


def _single_block_set(res_type, p, a, b, d, w):
    N = onp.NpRegistration()
    N.set_correspondences(res_type, p[None], a[None], None if b is None else b[None], None if d is None else np.array([d]),
                          np.array([w]))
    return N


@pytest.mark.parametrize("res_type", [0, 1, 2])
def test_jacobians_are_left_perturbation_derivatives(res_type):
    """registration.cpp:36-42, :73-83, :105-112 against central differences of the residual under
    x <- log(exp(delta) exp(x)).  For the plane functor the Jacobian is w times the derivative of the
    UNWEIGHTED residual (reference quirk)."""
    x = rand_se3(1.0, 0.4)
    p = RNG.normal(size=3) * 5
    a = RNG.normal(size=3); b = a + 0.2 * np.array([0.1, 0.2, 0.97]); n = a / np.linalg.norm(a)
    w = 0.7
    N = _single_block_set(res_type, p, n if res_type == 0 else a, b if res_type == 1 else None,
                          0.3 if res_type == 0 else None, w)
    cs = [s for s in N.sets if s is not None][0]
    r0, J, _ = onp.residual_blocks(cs, onp.se3_exp(x))
    h = 1e-6
    Jn = np.zeros_like(J[0])
    for k in range(6):
        d = np.zeros(6); d[k] = h
        rp, _, _ = onp.residual_blocks(cs, onp.se3_exp(onp.plus(x, d)))
        rm, _, _ = onp.residual_blocks(cs, onp.se3_exp(onp.plus(x, -d)))
        Jn[:, k] = (rp[0] - rm[0]) / (2 * h)
    scale = w if res_type == 0 else 1.0
    np.testing.assert_allclose(J[0], Jn * scale, atol=2e-6 * max(1.0, np.abs(Jn).max()))


@pytest.mark.parametrize("res_type", [0, 1, 2])
def test_oracle_accumulate_matches_numpy(res_type):
    from tloam_amd import synth
    sets, x_true, x_eval = synth.make_prebuilt(seed=5, n_plane=300, n_line=200, n_point=100, weights="timing")
    O = ob.Oracle(); N = onp.NpRegistration()
    p, a, b, d, w = sets[res_type]
    O.set_correspondences(res_type, p, a, b, d, w); N.set_correspondences(res_type, p, a, b, d, w)
    Ho, go, co = O.accumulate(x_eval)
    Hn, gn, cn = N.accumulate(x_eval)
    np.testing.assert_allclose(Ho, Hn, rtol=1e-11, atol=1e-11 * np.abs(Hn).max())
    np.testing.assert_allclose(go, gn, rtol=1e-11, atol=1e-11 * np.abs(gn).max())
    assert abs(co - cn) < 1e-12 * max(1.0, abs(cn))
    kind = {0: 0, 1: 2, 2: 3}[res_type]
    np.testing.assert_allclose(O.get_costs(res_type), N.sets[kind].cost, rtol=1e-11, atol=1e-20)
