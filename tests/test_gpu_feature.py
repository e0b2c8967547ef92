"""-m gpu: PCA feature extraction on the device (tloam_pca_info / tloam_extract_planar_sphere, SURVEY 8(f)
next-2) against the CPU restatement -- the same grid search, (distance, index) order and Jacobi eigen solve,
so every per-point value and every index list is compared bit for bit -- through the C ABI."""
import numpy as np
import pytest

from oracle import binding as ob
from tloam_amd import synth_submap as ss

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("seed,n", [(0, 1500), (1, 4000), (2, 12000)])
def test_pca_info_bit_exact(hip_module, seed, n):
    p = ss.feature_cloud(seed, n=n)
    H = hip_module.HipRegistration()
    g = H.pca_info(p)
    o = ob.pca_info(p)
    assert _same(g["num_sum"], o["num_sum"])
    assert _same(g["neigh"], o["neigh"])
    for k in ("flatness", "cvr", "sphericity", "normal"):
        assert _same(g[k], o[k]), (k, np.nanmax(np.abs(g[k] - o[k])))
    assert (o["num_sum"] > 0).sum() > 0.8 * n
    H.close()


@pytest.mark.parametrize("seed,cfg", [(0, {}), (1, dict(planar_num=40, sphere_num=3)),
                                      (2, dict(K=12, min_neigh=6, radius=0.15, planar_scan_thres=0.9, cvr_scan=0.05)),
                                      (3, dict(planar_vertic_thres=2.0, cvr_submap=0.01))])
def test_index_lists_bit_exact(hip_module, seed, cfg):
    p = ss.feature_cloud(seed, n=5000)
    H = hip_module.HipRegistration()
    g = H.extract_planar_sphere(p, hip_module.default_feature_config(**cfg))
    o = ob.extract_planar_sphere(p, ob.make_feature_config(**cfg))
    for i, name in enumerate(("planar_scan", "planar_submap", "sphere_scan", "sphere_submap")):
        assert _same(g[i], o[i]), (name, len(g[i]), len(o[i]))
    assert len(o[1]) > 100
    H.close()


def test_ranking_with_exact_ties_and_many_candidates(hip_module):
    """The ranking (std::sort descending by flatness, ties by ascending index) works bucket by bucket on the device.  A cloud and
    its point reflection have bit-identical PCA values -- every candidate ties with its mirror image --, and 16 k points give
    lists of several blocks: the four index lists as the oracle's, a cloud whose candidates all share ONE flatness value too."""
    p = ss.feature_cloud(4, n=8000)
    cloud = np.ascontiguousarray(np.concatenate([p, -p]))
    H = hip_module.HipRegistration()
    for cfg in ({}, dict(planar_num=50, sphere_num=10, planar_submap_thres=0.3, cvr_submap=0.05)):
        g = H.extract_planar_sphere(cloud, hip_module.default_feature_config(**cfg))
        o = ob.extract_planar_sphere(cloud, ob.make_feature_config(**cfg))
        assert len(o[1]) > 2000 and len(o[1]) % 2 == 0
        for i, name in enumerate(("planar_scan", "planar_submap", "sphere_scan", "sphere_submap")):
            assert _same(g[i], o[i]), (name, len(g[i]), len(o[i]))
    # a lattice: every interior point has the same neighbourhood up to translation by exactly representable steps
    a = np.arange(24) * 0.0625
    lat = np.ascontiguousarray(np.array([(x, y, 0.0) for x in a for y in a]))
    g, o = H.extract_planar_sphere(lat, hip_module.default_feature_config(planar_vertic_thres=2.0, planar_submap_thres=-1.0)), \
        ob.extract_planar_sphere(lat, ob.make_feature_config(planar_vertic_thres=2.0, planar_submap_thres=-1.0))
    assert len(o[1]) > 300
    for a_, b_ in zip(g, o):
        assert _same(a_, b_)
    # 3 000 collinear points: every flatness is exactly 0 -- ONE crowded bucket, the blocks count through LDS over all of it
    x = np.arange(3000) * 2.0 ** -7
    line = np.ascontiguousarray(np.column_stack([x, np.zeros_like(x), np.zeros_like(x)]))
    cfg = dict(planar_vertic_thres=2.0, planar_submap_thres=-1.0, planar_num=10)
    g, o = H.extract_planar_sphere(line, hip_module.default_feature_config(**cfg)), ob.extract_planar_sphere(line, ob.make_feature_config(**cfg))
    assert len(o[1]) == 3000 and list(o[1][:5]) == [0, 1, 2, 3, 4]
    for a_, b_ in zip(g, o):
        assert _same(a_, b_)
    H.close()


def test_hip_against_golden(hip_module):
    from test_feature_oracle import check_against_golden
    H = hip_module.HipRegistration()
    check_against_golden(lambda p, cfg: H.pca_info(p, hip_module.default_feature_config(**cfg)),
                         lambda p, cfg: H.extract_planar_sphere(p, hip_module.default_feature_config(**cfg)))
    H.close()


def test_degenerate_inputs(hip_module):
    H = hip_module.HipRegistration()
    line = np.column_stack([np.linspace(0, 0.1, 11), np.zeros(11), np.zeros(11)])
    dup = np.repeat(np.array([[1.0, 2.0, 3.0]]), 30, axis=0)
    for cloud in (line, line[:10], dup, np.array([[0.0, 0.0, 0.0]])):
        g, o = H.pca_info(cloud), ob.pca_info(cloud)
        for k in g:
            assert _same(g[k], o[k]), k
        gl, ol = H.extract_planar_sphere(cloud), ob.extract_planar_sphere(cloud)
        assert all(_same(a, b) for a, b in zip(gl, ol))
    assert all(len(x) == 0 for x in H.extract_planar_sphere(np.zeros((0, 3))))
    with pytest.raises(hip_module.TloamHipError):
        H.pca_info(line, hip_module.default_feature_config(K=21))          # K <= 20 on the device
    H.close()


def test_feature_lists_feed_the_submap(hip_module):
    """The three widened rows chained: extract on the device, select the clouds, update the device submap."""
    p = ss.feature_cloud(5, n=6000)
    H = hip_module.HipRegistration()
    ps, pm, s_scan, s_sub = H.extract_planar_sphere(p)
    cl = ss.frame_clouds(5, 0, n=(10, 10, 300, 400))
    H.submap_init(p[pm], p[s_sub], cl[2], cl[3])
    H.submap_update(ss.frame_pose(1), p[pm], p[s_sub], cl[2], cl[3])
    assert len(H.get_target(0)) == len(pm) and len(H.get_target(3)) == len(pm)    # sphere submap <- planar buffer
    H.close()


def test_near_ties_take_the_exact_path(hip_module):
    """The neighbour walk orders candidates by a packed key -- the squared distance with its low mantissa bits
    replaced by the candidate's position -- and must redo a query exactly when two kept distances agree in every
    bit the key keeps.  Planted near-ties (relative difference 3e-15 .. 3e-13, below the ~1e-12 resolution of the key),
    exact ties (mirror points, duplicates) and ordinary points: every neighbour list in the oracle's exact
    (distance, index) order, every PCA value bit for bit."""
    rng = np.random.default_rng(7)
    base = ss.feature_cloud(5, n=3000)
    extra = []
    for i in rng.choice(len(base), 400, replace=False):
        q = base[i]
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        v = np.cross(u, rng.normal(size=3)); v /= np.linalg.norm(v)
        r = rng.uniform(0.01, 0.08)
        eps = 10.0 ** rng.uniform(-14.5, -12.5)
        extra += [q + r * u, q + r * (1.0 + eps) * v]          # near-tie as seen from q
        if i % 3 == 0:
            extra += [q - r * u]                                  # exact tie with q + r*u (to rounding of q +- r*u)
        if i % 5 == 0:
            extra += [q + r * u]                                  # exact duplicate
    p = np.ascontiguousarray(np.vstack([base, np.array(extra)]))
    p = p[rng.permutation(len(p))]                                # positions in the cell-sorted records unrelated to distance
    H = hip_module.HipRegistration()
    for cfg in ({}, dict(K=12, min_neigh=4)):
        g = H.pca_info(p, hip_module.default_feature_config(**cfg))
        o = ob.pca_info(p, ob.make_feature_config(**cfg))
        assert _same(g["neigh"], o["neigh"])
        assert _same(g["num_sum"], o["num_sum"])
        for k in ("flatness", "cvr", "sphericity", "normal"):
            assert _same(g[k], o[k]), k
    H.close()


@pytest.mark.parametrize("radius", [0.0, 1e-9, 50.0, 1e6, float("inf")])
def test_search_radius_at_the_ends_of_its_range(hip_module, radius):
    """`assert(r_ >= 0.0 ...)` (feature_extract.cpp:55) admits a radius of ZERO -- SearchHybrid then finds nobody, every point comes
    out with no neighbour -- and an infinite one (plain k-NN).  Until round 5 a zero radius left the search grid empty and the
    walk faulted on it (tests/tools/fuzz_call_order.py); NaN and negative radii are TLOAM_E_INVALID."""
    p = ss.feature_cloud(3, n=2500)
    H = hip_module.HipRegistration()
    g = H.pca_info(p, hip_module.default_feature_config(radius=radius))
    o = ob.pca_info(p, ob.make_feature_config(radius=radius))
    for k in ("num_sum", "neigh", "flatness", "cvr", "sphericity", "normal"):
        assert _same(g[k], o[k]), k
    assert (g["num_sum"].max() == 0) == (radius < 1e-6)
    lists_g = H.extract_planar_sphere(p, hip_module.default_feature_config(radius=radius))
    lists_o = ob.extract_planar_sphere(p, ob.make_feature_config(radius=radius))
    for a, b in zip(lists_g, lists_o):
        assert np.array_equal(a, b)
    for bad in (-1.0, float("nan")):
        with pytest.raises(hip_module.TloamHipError, match="INVALID"):
            H.pca_info(p, hip_module.default_feature_config(radius=bad))
    H.close()


@pytest.mark.parametrize("n", [1, 2, 100, 3000])
@pytest.mark.parametrize("val", [np.nan, np.inf])
def test_cloud_without_a_finite_point(hip_module, n, val):
    """Every coordinate NaN / infinite: there are no bounds to size a search grid from (a GPU memory fault until round 5:
    tests/tools/fuzz_call_order.py, a one-point cloud whose point is NaN) and nobody has a neighbour.  Run AFTER an ordinary
    cloud on the same context, so that whatever is not written would show the previous call's values."""
    H = hip_module.HipRegistration()
    H.pca_info(ss.feature_cloud(5, n=4000))
    a = np.full((n, 3), val)
    g = H.pca_info(a)
    o = ob.pca_info(a)
    for k in ("num_sum", "neigh", "flatness", "cvr", "sphericity", "normal"):
        assert _same(g[k], o[k]), k
    assert g["num_sum"].max() == 0 and (g["neigh"] == -1).all()
    for x, y in zip(H.extract_planar_sphere(a), ob.extract_planar_sphere(a)):
        assert np.array_equal(x, y)
    H.close()
