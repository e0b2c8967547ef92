"""-m gpu: device-resident submap maintenance (tloam_submap_init / tloam_submap_update)
against the CPU restatement -- bit-exact, cloud for cloud and frame for frame -- through the C ABI."""
import numpy as np
import pytest

from conftest import pose_delta
from oracle import binding as ob
from test_submap_oracle import replay_golden
from tloam_amd import synth, synth_submap as ss

pytestmark = pytest.mark.gpu


class _HipSubmap:
    """The HIP context driven through the submap entry points, with the oracle's call surface."""

    def __init__(self, reg, cfg):
        self.reg = reg
        self.H = reg.HipRegistration()
        self.cfg = reg.default_submap_config(**cfg)

    def init(self, *cl):
        return self.H.submap_init(*cl, cfg=self.cfg)

    def update(self, T, *cl):
        return self.H.submap_update(T, *cl)

    def get(self, k):
        return self.H.get_target(k)


def test_hip_replays_golden_bit_exact(hip_module):
    replay_golden(lambda cfg: _HipSubmap(hip_module, cfg), lambda S, k: S.get(k))


@pytest.mark.parametrize("seed,crop", [(0, 100.0), (1, 30.0), (2, 12.0)])
def test_hip_vs_oracle_sequences(hip_module, seed, crop):
    cfg = dict(edge_crop_box_length=crop, ground_crop_box_length=crop * 0.8, planar_frame_size=2 + seed)
    A = _HipSubmap(hip_module, cfg)
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    for f in range(7):
        cl = ss.frame_clouds(seed, f)
        if f == 0:
            A.init(*cl); B.init(*cl)
        else:
            T = ss.frame_pose(f, step=3.0, yaw_rate=0.04)
            A.update(T, *cl); B.update(T, *cl)
        for k in range(4):
            a, b = A.get(k), B.get(k)
            assert a.shape == b.shape, (f, k, a.shape, b.shape)
            assert np.array_equal(a, b), (f, k, np.abs(a - b).max())


def test_crowded_voxels_and_duplicates(hip_module):
    """Hundreds of members per voxel (raw ground near the sensor), exact duplicates, one-voxel clouds: the
    accumulation must still run in index order."""
    rng = np.random.default_rng(5)
    ground = np.concatenate([rng.uniform(-0.4, 0.4, (3000, 3)), np.repeat(rng.uniform(-5, 5, (20, 3)), 40, axis=0),
                             rng.uniform(-30, 30, (2000, 3)) * [1, 1, 0.02]])
    ground = ground.astype(np.float32).astype(np.float64)
    edge = np.repeat(np.array([[1.0, 2.0, 3.0]]), 257, axis=0)
    planar = rng.uniform(-10, 10, (64, 3)); sphere = rng.uniform(-10, 10, (17, 3))
    A = _HipSubmap(hip_module, {}); B = ob.OracleSubmap()
    A.init(planar, sphere, edge, ground); B.init(planar, sphere, edge, ground)
    for k in range(4):
        assert np.array_equal(A.get(k), B.get(k)), k
    T = ss.frame_pose(2)
    A.update(T, planar, sphere, edge, ground); B.update(T, planar, sphere, edge, ground)
    for k in range(4):
        assert np.array_equal(A.get(k), B.get(k)), k


def test_odd_cloud_sizes_and_one_point_clouds(hip_module):
    """The staged clouds start on 16-byte boundaries and are read in 16-byte steps: odd point counts (a padding double behind
    the cloud), clouds of one point, clouds that end inside a block's first pair."""
    A = _HipSubmap(hip_module, {}); B = ob.OracleSubmap()
    sizes = [(51, 21, 101, 103), (1, 1, 1, 1), (257, 3, 513, 255), (255, 7, 1, 769), (3, 5, 771, 1)]
    for f, n in enumerate(sizes):
        cl = ss.frame_clouds(11, f, n=n)
        if f == 0:
            A.init(*cl); B.init(*cl)
        else:
            T = ss.frame_pose(f, step=1.0, yaw_rate=0.02)
            A.update(T, *cl); B.update(T, *cl)
        for k in range(4):
            a, b = A.get(k), B.get(k)
            assert a.shape == b.shape and np.array_equal(a, b), (f, k)


def test_empty_and_everything_cropped(hip_module):
    cl = ss.frame_clouds(9, 0, n=(50, 20, 100, 100))
    A = _HipSubmap(hip_module, dict(edge_crop_box_length=1.0, ground_crop_box_length=1.0))
    B = ob.OracleSubmap(ob.make_submap_config(edge_crop_box_length=1.0, ground_crop_box_length=1.0))
    A.init(*cl); B.init(*cl)
    far = np.eye(4); far[:3, 3] = [500.0, 0.0, 0.0]          # the box leaves every old point behind
    empty = np.zeros((0, 3))
    A.update(far, cl[0], cl[1], empty, empty); B.update(far, cl[0], cl[1], empty, empty)
    for k in range(4):
        assert np.array_equal(A.get(k), B.get(k)), k
    assert len(A.get(2)) == 0 and len(A.get(1)) == 0
    A.update(far, empty, empty, cl[2], cl[3]); B.update(far, empty, empty, cl[2], cl[3])
    for k in range(4):
        assert np.array_equal(A.get(k), B.get(k)), k


def test_update_before_init_is_refused(hip_module):
    H = hip_module.HipRegistration()
    cl = ss.frame_clouds(0, 0, n=(10, 5, 20, 20))
    with pytest.raises(hip_module.TloamHipError):
        H.submap_update(np.eye(4), *cl)
    H.close()


def test_scan_match_on_the_device_submap(hip_module):
    """End to end: targets produced ON the device by the submap path feed scan_match without a host round
    trip; the pose equals the oracle's, whose targets came from the oracle's submap."""
    sc = synth.make_scene(seed=2, n_src=synth.SMALL_SRC, n_tgt=synth.SMALL_TGT)
    H = hip_module.HipRegistration()
    O = ob.Oracle()
    S = ob.OracleSubmap()
    tgt = [sc.target.cloud(k) for k in range(4)]
    # first frame: submap = the scene's target clouds (identity pose); then one update with a far-away tiny scan
    H.submap_init(tgt[0], tgt[3], tgt[2], tgt[1]); S.init(tgt[0], tgt[3], tgt[2], tgt[1])
    extra = ss.frame_clouds(1, 1, n=(30, 10, 40, 60))
    T1 = np.eye(4); T1[:3, 3] = [0.3, 0.1, 0.0]
    H.submap_update(T1, tgt[0], extra[1], extra[2], extra[3]); S.update(T1, tgt[0], extra[1], extra[2], extra[3])
    for k in range(4):
        H.set_source(k, sc.source.cloud(k)); O.set_source(k, sc.source.cloud(k))
        O.set_target(k, S.get(k))
        assert np.array_equal(H.get_target(k), S.get(k))
    rc_h, T_h, st_h = H.scan_match(sc.T_pred)
    rc_o, T_o, st_o = O.scan_match(sc.T_pred)
    assert rc_h == rc_o == 0
    dt, dr = pose_delta(T_h, T_o)
    assert dt < 1e-6 and dr < 1e-6        # north-star tolerance; in practice ~1e-12
    assert st_h["n_corr"] == st_o["n_corr"]
    H.close()


def test_look_back_time_out_switches_to_start_tickets(hip_module):
    """k_vox_emit's blocks wait for the blocks in front of them (look-back scan); they take their places from
    blockIdx only while the whole grid is resident on the device, the wait is bounded, and one that runs out raises a fault word:
    the update in progress reports TLOAM_E_HIP (its submap is undefined) and the context's emit blocks take start tickets from
    then on.  The word is raised by hand here (tloam_debug_raise_fault); after a new init the ticketed form must reproduce the
    oracle bit for bit."""
    A = _HipSubmap(hip_module, {}); B = ob.OracleSubmap()
    cl0 = ss.frame_clouds(5, 0)
    A.init(*cl0); B.init(*cl0)
    assert A.H.L.tloam_debug_raise_fault(A.H.h, 1) == 0
    with pytest.raises(hip_module.TloamHipError):
        A.update(ss.frame_pose(1, step=1.0, yaw_rate=0.02), *ss.frame_clouds(5, 1))
    A.init(*cl0)
    for f in range(1, 6):
        T = ss.frame_pose(f, step=1.0, yaw_rate=0.02)
        cl = ss.frame_clouds(5, f)
        A.update(T, *cl); B.update(T, *cl)
        for k in range(4):
            assert np.array_equal(A.get(k), B.get(k)), (f, k)


def test_non_finite_points_in_the_scans(hip_module):
    """NaN and +-Inf coordinates in the edge / ground scans.  Behind the crop box (`updateSubmap`, front_end.cpp:246-262) they
    disappear: the box's comparisons are false for a NaN, an infinity is outside every box.  Where a cloud is voxel-sampled
    WITHOUT a crop (the first-frame branch, :283-304) an infinite extent is Open3D's "voxel_size is too small" error: a status
    code here, in the restatement and on the device alike, and the submap can be initialised again afterwards."""
    cfg = dict(edge_crop_box_length=25.0, ground_crop_box_length=20.0, planar_frame_size=3)
    A = _HipSubmap(hip_module, cfg)
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    rng = np.random.default_rng(11)

    def spoil(clouds):
        out = [c.copy() for c in clouds]
        for k in (2, 3):                       # edge, ground (the planar / sphere selections feed a kd-tree build: kept finite)
            c = out[k]
            j = rng.choice(len(c), 6, replace=False)
            c[j[0], 0] = np.nan; c[j[1], 2] = np.nan; c[j[2]] = np.nan
            c[j[3], 1] = np.inf; c[j[4], 0] = -np.inf; c[j[5]] = (np.inf, -np.inf, np.nan)
        return out

    def status(fn, *args):
        try:
            rc = fn(*args)
            return 0 if rc is None else int(rc)
        except hip_module.TloamHipError as e:
            assert "TLOAM_E_INVALID" in str(e), e
            return -1

    first = ss.frame_clouds(4, 0)
    ra, rb = status(A.init, *spoil(first)), status(B.init, *spoil(first))
    assert (ra != 0) == (rb != 0), (ra, rb)
    if ra != 0:                                  # refused alike: start over with a finite first frame
        assert status(A.init, *first) == 0 and status(B.init, *first) == 0
    for f in range(1, 6):
        cl = spoil(ss.frame_clouds(4, f))
        T = ss.frame_pose(f, step=2.0, yaw_rate=0.03)
        ra, rb = status(A.update, T, *cl), status(B.update, T, *cl)
        assert (ra != 0) == (rb != 0), (f, ra, rb)
        assert ra == 0, f                        # behind the crop box the non-finite points are simply gone
        for k in range(4):
            a, b = A.get(k), B.get(k)
            assert a.shape == b.shape, (f, k, a.shape, b.shape)
            assert np.array_equal(a, b, equal_nan=True), (f, k)
            assert np.isfinite(a).all(), (f, k)


def test_non_finite_pose_is_a_status_code(hip_module):
    """`tloam_submap_update` with a NaN / an infinity anywhere in the pose: TLOAM_E_BAD_POSE, the submap untouched and usable
    (until round 5 the NaN crop box ended in a GPU memory fault -- found by tests/tools/fuzz_call_order.py).  Any FINITE 4x4 is
    taken as Open3D's Transform takes it, scaled rotation and projective last row included."""
    cfg = dict(edge_crop_box_length=40.0, ground_crop_box_length=30.0, planar_frame_size=2)
    A = _HipSubmap(hip_module, cfg)
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    cl = ss.frame_clouds(6, 0)
    A.init(*cl); B.init(*cl)
    before = [A.get(k) for k in range(4)]
    for r, c_, v in ((0, 3, np.nan), (1, 3, np.inf), (2, 3, -np.inf), (1, 1, np.nan), (3, 3, np.inf), (3, 0, np.nan)):
        T = ss.frame_pose(1, step=2.0, yaw_rate=0.03).copy()
        T[r, c_] = v
        with pytest.raises(hip_module.TloamHipError, match="BAD_POSE"):
            A.update(T, *ss.frame_clouds(6, 1))
    for k in range(4):
        assert np.array_equal(A.get(k), before[k]), k
    T = ss.frame_pose(1, step=2.0, yaw_rate=0.03).copy()
    T[:3, :3] *= 1.2
    T[3, :] = (0.001, -0.002, 0.0005, 1.1)
    cl = ss.frame_clouds(6, 1)
    A.update(T, *cl); B.update(T, *cl)
    for k in range(4):
        a, b = A.get(k), B.get(k)
        assert a.shape == b.shape and np.array_equal(a, b), k


def test_submaps_that_outgrow_their_arrays(hip_module):
    """A submap initialised from a handful of points and then fed scans a hundred times that size: the edge / ground arrays
    are input (the old submap) and output of every update and have to grow while keeping their points (round 5: the one-launch
    front used to read the old points from a block that had just been freed).  Other contexts allocate and free blocks of the
    same sizes in between, so a freed block does not stay intact by luck."""
    cfg = dict(edge_crop_box_length=200.0, ground_crop_box_length=200.0, planar_frame_size=2, edge_down_sample_submap=0.05,
               ground_down_sample_submap=0.05, ground_down_sample=0.05)
    A = _HipSubmap(hip_module, cfg)
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    rng = np.random.default_rng(21)
    sizes = [12, 40, 700, 9000, 300, 30000, 50]
    for f, n in enumerate(sizes):
        cl = [np.ascontiguousarray(rng.uniform(-60, 60, (n + 3 * k, 3)).astype(np.float32).astype(np.float64)) for k in range(4)]
        if f == 0:
            A.init(*cl); B.init(*cl)
        else:
            T = ss.frame_pose(f, step=1.0, yaw_rate=0.02)
            A.update(T, *cl); B.update(T, *cl)
        # stir the allocator: a context that takes and returns blocks of this frame's sizes
        X = hip_module.HipRegistration()
        for k in range(4):
            X.set_target(k, cl[k]); X.set_source(k, cl[k])
        X.close()
        for k in range(4):
            a, b = A.get(k), B.get(k)
            assert a.shape == b.shape, (f, k, a.shape, b.shape)
            assert np.array_equal(a, b), (f, k)


@pytest.mark.parametrize("seed,trial", [(2, 21), (1, 358)])
def test_sizes_of_consecutive_updates_that_xor_alike(hip_module, seed, trial):
    """The two sizes of an update come back through a pinned result segment whose check word used to be the sequence number XOR
    the payload words: when the check word had arrived and the payload had not, the segment still checked if the old and the new
    payload XORed alike -- (58, 58) after (0, 0); (459, 16) after (458, 17) -- and the update's clouds were cut to the sizes of
    the update before (2-4 % of fresh contexts; found by tests/tools/stress_rows.py, whose two failing sequences these are).
    Every payload word now enters the check through a position-dependent mix (tl_common.hpp seg_word).  300 fresh contexts each."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("stress_rows", os.path.join(os.path.dirname(__file__), "tools", "stress_rows.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    rng = np.random.default_rng([seed, trial, 1])
    extent = float(rng.choice([5.0, 20.0, 60.0]))
    vox = float(rng.choice([0.02, 0.1, 0.3, 0.45, 1.0, 5.0]))
    cfg = dict(edge_crop_box_length=float(rng.choice([extent * 0.3, extent, extent * 3])),
               ground_crop_box_length=float(rng.choice([extent * 0.3, extent, extent * 3])),
               planar_frame_size=int(rng.integers(1, 6)), sphere_frame_size=int(rng.integers(1, 6)),
               edge_down_sample_submap=vox, ground_down_sample_submap=float(rng.choice([vox, vox * 1.5])), ground_down_sample=vox)
    frames, poses, T = [], [], np.eye(4)
    for f in range(int(rng.integers(3, 9))):
        cl = []
        for k in range(4):
            c = sr.rand_cloud(rng, int(rng.choice([0, 1, 2, 7, 60, 700, 6000])), extent, vox)
            if k < 2 or f == 0:
                c = c[np.isfinite(c).all(1)]
            cl.append(np.ascontiguousarray(c))
        if f > 0:
            T = T @ ss._se3_exp((rng.uniform(0, 2), rng.normal(0, 0.2), rng.normal(0, 0.05), rng.normal(0, 0.02), rng.normal(0, 0.02),
                                 rng.normal(0, 0.3)))
        frames.append(cl)
        poses.append(T.copy())
    B = ob.OracleSubmap(ob.make_submap_config(**cfg))
    want = []
    for f, cl in enumerate(frames):
        assert (B.init(*cl) if f == 0 else B.update(poses[f], *cl)) == 0
        want.append([B.get(k) for k in range(4)])
    # the payload of every call's result segment: (edge, ground) sizes of an update; the first frame's job is the ground cloud alone
    sizes = [(len(want[0][1]), 0)] + [(len(w[2]), len(w[1])) for w in want[1:]]
    assert any(a != b and (a[0] ^ a[1]) == (b[0] ^ b[1]) for a, b in zip(sizes, sizes[1:])), sizes   # the case this test is about
    hcfg = hip_module.default_submap_config(**cfg)
    for rep in range(300):
        H = hip_module.HipRegistration()
        for f, cl in enumerate(frames):
            if f == 0:
                H.submap_init(*cl, cfg=hcfg)
            else:
                H.submap_update(poses[f], *cl)
            for k in range(4):
                a = H.get_target(k)
                assert a.shape == want[f][k].shape and np.array_equal(a, want[f][k], equal_nan=True), (rep, f, k, a.shape, want[f][k].shape)
        H.close()
