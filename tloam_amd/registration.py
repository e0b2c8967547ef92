"""Host-side mirror of the reference's registration plugin, bound to the C ABI (include/tloam_hip.h).

`HipRegistration` follows tloam::RegistrationInterface
(include/tloam/models/registration/registration_interface.hpp:40-48):

    setInputSource(Frame&)            -> set_input_source(frame)
    setInputTarget(Frame&)            -> set_input_target(frame)
    scanMatching(out, predict, pose)  -> scan_matching(predict, ...)  -> (ok, result_pose)
    getFitnessScore()                 -> get_fitness_score()

and is selected the way FrontEnd::initRegistraton selects "TLS" (front_end.cpp:155-167): see
`make_registration("TLS_HIP", cfg)`.  All computation happens in libtloam_hip.so (hand-written HIP
kernels for gfx950); this module only marshals numpy arrays into the C ABI through ctypes.  There is
no CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import sys
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TLOAM_HIP_LIB") or os.path.join(_HERE, "libtloam_hip.so")  # env override: tuning builds only

KIND_PLANAR, KIND_GROUND, KIND_EDGE, KIND_SPHERE = 0, 1, 2, 3
RES_PLANE, RES_LINE, RES_POINT = 0, 1, 2

STATUS = {0: "TLOAM_OK", -1: "TLOAM_E_INVALID", -2: "TLOAM_E_TOO_FEW_POINTS", -3: "TLOAM_E_BAD_POSE",
          -4: "TLOAM_E_HIP", -5: "TLOAM_E_RCCL", -6: "TLOAM_E_NOT_READY", -7: "TLOAM_E_WEIGHT_RANGE"}


class SubmapConfig(C.Structure):
    """tloam_submap_config: the submap keys of config/mapping/lidar_odometry.yaml:6-17."""
    _fields_ = [("planar_frame_size", C.c_int32), ("sphere_frame_size", C.c_int32),
                ("edge_crop_box_length", C.c_double), ("ground_crop_box_length", C.c_double),
                ("edge_down_sample_submap", C.c_double), ("ground_down_sample_submap", C.c_double),
                ("ground_down_sample", C.c_double)]


class FeatureConfig(C.Structure):
    """tloam_feature_config: the `feature:` block of config/mapping/feature.yaml."""
    _fields_ = [("radius", C.c_double), ("K", C.c_int32), ("min_neigh", C.c_int32), ("planar_num", C.c_int32),
                ("sphere_num", C.c_int32), ("cvr_scan", C.c_double), ("cvr_submap", C.c_double),
                ("planar_scan_thres", C.c_double), ("planar_submap_thres", C.c_double),
                ("planar_vertic_thres", C.c_double)]


class TlsConfig(C.Structure):
    """tloam_tls_config: the 16 keys of the `TLS:` block (config/mapping/lidar_odometry.yaml:23-39)."""
    _fields_ = [
        ("k_corr", C.c_int32), ("factor_num", C.c_int32),
        ("edge_dist_thres", C.c_double), ("edge_dir_thres", C.c_double),
        ("edge_maxnum", C.c_int32), ("sphere_maxnum", C.c_int32),
        ("sphere_dist_thres", C.c_double), ("planar_dist_thres", C.c_double),
        ("planar_maxnum", C.c_int32), ("ground_maxnum", C.c_int32),
        ("ground_dist_thres", C.c_double),
        ("max_iterations", C.c_int32), ("reserved0", C.c_int32),
        ("cost_threshold", C.c_double), ("gnc_factor", C.c_double),
        ("noise_bound", C.c_double), ("fitness_thres", C.c_double),
    ]


class CtxInfo(C.Structure):
    """tloam_ctx_info."""
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("device_cus", C.c_int32), ("comm_mode", C.c_int32),
                ("rank", C.c_int32), ("nranks", C.c_int32), ("rccl_comm_count", C.c_int32), ("rccl_comm_rank", C.c_int32),
                ("fallbacks_taken", C.c_int32), ("fallback_events", C.c_int32), ("k3_grid", C.c_int32), ("k3_single", C.c_int32),
                ("one_launch_solve", C.c_int32), ("loopback", C.c_int32), ("direct_set", C.c_int32), ("set_stale", C.c_int32),
                ("k3_wide", C.c_int32)]


class Stats(C.Structure):
    """tloam_stats."""
    _fields_ = [
        ("outer_iterations", C.c_int32), ("gn_evaluations", C.c_int32),
        ("gn_iterations", C.c_int32), ("accepted_steps", C.c_int32),
        ("n_corr", C.c_int32 * 4), ("converged_early", C.c_int32), ("weight_range_violations", C.c_int32),
        ("kind_cost", C.c_double * 4), ("mu", C.c_double), ("solver_cost", C.c_double),
        ("se3", C.c_double * 6),
        ("gn_sweeps", C.c_int32), ("host_wait_us", C.c_int32),
    ]

    def as_dict(self):
        # (on the per-frame path of every caller: array fields as slices / a buffer view, ~2 us instead of ~4)
        return {"outer_iterations": self.outer_iterations, "gn_evaluations": self.gn_evaluations,
                "gn_iterations": self.gn_iterations, "accepted_steps": self.accepted_steps,
                "n_corr": self.n_corr[:], "converged_early": self.converged_early,
                "bad_weights": self.weight_range_violations, "kind_cost": self.kind_cost[:], "mu": self.mu,
                "solver_cost": self.solver_cost, "se3": np.frombuffer(self.se3, dtype=np.float64).copy(),
                "gn_sweeps": self.gn_sweeps, "host_wait_us": self.host_wait_us}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

_lib = None


class TloamHipError(RuntimeError):
    pass


def load_library():
    """dlopen libtloam_hip.so (built by tloam_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TloamHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in HBM (effective if HIP is not up yet)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    sz = C.c_size_t
    sig = {
        "tloam_abi_version": (C.c_int, []),
        "tloam_status_string": (C.c_char_p, [C.c_int]),
        "tloam_last_error": (C.c_char_p, [vp]),
        "tloam_default_config": (None, [C.POINTER(TlsConfig)]),
        "tloam_create": (C.c_int, [C.POINTER(TlsConfig), C.c_int, C.POINTER(vp)]),
        "tloam_destroy": (None, [vp]),
        "tloam_set_source": (C.c_int, [vp, C.c_int, dp, sz]),
        "tloam_set_target": (C.c_int, [vp, C.c_int, dp, sz]),
        "tloam_set_source_frame": (C.c_int, [vp, C.POINTER(dp), C.POINTER(sz)]),
        "tloam_set_target_frame": (C.c_int, [vp, C.POINTER(dp), C.POINTER(sz)]),
        "tloam_frame_stash": (C.c_int, [vp, C.c_int]),
        "tloam_frame_select": (C.c_int, [vp, C.c_int]),
        "tloam_scan_match": (C.c_int, [vp, dp, dp, dp, dp, sz, C.POINTER(Stats)]),
        "tloam_sm_begin": (C.c_int, [vp, dp, dp]),
        "tloam_sm_outer": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(Stats)]),
        "tloam_sm_end": (C.c_int, [vp, dp, C.POINTER(Stats)]),
        "tloam_fitness": (C.c_int, [vp, dp, dp]),
        "tloam_get_correspondences": (C.c_int, [vp, C.c_int, sz, C.POINTER(sz), ip, dp, dp, dp, dp, dp]),
        "tloam_get_weights": (C.c_int, [vp, C.c_int, sz, C.POINTER(sz), dp]),
        "tloam_knn": (C.c_int, [vp, C.c_int, dp, sz, C.c_double, C.c_int, ip, dp, ip]),
        "tloam_set_correspondences": (C.c_int, [vp, C.c_int, sz, dp, dp, dp, dp, dp]),
        "tloam_accumulate": (C.c_int, [vp, dp, dp, dp, dp]),
        "tloam_get_costs": (C.c_int, [vp, C.c_int, sz, C.POINTER(sz), dp]),
        "tloam_get_normal_equations": (C.c_int, [vp, dp, dp, dp]),
        "tloam_comm_mailbox_export": (C.c_int, [vp, vp]),
        "tloam_comm_init_mailbox": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "tloam_solve": (C.c_int, [vp, dp, C.POINTER(Stats)]),
        "tloam_time_accumulate": (C.c_int, [vp, dp, C.c_int, dp]),
        "tloam_time_sharded_sweep": (C.c_int, [vp, dp, C.c_int, C.c_int, dp]),
        "tloam_time_build": (C.c_int, [vp, C.c_int, dp, C.POINTER(C.c_int64)]),
        "tloam_k3_timer": (C.c_int, [vp, C.c_int, dp, C.POINTER(C.c_int64), dp]),
        "tloam_k3_timer_all": (C.c_int, [vp, dp, C.POINTER(C.c_int64)]),
        "tloam_k3_span": (C.c_int, [vp, C.c_int, dp, C.POINTER(C.c_int64)]),
        "tloam_gn_iter_timer": (C.c_int, [vp, C.c_int, dp, C.POINTER(C.c_int64)]),
        "tloam_time_read_stream": (C.c_int, [vp, sz, C.c_int, dp]),
        "tloam_get_info": (C.c_int, [vp, C.POINTER(CtxInfo)]),
        "tloam_debug_state": (C.c_int, [vp, dp, C.c_int]),
        "tloam_debug_se3": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "tloam_debug_partials": (C.c_int, [vp, dp, C.c_int]),
        "tloam_debug_raise_fault": (C.c_int, [vp, C.c_int]),
        "tloam_submap_default_config": (None, [C.POINTER(SubmapConfig)]),
        "tloam_submap_init": (C.c_int, [vp, C.POINTER(SubmapConfig), dp, sz, dp, sz, dp, sz, dp, sz]),
        "tloam_submap_update": (C.c_int, [vp, dp, dp, sz, dp, sz, dp, sz, dp, sz]),
        "tloam_get_target": (C.c_int, [vp, C.c_int, sz, C.POINTER(sz), dp]),
        "tloam_feature_default_config": (None, [C.POINTER(FeatureConfig)]),
        "tloam_pca_info": (C.c_int, [vp, C.POINTER(FeatureConfig), dp, sz, dp, dp, dp, dp, ip, ip]),
        "tloam_extract_planar_sphere": (C.c_int, [vp, C.POINTER(FeatureConfig), dp, sz, ip, C.POINTER(sz), ip,
                                                  C.POINTER(sz), ip, C.POINTER(sz), ip, C.POINTER(sz)]),
        "tloam_rccl_unique_id": (C.c_int, [vp]),
        "tloam_comm_init_rccl": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "tloam_comm_init_callback": (C.c_int, [vp, C.c_int, C.c_int, ALLREDUCE_FN, vp]),
        "tloam_shard_range": (None, [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
        "tloam_shard_ranges_frame": (None, [C.POINTER(sz), C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
        "tloam_se3_exp": (C.c_int, [dp, dp]),
        "tloam_se3_log": (C.c_int, [dp, dp]),
        "tloam_se3_plus": (C.c_int, [dp, dp, dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "tloam_abi_version", "tloam_status_string", "tloam_last_error", "tloam_default_config", "tloam_create",
    "tloam_destroy", "tloam_set_source", "tloam_set_target", "tloam_set_source_frame", "tloam_set_target_frame",
    "tloam_frame_stash", "tloam_frame_select", "tloam_scan_match", "tloam_sm_begin",
    "tloam_sm_outer", "tloam_sm_end", "tloam_fitness", "tloam_get_correspondences", "tloam_get_weights",
    "tloam_knn", "tloam_set_correspondences", "tloam_accumulate", "tloam_get_costs", "tloam_get_normal_equations",
    "tloam_solve",
    "tloam_time_accumulate", "tloam_time_sharded_sweep", "tloam_time_build", "tloam_k3_timer", "tloam_k3_timer_all", "tloam_k3_span", "tloam_gn_iter_timer", "tloam_time_read_stream", "tloam_get_info", "tloam_debug_state", "tloam_debug_partials", "tloam_debug_se3",
    "tloam_debug_raise_fault",
    "tloam_submap_default_config", "tloam_submap_init", "tloam_submap_update", "tloam_get_target",
    "tloam_feature_default_config", "tloam_pca_info", "tloam_extract_planar_sphere", "tloam_rccl_unique_id", "tloam_comm_init_rccl",
    "tloam_comm_mailbox_export", "tloam_comm_init_mailbox",
    "tloam_comm_init_callback", "tloam_shard_range", "tloam_shard_ranges_frame", "tloam_se3_exp", "tloam_se3_log", "tloam_se3_plus",
)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _aos(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, 3))


def _colmajor(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(-1)


def default_config(**over) -> TlsConfig:
    cfg = TlsConfig()
    load_library().tloam_default_config(C.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


def shard_range(n, rank, nranks):
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    load_library().tloam_shard_range(C.c_size_t(n), int(rank), int(nranks), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_ranges_frame(n4, rank, nranks):
    """[(lo, hi)] x 4: the blocks of a whole Frame a sharded context keeps (tloam_shard_ranges_frame)."""
    n = (C.c_size_t * 4)(*[int(v) for v in n4])
    lo, hi = (C.c_size_t * 4)(), (C.c_size_t * 4)()
    load_library().tloam_shard_ranges_frame(n, int(rank), int(nranks), lo, hi)
    return [(int(lo[k]), int(hi[k])) for k in range(4)]


def se3_exp(x):
    T = np.zeros(16)
    rc = load_library().tloam_se3_exp(_dp(np.ascontiguousarray(x, float)), _dp(T))
    if rc:
        raise TloamHipError(STATUS.get(rc, rc))
    return T.reshape(4, 4).T.copy()


def se3_log(T):
    x = np.zeros(6)
    rc = load_library().tloam_se3_log(_dp(_colmajor(T)), _dp(x))
    if rc:
        raise TloamHipError(STATUS.get(rc, rc))
    return x


def se3_plus(x, delta):
    o = np.zeros(6)
    load_library().tloam_se3_plus(_dp(np.ascontiguousarray(x, float)), _dp(np.ascontiguousarray(delta, float)), _dp(o))
    return o


class HipRegistration:
    """MI355X-native drop-in for tloam::LocalRegistration behind RegistrationInterface."""

    def __init__(self, cfg: TlsConfig | None = None, device: int = 0):
        self.L = load_library()
        self.cfg = cfg if cfg is not None else default_config()
        self.h = C.c_void_p()
        rc = self.L.tloam_create(C.byref(self.cfg), int(device), C.byref(self.h))
        if rc != 0:
            raise TloamHipError(f"tloam_create: {STATUS.get(rc, rc)} -- a gfx950 (MI355X) device is required; "
                                "this path has no CPU fallback")
        self._n = {}
        self._cb = None
        # marshalling buffers of scan_match (column-major 4x4 in / out), allocated once: the binding adds ~3 us per
        # call instead of ~10
        self._pred_buf, self._res_buf = (C.c_double * 16)(), (C.c_double * 16)()
        self._pred_view = np.frombuffer(self._pred_buf).reshape(4, 4).T
        self._res_view = np.frombuffer(self._res_buf).reshape(4, 4).T

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.tloam_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.L.tloam_last_error(self.h)
            raise TloamHipError(f"{what}: {STATUS.get(rc, rc)} {msg.decode() if msg else ''}")

    # ---- RegistrationInterface ------------------------------------------------------------
    def _note(self, rc, what):
        """entry points whose status the caller inspects (they do not raise): a HIP / exchange failure leaves its text on stderr"""
        if rc in (-4, -5):
            sys.stderr.write(f"[tloam_amd] {what}: {STATUS.get(rc, rc)}: {self.L.tloam_last_error(self.h).decode(errors='replace')}\n")
        return rc

    def _frame_call(self, fn, name, tag, frame):
        clouds = [_aos(frame.cloud(k)) for k in range(4)]
        ptrs = (C.POINTER(C.c_double) * 4)(*[_dp(a) for a in clouds])
        ns = (C.c_size_t * 4)(*[len(a) for a in clouds])
        for k in range(4):
            self._n[(tag, k)] = len(clouds[k])
        self._check(fn(self.h, ptrs, ns), name)
        return True

    def set_input_source(self, frame) -> bool:
        """setInputSource (registration.cpp:232-239): the four clouds of the Frame in one call."""
        return self._frame_call(self.L.tloam_set_source_frame, "tloam_set_source_frame", "s", frame)

    def set_input_target(self, frame) -> bool:
        """setInputTarget (registration.cpp:241-248)."""
        return self._frame_call(self.L.tloam_set_target_frame, "tloam_set_target_frame", "t", frame)

    def scan_matching(self, predict_pose, omega_perturb=None, scan_cloud=None):
        """scanMatching (registration.cpp:879-1133) -> (True, result_pose 4x4).  `scan_cloud`
        ((n,3) float64, C-contiguous) is transformed in place like out_result_.scan_cloud."""
        rc, T, st = self.scan_match(predict_pose, omega_perturb, scan_cloud)
        self._check(rc, "tloam_scan_match")
        self.last_stats = st
        return True, T

    def get_fitness_score(self):
        """getFitnessScore (registration.cpp:257-296)."""
        f, r = C.c_double(0), C.c_double(0)
        self._check(self.L.tloam_fitness(self.h, C.byref(f), C.byref(r)), "tloam_fitness")
        return f.value, r.value

    # ---- C ABI, one to one --------------------------------------------------------------------
    def set_source(self, kind, xyz):
        a = _aos(xyz)
        self._n[("s", kind)] = len(a)
        rc = self.L.tloam_set_source(self.h, int(kind), _dp(a), len(a))
        self._check(rc, "tloam_set_source")
        return rc

    def set_target(self, kind, xyz):
        a = _aos(xyz)
        self._n[("t", kind)] = len(a)
        rc = self.L.tloam_set_target(self.h, int(kind), _dp(a), len(a))
        self._check(rc, "tloam_set_target")
        return rc

    def set_frames(self, source, target):
        self.set_input_source(source)
        self.set_input_target(target)

    def frame_stash(self, slot):
        """tloam_frame_stash: the registered clouds move into slot `slot` of the HBM frame store."""
        self._check(self.L.tloam_frame_stash(self.h, int(slot)), "tloam_frame_stash")

    def frame_select(self, slot):
        """tloam_frame_select: the clouds of `slot` become the registered ones (-1: the context's own); O(1)."""
        rc = self.L.tloam_frame_select(self.h, int(slot))
        if rc != 0:
            self._check(rc, "tloam_frame_select")

    def scan_match(self, predict, omega=None, scan=None):
        self._pred_view[...] = predict
        self._res_view[...] = 0.0
        st = Stats()
        om = None if omega is None else np.ascontiguousarray(omega, float)
        if scan is not None:
            assert scan.dtype == np.float64 and scan.flags.c_contiguous
        rc = self.L.tloam_scan_match(self.h, self._pred_buf, _dp(om), self._res_buf, _dp(scan),
                                     0 if scan is None else len(scan), C.byref(st))
        self._note(rc, "tloam_scan_match")
        return rc, self._res_view.copy(), st.as_dict()

    def sm_begin(self, predict, omega=None):
        om = None if omega is None else np.ascontiguousarray(omega, float)
        return self._note(self.L.tloam_sm_begin(self.h, _dp(_colmajor(predict)), _dp(om)), "tloam_sm_begin")

    def sm_outer(self):
        done = C.c_int(0)
        st = Stats()
        rc = self._note(self.L.tloam_sm_outer(self.h, C.byref(done), C.byref(st)), "tloam_sm_outer")
        return rc, bool(done.value), st.as_dict()

    def sm_end(self):
        res = np.zeros(16)
        st = Stats()
        rc = self._note(self.L.tloam_sm_end(self.h, _dp(res), C.byref(st)), "tloam_sm_end")
        return rc, res.reshape(4, 4).T.copy(), st.as_dict()

    # ---- device-resident submap (FrontEnd::updateSubmap, front_end.cpp:201-275 / :283-304)
    def submap_init(self, planar, sphere, edge, ground, cfg: SubmapConfig | None = None):
        cl = [_aos(x) for x in (planar, sphere, edge, ground)]
        args = []
        for a in cl:
            args += [_dp(a), len(a)]
        rc = self.L.tloam_submap_init(self.h, C.byref(cfg) if cfg is not None else None, *args)
        self._check(rc, "tloam_submap_init")
        return rc

    def submap_update(self, pose, planar, sphere, edge, ground):
        m = np.ascontiguousarray(np.asarray(pose, float).reshape(4, 4).T.ravel())  # column-major
        cl = [_aos(x) for x in (planar, sphere, edge, ground)]
        args = []
        for a in cl:
            args += [_dp(a), len(a)]
        rc = self.L.tloam_submap_update(self.h, _dp(m), *args)
        self._check(rc, "tloam_submap_update")
        return rc

    def get_target(self, kind):
        n = C.c_size_t(0)
        self.L.tloam_get_target(self.h, int(kind), 0, C.byref(n), None)
        out = np.zeros((max(n.value, 1), 3))
        if n.value:
            self._check(self.L.tloam_get_target(self.h, int(kind), n.value, C.byref(n), _dp(out)), "tloam_get_target")
        self._n[("t", kind)] = n.value
        return out[: n.value].copy()

    # ---- PCA feature extraction (featureExtract::calculatePCAInfo / extractPlanarSphere)
    def pca_info(self, xyz, cfg: FeatureConfig | None = None):
        cfg = cfg or default_feature_config()
        a = _aos(xyz)
        n, K = len(a), cfg.K
        out = dict(flatness=np.zeros(n), cvr=np.zeros(n), sphericity=np.zeros(n), normal=np.zeros((max(n, 1), 3)),
                   num_sum=np.zeros(n, np.int32), neigh=np.zeros((max(n, 1), K), np.int32))
        rc = self.L.tloam_pca_info(self.h, C.byref(cfg), _dp(a), n, _dp(out["flatness"]), _dp(out["cvr"]),
                                   _dp(out["sphericity"]), _dp(out["normal"]), _ip(out["num_sum"]), _ip(out["neigh"]))
        self._check(rc, "tloam_pca_info")
        out["normal"] = out["normal"][:n]
        out["neigh"] = out["neigh"][:n]
        return out

    def extract_planar_sphere(self, xyz, cfg: FeatureConfig | None = None):
        """-> (planar_scan_index, planar_submap_index, sphere_scan_index, sphere_submap_index)"""
        cfg = cfg or default_feature_config()
        a = _aos(xyz)
        n = len(a)
        lists = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
        cnt = [C.c_size_t(0) for _ in range(4)]
        args = []
        for l, k in zip(lists, cnt):
            args += [_ip(l), C.byref(k)]
        rc = self.L.tloam_extract_planar_sphere(self.h, C.byref(cfg), _dp(a), n, *args)
        self._check(rc, "tloam_extract_planar_sphere")
        return tuple(l[: k.value].copy() for l, k in zip(lists, cnt))

    def fitness(self):
        f, r = C.c_double(0), C.c_double(0)
        rc = self.L.tloam_fitness(self.h, C.byref(f), C.byref(r))
        return rc, f.value, r.value

    def get_correspondences(self, kind, capacity=None):
        cap = int(capacity or max(self._n.get(("s", kind), 0), self._n.get(("c", kind), 0), 1))
        n = C.c_size_t(0)
        idx = np.zeros(cap, np.int32); a = np.zeros((cap, 3)); b = np.zeros((cap, 3))
        d = np.zeros(cap); w = np.zeros(cap); cost = np.zeros(cap)
        rc = self.L.tloam_get_correspondences(self.h, int(kind), cap, C.byref(n), _ip(idx), _dp(a), _dp(b),
                                              _dp(d), _dp(w), _dp(cost))
        self._check(rc, "tloam_get_correspondences")
        m = n.value
        return dict(idx=idx[:m], a=a[:m], b=b[:m], d=d[:m], w=w[:m], cost=cost[:m])

    def get_weights(self, kind):
        cap = max(self._n.get(("s", kind), 0), 1)
        n = C.c_size_t(0)
        w = np.zeros(cap)
        self._check(self.L.tloam_get_weights(self.h, int(kind), cap, C.byref(n), _dp(w)), "tloam_get_weights")
        return w[:n.value]

    def knn(self, kind, queries, radius, k):
        q = _aos(queries)
        idx = np.zeros((len(q), k), np.int32); d2 = np.zeros((len(q), k)); cnt = np.zeros(len(q), np.int32)
        rc = self.L.tloam_knn(self.h, int(kind), _dp(q), len(q), float(radius), int(k), _ip(idx), _dp(d2), _ip(cnt))
        self._check(rc, "tloam_knn")
        return idx, d2, cnt

    def set_correspondences(self, res_type, p, a, b=None, d=None, w=None):
        p = _aos(p); a = _aos(a)
        b = None if b is None else _aos(b)
        d = None if d is None else np.ascontiguousarray(d, float)
        w = np.ones(len(p)) if w is None else np.ascontiguousarray(w, float)
        kind = {RES_PLANE: KIND_PLANAR, RES_LINE: KIND_EDGE, RES_POINT: KIND_SPHERE}[res_type]
        self._n[("c", kind)] = len(p)
        rc = self.L.tloam_set_correspondences(self.h, int(res_type), len(p), _dp(p), _dp(a), _dp(b), _dp(d), _dp(w))
        self._check(rc, "tloam_set_correspondences")
        return rc

    def accumulate(self, se3):
        x = np.ascontiguousarray(se3, float)
        H = np.zeros(36); g = np.zeros(6); cost = C.c_double(0)
        self._check(self.L.tloam_accumulate(self.h, _dp(x), _dp(H), _dp(g), C.byref(cost)), "tloam_accumulate")
        return H.reshape(6, 6), g, cost.value

    def get_normal_equations(self):
        """(H 6x6, g, cost) the minimiser held at the accepted iterate when its last Solve returned."""
        H = np.zeros(36); g = np.zeros(6); cost = C.c_double(0)
        self._check(self.L.tloam_get_normal_equations(self.h, _dp(H), _dp(g), C.byref(cost)), "tloam_get_normal_equations")
        return H.reshape(6, 6), g, cost.value

    def get_costs(self, res_type):
        kind = {RES_PLANE: KIND_PLANAR, RES_LINE: KIND_EDGE, RES_POINT: KIND_SPHERE}[res_type]
        cap = max(self._n.get(("c", kind), 0), 1)
        n = C.c_size_t(0)
        c = np.zeros(cap)
        self._check(self.L.tloam_get_costs(self.h, int(res_type), cap, C.byref(n), _dp(c)), "tloam_get_costs")
        return c[:n.value]

    def solve(self, se3):
        x = np.array(se3, float)
        st = Stats()
        self._check(self.L.tloam_solve(self.h, _dp(x), C.byref(st)), "tloam_solve")
        return x, st.as_dict()

    def time_accumulate(self, se3, launches=100):
        x = np.ascontiguousarray(se3, float)
        us = C.c_double(0)
        self._check(self.L.tloam_time_accumulate(self.h, _dp(x), int(launches), C.byref(us)), "tloam_time_accumulate")
        return us.value

    def time_build(self, launches=20):
        """(mean us per launch, queries per launch) of the correspondence-search kernel on the last frame's state."""
        us = C.c_double(0); n = C.c_int64(0)
        self._check(self.L.tloam_time_build(self.h, int(launches), C.byref(us), C.byref(n)), "tloam_time_build")
        return us.value, n.value

    def time_sharded_sweep(self, se3, launches=50, with_exchange=True):
        """collective: every rank calls it with the same arguments (tloam_time_sharded_sweep)."""
        x = np.ascontiguousarray(se3, float)
        us = C.c_double(0)
        self._check(self.L.tloam_time_sharded_sweep(self.h, _dp(x), int(launches), int(bool(with_exchange)), C.byref(us)),
                    "tloam_time_sharded_sweep")
        return us.value

    def k3_timer(self, reset=False):
        us = C.c_double(0); n = C.c_int64(0); b = C.c_double(0)
        self._check(self.L.tloam_k3_timer(self.h, int(bool(reset)), C.byref(us), C.byref(n), C.byref(b)), "tloam_k3_timer")
        return us.value, n.value, b.value

    def k3_timer_all(self):
        us = C.c_double(0); n = C.c_int64(0)
        self._check(self.L.tloam_k3_timer_all(self.h, C.byref(us), C.byref(n)), "tloam_k3_timer_all")
        return us.value, n.value

    def k3_span(self, reset=False):
        """(total us, launches) of the streaming span of the one-launch GN iterations (tloam_k3_span)."""
        us = C.c_double(0); n = C.c_int64(0)
        self._check(self.L.tloam_k3_span(self.h, int(bool(reset)), C.byref(us), C.byref(n)), "tloam_k3_span")
        return us.value, n.value

    def gn_iter_timer(self, reset=False):
        """(total us, periods) of the GN iterations as the device clocks them (tloam_gn_iter_timer); the first call arms it."""
        us = C.c_double(0); n = C.c_int64(0)
        self._check(self.L.tloam_gn_iter_timer(self.h, int(bool(reset)), C.byref(us), C.byref(n)), "tloam_gn_iter_timer")
        return us.value, n.value

    def time_read_stream(self, nbytes, launches=20):
        """GB/s of a read stream with the sweep's access pattern over ~nbytes (tloam_time_read_stream)."""
        g = C.c_double(0)
        self._check(self.L.tloam_time_read_stream(self.h, int(nbytes), int(launches), C.byref(g)), "tloam_time_read_stream")
        return g.value

    def info(self):
        """tloam_get_info as a dict."""
        ci = CtxInfo()
        self._check(self.L.tloam_get_info(self.h, C.byref(ci)), "tloam_get_info")
        return {f: getattr(ci, f) for f, _ in CtxInfo._fields_ if f != "reserved"}

    # ---- multi-GPU ---------------------------------------------------------------------------
    def comm_init_rccl(self, rank, nranks, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.L.tloam_comm_init_rccl(self.h, int(rank), int(nranks), C.cast(buf, C.c_void_p)),
                    "tloam_comm_init_rccl")

    def comm_mailbox_export(self) -> bytes:
        """64 bytes (hipIpcMemHandle_t) of this context's exchange buffer; all-gather them in rank order."""
        buf = C.create_string_buffer(64)
        self._check(self.L.tloam_comm_mailbox_export(self.h, C.cast(buf, C.c_void_p)), "tloam_comm_mailbox_export")
        return bytes(buf.raw)

    def comm_init_mailbox(self, rank, nranks, handles):
        """handles: the nranks 64-byte handles in rank order (this rank's own entry is ignored)."""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == 64 * int(nranks)
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.L.tloam_comm_init_mailbox(self.h, int(rank), int(nranks), C.cast(buf, C.c_void_p)),
                    "tloam_comm_init_mailbox")

    def comm_init_callback(self, rank, nranks, fn):
        """fn(device_ptr:int, count:int, stream:int) -> 0 ; must sum-all-reduce `count` doubles in place."""
        def tramp(user, dev, count, stream):
            try:
                return int(fn(dev, count, stream) or 0)
            except Exception:  # never unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        self._check(self.L.tloam_comm_init_callback(self.h, int(rank), int(nranks), self._cb, None),
                    "tloam_comm_init_callback")


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = load_library().tloam_rccl_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:
        raise TloamHipError(f"tloam_rccl_unique_id: {STATUS.get(rc, rc)}")
    return buf.raw


def default_feature_config(**over) -> FeatureConfig:
    cfg = FeatureConfig()
    load_library().tloam_feature_default_config(C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def default_submap_config(**over) -> SubmapConfig:
    cfg = SubmapConfig()
    load_library().tloam_submap_default_config(C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def make_registration(method: str, cfg: TlsConfig | None = None, device: int = 0):
    """FrontEnd::initRegistraton (front_end.cpp:155-167) keyed on `local_registration_method`."""
    if method == "TLS_HIP":
        return HipRegistration(cfg, device)
    raise ValueError("Other methods are not yet supported")  # the reference's message, front_end.cpp:163
