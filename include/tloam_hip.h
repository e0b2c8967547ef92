/*
 * tloam_hip.h -- C ABI of the MI355X-native T-LOAM pose-optimisation path.
 *
 * This library replaces ONE path of the reference: LocalRegistration::scanMatching and
 * what it calls (reference: src/models/registration/registration.cpp:879-1133), i.e. the
 * four KDTreeFlann::SearchHybrid correspondence builders (:427-505, :517-559, :571-635,
 * :714-778), the three Ceres cost functors (:19-117), the SE(3) local parameterisation
 * (:162-179), ceres::Solve as configured at :1036-1047, the GNC-TLS weight update
 * (:858-876) and getFitnessScore (:257-296).  It sits behind the reference's plugin
 * boundary tloam::RegistrationInterface
 * (include/tloam/models/registration/registration_interface.hpp:40-48); the C++ adapter
 * that marshals a reference `Frame` into these calls is adapters/hip_registration.hpp and
 * the binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / Eigen / Open3D types; pointers + sizes only.
 *   - every function returns TLOAM_OK (0) or a negative tloam_status; nothing throws or
 *     aborts (the reference asserts / SOPHUS_ENSUREs instead, registration.cpp:928-929,
 *     sophus/se3.hpp:497-504).
 *   - point clouds are borrowed for the duration of the call as contiguous AoS double[3]
 *     (== open3d::geometry::PointCloud2::points_.data(), PointCloud2.hpp:396) and copied
 *     to HBM as SoA.  4x4 poses are column-major doubles (== Eigen::Isometry3d::matrix()).
 *   - one context = one device + one HIP stream; not thread-safe (the reference has a
 *     single caller thread, lidar_odometry_nodelet.cpp:57-63).
 *   - se(3) vectors are (upsilon[3], omega[3]) -- translation part first, as
 *     registration.hpp:327-329.
 *   - there is NO CPU fallback: every entry point that computes needs a gfx950 device and
 *     returns TLOAM_E_HIP if none is usable.
 */
#ifndef TLOAM_HIP_H
#define TLOAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TLOAM_ABI_VERSION 8  /* 2: tloam_stats gained gn_sweeps; submap + feature entry points
                               * 3: tloam_set_source_frame / tloam_set_target_frame; tloam_stats.host_wait_us
                               * 4: tloam_get_normal_equations; tloam_comm_mailbox_*; tloam_stats.reserved0 ->
                               *    weight_range_violations (same slot), TLOAM_E_WEIGHT_RANGE is returned
                               * 5: tloam_frame_stash / tloam_frame_select (frames staged in HBM ahead of their solve)
                               * 6: tloam_k3_span, tloam_shard_ranges_frame
                               * 7: tloam_debug_raise_fault
                               * 8: tloam_get_info, tloam_gn_iter_timer, tloam_time_read_stream; a mailbox / RCCL set-up with nranks == 1
                               *    is a loop-back (the sharded forms run); a cloud holds at most 2^28 points (was 2^29) */

/* feature kinds; order = the builder order of registration.cpp:981-992 */
#define TLOAM_KIND_PLANAR 0 /* addSurfCostFactor    -> point-to-plane  */
#define TLOAM_KIND_GROUND 1 /* addGroundCostFactor  -> point-to-plane  */
#define TLOAM_KIND_EDGE 2   /* addEdgeCostFactor    -> point-to-line   */
#define TLOAM_KIND_SPHERE 3 /* addSphereCostFactor  -> point-to-point  */
#define TLOAM_NUM_KINDS 4

/* residual types of pre-built correspondence sets (tloam_set_correspondences) */
#define TLOAM_RES_PLANE 0 /* PointToPlaneErr registration.cpp:96-117  */
#define TLOAM_RES_LINE 1  /* PointToLineErr  registration.cpp:55-88   */
#define TLOAM_RES_POINT 2 /* PointToPointErr registration.cpp:19-47   */
#define TLOAM_NUM_RES 3

typedef enum tloam_status {
  TLOAM_OK = 0,
  TLOAM_E_INVALID = -1,        /* null pointer / bad enum / bad size                        */
  TLOAM_E_TOO_FEW_POINTS = -2, /* < 10 points in one of the 8 clouds (registration.cpp:928) */
  TLOAM_E_BAD_POSE = -3,       /* predict pose is not a rigid transform (sophus se3.hpp:497), or its translation is
                                  not finite (the reference goes on and returns a NaN pose)                  */
  TLOAM_E_HIP = -4,            /* HIP runtime error / no device                             */
  TLOAM_E_RCCL = -5,           /* RCCL error / librccl not loadable                         */
  TLOAM_E_NOT_READY = -6,      /* call sequence violated (e.g. outer step before begin)     */
  TLOAM_E_WEIGHT_RANGE = -7    /* a GNC weight left [0,1]: the reference's assert at :871, live in its build
                                * (CMakeLists.txt:5-6 set no NDEBUG).  Returned by tloam_sm_outer for the
                                * iteration it happened in and by tloam_scan_match AFTER the whole solve ran:
                                * result pose and stats are written, the weights are used as computed (what
                                * an NDEBUG build of the reference would do); the caller decides           */
} tloam_status;

/* The 16 keys of the `TLS:` block, config/mapping/lidar_odometry.yaml:23-39, read by
 * LocalRegistration::initConfig (registration.cpp:212-230). */
typedef struct tloam_tls_config {
  int32_t k_corr;     /* unused on the live path (only the dead PlaneToPlane builders) */
  int32_t factor_num; /* 2: planar+ground, 3: +edge, 4: +sphere (registration.hpp:144-148) */
  double edge_dist_thres;
  double edge_dir_thres;
  int32_t edge_maxnum;
  int32_t sphere_maxnum;
  double sphere_dist_thres;
  double planar_dist_thres;
  int32_t planar_maxnum;
  int32_t ground_maxnum;
  double ground_dist_thres;
  int32_t max_iterations;
  int32_t reserved0;
  double cost_threshold;
  double gnc_factor;
  double noise_bound;
  double fitness_thres;
} tloam_tls_config;

/* Fills *cfg with the shipped values of lidar_odometry.yaml:23-39. */
void tloam_default_config(tloam_tls_config* cfg);

/* What one scan_match (or one outer GNC iteration) did. */
typedef struct tloam_stats {
  int32_t outer_iterations;       /* GNC iterations executed (<= max_iterations)              */
  int32_t gn_evaluations;         /* solver evaluations (Ceres `Evaluate` calls of the minimiser) */
  int32_t gn_iterations;          /* trust-region iterations attempted (<= 4 per outer)       */
  int32_t accepted_steps;         /* ... of which accepted                                    */
  int32_t n_corr[TLOAM_NUM_KINDS];/* factors added in the LAST outer iteration, per kind      */
  int32_t converged_early;        /* 1 if the planar-cost plateau test broke the loop (:1108) */
  int32_t weight_range_violations;/* GNC weights that left [0,1] in this scan_match (the reference asserts, :871) */
  double kind_cost[TLOAM_NUM_KINDS]; /* side-channel cost sums of the last iteration (:1091-1094) */
  double mu;                      /* GNC mu after the last update (:1089)                      */
  double solver_cost;             /* Ceres-style cost 0.5*sum(rho) at the final iterate        */
  double se3[6];                  /* final tangent vector `parameters` (registration.hpp:328)  */
  int32_t gn_sweeps;              /* residual+Jacobian sweeps actually executed (K3 launches that did
                                   * work): <= gn_evaluations -- an evaluation of a point that is bit-identical
                                   * to the one just evaluated (a rejected step retried inside a halved trust
                                   * region) is served from the totals already on the device */
  int32_t host_wait_us;        /* microseconds of this scan_match the calling thread spent waiting for the device (0 in the oracle) */
} tloam_stats;

typedef struct tloam_ctx tloam_ctx;

/* ---- lifetime ------------------------------------------------------------------------ */
/* LocalRegistration::LocalRegistration(config["TLS"]) (registration.cpp:182-206). */
int tloam_create(const tloam_tls_config* cfg, int device_id, tloam_ctx** out);
void tloam_destroy(tloam_ctx* ctx);
int tloam_abi_version(void);
const char* tloam_status_string(int status);
/* text of the last HIP/RCCL error seen by this context ("" if none) */
const char* tloam_last_error(const tloam_ctx* ctx);

/* Sizes: a cloud, a correspondence set or a query batch holds at most 2^28 points (slots, cells and ranks are 32-bit integers on
 * the device, and the four kinds of a frame share one slot space); more is TLOAM_E_INVALID at the entry point. */
/* What a context is and what has happened to it -- read by the bench (so that a multi-GPU line says what it ran on), by the
 * co-residency test and by anybody who wants to know whether a context has left its fast forms. */
#define TLOAM_FALLBACK_SCAN 1   /* a single-pass look-back scan timed out: multi-launch scans from then on            */
#define TLOAM_FALLBACK_VOXEL 2  /* the voxel down-sampling's look-back timed out: start tickets from then on          */
#define TLOAM_FALLBACK_SOLVE 4  /* the one-launch Solve's in-launch hand-over timed out: one launch per GN iteration  */
typedef struct tloam_ctx_info {
  int32_t abi_version;
  int32_t device;
  int32_t device_cus;       /* hipDeviceAttributeMultiprocessorCount: what the forms that need all their blocks resident are sized by */
  int32_t comm_mode;        /* 0 one rank, 1 caller's all-reduce, 2 RCCL, 3 peer mailbox                               */
  int32_t rank, nranks;
  int32_t rccl_comm_count;  /* ncclCommCount of the context's communicator; -1: no RCCL communicator                    */
  int32_t rccl_comm_rank;   /* ncclCommUserRank; -1 likewise                                                            */
  int32_t fallbacks_taken;  /* TLOAM_FALLBACK_* bits: bounded in-launch waits that ran out and moved the context to a form
                             * that waits for nothing (knobs of the environment are not reported here)                  */
  int32_t fallback_events;  /* how many times that happened (each one cost a ~1-2 s wait and a re-run)                   */
  int32_t k3_grid;          /* blocks of the residual/Jacobian sweep over the current correspondence set (= rows it leaves) */
  int32_t k3_single;        /* 1: one wave per chunk (KITTI-size sets), 0: the streaming form                           */
  int32_t one_launch_solve; /* 1: a ceres::Solve on the current set runs as ONE launch (k_solve_all)                    */
  int32_t loopback;         /* 1: a mailbox / RCCL set-up with nranks == 1 -- the sharded launch forms run, the exchange is
                             * a loop-back                                                                               */
  int32_t direct_set;       /* 1: the frame in progress / the last one keeps its factors as a DIRECT set (large frames whose caps
                             * cannot bind: rows in the search's own order, no compaction -- DESIGN.md section 4)              */
  int32_t set_stale;        /* 1: ... and its rows will be rebuilt before a getter reads them (the loop ended beside a search
                             * that had already run)                                                                          */
  int32_t k3_wide;          /* 1: the streaming sweep goes out as blocks of EIGHT waves, one per CU (full-chip grids: half the rows
                             * for the block that folds them), 0: four waves                                                   */
} tloam_ctx_info;
int tloam_get_info(tloam_ctx* ctx, tloam_ctx_info* out);

/* ---- inputs: RegistrationInterface::setInputSource / setInputTarget -------------------
 * (registration.cpp:232-248).  The reference keeps shared_ptrs; here the cloud is copied
 * to HBM (AoS -> SoA on device).  In a sharded context (tloam_comm_*) every rank passes
 * the FULL cloud and the context keeps its contiguous index block.
 * Non-finite coordinates (NaN, +-inf) are taken as they are, like the reference takes them (it hands every point to nanoflann,
 * whose result set never admits a NaN / inf distance): such a SOURCE point is never matched -- it keeps its index and its
 * weight of 1 --, such a TARGET point is never anybody's neighbour and does not extend the search grid.  No status is raised;
 * the result is the reference's (tests/test_gpu_parity.py::test_non_finite_points_are_never_matched). */
int tloam_set_source(tloam_ctx* ctx, int kind, const double* xyz_aos, size_t n);
int tloam_set_target(tloam_ctx* ctx, int kind, const double* xyz_aos, size_t n);
/* The same for the four clouds of a tloam::Frame at once (registration_interface.hpp:19-38; setInputSource /
 * setInputTarget take a Frame, registration.cpp:232-248), indexed by TLOAM_KIND_*.  xyz_aos[k] may be NULL when n[k] == 0.
 * tloam_set_target_frame synchronises once per frame instead of once per cloud, then enqueues the build of the four search
 * grids over the new targets without waiting for it (the next tloam_scan_match / tloam_sm_begin uses them; until then the
 * context's search structures -- what tloam_fitness sees -- remain those of the last scanMatching).  tloam_set_source_frame does not wait for the
 * device at all: it returns when the borrowed buffers have been copied OUT (into pinned staging; one copy is enqueued on the
 * context's stream behind it), so they may be reused at once and the next call on the context is ordered behind the copy.
 * SHARDED contexts: the four source clouds of a frame must reach EVERY rank through the SAME entry point -- all four through
 * tloam_set_source_frame (the frame is cut as one line, tloam_shard_ranges_frame: a rank holds one or two kinds and builds only
 * those kinds' search grids) or each through tloam_set_source (every cloud cut by itself, tloam_shard_range) -- never a mix
 * within a frame, and the same choice on all ranks: the two rules give a rank different index blocks, and a mix leaves source
 * points unowned or owned twice.  A kind a rank holds no source points of has no search structure on that rank
 * (tloam_fitness skips it there). */
int tloam_set_source_frame(tloam_ctx* ctx, const double* const xyz_aos[4], const size_t n[4]);
int tloam_set_target_frame(tloam_ctx* ctx, const double* const xyz_aos[4], const size_t n[4]);
/* Frames staged ahead of their solve.  The reference's caller hands a Frame over and solves it at once (front_end.cpp:314,
 * :321); a caller that receives scans while the previous solve is still running -- or a replay / benchmark that wants its
 * frames resident in HBM before the clock starts -- can hand frames over early and activate them later at no cost:
 *   tloam_frame_stash(ctx, slot)   moves the clouds currently registered with the context (whatever tloam_set_source* /
 *                                  tloam_set_target* left there: eight clouds, their bounds) into slot `slot` (>= 0) of a frame
 *                                  store kept in HBM; the context is left without registered clouds.  A slot that was in
 *                                  use is overwritten.  Stashing the slot that is currently selected keeps the frame in
 *                                  that slot (including what tloam_set_* wrote since) and makes the context's own clouds
 *                                  the registered ones again (= select -1); with ANOTHER slot selected: TLOAM_E_NOT_READY.
 *   tloam_frame_select(ctx, slot)  makes the clouds of `slot` the registered ones (buffers are exchanged, nothing is copied or
 *                                  synchronised); slot -1 = back to the context's own.  While a slot is selected,
 *                                  tloam_set_source* / tloam_set_target* write into that slot's frame.
 * TLOAM_E_NOT_READY between tloam_sm_begin and tloam_sm_end, TLOAM_E_INVALID for an unknown slot. */
int tloam_frame_stash(tloam_ctx* ctx, int slot);
int tloam_frame_select(tloam_ctx* ctx, int slot);

/* ---- RegistrationInterface::scanMatching (registration.cpp:879-1133) -------------------
 * predict/result: 4x4 column-major.  omega_perturb3: the unit vector the reference draws
 * with Eigen::Vector3d::Random() when |omega| < 1e-2 (:884-886); NULL = (0,0,1).
 * scan_xyz_aos/n_scan: optional cloud transformed IN PLACE by the result pose
 * (out_result_.scan_cloud, :1126-1128); NULL/0 to skip.  stats may be NULL. */
int tloam_scan_match(tloam_ctx* ctx, const double predict_colmajor[16],
                     const double* omega_perturb3_or_null, double result_colmajor[16],
                     double* scan_xyz_aos_or_null, size_t n_scan, tloam_stats* stats);

/* The same, one outer GNC iteration at a time (tests inspect the state in between).
 * begin -> outer x N (until *done) -> end. */
int tloam_sm_begin(tloam_ctx* ctx, const double predict_colmajor[16],
                   const double* omega_perturb3_or_null);
int tloam_sm_outer(tloam_ctx* ctx, int* done, tloam_stats* stats);
int tloam_sm_end(tloam_ctx* ctx, double result_colmajor[16], tloam_stats* stats);

/* ---- RegistrationInterface::getFitnessScore (registration.cpp:257-296) ----------------- */
int tloam_fitness(tloam_ctx* ctx, double* fitness, double* rmse);

/* ---- introspection (parity tests) ------------------------------------------------------
 * Correspondences built by the last outer iteration for `kind`, in source-index order.
 * src_index[n]; a[3n] = plane normal | line point a | target point; b[3n] = line point b
 * (edge only; else untouched); d[n] = plane offset (planar/ground only); w[n] = weight
 * captured at build time; cost[n] = side-channel cost after the last sweep.  Any output
 * pointer may be NULL.  capacity = elements available; returns count via *n. */
int tloam_get_correspondences(tloam_ctx* ctx, int kind, size_t capacity, size_t* n,
                              int32_t* src_index, double* a_aos, double* b_aos, double* d,
                              double* w, double* cost);
/* current per-source-point GNC weights of `kind` (this rank's block if sharded) */
int tloam_get_weights(tloam_ctx* ctx, int kind, size_t capacity, size_t* n, double* w);
/* exact hybrid search on the device structure of `kind` (KDTreeFlann::SearchHybrid, B.2):
 * for each of nq queries (AoS) the k nearest targets with squared distance < radius^2,
 * ascending; out_idx[nq*k] (-1 padded), out_d2[nq*k], out_cnt[nq]. */
int tloam_knn(tloam_ctx* ctx, int kind, const double* queries_aos, size_t nq, double radius,
              int k, int32_t* out_idx, double* out_d2, int32_t* out_cnt);

/* ---- pre-built correspondence sets (roofline / parity of K3 and the solver) ------------
 * res_type TLOAM_RES_*: p = source point (scan frame); a = normal | line a | target;
 * b = line b (LINE only, else NULL); d = plane offset (PLANE only, else NULL); w = weights.
 * Replaces whatever the builders produced.  Sharded contexts keep their index block. */
int tloam_set_correspondences(tloam_ctx* ctx, int res_type, size_t n, const double* p_aos,
                              const double* a_aos, const double* b_aos, const double* d,
                              const double* w);
/* One residual+Jacobian sweep at se3 (K3): H = sum rho' J^T J (row-major 6x6),
 * g = sum rho' J^T r, cost = sum 0.5*log(1+|r|^2)  (Ceres evaluator + CauchyLoss(1.0)
 * corrector, registration.cpp:970).  Also refreshes the side-channel costs.  In a sharded
 * context the outputs are the all-reduced totals. */
int tloam_accumulate(tloam_ctx* ctx, const double se3[6], double H_rowmajor[36], double g[6],
                     double* cost);
/* Parity probe of the per-outer-iteration linear system: the robustified normal equations the minimiser
 * held when its last Solve returned -- H = sum rho' J^T J (row-major 6x6), g = sum rho' J^T r, and the cost
 * 0.5 sum rho, all at the ACCEPTED iterate (tloam_stats.se3).  In a sharded context: the all-reduced totals. */
int tloam_get_normal_equations(tloam_ctx* ctx, double H_rowmajor[36], double g[6], double* cost);
/* side-channel costs of the current set, per residual type (after tloam_accumulate/solve) */
int tloam_get_costs(tloam_ctx* ctx, int res_type, size_t capacity, size_t* n, double* cost);
/* One ceres::Solve as configured at registration.cpp:1036-1047 on the current set:
 * se3_inout is `parameters`.  stats->gn_* filled. */
int tloam_solve(tloam_ctx* ctx, double se3_inout[6], tloam_stats* stats);
/* Timing helper for the bench: `launches` back-to-back K3 sweeps at se3 on the context's
 * stream bracketed by HIP events; returns the mean kernel-pair time in microseconds. */
int tloam_time_accumulate(tloam_ctx* ctx, const double se3[6], int launches, double* mean_us);
/* Timing helper for the bench: `launches` back-to-back runs of the correspondence-search kernel (K1 + K2:
 * SearchHybrid + the four builders, registration.cpp:427-635, :714-778) over the source points of the last
 * scan_match -- same pose, grids and query order -- between one HIP event pair; *queries = points searched. */
int tloam_time_build(tloam_ctx* ctx, int launches, double* mean_us, int64_t* queries);
/* Sharded contexts, collective (every rank calls it with the same arguments): `launches` sweeps over this rank's
 * block of the current set, each followed -- with_exchange != 0 -- by the exchange of the 48 doubles exactly as a GN
 * iteration performs it (mailbox: posted by the sweep's last block and gathered by a one-wave kernel; RCCL / callback:
 * all-reduce), one HIP event pair around the lot.  The difference with / without is the latency the exchange adds. */
int tloam_time_sharded_sweep(tloam_ctx* ctx, const double se3[6], int launches, int with_exchange, double* mean_us);
/* accumulated HIP-event time (us) and launch count of the K3 sweeps since the last reset; the first
 * call arms the timer: each K3 dispatch then carries a HIP start/stop event pair bound to the
 * dispatch packet (hipExtLaunchKernelGGL), i.e. the elapsed time is the kernel duration itself */
int tloam_k3_timer(tloam_ctx* ctx, int reset, double* total_us, int64_t* launches,
                   double* algorithmic_bytes);
/* the same over EVERY K3 launch, including the no-op launches enqueued after a solver tolerance exit
 * (the population a kernel trace averages over) */
int tloam_k3_timer_all(tloam_ctx* ctx, double* total_us, int64_t* launches);
/* One-launch GN iterations of large sets (k3_sweep_step: sweep + row fold + minimiser step in one dispatch): the STREAMING
 * span of those launches -- first wave in to last block row out, by the device's 100 MHz wall clock, without the serial
 * tail -- accumulated on the device since the last reset; synchronises the stream.  launches counts working sweeps only
 * (a launch that finds the Solve finished returns before the span is taken). */
int tloam_k3_span(tloam_ctx* ctx, int reset, double* total_us, int64_t* launches);
/* The period of a GN iteration (sweep + reduction + exchange + 6x6 step + pose update) as the DEVICE clocks it:
 * every kernel that ends an iteration stamps the 100 MHz wall clock when its step is done, and the time between two consecutive
 * stamps of ONE Solve is added up -- launch boundaries, fold, exchange and step included; a Solve's first iteration has no stamp
 * to start from and is not counted.  The first call arms the counter (until then the kernels skip it); synchronises the stream. */
int tloam_gn_iter_timer(tloam_ctx* ctx, int reset, double* total_us, int64_t* iterations);
/* On-box bandwidth of a READ stream with the sweep's access pattern (eight fp64 streams, 16-byte loads, persistent waves, two
 * blocks per CU): `launches` passes over ~`bytes`, one HIP event pair; *gbps = bytes read / s / 1e9.  bytes >> 256 MiB: from HBM;
 * bytes = a sweep's 75 MB: from the Infinity Cache, as the sweeps of a Solve.  Allocates and frees its own buffer. */
int tloam_time_read_stream(tloam_ctx* ctx, size_t bytes, int launches, double* gbps);
/* Test aid: the DEVICE SE(3) arithmetic the minimiser step uses (vendored-Sophus restatements sophus/so3.hpp:583-619,
 * se3.hpp:761-785 exp; so3.hpp:247-290, se3.hpp:223-256 log; registration.cpp:162-173 Plus), n items.  out26 per item:
 * [0..6] exp(delta) as (qw qx qy qz tx ty tz), [7..12] log(exp(x)), [13..18] Plus(x, delta), [19..25] exp(x) (shared form). */
int tloam_debug_se3(tloam_ctx* ctx, int n, const double* x6, const double* delta6, double* out26);
/* debugging aid: raw copy of the device-resident minimiser state; returns its size in doubles */
int tloam_debug_state(tloam_ctx* ctx, double* out, int n_doubles);
/* development aid: the per-block partial rows of the last K3 launch (32 doubles per block); returns the
 * number of rows.  Columns 28..31 carry in-kernel timestamps in builds with -DTLOAM_K3_PROFILE. */
int tloam_debug_partials(tloam_ctx* ctx, double* out, int n_doubles);
/* Test hook.  A few kernels spin on blocks of their own launch (the single-pass look-back scans of 1 M-class tables, the
 * voxel down-sampling's look-back); the host only picks those forms where the device's CU count says every block is
 * resident at once, their waits are bounded (~1 s) all the same, and a wait that runs out raises a word in pinned host
 * memory: the call in progress returns TLOAM_E_HIP -- tloam_scan_match runs the frame again by itself -- and the context
 * uses the forms that wait for nothing (multi-launch scans / start tickets) from then on.  This raises the word by hand:
 * which = 0 look-back scan, 1 voxel down-sampling. */
int tloam_debug_raise_fault(tloam_ctx* ctx, int which);

/* ---- submap maintenance on the device ---------------------------------------
 * FrontEnd::updateSubmap (front_end.cpp:201-275) and the first-frame branch of updateLidarOdometry
 * (front_end.cpp:283-304) restated on the device, so that the four target clouds never leave HBM between
 * frames: Transform (PointCloud2.cpp:71-75) -> += (:96-132) -> Crop (:551-559) -> VoxelDownSample (:358-403).
 * The result is installed as the registration target (the reference's setInputTarget(submap),
 * front_end.cpp:267) without a host round trip.  Quirk kept: the sphere submap is rebuilt from the PLANAR
 * frame buffer (front_end.cpp:221 iterates submap_planar_buffer).  Voxels are emitted in order of first
 * occurrence (the reference's order is std::unordered_map's, i.e. unspecified). */
typedef struct tloam_submap_config {
  int32_t planar_frame_size;        /* lidar_odometry.yaml:13  (3)   */
  int32_t sphere_frame_size;        /* lidar_odometry.yaml:12  (3)   */
  double edge_crop_box_length;      /* lidar_odometry.yaml:16  (100) */
  double ground_crop_box_length;    /* lidar_odometry.yaml:17  (100) */
  double edge_down_sample_submap;   /* lidar_odometry.yaml:9   (0.3) */
  double ground_down_sample_submap; /* lidar_odometry.yaml:7   (0.45)*/
  double ground_down_sample;        /* lidar_odometry.yaml:6   (0.3), first frame only (front_end.cpp:287) */
} tloam_submap_config;
void tloam_submap_default_config(tloam_submap_config* cfg);
/* first frame (front_end.cpp:283-304): edge += edge cloud; ground += ground cloud->VoxelDownSample(
 * ground_down_sample); planar / sphere += the submap selections; setInputTarget(submap). */
int tloam_submap_init(tloam_ctx* ctx, const tloam_submap_config* cfg, const double* planar_submap_xyz, size_t n_planar,
                      const double* sphere_submap_xyz, size_t n_sphere, const double* edge_xyz, size_t n_edge,
                      const double* ground_xyz, size_t n_ground);
/* every later frame (front_end.cpp:201-275) with lidar_odom_pose = pose_colmajor: planar/sphere frame
 * buffers, edge/ground accumulate -> crop around the pose's translation -> voxel grid; setInputTarget.
 * The pose is taken as Open3D's Transform takes a 4x4 (no orthogonality test on this path: a scaled rotation or a projective
 * last row is applied as it stands); a NaN or an infinity anywhere in it is TLOAM_E_BAD_POSE and leaves the submap as it was
 * (the reference would go on with NaN clouds).  TLOAM_E_NOT_READY before tloam_submap_init. */
int tloam_submap_update(tloam_ctx* ctx, const double pose_colmajor[16], const double* planar_submap_xyz,
                        size_t n_planar, const double* sphere_submap_xyz, size_t n_sphere,
                        const double* edge_scan_xyz, size_t n_edge, const double* ground_scan_xyz, size_t n_ground);
/* the target cloud of `kind` as the device holds it (AoS out); n receives the size even when capacity is
 * too small (then nothing is copied and TLOAM_E_INVALID is returned) */
int tloam_get_target(tloam_ctx* ctx, int kind, size_t capacity, size_t* n, double* xyz_aos);


/* ---- PCA feature extraction on the device ------------------------------------
 * featureExtract::calculatePCAInfo (src/models/feature_extraction/feature_extract.cpp:47-122) and
 * featureExtract::extractPlanarSphere (:133-197): hybrid k-NN (r, K) of every point in its own cloud,
 * 3x3 covariance from nine cumulants, ascending eigen decomposition, cvr / flatness / sphericity, then the
 * planar / sphere candidate lists ranked by flatness (descending; ties by ascending index -- the reference's
 * std::sort is unstable there).  Quirks kept: the sphere lists are ranked by FLATNESS (:162) and hold sort
 * RANKS, not point indices (:186,:188).  K <= 20. */
typedef struct tloam_feature_config {
  double radius;               /* feature.yaml: radius 0.2 */
  int32_t K;                   /* 20 */
  int32_t min_neigh;           /* 10 */
  int32_t planar_num;          /* 500 */
  int32_t sphere_num;          /* 300 */
  double cvr_scan;             /* 0.25 */
  double cvr_submap;           /* 0.15 */
  double planar_scan_thres;    /* 0.75 */
  double planar_submap_thres;  /* 0.65 */
  double planar_vertic_thres;  /* 0.25 */
} tloam_feature_config;
void tloam_feature_default_config(tloam_feature_config* cfg);
/* per-point PCAInfo (feature_extract.hpp:33-39); any output pointer may be NULL.  Points that are skipped
 * (no more than min_neigh neighbours) keep the value-initialised zeros of the reference; neigh_index is
 * n x K, padded with -1.  radius >= 0 and 3 <= K <= 20, else TLOAM_E_INVALID (assert(r_ >= 0.0 && K_ >= 3),
 * feature_extract.cpp:55): a radius of exactly 0 finds nobody, an infinite one is plain k-NN; a cloud without a single finite
 * point has no search structure and every point keeps the zeros. */
int tloam_pca_info(tloam_ctx* ctx, const tloam_feature_config* cfg, const double* xyz_aos, size_t n,
                   double* flatness, double* cvr, double* sphericity, double* normal_aos, int32_t* num_sum,
                   int32_t* neigh_index);
/* the four index lists of extractPlanarSphere; each output array must hold n entries */
int tloam_extract_planar_sphere(tloam_ctx* ctx, const tloam_feature_config* cfg, const double* xyz_aos, size_t n,
                                int32_t* planar_scan_index, size_t* n_planar_scan, int32_t* planar_submap_index,
                                size_t* n_planar_submap, int32_t* sphere_scan_index, size_t* n_sphere_scan,
                                int32_t* sphere_submap_index, size_t* n_sphere_submap);

/* ---- multi-GPU: correspondence set sharded over ranks, one all-reduce per sweep --------
 * (nothing in the reference).  Call before set_source / set_correspondences.
 * (a) native RCCL over xGMI: unique_id = the 128 bytes of an ncclUniqueId made on rank 0
 *     by tloam_rccl_unique_id and broadcast by the launcher. */
int tloam_rccl_unique_id(void* out128);
int tloam_comm_init_rccl(tloam_ctx* ctx, int rank, int nranks, const void* unique_id128);
/* nranks == 1 in (a) and (c): the context exchanges with itself -- every launch of the sharded forms runs (fused sweep + post,
 * gather + step, the side exchanges of the caps and the cost sums), the all-reduce / the mailbox is a loop-back, and the results
 * are those of the single-rank forms bit for bit.  It is how the sharded forms are timed at shard size on ONE GPU and how a
 * one-rank RCCL communicator gets to carry the all-reduce. */
/* (b) caller-provided sum all-reduce on a DEVICE buffer of `count` doubles, enqueued on (or
 *     synchronised with) `hip_stream`; returns 0 on success.  Used by the gloo-backed tests
 *     and by hosts that already own a communicator. */
typedef int (*tloam_allreduce_fn)(void* user, double* device_buf, int count, void* hip_stream);
int tloam_comm_init_callback(tloam_ctx* ctx, int rank, int nranks, tloam_allreduce_fn fn,
                             void* user);
/* (c) one-shot peer exchange ("mailbox") over xGMI, no collective library on the data path.  Every rank calls
 *     tloam_comm_mailbox_export (64 bytes = a hipIpcMemHandle_t of its small fine-grained buffer), the launcher
 *     all-gathers the handles (rank order), every rank calls tloam_comm_init_mailbox with all of them.  A sharded
 *     GN iteration is then TWO launches: the sweep, whose last block stores the 48 doubles into every rank's
 *     buffer, and the step, which adds the ranks' rows in rank order (bit-identical on all ranks).  A peer that
 *     never posts makes the waiting kernel give up after ~2 s: TLOAM_E_RCCL from the scan_match in progress.
 *     One process per rank (HIP IPC does not open a handle in the process that made it). */
int tloam_comm_mailbox_export(tloam_ctx* ctx, void* handle64_out);
int tloam_comm_init_mailbox(tloam_ctx* ctx, int rank, int nranks, const void* handles64_by_rank);
/* contiguous index block [*lo,*hi) of n items owned by `rank` of `nranks` (pure function) */
void tloam_shard_range(size_t n, int rank, int nranks, size_t* lo, size_t* hi);
/* The blocks of a whole Frame (what tloam_set_source_frame keeps in a sharded context): the four clouds laid end to end, the
 * line cut into nranks equal pieces.  Per kind still contiguous index blocks in rank order; every rank the same number of
 * source points; a rank touches one or two kinds and builds only those kinds' search grids. */
void tloam_shard_ranges_frame(const size_t n[4], int rank, int nranks, size_t lo[4], size_t hi[4]);

/* ---- SE(3) helpers (host; the ~150 lines of vendored Sophus the path uses) -------------
 * se3.hpp:761-785 (exp), :223-256 (log), :497-504 (from matrix), registration.cpp:162-173 */
int tloam_se3_exp(const double se3[6], double T_colmajor[16]);
int tloam_se3_log(const double T_colmajor[16], double se3[6]);
int tloam_se3_plus(const double x[6], const double delta[6], double x_plus_delta[6]);

#ifdef __cplusplus
}
#endif
#endif /* TLOAM_HIP_H */
