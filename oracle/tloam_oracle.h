/*
 * tloam_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, no dependencies) of the reference's pose-optimisation path,
 * used only as the parity checker by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  Nothing under tloam_amd/ may include, link or call it.
 *
 * PARITY UNPINNED: the reference ships no test, golden vector or known-answer fixture for
 * this path and cannot be compiled here (it needs Eigen,
 * Ceres 2.0, Open3D 0.12, yaml-cpp and ROS, none of which exist in this image; writing
 * stand-ins for them is not allowed).  The arithmetic that lives in un-vendored third
 * parties is restated from their published algorithms:
 *   - Ceres Solver 2.0 (README.md:111): trust_region_minimizer.cc, dogleg_strategy.cc,
 *     corrector.cc, loss_function.cc(CauchyLoss), residual_block.cc -- see orc_ceres_solve.
 *   - Open3D 0.12.0 (README.md:83) KDTreeFlann::SearchHybrid (nanoflann): exact k-NN,
 *     ascending, cut at squared distance < radius^2 -- see orc_knn_hybrid.
 *   - Eigen3 SelfAdjointEigenSolver<Matrix3d> (registration.cpp:476-479): any solver
 *     accurate to ~1e-15 is equivalent away from the two gates; cyclic Jacobi here.
 *   - Eigen3 Quaternion(Matrix3) / toRotationMatrix (inside the Sophus ctor/matrix()).
 * The oracle is cross-checked against an INDEPENDENT numpy/scipy restatement
 * (oracle/oracle_np.py: cKDTree, numpy.linalg.eigh, numpy.linalg.solve) and both against
 * analytic known-answer cases; the committed vectors under tests/golden/ come from the
 * numpy restatement.
 */
#ifndef TLOAM_ORACLE_H
#define TLOAM_ORACLE_H

#include "../include/tloam_hip.h" /* only for the POD types tloam_tls_config / tloam_stats */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* ---- SE(3) (vendored Sophus restated) ---- */
void orc_so3_exp(const double w[3], double q_wxyz[4], double* theta);       /* so3.hpp:583-619 */
void orc_so3_log(const double q_wxyz[4], double w[3], double* theta);       /* so3.hpp:247-290 */
void orc_se3_exp(const double a[6], double q_wxyz[4], double t[3]);         /* se3.hpp:761-785 */
void orc_se3_log(const double q_wxyz[4], const double t[3], double a[6]);   /* se3.hpp:223-256 */
void orc_se3_act(const double q[4], const double t[3], const double p[3], double out[3]); /* se3.hpp:321-324 */
int orc_se3_from_matrix(const double M_colmajor[16], double q[4], double t[3]); /* se3.hpp:497-504 */
void orc_se3_to_matrix(const double q[4], const double t[3], double M_colmajor[16]);
void orc_plus(const double x[6], const double delta[6], double out[6]);     /* registration.cpp:162-173 */

/* ---- small geometry ---- */
void orc_fit_plane(const double* pts_aos, int n, double plane[4]);          /* registration.cpp:303-368 */
void orc_eig3_sym(const double cov_rowmajor[9], double evals_asc[3], double evecs_cols[9]);
/* KDTreeFlann::SearchHybrid semantics, brute force (tie-break: lower index first) */
int orc_knn_hybrid_brute(const double* tgt_aos, int n, const double q[3], double radius, int k,
                         int* idx, double* d2);

/* ---- context mirroring the C ABI of include/tloam_hip.h ---- */
int orc_create(const tloam_tls_config* cfg, orc_ctx** out);
void orc_destroy(orc_ctx* c);
void orc_set_threads(orc_ctx* c, int builder_threads, int eval_threads);
void orc_set_eval_grain(orc_ctx* c, int blocks_per_thread);   /* evaluator team = min(eval_threads, blocks / grain); default 256, 0 = all */
int orc_set_source(orc_ctx* c, int kind, const double* xyz_aos, size_t n);
int orc_set_target(orc_ctx* c, int kind, const double* xyz_aos, size_t n);
int orc_scan_match(orc_ctx* c, const double predict[16], const double* omega3, double result[16],
                   double* scan_xyz, size_t n_scan, tloam_stats* stats);
int orc_sm_begin(orc_ctx* c, const double predict[16], const double* omega3);
int orc_sm_outer(orc_ctx* c, int* done, tloam_stats* stats);
int orc_sm_end(orc_ctx* c, double result[16], tloam_stats* stats);
int orc_fitness(orc_ctx* c, double* fitness, double* rmse);
int orc_get_correspondences(orc_ctx* c, int kind, size_t capacity, size_t* n, int32_t* src_index,
                            double* a_aos, double* b_aos, double* d, double* w, double* cost);
int orc_get_weights(orc_ctx* c, int kind, size_t capacity, size_t* n, double* w);
int orc_knn(orc_ctx* c, int kind, const double* q_aos, size_t nq, double radius, int k,
            int32_t* out_idx, double* out_d2, int32_t* out_cnt);
int orc_set_correspondences(orc_ctx* c, int res_type, size_t n, const double* p, const double* a,
                            const double* b, const double* d, const double* w);
/* shard helper for the multi-rank tests: restrict the pre-built sets to [lo,hi) per type */
int orc_accumulate(orc_ctx* c, const double se3[6], double H[36], double g[6], double* cost);
int orc_get_costs(orc_ctx* c, int res_type, size_t capacity, size_t* n, double* cost);
int orc_get_normal_equations(orc_ctx* c, double H[36], double g[6], double* cost);
int orc_solve(orc_ctx* c, double se3_inout[6], tloam_stats* stats);
/* the minimiser's trace since the last orc_sm_begin / orc_solve: rows of ORC_TRACE_COLS doubles (layout at trace_push in
 * tloam_oracle.c: Solve index, iteration, cost, cost change, accepted, radius, |dx|, step quality, gradient max norm, exit kind,
 * x[6]); copies min(rows, cap_rows) rows, returns the number of rows recorded */
#define ORC_TRACE_ROWS 256
#define ORC_TRACE_COLS 16
int orc_get_trace(orc_ctx* c, double* out_rows, int cap_rows);

#ifdef __cplusplus
}
#endif
#endif
