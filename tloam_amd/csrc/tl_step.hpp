// tl_step.hpp -- K5: the Ceres-configured trust-region / dogleg step on the 6x6 system (device side of
// ceres::Solve, registration.cpp:1036-1047) and PoseSE3Parameterization::Plus (:162-173).
// Included by tl_gn.hip after the K3 helpers (fast_rcp, fast_rsqrt, ut).
//
// ONE WAVE, WAVE-UNIFORM: every lane carries the same values and executes the same instruction stream, so the Cholesky
// factorisation, the solves and the quadratic forms are plain register arithmetic with no cross-lane exchange on the
// critical path.  A wave64 VALU instruction occupies the 16-lane SIMD for four cycles whatever the lanes hold, and a
// GN iteration of a KITTI-size frame is ONE launch whose tail is this function -- its cost is the NUMBER of vector
// instructions it executes.  Round 2's form executed ~2.7 k of them (~8 us): two thirds were not arithmetic but the
// price of ~120 live doubles carried through a loop nest (accumulator-register shuttling, phi copies, selects, lane
// reads).  This form is built to execute as few as possible:
//   * the LDS copy of the state (`sm`) is the single home of everything vector-valued; registers hold scalars and
//     the values of the block being computed.  State changes are LDS stores of lane 0 (one exec-mask region per
//     block); the device state is updated at the very end by ONE coalesced copy of the LDS image (3 store
//     instructions instead of ~90 scattered lane-0 stores);
//   * the retry of a rejected step inside the halved trust region (four times per Solve from the second
//     outer iteration on) re-creates, input for input, the candidate the state already holds.  Whether the state's
//     candidate is the Gauss-Newton step of the current dogleg data is kept as a flag (`cand_gn`) with the step's
//     norm (`gn_norm`), so that retry is scalar bookkeeping -- no 6x6 work, no exp/log, no bitwise comparison;
//   * the dogleg data are kept in the form the hot path needs: D^2 (the clamped diagonal), the scaled-space
//     Gauss-Newton step -y and its D-norm; sqrt(D), the gradient D^-1 gs and the subspace model are formed only when a
//     step actually leaves the trust region (rare);
//   * the two SE(3) "Plus" evaluations a step needs run in lockstep (lane 0 the candidate, lane 1 Ceres'
//     projected-gradient point) and their results leave the lanes through LDS stores / one lane read;
//   * the IEEE divisions / square roots of the chain are v_rcp_f64 / v_rsq_f64 + one Newton step (<= 1-2 ulp); every
//     DECISION (step quality, tolerances) keeps the exact division.
#pragma once

#include "tl_common.hpp"

namespace tl {

struct Vec2 { double x, y; };
// minimum of 0.5 x^T B x + g^T x on |x| = r (dogleg_strategy.cc FindMinimumOnTrustRegionBoundary;
// Ceres roots a quartic -- the global minimiser is unique, here bracketed by sampling the angle
// and polished by bisection on the tangential derivative).  Rare branch: only when the
// Gauss-Newton step leaves the trust region.
__device__ __forceinline__ Vec2 min_on_circle(double B0, double B1, double B2, double B3, double g0, double g1, double r) {
  const int NS = 720;
  double best = 1e300, bth = 0.0;
  const double b01 = 0.5 * (B1 + B2);
  for (int i = 0; i < NS; ++i) {
    const double th = 2.0 * kPi * i / NS;
    const double cx = r * cos(th), sx = r * sin(th);
    const double f = 0.5 * (B0 * cx * cx + 2.0 * b01 * cx * sx + B3 * sx * sx) + g0 * cx + g1 * sx;
    if (f < best) { best = f; bth = th; }
  }
  double lo = bth - 2.0 * kPi / NS, hi = bth + 2.0 * kPi / NS;
  for (int it = 0; it < 200; ++it) {
    const double th = 0.5 * (lo + hi);
    const double cx = r * cos(th), sx = r * sin(th);
    const double gx = B0 * cx + b01 * sx + g0;
    const double gy = b01 * cx + B3 * sx + g1;
    const double df = gx * (-sx) + gy * cx;
    if (df > 0.0) hi = th; else lo = th;
    if (hi - lo < 1e-16 * (1.0 + fabs(th))) break;
  }
  const double th = 0.5 * (lo + hi);
  return Vec2{r * cos(th), r * sin(th)};
}

#ifdef TLOAM_STEP_PROFILE
#define TL_STAMP(i) if (lane == 0) st->dbg[i] = (double)__builtin_readcyclecounter();
#else
#define TL_STAMP(i)
#endif

__device__ __forceinline__ double fsqrt(double x) {  // x >= 0
  const double r = fast_rsqrt(x);
  return x > 0.0 ? x * r : x;
}
__device__ __forceinline__ double dot6(const double a[6], const double b[6]) {   // three independent two-term chains
  return (__builtin_fma(a[1], b[1], a[0] * b[0]) + __builtin_fma(a[3], b[3], a[2] * b[2])) + __builtin_fma(a[5], b[5], a[4] * b[4]);
}
// cross / rotate / quaternion product with every multiply-add spelled out (this unit is compiled with -ffp-contract=off)
__device__ __forceinline__ Vec3 cross_f(Vec3 a, Vec3 b) {
  return {__builtin_fma(a.y, b.z, -(a.z * b.y)), __builtin_fma(a.z, b.x, -(a.x * b.z)), __builtin_fma(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ Vec3 rotate_f(const Pose& T, Vec3 p) {   // so3.hpp:358-367: p + w uv + v x uv, uv = 2 (v x p)
  const Vec3 v{T.qx, T.qy, T.qz};
  Vec3 uv = cross_f(v, p);
  uv = uv + uv;
  const Vec3 c2 = cross_f(v, uv);
  return {__builtin_fma(T.qw, uv.x, p.x) + c2.x, __builtin_fma(T.qw, uv.y, p.y) + c2.y, __builtin_fma(T.qw, uv.z, p.z) + c2.z};
}
__device__ __forceinline__ Vec3 axpy2(Vec3 u, double c1, Vec3 w1, double c2, Vec3 w2) {   // u + c1 w1 + c2 w2
  return {__builtin_fma(c2, w2.x, __builtin_fma(c1, w1.x, u.x)), __builtin_fma(c2, w2.y, __builtin_fma(c1, w1.y, u.y)),
          __builtin_fma(c2, w2.z, __builtin_fma(c1, w1.z, u.z))};
}
// a^T M b with M symmetric, stored as its upper triangle (ut)
__device__ __forceinline__ double quad6(const double a[6], const double M[21], const double b[6]) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double row = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) row = __builtin_fma(M[i <= j ? ut(i, j) : ut(j, i)], b[j], row);
    acc = __builtin_fma(a[i], row, acc);
  }
  return acc;
}
// (A) y = b, A symmetric positive definite (upper triangle in Au, destroyed).  false: a pivot is not positive /
// finite or the result is not finite (Ceres: LINEAR_SOLVER_FAILURE).
__device__ __forceinline__ bool chol6_uniform(double Au[21], const double b[6], double y[6]) {
  bool ok = true;
  double inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double pivot = Au[ut(k, k)];
    if (!(pivot > 0.0) || !isfinite(pivot)) ok = false;
    inv[k] = fast_rsqrt(pivot);                      // 1 / l_kk
#pragma unroll
    for (int j = k + 1; j < 6; ++j) Au[ut(k, j)] *= inv[k];   // row k of L^T
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) Au[ut(i, j)] = __builtin_fma(-Au[ut(k, i)], Au[ut(k, j)], Au[ut(i, j)]);
  }
  double z[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) z[k] = b[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) {   // L z = b
    z[k] *= inv[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) z[i] = __builtin_fma(-Au[ut(k, i)], z[k], z[i]);
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {  // L^T y = z
    double t = z[k];
#pragma unroll
    for (int j = k + 1; j < 6; ++j) t = __builtin_fma(-Au[ut(k, j)], y[j], t);
    y[k] = t * inv[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (!isfinite(y[k])) ok = false;
  return ok;
}

// exp / product / log with the reciprocals and square roots of the chain on v_rcp / v_rsq (+ Newton)
// (sophus so3.hpp:583-619 + se3.hpp:761-785; so3.hpp:325-340 + se3.hpp:304-309; so3.hpp:247-290 + se3.hpp:223-256)
__device__ __forceinline__ Pose exp_fast2(const double a[6]) {
  const double ox = a[3], oy = a[4], oz = a[5];
  const double theta_sq = __builtin_fma(oz, oz, __builtin_fma(oy, oy, ox * ox));
  Pose T;
  const Vec3 om{ox, oy, oz}, u{a[0], a[1], a[2]};
  if (theta_sq < kSophusEps * kSophusEps) {
    const double theta_po4 = theta_sq * theta_sq;
    const double imag = __builtin_fma(1.0 / 3840.0, theta_po4, __builtin_fma(-(1.0 / 48.0), theta_sq, 0.5));
    T.qw = __builtin_fma(1.0 / 384.0, theta_po4, __builtin_fma(-(1.0 / 8.0), theta_sq, 1.0));
    T.qx = imag * ox; T.qy = imag * oy; T.qz = imag * oz;
    const Vec3 t = rotate_f(T, u);
    T.tx = t.x; T.ty = t.y; T.tz = t.z;
    return T;
  }
  const double inv_theta = fast_rsqrt(theta_sq);
  const double theta = theta_sq * inv_theta;
  double sh, ch;
  sincos(0.5 * theta, &sh, &ch);
  const double imag = sh * inv_theta;
  T.qw = ch; T.qx = imag * ox; T.qy = imag * oy; T.qz = imag * oz;
  Vec3 t;
  if (theta < kSophusEps) {
    t = rotate_f(T, u);
  } else {
    const double c1 = 2.0 * imag * imag;                                            // (1 - cos t) / t^2
    const double c2 = __builtin_fma(-2.0 * sh, ch, theta) * inv_theta * inv_theta * inv_theta;  // (t - sin t) / t^3
    const Vec3 w1 = cross_f(om, u);
    const Vec3 w2 = cross_f(om, w1);
    t = axpy2(u, c1, w1, c2, w2);
  }
  T.tx = t.x; T.ty = t.y; T.tz = t.z;
  return T;
}
__device__ __forceinline__ Pose compose_fast2(const Pose& A, const Pose& B) {
  Pose C;
  C.qw = __builtin_fma(-A.qz, B.qz, __builtin_fma(-A.qy, B.qy, __builtin_fma(-A.qx, B.qx, A.qw * B.qw)));
  C.qx = __builtin_fma(-A.qz, B.qy, __builtin_fma(A.qy, B.qz, __builtin_fma(A.qx, B.qw, A.qw * B.qx)));
  C.qy = __builtin_fma(-A.qx, B.qz, __builtin_fma(A.qz, B.qx, __builtin_fma(A.qy, B.qw, A.qw * B.qy)));
  C.qz = __builtin_fma(-A.qy, B.qx, __builtin_fma(A.qx, B.qy, __builtin_fma(A.qz, B.qw, A.qw * B.qz)));
  const double il = fast_rsqrt(__builtin_fma(C.qz, C.qz, __builtin_fma(C.qy, C.qy, __builtin_fma(C.qx, C.qx, C.qw * C.qw))));
  C.qw *= il; C.qx *= il; C.qy *= il; C.qz *= il;
  const Vec3 rt = rotate_f(A, Vec3{B.tx, B.ty, B.tz});
  C.tx = A.tx + rt.x; C.ty = A.ty + rt.y; C.tz = A.tz + rt.z;
  return C;
}
__device__ __forceinline__ void log_fast2(const Pose& T, double a[6]) {
  const double squared_n = __builtin_fma(T.qz, T.qz, __builtin_fma(T.qy, T.qy, T.qx * T.qx));
  const double w = T.qw;
  double f, theta, c2;
  if (squared_n < kSophusEps * kSophusEps) {
    const double iw = fast_rcp(w);
    f = __builtin_fma(-(2.0 / 3.0) * squared_n, iw * iw * iw, 2.0 * iw);
    theta = 2.0 * squared_n * iw;
    c2 = 1.0 / 12.0;
  } else {
    const double in_ = fast_rsqrt(squared_n);   // 1 / n
    const double n = squared_n * in_;
    if (fabs(w) < kSophusEps) {
      f = (w > 0.0) ? kPi * in_ : -kPi * in_;
      theta = f * n;
      c2 = fast_rcp(theta * theta);  // cos(theta/2) -> 0
    } else {
      f = 2.0 * atan(n * fast_rcp(w)) * in_;
      theta = f * n;
      c2 = (fabs(theta) < kSophusEps) ? 1.0 / 12.0 : __builtin_fma(-0.5 * theta * w, in_, 1.0) * fast_rcp(theta * theta);
    }
  }
  const Vec3 om{f * T.qx, f * T.qy, f * T.qz};
  const Vec3 t{T.tx, T.ty, T.tz};
  const Vec3 w1 = cross_f(om, t);
  const Vec3 w2 = cross_f(om, w1);
  const Vec3 ups = axpy2(t, -0.5, w1, c2, w2);
  a[0] = ups.x; a[1] = ups.y; a[2] = ups.z;
  a[3] = om.x;  a[4] = om.y;  a[5] = om.z;
}

__device__ __forceinline__ double rdlane(double v, int lane_const) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_const);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_const);
  return __hiloint2double(hi, lo);
}

// words of GnState the step owns: everything in front of the outer loop's device-side control
constexpr int kStepWords = (int)(offsetof(GnState, x_build) / 8);
static_assert(offsetof(GnState, x_build) % 8 == 0, "the minimiser part of GnState is a whole number of words");

// ================================================================================================
//  Consume one reduced sweep (tot: H upper triangle 0..20, g 21..26, cost 27 -- in LDS) and run the minimiser until
//  the next sweep is needed or it is done.  Mirrors trust_region_minimizer.cc Minimize(): IterationZero, then per
//  iteration ComputeTrustRegionStep -> candidate -> tolerances -> IsStepSuccessful -> Handle(Un)SuccessfulStep, with
//  DoglegStrategy (SUBSPACE_DOGLEG) inlined.  The candidate sweep is fused (cost + Jacobian in one pass): Ceres
//  evaluates the candidate cost-only and re-evaluates an accepted point with Jacobians -- same numbers, half the
//  traffic.  The gradient-tolerance test of a freshly accepted point is evaluated together with the next candidate
//  and, if it fires, the speculative iteration is rolled back.
//  Evaluation reuse: when the minimiser asks for the evaluation of the point whose totals are in `tot` again -- a
//  rejected Gauss-Newton step retried inside the halved trust region is the same step -- the answer is already here:
//  count the evaluation and go round again instead of waiting for another sweep.  Residuals, Jacobians and
//  side-channel costs are pure functions of the point, so nothing observable changes (GnState::no_eval_reuse
//  switches it off: then the same candidate is swept again).
//  `sm`: the state as of the start of the launch, in LDS; `scr`: >= 32 doubles of LDS scratch.  One wave.
// ================================================================================================
struct StepVars {  // the wave-uniform scalars of the minimiser; everything vector-valued stays in the LDS image
  int phase, iteration, invalid, step_successful, reuse, subspace_1d, cand_gn, done, iters;
  double gmax, mcc, radius, mu, step_norm, gn_norm;
  bool need_gmax, take;
};
enum StepResult : int { SR_CANDIDATE = 0, SR_DONE = 1, SR_RETRY = 2 };
constexpr double kGradientTolerance = 1e-10;

// One pass of ComputeTrustRegionStep + candidate: the step from the (H, g) of the accepted point -- fresh dogleg data or
// the data of the rejected step reused --, its model cost change, Plus.  SR_RETRY: the step was invalid
// (HandleInvalidStep: mu *= 10, nothing reused) and the minimiser's loop goes round again.
__device__ __forceinline__ StepResult step_compute(StepVars& v, GnState* st, const double* tot, int lane, GnState* sm, double* scr) {
  constexpr int max_consecutive_invalid = 5;
  // (H, g) of the accepted point, packed: the totals of this sweep, or -- a step recomputed on the system of an
  // earlier sweep: rare -- gathered from the state
  const double* hp = tot;
  if (!v.take) {
    const int mi = lane < 36 ? lane / 6 : 0, mj = lane < 36 ? lane - mi * 6 : 0;
    if (lane < 36 && mi <= mj) scr[mi * 6 - (mi * (mi - 1)) / 2 + (mj - mi)] = sm->H[lane];
    if (lane < 6) scr[21 + lane] = sm->g[lane];
    hp = scr;
  }
  double Sv[6], gs[6], Hs[21];
#pragma unroll
  for (int i = 0; i < 6; ++i) Sv[i] = sm->S[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    gs[i] = Sv[i] * hp[21 + i];
#pragma unroll
    for (int j = i; j < 6; ++j) Hs[ut(i, j)] = Sv[i] * hp[ut(i, j)] * Sv[j];
  }
  bool lin_ok = true;
  double stp[6];   // the Gauss-Newton step in the Jacobi-scaled space: -y, (Hs + mu D^2) y = gs
  if (!v.reuse) {  // DoglegStrategy::ComputeStep, fresh
    v.reuse = 1;
    double D2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) D2[i] = fmin(fmax(Hs[ut(i, i)], 1e-6), 1e32);  // (min_diagonal_, max_diagonal_)^2
    // ComputeGaussNewtonStep: on failure mu *= 10 while mu < max_mu (1.0)
    bool ok = false;
    double y[6] = {0, 0, 0, 0, 0, 0};
    while (v.mu < 1.0) {
      double A[21];
#pragma unroll
      for (int i = 0; i < 21; ++i) A[i] = Hs[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) A[ut(i, i)] = __builtin_fma(v.mu, D2[i], A[ut(i, i)]);
      if (chol6_uniform(A, gs, y)) { ok = true; break; }
      v.mu *= 10.0;
    }
    if (ok) {
      double q = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) { stp[i] = -y[i]; q = __builtin_fma(D2[i] * stp[i], stp[i], q); }
      v.gn_norm = fsqrt(q);   // | -D y |: the norm Ceres compares with the radius
    } else {
      lin_ok = false;
#pragma unroll
      for (int i = 0; i < 6; ++i) stp[i] = sm->gn[i];
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { sm->D[i] = D2[i]; sm->gn[i] = stp[i]; }
    }
    v.subspace_1d = -1;  // ComputeSubspaceModel is deferred until a step actually leaves the trust region
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) stp[i] = sm->gn[i];
  }
  TL_STAMP(3)
  double step[6] = {0, 0, 0, 0, 0, 0};
  bool valid = false;
  if (lin_ok && v.gn_norm == 0.0 && dot6(gs, gs) == 0.0) lin_ok = false;  // rank-0 subspace (Ceres: failure)
  if (lin_ok) {  // ComputeSubspaceDoglegStep
    if (v.gn_norm <= v.radius) {
#pragma unroll
      for (int i = 0; i < 6; ++i) step[i] = stp[i];   // gn / D
      v.step_norm = v.gn_norm;
      v.cand_gn = 1;
    } else {
      // the Gauss-Newton step leaves the trust region (rare): D, the gradient D^-1 gs and the D-scaled step, then
      // the 2-D subspace model, which lives in the LDS copy
      v.cand_gn = 0;
      double D[6], iD[6], grad[6], gnv[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        D[i] = fsqrt(sm->D[i]);
        iD[i] = fast_rcp(D[i]);
        grad[i] = gs[i] * iD[i];      // ComputeGradient
        gnv[i] = D[i] * stp[i];       // gauss_newton_step_ = -D y
      }
      if (v.subspace_1d < 0) {
        // ComputeSubspaceModel: orthonormal basis of span{grad, gn}, larger column first; g and B of the 2-D model
        const double n0 = fsqrt(dot6(grad, grad)), gnn = v.gn_norm;
        const bool gfirst = n0 >= gnn;
        const double nf = gfirst ? n0 : gnn, ns = gfirst ? gnn : n0;
        const double inf_ = fast_rcp(nf);
        double u0[6], u1[6], second[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { u0[i] = (gfirst ? grad[i] : gnv[i]) * inf_; second[i] = gfirst ? gnv[i] : grad[i]; }
        const double proj = dot6(u0, second);
#pragma unroll
        for (int i = 0; i < 6; ++i) u1[i] = second[i] - proj * u0[i];
        const double nr = fsqrt(dot6(u1, u1));
        if (ns == 0.0 || nr <= 1e-14 * nf) {
          v.subspace_1d = 1;
        } else {
          v.subspace_1d = 0;
          const double inr = fast_rcp(nr);
          double v0[6], v1[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) { u1[i] *= inr; v0[i] = u0[i] * iD[i]; v1[i] = u1[i] * iD[i]; }
          const double b0 = quad6(v0, Hs, v0), b1 = quad6(v0, Hs, v1), b3 = quad6(v1, Hs, v1);
          const double g0 = dot6(u0, grad), g1 = dot6(u1, grad);
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) { sm->U[i] = u0[i]; sm->U[6 + i] = u1[i]; }
            sm->sg[0] = g0; sm->sg[1] = g1; sm->sB[0] = b0; sm->sB[1] = b1; sm->sB[2] = b1; sm->sB[3] = b3;
          }
        }
      }
      if (v.subspace_1d) {
        const double k = -v.radius * fast_rsqrt(dot6(grad, grad));
#pragma unroll
        for (int i = 0; i < 6; ++i) step[i] = k * grad[i] * iD[i];
      } else {
        const Vec2 m2 = min_on_circle(sm->sB[0], sm->sB[1], sm->sB[1], sm->sB[3], sm->sg[0], sm->sg[1], v.radius);
#pragma unroll
        for (int i = 0; i < 6; ++i) step[i] = (sm->U[i] * m2.x + sm->U[6 + i] * m2.y) * iD[i];
      }
      v.step_norm = v.radius;
    }
    v.mcc = __builtin_fma(-0.5, quad6(step, Hs, step), -dot6(step, gs));  // model_cost_change_
    valid = v.mcc > 0.0;
  }
  TL_STAMP(4)
  // ---- candidate Plus(x, delta) on lane 0 (the even lanes) and projected-gradient point Plus(x, -g) on lane 1
  if (valid || v.need_gmax) {
    double din[6], x[6];
    const bool odd = (lane & 1) != 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      din[i] = odd ? -hp[21 + i] : (valid ? step[i] * Sv[i] : 0.0);
      x[i] = sm->x[i];
    }
    const Pose T_cur = sm->T_cur;
    const Pose C = compose_fast2(exp_fast2(din), T_cur);  // exp(in) * exp(x)   registration.cpp:162-173
    double out[6];
    log_fast2(C, out);
    if (v.need_gmax) {
      v.need_gmax = false;
      double m = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) m = fmax(m, fabs(x[i] - out[i]));
      v.gmax = rdlane(m, 1);  // || x - Plus(x, -g) ||_inf
      if (v.gmax <= kGradientTolerance) {  // the accepted point was already converged: roll back
        v.iteration--;
        v.iters--;
        v.done = 1;
        return SR_DONE;
      }
    }
    if (valid && lane == 0) {   // the candidate the next sweep evaluates
#pragma unroll
      for (int i = 0; i < 6; ++i) sm->x_cand[i] = out[i];
      sm->T_eval = C;
      sm->Rt_eval = to_rt(C);
    }
  }
  if (!valid) {  // HandleInvalidStep -> DoglegStrategy::StepIsInvalid
    v.cand_gn = 0;
    if (++v.invalid >= max_consecutive_invalid) { v.done = 1; return SR_DONE; }
    v.mu *= 10.0;
    v.reuse = 0;
    v.step_successful = 0;
    return SR_RETRY;
  }
  v.invalid = 0;
  v.phase = PH_CAND;
  return SR_CANDIDATE;  // the next K3 sweep evaluates x_cand
}

// ---- the decision a reduced sweep allows, before anything is computed: tolerances, then the step quality --------------------
// (ParameterToleranceReached / FunctionToleranceReached / IsStepSuccessful, trust_region_minimizer.cc).  Pure function of the LDS
// image and the sweep's cost.
struct StepDecision {
  bool iter0;      // the sweep was the first evaluation of the Solve (IterationZero)
  bool done_tol;   // a tolerance exit: the candidate is NOT applied
  bool accepted;   // HandleSuccessfulStep
  double rel;      // step quality (0 unless it was evaluated)
};
__device__ __forceinline__ StepDecision step_decide(const GnState* sm, double cost) {
  constexpr double function_tolerance = 1e-6, parameter_tolerance = 1e-8, min_relative_decrease = 1e-3;
  StepDecision d{sm->phase == PH_ITER0, false, false, 0.0};
  if (d.iter0) return d;
  const double x_cost = sm->x_cost, x_norm = sm->x_norm;
  double dx[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) dx[i] = sm->x[i] - sm->x_cand[i];
  if (fsqrt(dot6(dx, dx)) <= parameter_tolerance * (x_norm + parameter_tolerance)) d.done_tol = true;   // ParameterToleranceReached
  else if (fabs(x_cost - cost) <= function_tolerance * x_cost) d.done_tol = true;                        // FunctionToleranceReached
  else {
    d.rel = (x_cost - cost) / sm->model_cost_change;   // TrustRegionStepEvaluator::StepQuality (a decision: exact division)
    d.accepted = d.rel > min_relative_decrease;
  }
  return d;
}

// FinalizeIterationAndCheckIfMinimizerCanContinue (the gradient test is deferred while need_gmax): true = go on
__device__ __forceinline__ bool step_can_continue(StepVars& v) {
  constexpr double min_trust_region_radius = 1e-32;
  constexpr int max_num_iterations = 4;
  if (v.iteration >= max_num_iterations || (!v.need_gmax && v.step_successful && v.gmax <= kGradientTolerance) ||
      v.radius <= min_trust_region_radius) {
    v.done = 1;
    return false;
  }
  v.iteration++;
  v.iters++;
  return true;
}

// WRITE_GLOBAL = false (k_solve_all: every block advances its own LDS image of the state, one of them writes it out): the
// device state `st` is not touched (wave-uniform run-time flag: ONE inlined copy of the step per kernel).
__device__ __forceinline__ void gn_consume(GnState* st, const double* tot /* LDS */, int lane, GnState* sm /* LDS */,
                                           double* scr /* LDS */, const bool WRITE_GLOBAL = true) {
  TL_STAMP(1)
  StepVars v;
  v.phase = sm->phase; v.iteration = sm->iteration; v.invalid = sm->invalid; v.step_successful = sm->step_successful;
  v.reuse = sm->reuse; v.subspace_1d = sm->subspace_1d; v.cand_gn = sm->cand_gn; v.done = 0; v.iters = sm->gn_iterations;
  v.gmax = sm->gmax; v.mcc = sm->model_cost_change; v.radius = sm->radius; v.mu = sm->mu; v.step_norm = sm->step_norm;
  v.gn_norm = sm->gn_norm;
  v.need_gmax = false;
  v.take = false;   // the totals of this sweep become the system of the accepted point
  int evals = sm->gn_evaluations + 1, accepted = sm->accepted_steps;
  const int sweeps = sm->gn_sweeps + 1;
  double x_cost = sm->x_cost, x_norm = sm->x_norm;
  const bool eval_reuse = sm->no_eval_reuse == 0;
  const double cost = tot[27];
  if (v.phase == PH_ITER0) {
    x_cost = cost;
    v.take = true;
    double S[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      S[i] = fast_rcp(1.0 + fsqrt(tot[ut(i, i)]));  // jacobi_scaling, fixed at iteration 0 of the Solve
      x[i] = sm->x[i];
    }
    x_norm = fsqrt(dot6(x, x));
    v.step_successful = 1;
    v.need_gmax = true;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) sm->S[i] = S[i];
    }
  } else {
    const StepDecision d = step_decide(sm, cost);
    if (d.done_tol) v.done = 1;                     // ParameterToleranceReached / FunctionToleranceReached
    else {
      const double rel = d.rel;                     // TrustRegionStepEvaluator::StepQuality
      if (d.accepted) {                             // HandleSuccessfulStep: x = candidate
        double xc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) xc[i] = sm->x_cand[i];
        if (lane < 6) sm->x[lane] = sm->x_cand[lane];
        if (lane < 7) reinterpret_cast<double*>(&sm->T_cur)[lane] = reinterpret_cast<const double*>(&sm->T_eval)[lane];
        x_norm = fsqrt(dot6(xc, xc));
        x_cost = cost;
        v.take = true;
        v.step_successful = 1;
        accepted++;
        v.need_gmax = true;
        if (rel < 0.25) v.radius *= 0.5;                      // DoglegStrategy::StepAccepted
        if (rel > 0.75) v.radius = fmax(v.radius, 3.0 * v.step_norm);
        v.mu = fmax(1e-8, 2.0 * v.mu / 10.0);
        v.reuse = 0;
      } else {                                      // HandleUnsuccessfulStep / StepRejected
        v.step_successful = 0;
        v.radius *= 0.5;
        v.reuse = 1;
      }
    }
  }
  if (v.take) {
    // H (full 6x6) and g of the accepted point into the state, one element per lane
    const int mi = lane < 36 ? lane / 6 : 0, mj = lane < 36 ? lane - mi * 6 : 0;
    const int lo = mi < mj ? mi : mj, hi = mi < mj ? mj : mi;
    const double h = tot[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
    const double gg = tot[21 + (lane < 6 ? lane : 0)];
    if (lane < 36) sm->H[lane] = h;
    if (lane < 6) sm->g[lane] = gg;
  }
  TL_STAMP(2)
  // ---- the minimiser's loop.  First the part that needs no arithmetic: a rejected step retried with the dogleg data
  //      reused whose Gauss-Newton point is still inside the halved region is, input for input, the candidate the state
  //      already holds (x_cand, T_eval, Rt_eval, the model cost change).
  bool compute = false;
  while (!v.done) {
    if (!step_can_continue(v)) break;
    if (v.reuse && v.cand_gn && v.gn_norm <= v.radius && !v.need_gmax) {
      v.invalid = 0;
      v.phase = PH_CAND;
      if (!eval_reuse) break;   // development knob: the same candidate is swept again
      evals++;                  // served from the totals in hand: same point, same verdict -- rejected again
      v.radius *= 0.5;
      continue;
    }
    compute = true;
    break;
  }
  // ---- then at most one step computation on the straight-line path; an invalid step (rare) goes round a loop of its own
  if (compute) {
    StepResult r = step_compute(v, st, tot, lane, sm, scr);
    if (__builtin_expect(r == SR_RETRY, 0)) {
      while (step_can_continue(v)) {
        r = step_compute(v, st, tot, lane, sm, scr);
        if (r != SR_RETRY) break;
      }
    }
  }
  TL_STAMP(5)
  // ---- the scalars into the LDS image, then the image into the device state (one coalesced copy)
  if (lane == 0) {
    sm->phase = v.phase; sm->iteration = v.iteration; sm->invalid = v.invalid; sm->step_successful = v.step_successful;
    sm->reuse = v.reuse; sm->subspace_1d = v.subspace_1d; sm->done = v.done; sm->cand_gn = v.cand_gn;
    sm->gn_evaluations = evals; sm->gn_iterations = v.iters; sm->accepted_steps = accepted;
    sm->gn_sweeps = sweeps;
    sm->x_cost = x_cost; sm->x_norm = x_norm; sm->gmax = v.gmax; sm->model_cost_change = v.mcc;
    sm->radius = v.radius; sm->mu = v.mu; sm->step_norm = v.step_norm; sm->gn_norm = v.gn_norm;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (also a compiler barrier: the copy below reads the image as raw words)
  {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(sm);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(st);
    if (WRITE_GLOBAL) {
#pragma unroll
      for (int w = lane; w < kStepWords; w += 64) dst[w] = src[w];
    }
    // has this Solve ended somewhere else than where the factor set was built?  (what publish_and_rearm will find when it
    // compares x with x_build -- known here already, so the next search need not wait for the finish kernel)
    bool moved = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) moved = moved || (sm->x[i] != sm->x_build[i]);
    if (lane == 0) {
      sm->spec_build = (v.done && moved) ? 1 : 0;
      if (WRITE_GLOBAL) st->spec_build = sm->spec_build;
    }
  }
}
// what gn_consume(WRITE_GLOBAL = true) writes to the device state, as a step of its own: k_solve_all's lead block hands the
// candidate pose to its waves FIRST and copies the image out while they evaluate (its stepper wave holds no chunk) -- the copy
// (~0.3 us) was on the path every block's next row waits on
__device__ __forceinline__ void gn_write_back(GnState* st, const GnState* sm /* LDS */, int lane) {
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(sm);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(st);
#pragma unroll
  for (int w = lane; w < kStepWords; w += 64) dst[w] = src[w];
  if (lane == 0) st->spec_build = sm->spec_build;
}

}  // namespace tl
