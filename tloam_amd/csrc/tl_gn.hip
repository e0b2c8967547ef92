// tl_gn.hip -- K3 (residual + Jacobian + TLS weight + Cauchy correction + normal-equation reduction),
// K5 (the Ceres-configured trust-region / dogleg step on the 6x6 system, device resident) and
// K4 (GNC-TLS weight update) for gfx950.
//
// Replaces, on the device:
//   PointToPointErr::Evaluate  registration.cpp:19-47      PointToLineErr::Evaluate :55-88
//   PointToPlaneErr::Evaluate  registration.cpp:96-117     PoseSE3Parameterization::Plus :162-173
//   ceres::Problem/Solve as configured at registration.cpp:970-974, :1036-1047 (Ceres 2.0 evaluator,
//   CauchyLoss(1.0) + clamped Corrector, TrustRegionMinimizer, DoglegStrategy/SUBSPACE_DOGLEG,
//   DENSE_QR -> here Cholesky of the Jacobi-scaled 6x6 normal equations)
//   LocalRegistration::updateWeight registration.cpp:858-876 and the cost sums :1091-1094
//
// K3 is HBM-bound (72/88/64 algorithmic bytes per plane/line/point correspondence against
// ~190-260 fp64 flops): no MFMA -- the contraction is 6 wide.  Design: SoA streams read as
// 16-byte double2 per lane (1 KiB per wave instruction, fully coalesced), each wave walks
// 128-correspondence chunks, 28 fp64 accumulators per lane (21 upper-triangular H, 6 g, cost),
// wave reduction by cross-lane shuffles, 4-wave LDS combine, one 32-double partial row per block,
// then a fixed-order tree over the rows: bit-reproducible run to run (no atomics).
#include "tl_common.hpp"

namespace tl {

// ================================================================================================
//  K3
// ================================================================================================
struct Acc {
  double v[kAccN];
};

// upper-triangular index of (i,j), i<=j, row-major: 0..20
__host__ __device__ constexpr int ut(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// one residual row: r (already loss-corrected is NOT assumed): adds rho*J^T J and rho*J^T r
__device__ __forceinline__ void acc_row(Acc& a, const double J[6], double r, double rho) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double ji = rho * J[i];
    a.v[21 + i] += ji * r;
#pragma unroll
    for (int j = i; j < 6; ++j) a.v[ut(i, j)] += ji * J[j];
  }
}

// PointToPlaneErr::Evaluate (registration.cpp:96-117) through ResidualBlock::Evaluate + CauchyLoss(1)
__device__ __forceinline__ double eval_plane(const Pose& T, Vec3 p, Vec3 n, double d, double w, Acc& a) {
  const Vec3 pw = act(T, p);
  const double r = dot(n, pw) + d;           // :100 (unweighted)
  const Vec3 c = cross(pw, n);               // n^T (-hat(pw)) = (pw x n)^T   :110,:112
  const double J[6] = {n.x * w, n.y * w, n.z * w, c.x * w, c.y * w, c.z * w};
  const double s = r * r;                    // squared norm of the block
  const double sum = 1.0 + s;
  const double rho1 = 1.0 / sum;             // CauchyLoss(1): rho' ; rho'' < 0 -> clamped corrector
  a.v[27] += 0.5 * log(sum);
  acc_row(a, J, r, rho1);
  return s;                                  // side channel *cost = r^2  :101
}

// PointToLineErr::Evaluate (registration.cpp:55-88)
__device__ __forceinline__ double eval_line(const Pose& T, Vec3 p, Vec3 la, Vec3 lb, double w, Acc& a) {
  const Vec3 pw = act(T, p);
  const Vec3 nu = cross(pw - la, pw - lb);   // :62
  const Vec3 de = la - lb;                   // :63
  const double den = sqrt(dot(de, de));
  const double r0 = nu.x / den * w, r1 = nu.y / den * w, r2 = nu.z / den * w;  // :65-67
  const double side = (r0 + r1 + r2) * (r0 + r1 + r2);                          // :69
  // J = hat(lb - la) * [I w, -hat(pw) w] / |de|   :77-83.  With e = lb - la:
  //   rows of hat(e) = (0,-ez,ey), (ez,0,-ex), (-ey,ex,0);  hat(e) * (-hat(pw)) = -hat(e) hat(pw)
  const Vec3 e = lb - la;
  const double k = w / den;
  // -hat(e) hat(pw) = -(pw e^T - (e.pw) I)  ->  (e.pw) I - pw e^T  ... row i, col j: (e.pw) d_ij - e_j... see below
  // hat(a) hat(b) = b a^T - (a.b) I  =>  -hat(e) hat(pw) = (e.pw) I - pw e^T
  const double ep = dot(e, pw);
  const double J0[6] = {0.0, -e.z * k, e.y * k, (ep - pw.x * e.x) * k, (-pw.x * e.y) * k, (-pw.x * e.z) * k};
  const double J1[6] = {e.z * k, 0.0, -e.x * k, (-pw.y * e.x) * k, (ep - pw.y * e.y) * k, (-pw.y * e.z) * k};
  const double J2[6] = {-e.y * k, e.x * k, 0.0, (-pw.z * e.x) * k, (-pw.z * e.y) * k, (ep - pw.z * e.z) * k};
  const double s = r0 * r0 + r1 * r1 + r2 * r2;
  const double sum = 1.0 + s;
  const double rho1 = 1.0 / sum;
  a.v[27] += 0.5 * log(sum);
  acc_row(a, J0, r0, rho1);
  acc_row(a, J1, r1, rho1);
  acc_row(a, J2, r2, rho1);
  return side;
}

// PointToPointErr::Evaluate (registration.cpp:19-47)
__device__ __forceinline__ double eval_point(const Pose& T, Vec3 p, Vec3 q, double w, Acc& a) {
  const Vec3 pw = act(T, p);
  const double r0 = (q.x - pw.x) * w, r1 = (q.y - pw.y) * w, r2 = (q.z - pw.z) * w;  // :26-30
  const double side = (r0 + r1 + r2) * (r0 + r1 + r2);                                // :32
  // J = [-I w, hat(pw) w]  :39-40
  const double J0[6] = {-w, 0.0, 0.0, 0.0, -pw.z * w, pw.y * w};
  const double J1[6] = {0.0, -w, 0.0, pw.z * w, 0.0, -pw.x * w};
  const double J2[6] = {0.0, 0.0, -w, -pw.y * w, pw.x * w, 0.0};
  const double s = r0 * r0 + r1 * r1 + r2 * r2;
  const double sum = 1.0 + s;
  const double rho1 = 1.0 / sum;
  a.v[27] += 0.5 * log(sum);
  acc_row(a, J0, r0, rho1);
  acc_row(a, J1, r1, rho1);
  acc_row(a, J2, r2, rho1);
  return side;
}

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }

__global__ __launch_bounds__(256) void k3_accumulate(CorrView cv, GnState* __restrict__ st,
                                                     double* __restrict__ partials, int force) {
  __shared__ double red[4][kAccN];
  if (!force && st->done) return;  // after a tolerance exit the remaining launches are no-ops
  const Pose T = st->T_eval;       // exp(point), hoisted out of the per-block Evaluate (:22,:58,:98)
  int n[kKinds], chunk_end[kKinds];
  int total_chunks = 0;
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    n[k] = cv.seg_n[k];
    total_chunks += (n[k] + kChunk - 1) / kChunk;
    chunk_end[k] = total_chunks;
  }
  Acc a;
#pragma unroll
  for (int i = 0; i < kAccN; ++i) a.v[i] = 0.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave, W = gridDim.x * 4;
  for (int c = gw; c < total_chunks; c += W) {
    int kind = 0;
#pragma unroll
    for (int k = 0; k < kKinds - 1; ++k) kind += (c >= chunk_end[k]) ? 1 : 0;
    const int first = (kind == 0) ? 0 : chunk_end[kind - 1];
    const int local = (c - first) * kChunk + lane * 2;
    const int rem = n[kind] - local;  // >=2: both elements, 1: first only
    if (rem <= 0) continue;
    const CorrSeg& seg = cv.k[kind];
    const int j = local;
    const double2 px = ld2(seg.px + j), py = ld2(seg.py + j), pz = ld2(seg.pz + j);
    const double2 ax = ld2(seg.ax + j), ay = ld2(seg.ay + j), az = ld2(seg.az + j);
    const double2 w = ld2(seg.w + j);
    double c0, c1 = 0.0;
    if (kind <= TLOAM_KIND_GROUND) {
      const double2 d = ld2(seg.d + j);
      c0 = eval_plane(T, Vec3{px.x, py.x, pz.x}, Vec3{ax.x, ay.x, az.x}, d.x, w.x, a);
      if (rem > 1) c1 = eval_plane(T, Vec3{px.y, py.y, pz.y}, Vec3{ax.y, ay.y, az.y}, d.y, w.y, a);
    } else if (kind == TLOAM_KIND_EDGE) {
      const double2 bx = ld2(seg.bx + j), by = ld2(seg.by + j), bz = ld2(seg.bz + j);
      c0 = eval_line(T, Vec3{px.x, py.x, pz.x}, Vec3{ax.x, ay.x, az.x}, Vec3{bx.x, by.x, bz.x}, w.x, a);
      if (rem > 1) c1 = eval_line(T, Vec3{px.y, py.y, pz.y}, Vec3{ax.y, ay.y, az.y}, Vec3{bx.y, by.y, bz.y}, w.y, a);
    } else {
      c0 = eval_point(T, Vec3{px.x, py.x, pz.x}, Vec3{ax.x, ay.x, az.x}, w.x, a);
      if (rem > 1) c1 = eval_point(T, Vec3{px.y, py.y, pz.y}, Vec3{ax.y, ay.y, az.y}, w.y, a);
    }
    // the `mutable double* cost` side channel (registration.hpp:51,76,96): written on EVERY sweep
    if (rem > 1) *reinterpret_cast<double2*>(seg.cost + j) = double2{c0, c1};
    else seg.cost[j] = c0;
  }
  // wave reduction (fixed xor-free down tree), then 4 waves through LDS in wave order
#pragma unroll
  for (int i = 0; i < kAccN; ++i) {
    double v = a.v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAccStride) {
    double v = 0.0;
    if (threadIdx.x < kAccN) v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    partials[(size_t)blockIdx.x * kAccStride + threadIdx.x] = v;
  }
}

int k3_grid_for(int total_cap) {
  // one wave per 128-correspondence chunk up to a full-chip resident grid (256 CUs x 4 blocks)
  int waves = (total_cap + kChunk - 1) / kChunk;
  int blocks = (waves + 3) / 4;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) {
    // balance: every wave gets the same number of chunks
    const int per_wave = (waves + 4096 - 1) / 4096;
    const int need_waves = (waves + per_wave - 1) / per_wave;
    blocks = (need_waves + 3) / 4;
  }
  return blocks;
}
void launch_k3(const CorrView& cv, GnState* st, double* partials, int grid, bool force, hipStream_t s) {
  hipLaunchKernelGGL(k3_accumulate, dim3(grid), dim3(256), 0, s, cv, st, partials, force ? 1 : 0);
}

// ================================================================================================
//  fixed-order reduction of the per-block rows  (1 block x 1024 threads)
// ================================================================================================
__device__ __forceinline__ void reduce_rows(const double* __restrict__ partials, int rows, double* lds /*[32][33]*/,
                                            double* out32 /* LDS [32] */) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;  // 32 groups of 32 components
  double v = 0.0;
  for (int b = grp; b < rows; b += 32) v += partials[(size_t)b * kAccStride + comp];
  lds[grp * 33 + comp] = v;
  __syncthreads();
  if (threadIdx.x < 32) {
    double t = 0.0;
    for (int g = 0; g < 32; ++g) t += lds[g * 33 + threadIdx.x];
    out32[threadIdx.x] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void k_reduce(const double* __restrict__ partials, int rows,
                                                 const GnState* __restrict__ st, double* __restrict__ out48) {
  __shared__ double lds[32 * 33];
  __shared__ double tot[32];
  (void)st;
  reduce_rows(partials, rows, lds, tot);
  if (threadIdx.x < kReduceBuf) out48[threadIdx.x] = (threadIdx.x < kAccN) ? tot[threadIdx.x] : 0.0;
}
void launch_reduce(const double* partials, int grid, GnState* st, double* out48, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, partials, grid, st, out48);
}

// ================================================================================================
//  K5: Ceres TrustRegionMinimizer + DoglegStrategy on the 6x6 system (one lane)
// ================================================================================================
__device__ bool chol6_solve(const double A[36], const double b[6], double y[6]) {
  double L[36];
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    if (!(s > 0.0) || !isfinite(s)) return false;
    const double ljj = sqrt(s);
    L[j * 6 + j] = ljj;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i * 6 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = v / ljj;
    }
  }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= L[i * 6 + k] * z[k];
    z[i] = v / L[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    double v = z[i];
    for (int k = i + 1; k < 6; ++k) v -= L[k * 6 + i] * y[k];
    y[i] = v / L[i * 6 + i];
  }
  for (int i = 0; i < 6; ++i)
    if (!isfinite(y[i])) return false;
  return true;
}
__device__ double norm_n(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}
// minimum of 0.5 x^T B x + g^T x on |x| = r (dogleg_strategy.cc FindMinimumOnTrustRegionBoundary;
// Ceres roots a quartic -- the global minimiser is unique, here bracketed by sampling the angle
// and polished by bisection on the tangential derivative).  Rare branch: only when the
// Gauss-Newton step leaves the trust region.
__device__ void min_on_circle(const double B[4], const double g[2], double r, double x[2]) {
  const int NS = 720;
  double best = 1e300, bth = 0.0;
  const double b01 = 0.5 * (B[1] + B[2]);
  for (int i = 0; i < NS; ++i) {
    const double th = 2.0 * kPi * i / NS;
    const double cx = r * cos(th), sx = r * sin(th);
    const double f = 0.5 * (B[0] * cx * cx + 2.0 * b01 * cx * sx + B[3] * sx * sx) + g[0] * cx + g[1] * sx;
    if (f < best) { best = f; bth = th; }
  }
  double lo = bth - 2.0 * kPi / NS, hi = bth + 2.0 * kPi / NS;
  for (int it = 0; it < 200; ++it) {
    const double th = 0.5 * (lo + hi);
    const double cx = r * cos(th), sx = r * sin(th);
    const double gx = B[0] * cx + b01 * sx + g[0];
    const double gy = b01 * cx + B[3] * sx + g[1];
    const double df = gx * (-sx) + gy * cx;
    if (df > 0.0) hi = th; else lo = th;
    if (hi - lo < 1e-16 * (1.0 + fabs(th))) break;
  }
  const double th = 0.5 * (lo + hi);
  x[0] = r * cos(th);
  x[1] = r * sin(th);
}

// DoglegStrategy::ComputeStep (SUBSPACE_DOGLEG) on the Jacobi-scaled system; true = solver success
__device__ bool dogleg_compute_step(GnState& s, const double Hs[36], const double gs[6], double step[6]) {
  if (!s.reuse) {
    s.reuse = 1;
    for (int i = 0; i < 6; ++i) {
      double v = Hs[i * 6 + i];
      v = fmax(v, 1e-6);   // min_diagonal_
      v = fmin(v, 1e32);   // max_diagonal_
      s.D[i] = sqrt(v);
    }
    for (int i = 0; i < 6; ++i) s.grad[i] = gs[i] / s.D[i];
    {
      double v[6], Hv = 0.0, gg = 0.0;
      for (int i = 0; i < 6; ++i) v[i] = s.grad[i] / s.D[i];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Hv += v[i] * Hs[i * 6 + j] * v[j];
      for (int i = 0; i < 6; ++i) gg += s.grad[i] * s.grad[i];
      s.alpha = gg / Hv;
    }
    bool ok = false;
    double y[6];
    while (s.mu < 1.0) {  // max_mu_
      double A[36];
      for (int i = 0; i < 36; ++i) A[i] = Hs[i];
      for (int i = 0; i < 6; ++i) A[i * 6 + i] += s.mu * s.D[i] * s.D[i];
      if (chol6_solve(A, gs, y)) { ok = true; break; }
      s.mu *= 10.0;  // mu_increase_factor_
    }
    if (!ok) return false;
    for (int i = 0; i < 6; ++i) s.gn[i] = -s.D[i] * y[i];
    // ComputeSubspaceModel
    const double n0 = norm_n(s.grad, 6), n1 = norm_n(s.gn, 6);
    if (n0 == 0.0 && n1 == 0.0) return false;
    const bool gfirst = n0 >= n1;  // column pivoting: larger column first
    const double* first = gfirst ? s.grad : s.gn;
    const double* second = gfirst ? s.gn : s.grad;
    const double nf = gfirst ? n0 : n1, ns = gfirst ? n1 : n0;
    double u0[6], u1[6], proj = 0.0;
    for (int i = 0; i < 6; ++i) u0[i] = first[i] / nf;
    for (int i = 0; i < 6; ++i) proj += u0[i] * second[i];
    for (int i = 0; i < 6; ++i) u1[i] = second[i] - proj * u0[i];
    const double nr = norm_n(u1, 6);
    if (ns == 0.0 || nr <= 1e-14 * nf) {
      s.subspace_1d = 1;
    } else {
      s.subspace_1d = 0;
      for (int i = 0; i < 6; ++i) { u1[i] /= nr; s.U[i] = u0[i]; s.U[6 + i] = u1[i]; }
      s.sg[0] = s.sg[1] = 0.0;
      for (int i = 0; i < 6; ++i) { s.sg[0] += u0[i] * s.grad[i]; s.sg[1] += u1[i] * s.grad[i]; }
      double v0[6], v1[6], b00 = 0, b01 = 0, b11 = 0;
      for (int i = 0; i < 6; ++i) { v0[i] = u0[i] / s.D[i]; v1[i] = u1[i] / s.D[i]; }
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          b00 += v0[i] * Hs[i * 6 + j] * v0[j];
          b01 += v0[i] * Hs[i * 6 + j] * v1[j];
          b11 += v1[i] * Hs[i * 6 + j] * v1[j];
        }
      s.sB[0] = b00; s.sB[1] = b01; s.sB[2] = b01; s.sB[3] = b11;
    }
  }
  // ComputeSubspaceDoglegStep
  const double gnn = norm_n(s.gn, 6);
  if (gnn <= s.radius) {
    for (int i = 0; i < 6; ++i) step[i] = s.gn[i] / s.D[i];
    s.step_norm = gnn;
    return true;
  }
  if (s.subspace_1d) {
    const double gnorm = norm_n(s.grad, 6);
    for (int i = 0; i < 6; ++i) step[i] = -(s.radius / gnorm) * s.grad[i] / s.D[i];
    s.step_norm = s.radius;
    return true;
  }
  double m2[2];
  min_on_circle(s.sB, s.sg, s.radius, m2);
  for (int i = 0; i < 6; ++i) step[i] = (s.U[i] * m2[0] + s.U[6 + i] * m2[1]) / s.D[i];
  s.step_norm = s.radius;
  return true;
}

// || x - Plus(x, -g) ||_inf   (TrustRegionMinimizer::EvaluateGradientAndJacobian)
__device__ double grad_max_norm(const double x[6], const double g[6]) {
  double ng[6], xp[6], m = 0.0;
  for (int i = 0; i < 6; ++i) ng[i] = -g[i];
  se3_plus(x, ng, xp);
  for (int i = 0; i < 6; ++i) m = fmax(m, fabs(x[i] - xp[i]));
  return m;
}

__device__ void unpack_total(const double* tot, double* cost, double g[6], double H[36]) {
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      const double v = tot[ut(i, j)];
      H[i * 6 + j] = v;
      H[j * 6 + i] = v;
    }
  for (int i = 0; i < 6; ++i) g[i] = tot[21 + i];
  *cost = tot[27];
}

// Consume one reduced sweep and run the minimiser until the next sweep is needed (or it is done).
// Mirrors trust_region_minimizer.cc Minimize(): IterationZero, then per iteration
// ComputeTrustRegionStep -> candidate -> tolerances -> IsStepSuccessful -> Handle(Un)SuccessfulStep.
// The candidate sweep is fused (cost + Jacobian in one pass): Ceres evaluates the candidate
// cost-only and re-evaluates an accepted point with Jacobians; same numbers, half the traffic.
__device__ void gn_consume(GnState& s, const double* tot) {
  if (s.done) return;
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32;
  const int max_num_iterations = 4, max_consecutive_invalid = 5;
  double cost, g[6], H[36];
  unpack_total(tot, &cost, g, H);
  s.gn_evaluations++;
  if (s.phase == PH_ITER0) {
    s.x_cost = cost;
    for (int i = 0; i < 6; ++i) s.g[i] = g[i];
    for (int i = 0; i < 36; ++i) s.H[i] = H[i];
    for (int i = 0; i < 6; ++i) s.S[i] = 1.0 / (1.0 + sqrt(H[i * 6 + i]));  // jacobi_scaling, iteration 0 only
    s.x_norm = norm_n(s.x, 6);
    s.step_successful = 1;
    s.gmax = grad_max_norm(s.x, g);
  } else {
    const double candidate_cost = cost;
    double dx[6];
    for (int i = 0; i < 6; ++i) dx[i] = s.x[i] - s.x_cand[i];
    if (norm_n(dx, 6) <= parameter_tolerance * (s.x_norm + parameter_tolerance)) { s.done = 1; return; }
    if (fabs(s.x_cost - candidate_cost) <= function_tolerance * s.x_cost) { s.done = 1; return; }
    const double rel = (s.x_cost - candidate_cost) / s.model_cost_change;
    if (rel > min_relative_decrease) {
      for (int i = 0; i < 6; ++i) s.x[i] = s.x_cand[i];
      s.T_cur = s.T_eval;
      s.x_norm = norm_n(s.x, 6);
      s.x_cost = candidate_cost;
      for (int i = 0; i < 6; ++i) s.g[i] = g[i];
      for (int i = 0; i < 36; ++i) s.H[i] = H[i];
      s.gmax = grad_max_norm(s.x, g);
      s.step_successful = 1;
      s.accepted_steps++;
      if (rel < 0.25) s.radius *= 0.5;                               // DoglegStrategy::StepAccepted
      if (rel > 0.75) s.radius = fmax(s.radius, 3.0 * s.step_norm);
      s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
      s.reuse = 0;
    } else {
      s.step_successful = 0;                                         // StepRejected
      s.radius *= 0.5;
      s.reuse = 1;
    }
  }
  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (s.iteration >= max_num_iterations) { s.done = 1; return; }
    if (s.step_successful && s.gmax <= gradient_tolerance) { s.done = 1; return; }
    if (s.radius <= min_trust_region_radius) { s.done = 1; return; }
    s.iteration++;
    s.gn_iterations++;
    double Hs[36], gs[6], step[6];
    for (int i = 0; i < 6; ++i) {
      gs[i] = s.S[i] * s.g[i];
      for (int j = 0; j < 6; ++j) Hs[i * 6 + j] = s.S[i] * s.H[i * 6 + j] * s.S[j];
    }
    const bool lin_ok = dogleg_compute_step(s, Hs, gs, step);
    bool valid = false;
    if (lin_ok) {
      double sg = 0.0, sHs = 0.0;
      for (int i = 0; i < 6; ++i) {
        sg += step[i] * gs[i];
        for (int j = 0; j < 6; ++j) sHs += step[i] * Hs[i * 6 + j] * step[j];
      }
      s.model_cost_change = -sg - 0.5 * sHs;
      valid = s.model_cost_change > 0.0;
    }
    if (!valid) {  // HandleInvalidStep
      if (++s.invalid >= max_consecutive_invalid) { s.done = 1; return; }
      s.mu *= 10.0;
      s.reuse = 0;
      s.step_successful = 0;
      continue;
    }
    s.invalid = 0;
    double delta[6];
    for (int i = 0; i < 6; ++i) delta[i] = step[i] * s.S[i];
    se3_plus(s.x, delta, s.x_cand);      // PoseSE3Parameterization::Plus  registration.cpp:162-173
    s.T_eval = se3_exp(s.x_cand);        // what every Evaluate() then computes (:22,:58,:98)
    s.phase = PH_CAND;
    return;
  }
}

__global__ void k_solve_init(GnState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  GnState& s = *st;
  s.radius = 1e4;  // initial_trust_region_radius
  s.mu = 1e-8;     // min_mu_
  s.reuse = 0;
  s.subspace_1d = 0;
  s.phase = PH_ITER0;
  s.iteration = 0;
  s.invalid = 0;
  s.step_successful = 1;
  s.done = 0;
  s.T_eval = se3_exp(s.x);
  s.T_cur = s.T_eval;
}
void launch_solve_init(GnState* st, hipStream_t s) { hipLaunchKernelGGL(k_solve_init, dim3(1), dim3(64), 0, s, st); }

__global__ void k_set_eval(GnState* st, const double* se3) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[6];
  for (int i = 0; i < 6; ++i) a[i] = se3[i];
  st->T_eval = se3_exp(a);
}
void launch_set_eval(GnState* st, const double* se3_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_set_eval, dim3(1), dim3(64), 0, s, st, se3_dev);
}

__global__ void k_gn_step(GnState* st, const double* __restrict__ in48) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < kReduceBuf; ++i) st->total[i] = in48[i];
  gn_consume(*st, st->total);
}
void launch_gn_step(GnState* st, const double* in48, hipStream_t s) {
  hipLaunchKernelGGL(k_gn_step, dim3(1), dim3(64), 0, s, st, in48);
}

// single-GPU fast path: reduce the block rows and advance the minimiser in ONE launch
__global__ __launch_bounds__(1024) void k_reduce_and_step(const double* __restrict__ partials, int rows,
                                                          GnState* __restrict__ st) {
  __shared__ double lds[32 * 33];
  __shared__ double tot[32];
  if (st->done) return;
  reduce_rows(partials, rows, lds, tot);
  if (threadIdx.x == 0) {
    for (int i = 0; i < kAccN; ++i) st->total[i] = tot[i];
    gn_consume(*st, st->total);
  }
}
void launch_reduce_and_step(const double* partials, int grid, GnState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_and_step, dim3(1), dim3(1024), 0, s, partials, grid, st);
}

// ================================================================================================
//  K4: GNC-TLS weights (registration.cpp:858-876) + per-kind side-channel sums (:1091-1094)
// ================================================================================================
struct WeightArgs {
  CorrView cv;
  SlotView sv;
  WeightParams wp;
};
__global__ __launch_bounds__(256) void k_weights(WeightArgs A, double* __restrict__ partial) {
  __shared__ double red[4][8];
  double sum[kKinds] = {0, 0, 0, 0};
  double bad = 0.0;
  const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    const int n = A.cv.seg_n[k];
    const CorrSeg& seg = A.cv.k[k];
    for (int i = tid; i < n; i += stride) {
      const double c = seg.cost[i];
      sum[k] += c;
      if (!A.wp.active[k]) continue;
      if (c == 0) continue;                          // :862
      double w;
      if (c >= A.wp.th1) w = 0.0;                    // :865
      else if (c <= A.wp.th2) w = 1.0;               // :867
      else {
        w = sqrt(A.wp.noise_bound_sq * A.wp.mu * (A.wp.mu + 1) / c) - A.wp.mu;  // :870
        if (!(w >= 0.0 && w <= 1.0)) bad += 1.0;     // the reference asserts here (:871)
      }
      const int slot = A.sv.slot_off[k] + (seg.idx[i] - A.sv.src_lo[k]);
      A.sv.w_src[slot] = w;
    }
  }
  double v[5] = {sum[0], sum[1], sum[2], sum[3], bad};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    if (lane == 0) red[wave][i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = 0.0;
    if (threadIdx.x < 5) t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    partial[blockIdx.x * 8 + threadIdx.x] = t;
  }
}
void launch_weights(const CorrView& cv, const SlotView& sv, const WeightParams& wp, double* partial, int blocks,
                    hipStream_t s) {
  WeightArgs A;
  A.cv = cv;
  A.sv = sv;
  A.wp = wp;
  hipLaunchKernelGGL(k_weights, dim3(blocks), dim3(256), 0, s, A, partial);
}

// sums16 = [kind_cost x4, n_corr x4 (as doubles), bad, 0...]; all-reduced by the host when sharded
__global__ void k_outer_finish(const double* __restrict__ partial, int blocks, const int* __restrict__ seg_n,
                               double* __restrict__ sums16) {
  const int t = threadIdx.x;
  if (t < 16) {
    double v = 0.0;
    if (t < 4 || t == 8) {
      const int col = (t == 8) ? 4 : t;
      for (int b = 0; b < blocks; ++b) v += partial[b * 8 + col];
    } else if (t < 8) {
      v = (double)seg_n[t - 4];
    }
    sums16[t] = v;
  }
}
void launch_outer_finish(const double* partial, int blocks, const int* seg_n, GnState* st, double* sums16,
                         hipStream_t s) {
  (void)st;
  hipLaunchKernelGGL(k_outer_finish, dim3(1), dim3(64), 0, s, partial, blocks, seg_n, sums16);
}
__global__ void k_outer_publish(const double* __restrict__ sums16, GnState* st) {
  const int t = threadIdx.x;
  if (t < 4) {
    st->kind_cost[t] = sums16[t];
    st->n_corr[t] = (int)sums16[4 + t];
  }
  if (t == 0) st->bad_weights += (int)sums16[8];
}
void launch_outer_publish(const double* sums16, GnState* st, hipStream_t s) {
  hipLaunchKernelGGL(k_outer_publish, dim3(1), dim3(64), 0, s, sums16, st);
}

// PointCloud2::Transform (open3d PointCloud2.cpp:71-75): p <- (M * (p,1)).hnormalized()
struct Mat16 { double m[16]; };
__global__ void k_transform(double* aos, size_t n, Mat16 M) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double x = aos[3 * i], y = aos[3 * i + 1], z = aos[3 * i + 2];
    double o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = M.m[r] * x + M.m[4 + r] * y + M.m[8 + r] * z + M.m[12 + r];
    aos[3 * i] = o[0] / o[3];
    aos[3 * i + 1] = o[1] / o[3];
    aos[3 * i + 2] = o[2] / o[3];
  }
}
void launch_transform_cloud(double* aos, size_t n, const double M[16], hipStream_t s) {
  if (n == 0) return;
  Mat16 m;
  for (int i = 0; i < 16; ++i) m.m[i] = M[i];
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_transform, dim3(blocks), dim3(256), 0, s, aos, n, m);
}

}  // namespace tl
