/*
 * tloam_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.  See tloam_oracle.h for the scope,
 * the "parity unpinned" statement and the provenance of every third-party behaviour.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Plain C99 + libm (+ optional OpenMP for the timed CPU baseline).
 */
#define _POSIX_C_SOURCE 200809L   /* posix_memalign, pthreads (the file is -std=c99) */
#include "tloam_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#include <pthread.h>
#endif

#define SOPHUS_EPS 1e-10 /* sophus/common.hpp:93-95 Constants<double>::epsilon() */
#define ORC_PI 3.14159265358979323846

/* ============================================================================
 *  SE(3)  -- the ~150 lines of vendored Sophus that the hot path touches
 * ========================================================================== */

/* so3.hpp:583-619  SO3::expAndTheta */
void orc_so3_exp(const double w[3], double q[4], double* theta) {
  double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (theta_sq < SOPHUS_EPS * SOPHUS_EPS) {
    *theta = 0.0;
    double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    *theta = sqrt(theta_sq);
    double half = 0.5 * (*theta);
    imag = sin(half) / (*theta);
    real = cos(half);
  }
  q[0] = real;
  q[1] = imag * w[0];
  q[2] = imag * w[1];
  q[3] = imag * w[2];
}

/* so3.hpp:247-290  SO3::logAndTheta (atan-based) */
void orc_so3_log(const double q[4], double w[3], double* theta) {
  double squared_n = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double qw = q[0];
  double two_atan_nbyw_by_n;
  if (squared_n < SOPHUS_EPS * SOPHUS_EPS) {
    double squared_w = qw * qw;
    two_atan_nbyw_by_n = 2.0 / qw - (2.0 / 3.0) * squared_n / (qw * squared_w);
    *theta = 2.0 * squared_n / qw;
  } else {
    double n = sqrt(squared_n);
    if (fabs(qw) < SOPHUS_EPS) {
      two_atan_nbyw_by_n = (qw > 0.0) ? ORC_PI / n : -ORC_PI / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * atan(n / qw) / n;
    }
    *theta = two_atan_nbyw_by_n * n;
  }
  w[0] = two_atan_nbyw_by_n * q[1];
  w[1] = two_atan_nbyw_by_n * q[2];
  w[2] = two_atan_nbyw_by_n * q[3];
}

/* so3.hpp:671-680  SO3::hat, row-major 3x3 */
static void hat3(const double v[3], double M[9]) {
  M[0] = 0.0;   M[1] = -v[2]; M[2] = v[1];
  M[3] = v[2];  M[4] = 0.0;   M[5] = -v[0];
  M[6] = -v[1]; M[7] = v[0];  M[8] = 0.0;
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen Quaternion::toRotationMatrix (used by SO3::matrix()), row-major out */
static void quat_to_R(const double q[4], double R[9]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* se3.hpp:761-785  SE3::exp */
void orc_se3_exp(const double a[6], double q[4], double t[3]) {
  const double* ups = a;
  const double* om = a + 3;
  double theta;
  orc_so3_exp(om, q, &theta);
  double Om[9], Om2[9], V[9];
  hat3(om, Om);
  mat3_mul(Om, Om, Om2);
  if (theta < SOPHUS_EPS) {
    quat_to_R(q, V); /* "V = so3.matrix()" */
  } else {
    double theta_sq = theta * theta;
    double c1 = (1.0 - cos(theta)) / theta_sq;
    double c2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = c1 * Om[i] + c2 * Om2[i];
    V[0] += 1.0; V[4] += 1.0; V[8] += 1.0;
  }
  mat3_vec(V, ups, t);
}

/* se3.hpp:223-256  SE3::log */
void orc_se3_log(const double q[4], const double t[3], double a[6]) {
  double theta;
  orc_so3_log(q, a + 3, &theta);
  double Om[9], Om2[9], Vinv[9];
  hat3(a + 3, Om);
  mat3_mul(Om, Om, Om2);
  double c2;
  if (fabs(theta) < SOPHUS_EPS) {
    c2 = 1.0 / 12.0;
  } else {
    double half = 0.5 * theta;
    c2 = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
  }
  for (int i = 0; i < 9; ++i) Vinv[i] = -0.5 * Om[i] + c2 * Om2[i];
  Vinv[0] += 1.0; Vinv[4] += 1.0; Vinv[8] += 1.0;
  mat3_vec(Vinv, t, a);
}

/* so3.hpp:358-367 (quaternion sandwich) + se3.hpp:321-324 */
void orc_se3_act(const double q[4], const double t[3], const double p[3], double out[3]) {
  double uv[3], c2[3];
  cross3(q + 1, p, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(q + 1, uv, c2);
  out[0] = (p[0] + q[0] * uv[0] + c2[0]) + t[0];
  out[1] = (p[1] + q[0] * uv[1] + c2[1]) + t[1];
  out[2] = (p[2] + q[0] * uv[2] + c2[2]) + t[2];
}

/* so3.hpp:325-340 + se3.hpp:304-309: (q1,t1)*(q2,t2) */
static void se3_mul(const double q1[4], const double t1[3], const double q2[4], const double t2[3],
                    double q[4], double t[3]) {
  const double aw = q1[0], ax = q1[1], ay = q1[2], az = q1[3];
  const double bw = q2[0], bx = q2[1], by = q2[2], bz = q2[3];
  double zero[3] = {0, 0, 0}, rt[3];
  orc_se3_act(q1, zero, t2, rt);
  q[0] = aw * bw - ax * bx - ay * by - az * bz;
  q[1] = aw * bx + ax * bw + ay * bz - az * by;
  q[2] = aw * by + ay * bw + az * bx - ax * bz;
  q[3] = aw * bz + az * bw + ax * by - ay * bx;
  t[0] = t1[0] + rt[0];
  t[1] = t1[1] + rt[1];
  t[2] = t1[2] + rt[2];
  /* SO3Base::operator* normalises only if the squared norm drifts (so3.hpp multiplication
   * returns the raw product; SO3 ctor from a quaternion product normalises it). */
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double n = sqrt(n2);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* se3.hpp:497-504 SE3(Matrix4) -> so3.hpp:469-474 SO3(Matrix3) -> Eigen Quaternion(Matrix3).
 * Returns TLOAM_E_BAD_POSE where the reference would SOPHUS_ENSURE-abort. */
int orc_se3_from_matrix(const double M[16], double q[4], double t[3]) {
  /* column-major: M[c*4+r] */
#define MM(r, c) M[(c) * 4 + (r)]
  double lr = MM(3, 0) * MM(3, 0) + MM(3, 1) * MM(3, 1) + MM(3, 2) * MM(3, 2) +
              (MM(3, 3) - 1.0) * (MM(3, 3) - 1.0);
  if (!(lr < SOPHUS_EPS)) return TLOAM_E_BAD_POSE;
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = MM(r, c);
  /* isOrthogonal: ||R R^T - I||_F < eps ; det > 0 */
  double f2 = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * R[j * 3 + k];
      s -= (i == j) ? 1.0 : 0.0;
      f2 += s * s;
    }
  if (!(sqrt(f2) < SOPHUS_EPS)) return TLOAM_E_BAD_POSE;
  double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
               R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (!(det > 0.0)) return TLOAM_E_BAD_POSE;
  /* Eigen quaternion-from-matrix (Shoemake) */
  double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (R[7] - R[5]) * s;
    q[2] = (R[2] - R[6]) * s;
    q[3] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
  }
  t[0] = MM(0, 3); t[1] = MM(1, 3); t[2] = MM(2, 3);
#undef MM
  return TLOAM_OK;
}

void orc_se3_to_matrix(const double q[4], const double t[3], double M[16]) {
  double R[9];
  quat_to_R(q, R);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) M[c * 4 + r] = R[r * 3 + c];
  M[3] = M[7] = M[11] = 0.0;
  M[12] = t[0]; M[13] = t[1]; M[14] = t[2]; M[15] = 1.0;
}

/* registration.cpp:162-173  PoseSE3Parameterization::Plus : log(exp(delta) * exp(x)) */
void orc_plus(const double x[6], const double delta[6], double out[6]) {
  double qx[4], tx[3], qd[4], td[3], q[4], t[3];
  orc_se3_exp(x, qx, tx);
  orc_se3_exp(delta, qd, td);
  se3_mul(qd, td, qx, tx, q, t);
  orc_se3_log(q, t, out);
}

/* ============================================================================
 *  fitBestPlane  registration.cpp:303-368
 * ========================================================================== */
void orc_fit_plane(const double* pts, int n, double plane[4]) {
  if (n <= 0) { plane[0] = plane[1] = plane[2] = plane[3] = 0.0; return; }
  double total = (double)n;
  double c[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) { c[0] += pts[3 * i]; c[1] += pts[3 * i + 1]; c[2] += pts[3 * i + 2]; }
  c[0] /= total; c[1] /= total; c[2] /= total;
  double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
  for (int i = 0; i < n; ++i) {
    double a = pts[3 * i] - c[0], b = pts[3 * i + 1] - c[1], g = pts[3 * i + 2] - c[2];
    xx += a * a; xy += a * b; xz += a * g; yy += b * b; yz += b * g; zz += g * g;
  }
  xx /= total; xy /= total; xz /= total; yy /= total; yz /= total; zz /= total;
  double wd[3] = {0, 0, 0};
  {
    double det_x = yy * zz - yz * yz;
    double ax[3] = {det_x, xz * yz - xy * zz, xy * yz - xz * yy};
    double w = det_x * det_x;
    if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0) w = -w;
    wd[0] += ax[0] * w; wd[1] += ax[1] * w; wd[2] += ax[2] * w;
  }
  {
    double det_y = xx * zz - xz * xz;
    double ax[3] = {xz * yz - xy * zz, det_y, xy * xz - yz * xx};
    double w = det_y * det_y;
    if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0) w = -w;
    wd[0] += ax[0] * w; wd[1] += ax[1] * w; wd[2] += ax[2] * w;
  }
  {
    double det_z = xx * yy - xy * xy;
    double ax[3] = {xy * yz - xz * yy, xy * xz - yz * xx, det_z};
    double w = det_z * det_z;
    if (wd[0] * ax[0] + wd[1] * ax[1] + wd[2] * ax[2] < 0.0) w = -w;
    wd[0] += ax[0] * w; wd[1] += ax[1] * w; wd[2] += ax[2] * w;
  }
  double norm = sqrt(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
  if (norm == 0.0) { plane[0] = plane[1] = plane[2] = plane[3] = 0.0; return; }
  wd[0] /= norm; wd[1] /= norm; wd[2] /= norm;
  plane[0] = wd[0]; plane[1] = wd[1]; plane[2] = wd[2];
  plane[3] = -(wd[0] * c[0] + wd[1] * c[1] + wd[2] * c[2]);
}

/* ============================================================================
 *  3x3 symmetric eigen-decomposition (stands in for Eigen SelfAdjointEigenSolver,
 *  registration.cpp:476-479): cyclic Jacobi; eigenvalues ascending, unit eigenvectors in
 *  the COLUMNS of V (row-major storage V[r*3+c]).  Sign of an eigenvector is arbitrary
 *  (irrelevant downstream: swapping line points a<->b leaves H, g and the cost unchanged).
 * ========================================================================== */
void orc_eig3_sym(const double cov[9], double ev[3], double V[9]) {
  double A[9];
  memcpy(A, cov, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  /* entries whose squares leave the fp64 range: an exact power-of-two rescale first (tl_knn.hpp eig3_sym has the same branch) */
  int rescale = 0;
  {
    double mx = 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 3; ++j) mx = fmax(mx, fabs(A[i * 3 + j]));
    if (mx != 0.0 && mx < INFINITY && (mx < 1e-120 || mx > 1e120)) {
      (void)frexp(mx, &rescale);
      double f = ldexp(1.0, -rescale);
      for (int i = 0; i < 9; ++i) A[i] *= f;
    }
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-36 * dia) break; /* off-diagonal mass below fp64 resolution of the eigenvalues */
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double app = A[p * 3 + p], aqq = A[q * 3 + q];
        /* the rotation that annihilates a_pq: t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), tau = (a_qq - a_pp) / (2 a_pq),
         * c = 1 / sqrt(1 + t^2), s = t c -- written without tau and t: with d = a_qq - a_pp, b = 2 a_pq, h = hypot(d, b),
         * u = |d| + h:  t = sgn |b| / u,  c = u / hypot(u, b),  s = sgn |b| / hypot(u, b).  Same rotation, and the
         * dependent chain is two square roots and one division long instead of three divisions and two square roots
         * (the device restates exactly this, tl_knn.hpp: the per-query fit of K1 is one such chain after another). */
        double d = aqq - app, b = 2.0 * apq;
        double h = sqrt(d * d + b * b);
        double u = fabs(d) + h;
        double r = sqrt(u * u + b * b);
        if (!(r > 0.0 && r < INFINITY)) { /* d*d + b*b under-/overflowed: the rotation depends on d : b only (tl_knn.hpp has the same branch) */
          double sc = fmax(fabs(d), fabs(b));
          d /= sc; b /= sc;
          h = sqrt(d * d + b * b);
          u = fabs(d) + h;
          r = sqrt(u * u + b * b);
        }
        double cs = u / r, sn = fabs(b) / r;
        if (!(d == 0.0 || (d > 0.0) == (b > 0.0))) sn = -sn; /* sgn(tau), tau = +-0 counting as positive */
        /* A <- G^T A G */
        for (int k = 0; k < 3; ++k) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = cs * akp - sn * akq;
          A[k * 3 + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = cs * apk - sn * aqk;
          A[q * 3 + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = cs * vkp - sn * vkq;
          V[k * 3 + q] = sn * vkp + cs * vkq;
        }
      }
  }
  ev[0] = A[0]; ev[1] = A[4]; ev[2] = A[8];
  if (rescale != 0)
    for (int i = 0; i < 3; ++i) ev[i] = ldexp(ev[i], rescale);
  /* sort ascending (3 elements) */
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (ev[j] > ev[j + 1]) {
        double tmp = ev[j]; ev[j] = ev[j + 1]; ev[j + 1] = tmp;
        for (int k = 0; k < 3; ++k) {
          double tv = V[k * 3 + j]; V[k * 3 + j] = V[k * 3 + j + 1]; V[k * 3 + j + 1] = tv;
        }
      }
}

/* ============================================================================
 *  KDTreeFlann::SearchHybrid  (Open3D 0.12 / nanoflann)
 *  exact k-NN, ascending squared distance, then keep entries with d2 < radius^2.
 *  nanoflann L2_Simple_Adaptor: d2 = sum_i (a_i-b_i)^2 accumulated in dimension order.
 *  Ties (measure zero) are broken towards the lower target index.
 * ========================================================================== */
static inline double dist2(const double* a, const double* b) {
  double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  double r = d0 * d0;
  r += d1 * d1;
  r += d2 * d2;
  return r;
}
static inline void topk_insert(int k, int* cnt, int* idx, double* d2, int i, double d) {
  /* nanoflann only offers its result set a candidate whose distance is below the set's worst, which starts at the largest
   * double (KNNResultSet::worstDist): a NaN or infinite distance is never admitted -- also not while the set is still filling
   * (round 5: it used to be appended then, and the finite candidates that came after it stayed behind it) */
  if (!(d < DBL_MAX)) return;
  int n = *cnt;
  if (n == k && !(d < d2[n - 1] || (d == d2[n - 1] && i < idx[n - 1]))) return;
  int pos = (n < k) ? n : k - 1;
  while (pos > 0 && (d < d2[pos - 1] || (d == d2[pos - 1] && i < idx[pos - 1]))) {
    d2[pos] = d2[pos - 1];
    idx[pos] = idx[pos - 1];
    --pos;
  }
  d2[pos] = d;
  idx[pos] = i;
  if (n < k) *cnt = n + 1;
}
static int radius_cut(int cnt, const double* d2, double radius) {
  double r2 = radius * radius;
  int m = 0;
  while (m < cnt && d2[m] < r2) ++m;
  return m;
}

int orc_knn_hybrid_brute(const double* tgt, int n, const double q[3], double radius, int k, int* idx,
                         double* d2) {
  if (n <= 0 || k <= 0) return -1;
  int cnt = 0;
  for (int i = 0; i < n; ++i) topk_insert(k, &cnt, idx, d2, i, dist2(q, tgt + 3 * i));
  return radius_cut(cnt, d2, radius);
}

/* uniform grid (the oracle's stand-in for the kd-tree: same results, tractable at 1 M) */
typedef struct {
  int n;
  const double* pts; /* borrowed AoS */
  double cell, org[3];
  int dim[3];
  int* start; /* ncell+1 */
  int* order; /* target indices grouped by cell, ascending index within a cell */
} orc_grid;

static void grid_free(orc_grid* g) {
  free(g->start);
  free(g->order);
  memset(g, 0, sizeof(*g));
}
static inline int grid_coord(const orc_grid* g, double v, int ax) {
  double f = floor((v - g->org[ax]) / g->cell);
  if (!(f >= -2.0)) f = -2.0; /* (also a NaN coordinate: some cell, never a neighbour) */
  if (f > (double)g->dim[ax] + 1.0) f = (double)g->dim[ax] + 1.0;
  return (int)f;
}
static void grid_build(orc_grid* g, const double* pts, int n, double radius) {
  grid_free(g);
  g->n = n;
  g->pts = pts;
  if (n <= 0) return;
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      double v = pts[3 * i + a];
      if (!(fabs(v) < INFINITY)) continue; /* only finite coordinates extend the box (as k_bbox_all): a point with a NaN or
                                            * infinite coordinate still gets a cell and is never anybody's neighbour */
      if (v < lo[a]) lo[a] = v;
      if (v > hi[a]) hi[a] = v;
    }
  for (int a = 0; a < 3; ++a)
    if (lo[a] > hi[a]) lo[a] = hi[a] = 0.0; /* no finite coordinate on this axis */
  double cell = radius * (1.0 + 1e-6);
  for (;;) {
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) cells *= floor((hi[a] - lo[a]) / cell) + 1.0;
    if (cells <= 3.2e7) break;
    cell *= 1.5;
  }
  g->cell = cell;
  for (int a = 0; a < 3; ++a) {
    g->org[a] = lo[a];
    g->dim[a] = (int)floor((hi[a] - lo[a]) / cell) + 1;
  }
  size_t ncell = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
  g->start = (int*)calloc(ncell + 1, sizeof(int));
  g->order = (int*)malloc(sizeof(int) * (size_t)n);
  int* cid = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    int cx = grid_coord(g, pts[3 * i], 0), cy = grid_coord(g, pts[3 * i + 1], 1),
        cz = grid_coord(g, pts[3 * i + 2], 2);
    if (cx >= g->dim[0]) cx = g->dim[0] - 1;
    if (cy >= g->dim[1]) cy = g->dim[1] - 1;
    if (cz >= g->dim[2]) cz = g->dim[2] - 1;
    if (cx < 0) cx = 0;
    if (cy < 0) cy = 0;
    if (cz < 0) cz = 0;
    cid[i] = (cz * g->dim[1] + cy) * g->dim[0] + cx;
    g->start[cid[i] + 1]++;
  }
  for (size_t c = 0; c < ncell; ++c) g->start[c + 1] += g->start[c];
  int* fill = (int*)malloc(sizeof(int) * ncell);
  memcpy(fill, g->start, sizeof(int) * ncell);
  for (int i = 0; i < n; ++i) g->order[fill[cid[i]]++] = i;
  free(fill);
  free(cid);
}
static int grid_knn_hybrid(const orc_grid* g, const double q[3], double radius, int k, int* idx,
                           double* d2) {
  if (g->n <= 0 || k <= 0) return -1;
  int cnt = 0;
  int reach = (int)ceil(radius / g->cell);
  if (reach < 1) reach = 1;
  int c[3];
  for (int a = 0; a < 3; ++a) c[a] = grid_coord(g, q[a], a);
  for (int z = c[2] - reach; z <= c[2] + reach; ++z) {
    if (z < 0 || z >= g->dim[2]) continue;
    for (int y = c[1] - reach; y <= c[1] + reach; ++y) {
      if (y < 0 || y >= g->dim[1]) continue;
      int x0 = c[0] - reach, x1 = c[0] + reach;
      if (x0 < 0) x0 = 0;
      if (x1 >= g->dim[0]) x1 = g->dim[0] - 1;
      if (x0 > x1) continue;
      size_t base = ((size_t)z * g->dim[1] + y) * g->dim[0];
      int s = g->start[base + x0], e = g->start[base + x1 + 1];
      for (int j = s; j < e; ++j) {
        int i = g->order[j];
        topk_insert(k, &cnt, idx, d2, i, dist2(q, g->pts + 3 * i));
      }
    }
  }
  /* NB: the k nearest within the 27-cell neighbourhood that pass the radius cut are exactly
   * the k nearest overall that pass it, because every point with d2 < radius^2 lies in the
   * neighbourhood (cell >= radius). */
  return radius_cut(cnt, d2, radius);
}

/* ============================================================================
 *  context
 * ========================================================================== */
typedef struct {
  int n, cap;
  int32_t* idx; /* source index */
  double *p, *a, *b, *d, *w; /* p,a,b: AoS 3n */
  double* cost;              /* side channel of pre-built sets (no source slot) */
} orc_rset;

static int kind_res_type(int kind) {
  return (kind == TLOAM_KIND_EDGE) ? TLOAM_RES_LINE : (kind == TLOAM_KIND_SPHERE) ? TLOAM_RES_POINT : TLOAM_RES_PLANE;
}

struct orc_ctx {
  tloam_tls_config cfg;
  int builder_threads, eval_threads;
  int eval_grain;   /* residual blocks a thread of the evaluator gets at least (0: no lower bound), orc_set_eval_grain */
  double* src[4]; int nsrc[4];
  double* tgt[4]; int ntgt[4];
  double* tree_pts[4]; int tree_n[4]; /* deep copy at SetGeometry (registration.cpp:898-913) */
  orc_grid grid[4];
  double grid_radius[4];
  int trees_built;
  /* per-source-point state (registration.cpp:931-949) */
  double* weights[4];
  double* resid[4]; /* the `mutable double* cost` side channel targets */
  orc_rset set[4];
  int prebuilt;
  /* scanMatching state */
  double x[6]; /* `parameters` / se3_pose_ */
  int active, iter;
  double mu, noise_bound_sq;
  double prev_cost[4], cur_cost[4];
  tloam_stats stats;
  int bad_weights;
  double last_H[36], last_g[6], last_cost;
  /* per-minimiser-iteration trace of every Solve since sm_begin / orc_solve (orc_get_trace): what a ceres::IterationSummary
   * + the state after the iteration would show -- held against the REFERENCE's own Solves when tests/golden_ref/ exists */
  double trace[ORC_TRACE_ROWS][ORC_TRACE_COLS];
  int ntrace, nsolve;
  struct orc_pool* pool; /* the evaluator's persistent worker threads (eval_threads > 1), see eval_pool_* */
};

static void rset_reserve(orc_rset* s, int cap) {
  if (cap <= s->cap) return;
  s->idx = (int32_t*)realloc(s->idx, sizeof(int32_t) * (size_t)cap);
  s->p = (double*)realloc(s->p, sizeof(double) * 3 * (size_t)cap);
  s->a = (double*)realloc(s->a, sizeof(double) * 3 * (size_t)cap);
  s->b = (double*)realloc(s->b, sizeof(double) * 3 * (size_t)cap);
  s->d = (double*)realloc(s->d, sizeof(double) * (size_t)cap);
  s->w = (double*)realloc(s->w, sizeof(double) * (size_t)cap);
  s->cost = (double*)realloc(s->cost, sizeof(double) * (size_t)cap);
  s->cap = cap;
}
static void rset_free(orc_rset* s) {
  free(s->idx); free(s->p); free(s->a); free(s->b); free(s->d); free(s->w); free(s->cost);
  memset(s, 0, sizeof(*s));
}

static void eval_pool_destroy(orc_ctx* c);

int orc_create(const tloam_tls_config* cfg, orc_ctx** out) {
  if (!cfg || !out) return TLOAM_E_INVALID;
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  c->cfg = *cfg;
  c->builder_threads = 1;
  c->eval_threads = 1;
  c->eval_grain = 256;
  *out = c;
  return TLOAM_OK;
}
void orc_destroy(orc_ctx* c) {
  if (!c) return;
  for (int k = 0; k < 4; ++k) {
    free(c->src[k]); free(c->tgt[k]); free(c->tree_pts[k]); free(c->weights[k]); free(c->resid[k]);
    grid_free(&c->grid[k]);
    rset_free(&c->set[k]);
  }
  eval_pool_destroy(c);
  free(c);
}
void orc_set_threads(orc_ctx* c, int builder_threads, int eval_threads) {
  c->builder_threads = builder_threads < 1 ? 1 : builder_threads;
  eval_threads = eval_threads < 1 ? 1 : eval_threads;
  if (eval_threads != c->eval_threads) eval_pool_destroy(c);   /* (re)created at the next evaluation */
  c->eval_threads = eval_threads;
}
/* The evaluator's team for one sweep = min(eval_threads, blocks / grain), at least 1: Ceres itself has no such bound (every
 * ParallelFor goes to all num_threads workers), and at the reference's shape -- 128 threads for the 5.9 k blocks of a KITTI-cap
 * frame, 46 blocks each -- waking and collecting the team costs three times the work (19.5 against 6.2 ms per frame on the
 * 256-core bench host with all 128 taking part, round 5).  grain = 0 switches the bound off. */
void orc_set_eval_grain(orc_ctx* c, int blocks_per_thread) { c->eval_grain = blocks_per_thread < 0 ? 0 : blocks_per_thread; }
static int set_cloud(double** dst, int* n_out, const double* xyz, size_t n) {
  free(*dst);
  *dst = NULL;
  *n_out = 0;
  if (n > 0) {
    if (!xyz) return TLOAM_E_INVALID;
    *dst = (double*)malloc(sizeof(double) * 3 * n);
    memcpy(*dst, xyz, sizeof(double) * 3 * n);
    *n_out = (int)n;
  }
  return TLOAM_OK;
}
/* registration.cpp:232-239 */
int orc_set_source(orc_ctx* c, int kind, const double* xyz, size_t n) {
  if (!c || kind < 0 || kind >= 4) return TLOAM_E_INVALID;
  return set_cloud(&c->src[kind], &c->nsrc[kind], xyz, n);
}
/* registration.cpp:241-248 */
int orc_set_target(orc_ctx* c, int kind, const double* xyz, size_t n) {
  if (!c || kind < 0 || kind >= 4) return TLOAM_E_INVALID;
  return set_cloud(&c->tgt[kind], &c->ntgt[kind], xyz, n);
}

static double kind_radius(const orc_ctx* c, int kind) {
  switch (kind) {
    case TLOAM_KIND_PLANAR: return c->cfg.planar_dist_thres;
    case TLOAM_KIND_GROUND: return c->cfg.ground_dist_thres;
    case TLOAM_KIND_EDGE: return c->cfg.edge_dist_thres;
    default: return c->cfg.sphere_dist_thres;
  }
}
static int kind_maxnum(const orc_ctx* c, int kind) {
  switch (kind) {
    case TLOAM_KIND_PLANAR: return c->cfg.planar_maxnum;
    case TLOAM_KIND_GROUND: return c->cfg.ground_maxnum;
    case TLOAM_KIND_EDGE: return c->cfg.edge_maxnum;
    default: return c->cfg.sphere_maxnum;
  }
}
static int kind_active(const orc_ctx* c, int kind) {
  /* registration.cpp:979-1016 : factor_num 4 -> all; 3 -> planar,ground,edge; 2 -> planar,ground */
  int f = c->cfg.factor_num;
  if (f == 4) return 1;
  if (f == 3) return kind != TLOAM_KIND_SPHERE;
  if (f == 2) return kind == TLOAM_KIND_PLANAR || kind == TLOAM_KIND_GROUND;
  return 0;
}

/* ============================================================================
 *  correspondence builders  registration.cpp:427-505 (edge), :517-559 (sphere),
 *  :571-635 (planar), :714-778 (ground)
 * ========================================================================== */
static void build_kind(orc_ctx* c, int kind) {
  double q[4], t[3];
  orc_se3_exp(c->x, q, t); /* T = exp(se3_pose_)  :434 :524 :578 :721 */
  orc_rset* s = &c->set[kind];
  s->n = 0;
  const int n = c->nsrc[kind];
  const double radius = kind_radius(c, kind);
  const int maxnum = kind_maxnum(c, kind);
  const orc_grid* g = &c->grid[kind];
  const double* tp = c->tree_pts[kind];
  int num = 0; /* edge_num / surf_num / ground_num / sphere_sum */
  for (int i = 0; i < n; ++i) {
    const double* src = c->src[kind] + 3 * i;
    double pw[3];
    orc_se3_act(q, t, src, pw);
    int idx[5];
    double d2[5];
    if (kind == TLOAM_KIND_SPHERE) {
      /* :535-551 */
      if (grid_knn_hybrid(g, pw, radius, 1, idx, d2) > 0) {
        if (d2[0] > 0.2) continue; /* :536 compares a SQUARED distance with 0.2 */
        if (num >= maxnum) return;  /* :538 */
        rset_reserve(s, s->n + 1 > 64 ? (s->n + 1) * 2 : 64);
        int j = s->n++;
        s->idx[j] = i;
        memcpy(s->p + 3 * j, src, 24);
        memcpy(s->a + 3 * j, tp + 3 * idx[0], 24);
        s->w[j] = c->weights[kind][i]; /* captured by value :544 / registration.hpp:51 */
        s->cost[j] = 0.0;
      }
      num++; /* :551 -- counts every source point that did not `continue` */
      continue;
    }
    int cnt = grid_knn_hybrid(g, pw, radius, 5, idx, d2);
    if (cnt <= 0) continue;
    if (kind == TLOAM_KIND_EDGE) {
      if (cnt <= 3) continue;       /* :445 */
      if (num >= maxnum) return;    /* :448 */
      double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int m = 0; m < cnt; ++m) { /* :453-464 */
        const double* pt = tp + 3 * idx[m];
        cum[0] += pt[0]; cum[1] += pt[1]; cum[2] += pt[2];
        cum[3] += pt[0] * pt[0]; cum[4] += pt[0] * pt[1]; cum[5] += pt[0] * pt[2];
        cum[6] += pt[1] * pt[1]; cum[7] += pt[1] * pt[2]; cum[8] += pt[2] * pt[2];
      }
      for (int m = 0; m < 9; ++m) cum[m] /= (double)cnt; /* :465 */
      double cov[9];
      cov[0] = cum[3] - cum[0] * cum[0];
      cov[4] = cum[6] - cum[1] * cum[1];
      cov[8] = cum[8] - cum[2] * cum[2];
      cov[1] = cov[3] = cum[4] - cum[0] * cum[1];
      cov[2] = cov[6] = cum[5] - cum[0] * cum[2];
      cov[5] = cov[7] = cum[7] - cum[1] * cum[2];
      double ev[3], V[9];
      orc_eig3_sym(cov, ev, V);
      double dir[3] = {V[2], V[5], V[8]}; /* eigenvectors().col(2) :479 */
      if (ev[2] > 3 * ev[1] && fabs(dir[2]) > c->cfg.edge_dir_thres) { /* :481 */
        rset_reserve(s, s->n + 1 > 64 ? (s->n + 1) * 2 : 64);
        int j = s->n++;
        s->idx[j] = i;
        memcpy(s->p + 3 * j, src, 24);
        for (int m = 0; m < 3; ++m) {
          s->a[3 * j + m] = 0.1 * dir[m] + cum[m];  /* :483 */
          s->b[3 * j + m] = -0.1 * dir[m] + cum[m]; /* :484 */
        }
        s->w[j] = c->weights[kind][i];
        s->cost[j] = 0.0;
        num++; /* :492 */
      }
    } else {
      if (cnt <= 4) continue;    /* :589 / :732 */
      if (num >= maxnum) return; /* :592 / :735 */
      double near[15];
      for (int m = 0; m < 5; ++m) memcpy(near + 3 * m, tp + 3 * idx[m], 24);
      double plane[4];
      orc_fit_plane(near, 5, plane);
      int valid = 1;
      for (int m = 0; m < 5; ++m) { /* :605-613 signed test, no fabs */
        double dis = plane[0] * near[3 * m] + plane[1] * near[3 * m + 1] + plane[2] * near[3 * m + 2] + plane[3];
        if (dis > 0.2) { valid = 0; break; }
      }
      if (valid) {
        rset_reserve(s, s->n + 1 > 64 ? (s->n + 1) * 2 : 64);
        int j = s->n++;
        s->idx[j] = i;
        memcpy(s->p + 3 * j, src, 24);
        memcpy(s->a + 3 * j, plane, 24);
        s->d[j] = plane[3];
        s->w[j] = c->weights[kind][i];
        s->cost[j] = 0.0;
        num++; /* :623 / :766 */
      }
    }
  }
}

/* ============================================================================
 *  cost functors registration.cpp:19-47 / :55-88 / :96-117 evaluated through the Ceres
 *  residual-block path (residual_block.cc + CauchyLoss(1.0) + Corrector, SURVEY B.1 EVAL)
 * ========================================================================== */
typedef struct {
  double cost;
  double g[6];
  double H[36];
} orc_normal;

static void eval_block(int res_type, const double q[4], const double t[3], const double* p,
                       const double* a, const double* b, double d, double w, double* side_cost,
                       int want_J, orc_normal* acc) {
  double pw[3];
  orc_se3_act(q, t, p, pw);
  double r[3] = {0, 0, 0};
  double J[18]; /* row-major nres x 6 */
  int nres;
  double hat_pw[9];
  hat3(pw, hat_pw);
  if (res_type == TLOAM_RES_POINT) {
    nres = 3;
    for (int i = 0; i < 3; ++i) r[i] = (a[i] - pw[i]) * w;             /* :26-30 */
    *side_cost = pow(r[0] + r[1] + r[2], 2);                           /* :32 */
    if (want_J)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          J[i * 6 + j] = -((i == j) ? 1.0 : 0.0) * w;                  /* :39 */
          J[i * 6 + 3 + j] = hat_pw[i * 3 + j] * w;                    /* :40 */
        }
  } else if (res_type == TLOAM_RES_LINE) {
    nres = 3;
    double da[3] = {pw[0] - a[0], pw[1] - a[1], pw[2] - a[2]};
    double db[3] = {pw[0] - b[0], pw[1] - b[1], pw[2] - b[2]};
    double nu[3];
    cross3(da, db, nu);                                                 /* :62 */
    double de[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double den = sqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);
    for (int i = 0; i < 3; ++i) r[i] = nu[i] / den * w;                /* :65-67 */
    *side_cost = pow(r[0] + r[1] + r[2], 2);                           /* :69 */
    if (want_J) {
      double A[18]; /* dt_by_se3 = [I*w, -hat(pw)*w]  :77-78 */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A[i * 6 + j] = ((i == j) ? 1.0 : 0.0) * w;
          A[i * 6 + 3 + j] = -hat_pw[i * 3 + j] * w;
        }
      double re[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
      double S[9];
      hat3(re, S);                                                      /* :80-81 */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) {
          double s = 0.0;
          for (int k = 0; k < 3; ++k) s += S[i * 3 + k] * A[k * 6 + j];
          J[i * 6 + j] = s / den;                                       /* :83 */
        }
    }
  } else {
    nres = 1;
    r[0] = (a[0] * pw[0] + a[1] * pw[1] + a[2] * pw[2]) + d;           /* :100 (unweighted) */
    *side_cost = pow(r[0], 2);                                          /* :101 */
    if (want_J)
      for (int j = 0; j < 3; ++j) {
        J[j] = a[j] * w;                                                /* n^T (I w) :109,:112 */
        J[3 + j] = -(a[0] * hat_pw[0 * 3 + j] + a[1] * hat_pw[1 * 3 + j] + a[2] * hat_pw[2 * 3 + j]) * w; /* :110 */
      }
  }
  /* residual_block.cc: squared norm -> CauchyLoss(1.0): rho0 = log(1+s), rho1 = 1/(1+s),
   * rho2 = -rho1^2 <= 0 -> Corrector clamped branch: r *= sqrt(rho1), J *= sqrt(rho1). */
  double s = 0.0;
  for (int i = 0; i < nres; ++i) s += r[i] * r[i];
  double sum = 1.0 + s;
  double inv = 1.0 / sum;
  acc->cost += 0.5 * log(sum);
  if (!want_J) return;
  double rho1 = inv > DBL_MIN ? inv : DBL_MIN;
  double sr = sqrt(rho1);
  for (int i = 0; i < nres; ++i) {
    double ri = r[i] * sr;
    double Ji[6];
    for (int j = 0; j < 6; ++j) Ji[j] = J[i * 6 + j] * sr;
    for (int j = 0; j < 6; ++j) {
      acc->g[j] += Ji[j] * ri;
      for (int k = 0; k < 6; ++k) acc->H[j * 6 + k] += Ji[j] * Ji[k];
    }
  }
}

/* ---- the evaluator's threads (Ceres: Solver::Options::num_threads, registration.cpp:1044 = hardware_concurrency()/2) ----------
 * Ceres keeps ONE persistent thread pool for the life of the problem's context and hands every evaluation to it; so does this:
 * eval_threads - 1 workers created once (orc_set_threads -> first evaluation), the calling thread is worker 0.  An evaluation
 * publishes a job number; the workers -- spinning for a short while between the evaluations of a Solve, asleep on a condition
 * variable otherwise -- each take a FIXED contiguous slice of the concatenated block list (static schedule), add into their own
 * cache-line-padded accumulator, and count themselves out; the caller folds the accumulators in thread order.
 * (Round 4 used one OpenMP region per evaluation: at the reference's thread shape -- 4 builder tasks, 128 evaluator threads on a
 * 256-core host -- libgomp tore down and respawned 124 threads every time the team size changed between the two regions, and a
 * frame took 136-223 ms against 7-8 ms at 4 threads; BENCH_r04.json reference_thread_shape.) */
typedef struct {
  orc_normal acc;
  char pad[64 - sizeof(orc_normal) % 64];
} orc_part;
typedef struct orc_pool {
  int n;                        /* threads incl. the caller */
  int n_active;                 /* threads that take a slice of the current job (orc_set_eval_grain); the others skip it */
  int active_of[8];             /* n_active by job number & 7: a worker that is NOT part of a job may look at it arbitrarily late
                                 * (nobody waits for it), so it must find the team size that belongs to the job number it read --
                                 * see eval_pool_worker */
  pthread_t* th;
  orc_part* part;               /* [n], 64-byte aligned */
  pthread_mutex_t mu;
  pthread_cond_t cv;            /* threads of the current team that went to sleep between two jobs */
  pthread_cond_t cv_idle;       /* threads the current team size leaves out: parked until the team grows past them (or stop) */
  int parked, last_active;      /* (under mu) */
  unsigned long job;            /* job number, advanced by the caller (atomic) */
  int pending;                  /* workers still in the current job (atomic) */
  int sleepers;                 /* workers asleep on cv (under mu) */
  int stop;
  /* the job */
  orc_ctx* c;
  double q[4], t[3];
  int want_J;
  int total;                    /* blocks of the four sets, concatenated */
} orc_pool;
typedef struct { orc_pool* P; int tid; } orc_worker_arg;

static void eval_pool_slice(orc_pool* P, int tid, int n_active) {
  orc_ctx* c = P->c;
  orc_normal* acc = &P->part[tid].acc;
  memset(acc, 0, sizeof(*acc));
  const long long lo = (long long)P->total * tid / n_active, hi = (long long)P->total * (tid + 1) / n_active;
  long long base = 0;
  for (int kind = 0; kind < 4; ++kind) {
    orc_rset* s = &c->set[kind];
    const int rt = kind_res_type(kind);
    const long long a = lo > base ? lo - base : 0, b = (hi - base < s->n) ? hi - base : s->n;
    for (long long j = a; j < b; ++j) {
      double sc;
      eval_block(rt, P->q, P->t, s->p + 3 * j, s->a + 3 * j, s->b + 3 * j, s->d[j], s->w[j], &sc, P->want_J, acc);
      s->cost[j] = sc;
      if (!c->prebuilt) c->resid[kind][s->idx[j]] = sc;
    }
    base += s->n;
  }
}
static void* eval_pool_worker(void* argp) {
  orc_worker_arg* arg = (orc_worker_arg*)argp;
  orc_pool* P = arg->P;
  const int tid = arg->tid;
  free(arg);
  unsigned long seen = 0;
  int idle = 0;   /* this thread had no slice in the last job it looked at */
  for (;;) {
    if (idle) {
      /* not part of the team at its current size: PARKED, on a condition of its own -- woken when the team grows, not by every
       * job (105 parked threads woken for each of the 19 sweeps of a frame, each taking the mutex twice, held the caller up by
       * ~1 ms per sweep on the bench host: 29.7 ms per frame at the reference's shape against 5.9 at sixteen threads) */
      pthread_mutex_lock(&P->mu);
      P->parked++;
      while (!P->stop && !(__atomic_load_n(&P->active_of[__atomic_load_n(&P->job, __ATOMIC_ACQUIRE) & 7ul], __ATOMIC_ACQUIRE) > tid))
        pthread_cond_wait(&P->cv_idle, &P->mu);
      P->parked--;
      pthread_mutex_unlock(&P->mu);
      idle = 0;
    }
    /* wait for the next job: spin a little (the evaluations of a Solve follow each other within ~100 us), then sleep */
    int spins = 0;
    while (__atomic_load_n(&P->job, __ATOMIC_ACQUIRE) == seen && !__atomic_load_n(&P->stop, __ATOMIC_ACQUIRE)) {
      if (++spins < 20000) { __builtin_ia32_pause(); continue; }
      pthread_mutex_lock(&P->mu);
      P->sleepers++;
      while (__atomic_load_n(&P->job, __ATOMIC_ACQUIRE) == seen && !P->stop) pthread_cond_wait(&P->cv, &P->mu);
      P->sleepers--;
      pthread_mutex_unlock(&P->mu);
      spins = 0;
    }
    if (__atomic_load_n(&P->stop, __ATOMIC_ACQUIRE)) return NULL;
    /* Which job, and am I part of it?  The caller only waits for the threads that ARE part of a job, so a thread that is not may
     * read the job number just before the caller publishes the next one, and the team size just after: it would then take a
     * slice of a job it was never counted for and count itself out of the next one twice (the caller's count passes zero
     * unseen: a hang, met on the 256-core bench host).  So the team size is kept per job number (a ring of eight), and the pair
     * is only believed if the job number still stands afterwards: a thread that is part of job j finds it standing for as long
     * as it has not counted itself out; one that is not either sees the pair of j or, having been overtaken, looks again. */
    const unsigned long j = __atomic_load_n(&P->job, __ATOMIC_ACQUIRE);
    const int act = __atomic_load_n(&P->active_of[j & 7ul], __ATOMIC_ACQUIRE);
    if (__atomic_load_n(&P->job, __ATOMIC_ACQUIRE) != j) continue;   /* (seen unchanged: the wait loop falls through) */
    seen = j;
    idle = tid >= act;
    if (idle) continue;
    eval_pool_slice(P, tid, act);
    __atomic_fetch_sub(&P->pending, 1, __ATOMIC_ACQ_REL);
  }
}
static void eval_pool_destroy(orc_ctx* c) {
  orc_pool* P = c->pool;
  if (!P) return;
  pthread_mutex_lock(&P->mu);
  __atomic_store_n(&P->stop, 1, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&P->cv);
  pthread_cond_broadcast(&P->cv_idle);
  pthread_mutex_unlock(&P->mu);
  for (int i = 1; i < P->n; ++i) pthread_join(P->th[i], NULL);
  pthread_mutex_destroy(&P->mu);
  pthread_cond_destroy(&P->cv);
  pthread_cond_destroy(&P->cv_idle);
  free(P->th);
  free(P->part);
  free(P);
  c->pool = NULL;
}
static void eval_pool_run(orc_ctx* c, const double q[4], const double t[3], int want_J, orc_normal* out) {
  orc_pool* P = c->pool;
  if (!P) {
    P = (orc_pool*)calloc(1, sizeof(orc_pool));
    P->n = c->eval_threads;
    P->th = (pthread_t*)calloc((size_t)P->n, sizeof(pthread_t));
    if (posix_memalign((void**)&P->part, 64, sizeof(orc_part) * (size_t)P->n) != 0) abort();
    pthread_mutex_init(&P->mu, NULL);
    pthread_cond_init(&P->cv, NULL);
    pthread_cond_init(&P->cv_idle, NULL);
    P->last_active = P->n;
    P->c = c;
    for (int i = 1; i < P->n; ++i) {
      orc_worker_arg* a = (orc_worker_arg*)malloc(sizeof(orc_worker_arg));
      a->P = P; a->tid = i;
      if (pthread_create(&P->th[i], NULL, eval_pool_worker, a) != 0) { free(a); P->n = i; break; }   /* fewer threads: still correct */
    }
    c->pool = P;
  }
  memcpy(P->q, q, sizeof(P->q));
  memcpy(P->t, t, sizeof(P->t));
  P->want_J = want_J;
  P->total = c->set[0].n + c->set[1].n + c->set[2].n + c->set[3].n;
  P->n_active = P->n;
  if (c->eval_grain > 0) {
    const int by_work = P->total / c->eval_grain;
    P->n_active = by_work < 1 ? 1 : (by_work < P->n ? by_work : P->n);
  }
  __atomic_store_n(&P->active_of[(P->job + 1ul) & 7ul], P->n_active, __ATOMIC_RELEASE);
  __atomic_store_n(&P->pending, P->n_active - 1, __ATOMIC_RELEASE);
  __atomic_fetch_add(&P->job, 1ul, __ATOMIC_ACQ_REL);
  pthread_mutex_lock(&P->mu);
  if (P->sleepers > 0) pthread_cond_broadcast(&P->cv);
  if (P->parked > 0 && P->n_active > P->last_active) pthread_cond_broadcast(&P->cv_idle);   /* the team grew */
  P->last_active = P->n_active;
  pthread_mutex_unlock(&P->mu);
  eval_pool_slice(P, 0, P->n_active);
  while (__atomic_load_n(&P->pending, __ATOMIC_ACQUIRE) > 0) __builtin_ia32_pause();
  for (int th = 0; th < P->n_active; ++th) {   /* fold in thread order */
    const orc_normal* a = &P->part[th].acc;
    out->cost += a->cost;
    for (int m = 0; m < 6; ++m) out->g[m] += a->g[m];
    for (int m = 0; m < 36; ++m) out->H[m] += a->H[m];
  }
}

/* One evaluator sweep over all blocks at x.  Side channel: `resid[kind][src]` for built sets
 * (registration.cpp:486 passes &edge_residuals_(i)), set.cost[] for pre-built sets. */
static void evaluate(orc_ctx* c, const double x[6], int want_J, orc_normal* out) {
  double q[4], t[3];
  orc_se3_exp(x, q, t); /* the reference does this once per block (:22,:58,:98); hoisted */
  memset(out, 0, sizeof(*out));
  if (c->eval_threads > 1) {
    eval_pool_run(c, q, t, want_J, out);
    return;
  }
  for (int kind = 0; kind < 4; ++kind) {
    orc_rset* s = &c->set[kind];
    int rt = kind_res_type(kind);
    int n = s->n;
    for (int j = 0; j < n; ++j) {
      double sc;
      eval_block(rt, q, t, s->p + 3 * j, s->a + 3 * j, s->b + 3 * j, s->d[j], s->w[j], &sc, want_J, out);
      s->cost[j] = sc;
      if (!c->prebuilt) c->resid[kind][s->idx[j]] = sc;
    }
  }
}

/* ============================================================================
 *  ceres::Solve as configured at registration.cpp:1036-1047
 *  (trust_region_minimizer.cc + dogleg_strategy.cc, Ceres 2.0 -- restated from the published
 *  algorithm).  Linear algebra is on the 6x6 normal equations
 *  (Cholesky) instead of Ceres' dense QR of the Nx6 Jacobian: same solution in exact
 *  arithmetic, ~1e-10 relative difference in fp64 with the Jacobi scaling applied.
 * ========================================================================== */
static int chol6_solve(const double A[36], const double b[6], double y[6]) {
  double L[36];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
    for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
    if (!(s > 0.0) || !isfinite(s)) return 0;
    double ljj = sqrt(s);
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
    if (!isfinite(y[i])) return 0;
  return 1;
}

typedef struct {
  double radius, mu;
  int reuse;
  double D[6];       /* diagonal_ */
  double grad[6];    /* gradient_ = D^-1 gs */
  double gn[6];      /* gauss_newton_step_ (D-scaled) */
  double alpha;      /* Cauchy step length */
  int subspace_1d;
  double U[12];      /* subspace_basis_ 6x2 (column 0, column 1) */
  double sg[2], sB[4];
  double step_norm;  /* dogleg_step_norm_ */
} orc_dogleg;

static double vnorm(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

/* minimise 0.5 x^T B x + g^T x on the circle |x| = r  (dogleg_strategy.cc
 * FindMinimumOnTrustRegionBoundary; Ceres finds the roots of a quartic, here the unique
 * global minimiser is bracketed by sampling and polished with Newton on the angle). */
static void min_on_circle(const double B[4], const double g[2], double r, double x[2]) {
  const int NS = 720;
  double best = DBL_MAX, bth = 0.0;
  for (int i = 0; i < NS; ++i) {
    double th = 2.0 * ORC_PI * i / NS;
    double cx = r * cos(th), sx = r * sin(th);
    double f = 0.5 * (B[0] * cx * cx + (B[1] + B[2]) * cx * sx + B[3] * sx * sx) + g[0] * cx + g[1] * sx;
    if (f < best) { best = f; bth = th; }
  }
  double lo = bth - 2.0 * ORC_PI / NS, hi = bth + 2.0 * ORC_PI / NS;
  /* f'(th) changes sign across the bracket; bisect on f' */
  for (int it = 0; it < 200; ++it) {
    double th = 0.5 * (lo + hi);
    double cx = r * cos(th), sx = r * sin(th);
    /* df/dth with dx/dth = (-sx, cx) */
    double gx = B[0] * cx + 0.5 * (B[1] + B[2]) * sx + g[0];
    double gy = 0.5 * (B[1] + B[2]) * cx + B[3] * sx + g[1];
    double df = gx * (-sx) + gy * cx;
    if (df > 0.0) hi = th; else lo = th;
    if (hi - lo < 1e-16 * (1.0 + fabs(th))) break;
  }
  double th = 0.5 * (lo + hi);
  x[0] = r * cos(th);
  x[1] = r * sin(th);
}

/* DoglegStrategy::ComputeStep for SUBSPACE_DOGLEG on the Jacobi-scaled system (Hs, gs).
 * returns 1 = LINEAR_SOLVER_SUCCESS, 0 = LINEAR_SOLVER_FAILURE.  step = scaled step. */
static int dogleg_compute_step(orc_dogleg* s, const double Hs[36], const double gs[6], double step[6]) {
  if (!s->reuse) {
    s->reuse = 1;
    for (int i = 0; i < 6; ++i) {
      double v = Hs[i * 6 + i];
      if (v < 1e-6) v = 1e-6;  /* min_diagonal_ */
      if (v > 1e32) v = 1e32;  /* max_diagonal_ */
      s->D[i] = sqrt(v);
    }
    for (int i = 0; i < 6; ++i) s->grad[i] = gs[i] / s->D[i]; /* ComputeGradient */
    { /* ComputeCauchyPoint: alpha = |grad|^2 / |J D^-2 ... |^2 = |grad|^2 / (v^T Hs v), v = D^-1 grad */
      double v[6], Hv = 0.0;
      for (int i = 0; i < 6; ++i) v[i] = s->grad[i] / s->D[i];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Hv += v[i] * Hs[i * 6 + j] * v[j];
      double gg = 0.0;
      for (int i = 0; i < 6; ++i) gg += s->grad[i] * s->grad[i];
      s->alpha = gg / Hv;
    }
    /* ComputeGaussNewtonStep: (Hs + mu D^2) y = gs, retry with mu*10 while mu < max_mu(1.0) */
    int ok = 0;
    double y[6];
    while (s->mu < 1.0) {
      double A[36];
      memcpy(A, Hs, sizeof(A));
      for (int i = 0; i < 6; ++i) A[i * 6 + i] += s->mu * s->D[i] * s->D[i];
      if (chol6_solve(A, gs, y)) { ok = 1; break; }
      s->mu *= 10.0;
    }
    if (!ok) return 0;
    for (int i = 0; i < 6; ++i) s->gn[i] = -s->D[i] * y[i];
    /* ComputeSubspaceModel: orthonormal basis of span{grad, gn} */
    double u0[6], u1[6];
    double n0 = vnorm(s->grad, 6), n1 = vnorm(s->gn, 6);
    if (n0 == 0.0 && n1 == 0.0) return 0; /* rank 0 */
    const double* first = (n0 >= n1) ? s->grad : s->gn; /* column pivoting: larger norm first */
    const double* second = (n0 >= n1) ? s->gn : s->grad;
    double nf = (n0 >= n1) ? n0 : n1, ns = (n0 >= n1) ? n1 : n0;
    for (int i = 0; i < 6; ++i) u0[i] = first[i] / nf;
    double proj = 0.0;
    for (int i = 0; i < 6; ++i) proj += u0[i] * second[i];
    for (int i = 0; i < 6; ++i) u1[i] = second[i] - proj * u0[i];
    double nr = vnorm(u1, 6);
    if (nr <= 1e-14 * (ns > 0 ? nf : 1.0) || ns == 0.0) {
      s->subspace_1d = 1;
    } else {
      s->subspace_1d = 0;
      for (int i = 0; i < 6; ++i) u1[i] /= nr;
      for (int i = 0; i < 6; ++i) { s->U[i] = u0[i]; s->U[6 + i] = u1[i]; }
      s->sg[0] = s->sg[1] = 0.0;
      for (int i = 0; i < 6; ++i) { s->sg[0] += u0[i] * s->grad[i]; s->sg[1] += u1[i] * s->grad[i]; }
      double v0[6], v1[6];
      for (int i = 0; i < 6; ++i) { v0[i] = u0[i] / s->D[i]; v1[i] = u1[i] / s->D[i]; }
      double b00 = 0, b01 = 0, b11 = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          b00 += v0[i] * Hs[i * 6 + j] * v0[j];
          b01 += v0[i] * Hs[i * 6 + j] * v1[j];
          b11 += v1[i] * Hs[i * 6 + j] * v1[j];
        }
      s->sB[0] = b00; s->sB[1] = b01; s->sB[2] = b01; s->sB[3] = b11;
    }
  }
  /* ComputeSubspaceDoglegStep */
  double gnn = vnorm(s->gn, 6);
  if (gnn <= s->radius) {
    for (int i = 0; i < 6; ++i) step[i] = s->gn[i] / s->D[i];
    s->step_norm = gnn;
    return 1;
  }
  if (s->subspace_1d) {
    double gnorm = vnorm(s->grad, 6);
    for (int i = 0; i < 6; ++i) step[i] = -(s->radius / gnorm) * s->grad[i] / s->D[i];
    s->step_norm = s->radius;
    return 1;
  }
  double m2[2];
  min_on_circle(s->sB, s->sg, s->radius, m2);
  for (int i = 0; i < 6; ++i) step[i] = (s->U[i] * m2[0] + s->U[6 + i] * m2[1]) / s->D[i];
  s->step_norm = s->radius;
  return 1;
}

static double grad_max_norm(const double x[6], const double g[6]) {
  /* || x - Plus(x, -g) ||_inf  (trust_region_minimizer.cc EvaluateGradientAndJacobian) */
  double ng[6], xp[6], m = 0.0;
  for (int i = 0; i < 6; ++i) ng[i] = -g[i];
  orc_plus(x, ng, xp);
  for (int i = 0; i < 6; ++i) {
    double d = fabs(x[i] - xp[i]);
    if (d > m) m = d;
  }
  return m;
}

/* one row of the trace: [0] Solve index, [1] minimiser iteration, [2] cost after the iteration (x_cost), [3] cost change
 * (x_cost - candidate cost), [4] step accepted, [5] trust-region radius after the iteration, [6] |x - x_candidate|,
 * [7] step quality (relative decrease), [8] gradient max norm at x, [9] 0 regular | 1 ended by the parameter tolerance |
 * 2 by the function tolerance | 3 invalid step, [10..15] x after the iteration */
static void trace_push(orc_ctx* c, int iteration, double cost, double change, int ok, double radius, double step_norm, double rel,
                       double gmax, int kind, const double x[6]) {
  if (c->ntrace >= ORC_TRACE_ROWS) return;
  double* r = c->trace[c->ntrace++];
  r[0] = c->nsolve; r[1] = iteration; r[2] = cost; r[3] = change; r[4] = ok; r[5] = radius; r[6] = step_norm; r[7] = rel;
  r[8] = gmax; r[9] = kind;
  memcpy(r + 10, x, sizeof(double) * 6);
}
int orc_get_trace(orc_ctx* c, double* out, int cap_rows) {
  if (!c) return TLOAM_E_INVALID;
  int n = c->ntrace < cap_rows ? c->ntrace : cap_rows;
  if (out && n > 0) memcpy(out, c->trace, sizeof(double) * ORC_TRACE_COLS * (size_t)n);
  return c->ntrace;
}

static void ceres_solve(orc_ctx* c, double x[6], tloam_stats* st) {
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32;
  const int max_num_iterations = 4, max_consecutive_invalid = 5;
  orc_normal cur;
  orc_dogleg dl;
  memset(&dl, 0, sizeof(dl));
  dl.radius = 1e4; /* initial_trust_region_radius */
  dl.mu = 1e-8;    /* min_mu_ */
  /* IterationZero */
  evaluate(c, x, 1, &cur);
  st->gn_evaluations++;
  st->gn_sweeps++; /* the CPU restatement executes every evaluation */
  double x_cost = cur.cost;
  double S[6];
  for (int i = 0; i < 6; ++i) S[i] = 1.0 / (1.0 + sqrt(cur.H[i * 6 + i])); /* jacobi_scaling, iteration 0 only */
  double x_norm = vnorm(x, 6);
  int step_successful = 1;
  double gmax = grad_max_norm(x, cur.g);
  int iteration = 0, invalid = 0;
  trace_push(c, 0, x_cost, 0.0, 1, dl.radius, 0.0, 0.0, gmax, 0, x);
  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iteration >= max_num_iterations) break;
    if (step_successful && gmax <= gradient_tolerance) break;
    if (dl.radius <= min_trust_region_radius) break;
    iteration++;
    st->gn_iterations++;
    /* ComputeTrustRegionStep */
    double Hs[36], gs[6], step[6];
    for (int i = 0; i < 6; ++i) {
      gs[i] = S[i] * cur.g[i];
      for (int j = 0; j < 6; ++j) Hs[i * 6 + j] = S[i] * cur.H[i * 6 + j] * S[j];
    }
    int lin_ok = dogleg_compute_step(&dl, Hs, gs, step);
    int valid = 0;
    double model_cost_change = 0.0;
    if (lin_ok) {
      double sg = 0.0, sHs = 0.0;
      for (int i = 0; i < 6; ++i) {
        sg += step[i] * gs[i];
        for (int j = 0; j < 6; ++j) sHs += step[i] * Hs[i * 6 + j] * step[j];
      }
      model_cost_change = -sg - 0.5 * sHs;
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      /* HandleInvalidStep -> StepIsInvalid */
      trace_push(c, iteration, x_cost, 0.0, 0, dl.radius, 0.0, 0.0, gmax, 3, x);
      if (++invalid >= max_consecutive_invalid) break;
      dl.mu *= 10.0;
      dl.reuse = 0;
      step_successful = 0;
      continue;
    }
    invalid = 0;
    double delta[6], x_cand[6];
    for (int i = 0; i < 6; ++i) delta[i] = step[i] * S[i];
    orc_plus(x, delta, x_cand);
    /* ComputeCandidatePointAndEvaluateCost: cost-only sweep (side channel IS written) */
    orc_normal cand;
    evaluate(c, x_cand, 0, &cand);
    st->gn_evaluations++;
    st->gn_sweeps++;
    double candidate_cost = cand.cost;
    /* ParameterToleranceReached */
    double dx[6];
    for (int i = 0; i < 6; ++i) dx[i] = x[i] - x_cand[i];
    if (vnorm(dx, 6) <= parameter_tolerance * (x_norm + parameter_tolerance)) {
      trace_push(c, iteration, x_cost, x_cost - candidate_cost, 0, dl.radius, vnorm(dx, 6), 0.0, gmax, 1, x);
      break;
    }
    /* FunctionToleranceReached */
    if (fabs(x_cost - candidate_cost) <= function_tolerance * x_cost) {
      trace_push(c, iteration, x_cost, x_cost - candidate_cost, 0, dl.radius, vnorm(dx, 6), 0.0, gmax, 2, x);
      break;
    }
    /* IsStepSuccessful */
    double rel = (x_cost - candidate_cost) / model_cost_change;
    const double cost_change = x_cost - candidate_cost, step_norm_x = vnorm(dx, 6);
    if (rel > min_relative_decrease) {
      /* HandleSuccessfulStep: x = candidate; re-evaluate the same point WITH Jacobians */
      memcpy(x, x_cand, sizeof(double) * 6);
      x_norm = vnorm(x, 6);
      evaluate(c, x, 1, &cur);
      x_cost = cur.cost; /* == candidate_cost */
      gmax = grad_max_norm(x, cur.g);
      step_successful = 1;
      st->accepted_steps++;
      /* DoglegStrategy::StepAccepted */
      if (rel < 0.25) dl.radius *= 0.5;
      if (rel > 0.75) { double r3 = 3.0 * dl.step_norm; if (r3 > dl.radius) dl.radius = r3; }
      dl.mu = 2.0 * dl.mu / 10.0;
      if (dl.mu < 1e-8) dl.mu = 1e-8;
      dl.reuse = 0;
      trace_push(c, iteration, x_cost, cost_change, 1, dl.radius, step_norm_x, rel, gmax, 0, x);
    } else {
      /* HandleUnsuccessfulStep -> StepRejected */
      step_successful = 0;
      dl.radius *= 0.5;
      dl.reuse = 1;
      trace_push(c, iteration, x_cost, cost_change, 0, dl.radius, step_norm_x, rel, gmax, 0, x);
    }
  }
  c->nsolve++;
  st->solver_cost = x_cost;
  memcpy(c->last_H, cur.H, sizeof(cur.H));
  memcpy(c->last_g, cur.g, sizeof(cur.g));
  c->last_cost = x_cost;
}

/* ============================================================================
 *  scanMatching  registration.cpp:879-1133
 * ========================================================================== */
int orc_sm_begin(orc_ctx* c, const double predict[16], const double* omega3) {
  if (!c || !predict) return TLOAM_E_INVALID;
  for (int k = 0; k < 4; ++k) /* :928-929 (asserts in the reference) */
    if (c->nsrc[k] < 10 || c->ntgt[k] < 10) return TLOAM_E_TOO_FEW_POINTS;
  double q[4], t[3];
  int rc = orc_se3_from_matrix(predict, q, t);
  if (rc != TLOAM_OK) return rc;
  orc_se3_log(q, t, c->x); /* :881 */
  c->ntrace = 0; c->nsolve = 0;
  double on = sqrt(c->x[3] * c->x[3] + c->x[4] * c->x[4] + c->x[5] * c->x[5]);
  if (on < 1e-2) { /* :884-886; Random().normalized() replaced by an explicit input */
    double u[3] = {0.0, 0.0, 1.0};
    if (omega3) {
      double n = sqrt(omega3[0] * omega3[0] + omega3[1] * omega3[1] + omega3[2] * omega3[2]);
      if (n > 0.0) { u[0] = omega3[0] / n; u[1] = omega3[1] / n; u[2] = omega3[2] / n; }
    }
    c->x[3] = u[0] * 1e-4; c->x[4] = u[1] * 1e-4; c->x[5] = u[2] * 1e-4;
  }
  /* :889-915 four "kd-trees" (deep copies of the submap clouds) */
  for (int k = 0; k < 4; ++k) {
    free(c->tree_pts[k]);
    c->tree_n[k] = c->ntgt[k];
    c->tree_pts[k] = (double*)malloc(sizeof(double) * 3 * (size_t)c->ntgt[k]);
    memcpy(c->tree_pts[k], c->tgt[k], sizeof(double) * 3 * (size_t)c->ntgt[k]);
    c->grid_radius[k] = kind_radius(c, k);
    grid_build(&c->grid[k], c->tree_pts[k], c->tree_n[k], c->grid_radius[k]);
  }
  c->trees_built = 1;
  c->prebuilt = 0;
  for (int k = 0; k < 4; ++k) { /* :931-949 */
    free(c->weights[k]);
    free(c->resid[k]);
    c->weights[k] = (double*)malloc(sizeof(double) * (size_t)c->nsrc[k]);
    c->resid[k] = (double*)calloc((size_t)c->nsrc[k], sizeof(double));
    for (int i = 0; i < c->nsrc[k]; ++i) c->weights[k][i] = 1.0;
    c->prev_cost[k] = INFINITY; /* :952-959 */
    c->cur_cost[k] = INFINITY;
    c->set[k].n = 0;
  }
  c->mu = 1.0; /* :961 */
  c->noise_bound_sq = pow(c->cfg.noise_bound, 2);
  if (c->noise_bound_sq < 1e-16) c->noise_bound_sq = 1e-2; /* :963-964 */
  c->iter = 0;
  c->active = 1;
  c->bad_weights = 0;
  memset(&c->stats, 0, sizeof(c->stats));
  return TLOAM_OK;
}

/* registration.cpp:858-876 */
static void update_weight(orc_ctx* c, int kind, double th1, double th2, double mu) {
  double* w = c->weights[kind];
  const double* r = c->resid[kind];
  for (int i = 0; i < c->nsrc[kind]; ++i) {
    if (r[i] == 0) continue;
    if (r[i] >= th1) w[i] = 0.0;
    else if (r[i] <= th2) w[i] = 1.0;
    else {
      w[i] = sqrt(c->noise_bound_sq * mu * (mu + 1) / r[i]) - mu;
      if (!(w[i] >= 0.0 && w[i] <= 1.0)) c->bad_weights++; /* reference: assert (:871) */
    }
  }
}

int orc_sm_outer(orc_ctx* c, int* done, tloam_stats* stats) {
  if (!c || !c->active) return TLOAM_E_NOT_READY;
  const int iter = c->iter;
  if (iter >= c->cfg.max_iterations) { /* loop condition :966 */
    if (done) *done = 1;
    if (stats) *stats = c->stats;
    return TLOAM_OK;
  }
  /* :976-1020 builders (4 async tasks in the reference; order-independent) */
#ifdef _OPENMP
  if (c->builder_threads > 1) {
#pragma omp parallel for num_threads(c->builder_threads) schedule(dynamic, 1)
    for (int k = 0; k < 4; ++k) {
      if (kind_active(c, k)) build_kind(c, k); else c->set[k].n = 0;
    }
  } else
#endif
  for (int k = 0; k < 4; ++k) {
    if (kind_active(c, k)) build_kind(c, k); else c->set[k].n = 0;
  }
  if (iter == 0) { /* :1027-1033 ; ground excluded; slots are still all zero here */
    double mp = -INFINITY, me = -INFINITY, ms = -INFINITY;
    for (int i = 0; i < c->nsrc[TLOAM_KIND_PLANAR]; ++i) if (c->resid[TLOAM_KIND_PLANAR][i] > mp) mp = c->resid[TLOAM_KIND_PLANAR][i];
    for (int i = 0; i < c->nsrc[TLOAM_KIND_EDGE]; ++i) if (c->resid[TLOAM_KIND_EDGE][i] > me) me = c->resid[TLOAM_KIND_EDGE][i];
    for (int i = 0; i < c->nsrc[TLOAM_KIND_SPHERE]; ++i) if (c->resid[TLOAM_KIND_SPHERE][i] > ms) ms = c->resid[TLOAM_KIND_SPHERE][i];
    double mr = mp > me ? (mp > ms ? mp : ms) : (me > ms ? me : ms);
    c->mu = 1 / (2 * mr / c->noise_bound_sq - 1.0);
    if (c->mu <= 0) c->mu = 1e-10;
  }
  ceres_solve(c, c->x, &c->stats); /* :1036-1047 */
  double mu = c->mu;
  double th1 = (mu + 1) / mu * c->noise_bound_sq; /* :1049 */
  double th2 = mu / (mu + 1) * c->noise_bound_sq; /* :1050 */
  for (int k = 0; k < 4; ++k)
    if (kind_active(c, k)) update_weight(c, k, th1, th2, mu); /* :1053-1086 */
  c->mu = mu * exp((double)(iter + 1) * c->cfg.gnc_factor); /* :1089 */
  for (int k = 0; k < 4; ++k) { /* :1091-1094 */
    double s = 0.0;
    for (int i = 0; i < c->nsrc[k]; ++i) s += c->resid[k][i];
    c->cur_cost[k] = s;
    c->stats.kind_cost[k] = s;
    c->stats.n_corr[k] = c->set[k].n;
  }
  c->stats.outer_iterations = iter + 1;
  c->stats.mu = c->mu;
  memcpy(c->stats.se3, c->x, sizeof(double) * 6);
  const int weight_violation = c->bad_weights > c->stats.weight_range_violations;
  c->stats.weight_range_violations = c->bad_weights;
  double diff = fabs(c->cur_cost[TLOAM_KIND_PLANAR] - c->prev_cost[TLOAM_KIND_PLANAR]);
  int fin = 0;
  if (diff < c->cfg.cost_threshold) { /* :1108 */
    c->stats.converged_early = 1;
    fin = 1;
  } else {
    for (int k = 0; k < 4; ++k) { /* :1113-1121 */
      c->prev_cost[k] = c->cur_cost[k];
      memset(c->resid[k], 0, sizeof(double) * (size_t)c->nsrc[k]);
    }
    c->iter = iter + 1;
    if (c->iter >= c->cfg.max_iterations) fin = 1;
  }
  if (fin) c->iter = c->cfg.max_iterations;
  if (done) *done = fin;
  if (stats) *stats = c->stats;
  return weight_violation ? TLOAM_E_WEIGHT_RANGE : TLOAM_OK; /* the reference asserts (:871; no NDEBUG in its build) */
}

int orc_sm_end(orc_ctx* c, double result[16], tloam_stats* stats) {
  if (!c || !c->active) return TLOAM_E_NOT_READY;
  double q[4], t[3];
  orc_se3_exp(c->x, q, t);
  orc_se3_to_matrix(q, t, result); /* :1124 */
  if (stats) *stats = c->stats;
  c->active = 0;
  return TLOAM_OK;
}

int orc_scan_match(orc_ctx* c, const double predict[16], const double* omega3, double result[16],
                   double* scan_xyz, size_t n_scan, tloam_stats* stats) {
  int rc = orc_sm_begin(c, predict, omega3);
  if (rc != TLOAM_OK) return rc;
  int done = 0, weight_violation = 0;
  while (!done) {
    rc = orc_sm_outer(c, &done, NULL);
    if (rc == TLOAM_E_WEIGHT_RANGE) { weight_violation = 1; continue; } /* reported after the solve, like the HIP path */
    if (rc != TLOAM_OK) return rc;
  }
  rc = orc_sm_end(c, result, stats);
  if (rc != TLOAM_OK) return rc;
  if (scan_xyz && n_scan > 0) { /* :1126-1128  PointCloud2::Transform (PointCloud2.cpp:71-75): 4x4 * (p,1) */
    for (size_t i = 0; i < n_scan; ++i) {
      double* p = scan_xyz + 3 * i;
      double x = p[0], y = p[1], z = p[2];
      double o[4];
      for (int r = 0; r < 4; ++r) o[r] = result[0 * 4 + r] * x + result[1 * 4 + r] * y + result[2 * 4 + r] * z + result[3 * 4 + r];
      p[0] = o[0] / o[3]; p[1] = o[1] / o[3]; p[2] = o[2] / o[3];
    }
  }
  return weight_violation ? TLOAM_E_WEIGHT_RANGE : TLOAM_OK;
}

/* registration.cpp:257-296 */
int orc_fitness(orc_ctx* c, double* fitness, double* rmse) {
  if (!c || !fitness || !rmse) return TLOAM_E_INVALID;
  *fitness = 0.0;
  *rmse = 0.0;
  if (c->cfg.fitness_thres <= 0.0) return TLOAM_OK; /* :258-261 */
  const int order[4] = {TLOAM_KIND_EDGE, TLOAM_KIND_SPHERE, TLOAM_KIND_PLANAR, TLOAM_KIND_GROUND}; /* :287-290 */
  for (int o = 0; o < 4; ++o) {
    int k = order[o];
    double err = 0.0;
    int corr = 0;
    if (c->trees_built && c->tree_n[k] > 0) {
      orc_grid g;
      memset(&g, 0, sizeof(g));
      grid_build(&g, c->tree_pts[k], c->tree_n[k], c->cfg.fitness_thres);
      for (int i = 0; i < c->nsrc[k]; ++i) {
        int idx;
        double d2;
        if (grid_knn_hybrid(&g, c->src[k] + 3 * i, c->cfg.fitness_thres, 1, &idx, &d2) > 0) { /* :272 raw scan-frame point */
          err += d2; /* :273 adds the squared distance */
          corr++;
        }
      }
      grid_free(&g);
    }
    if (corr > 0) {
      *fitness += (double)corr / (double)c->nsrc[k];
      *rmse += sqrt(err / (double)corr);
    }
  }
  return TLOAM_OK;
}

/* ---- introspection -------------------------------------------------------- */
int orc_get_correspondences(orc_ctx* c, int kind, size_t capacity, size_t* n, int32_t* src_index,
                            double* a, double* b, double* d, double* w, double* cost) {
  if (!c || kind < 0 || kind >= 4 || !n) return TLOAM_E_INVALID;
  orc_rset* s = &c->set[kind];
  *n = (size_t)s->n;
  if ((size_t)s->n > capacity) return TLOAM_E_INVALID;
  for (int j = 0; j < s->n; ++j) {
    if (src_index) src_index[j] = s->idx[j];
    if (a) memcpy(a + 3 * j, s->a + 3 * j, 24);
    if (b && kind == TLOAM_KIND_EDGE) memcpy(b + 3 * j, s->b + 3 * j, 24);
    if (d && (kind == TLOAM_KIND_PLANAR || kind == TLOAM_KIND_GROUND)) d[j] = s->d[j];
    if (w) w[j] = s->w[j];
    if (cost) cost[j] = s->cost[j];
  }
  return TLOAM_OK;
}
int orc_get_weights(orc_ctx* c, int kind, size_t capacity, size_t* n, double* w) {
  if (!c || kind < 0 || kind >= 4 || !n) return TLOAM_E_INVALID;
  *n = (size_t)c->nsrc[kind];
  if (!c->weights[kind] || (size_t)c->nsrc[kind] > capacity) return TLOAM_E_INVALID;
  if (w) memcpy(w, c->weights[kind], sizeof(double) * (size_t)c->nsrc[kind]);
  return TLOAM_OK;
}
int orc_knn(orc_ctx* c, int kind, const double* q, size_t nq, double radius, int k, int32_t* out_idx,
            double* out_d2, int32_t* out_cnt) {
  if (!c || kind < 0 || kind >= 4 || k < 1 || k > 16) return TLOAM_E_INVALID;
  orc_grid g;
  memset(&g, 0, sizeof(g));
  grid_build(&g, c->tgt[kind], c->ntgt[kind], radius);
  for (size_t i = 0; i < nq; ++i) {
    int idx[16];
    double d2[16];
    int cnt = grid_knn_hybrid(&g, q + 3 * i, radius, k, idx, d2);
    if (cnt < 0) cnt = 0;
    for (int m = 0; m < k; ++m) {
      out_idx[i * k + m] = m < cnt ? idx[m] : -1;
      out_d2[i * k + m] = m < cnt ? d2[m] : 0.0;
    }
    out_cnt[i] = cnt;
  }
  grid_free(&g);
  return TLOAM_OK;
}

/* ---- pre-built sets ------------------------------------------------------- */
int orc_set_correspondences(orc_ctx* c, int res_type, size_t n, const double* p, const double* a,
                            const double* b, const double* d, const double* w) {
  if (!c || res_type < 0 || res_type >= 3) return TLOAM_E_INVALID;
  int kind = (res_type == TLOAM_RES_PLANE) ? TLOAM_KIND_PLANAR : (res_type == TLOAM_RES_LINE) ? TLOAM_KIND_EDGE : TLOAM_KIND_SPHERE;
  if (!c->prebuilt) {
    for (int k = 0; k < 4; ++k) c->set[k].n = 0;
    c->prebuilt = 1;
  }
  orc_rset* s = &c->set[kind];
  rset_reserve(s, (int)n > 0 ? (int)n : 1);
  s->n = (int)n;
  for (size_t j = 0; j < n; ++j) {
    s->idx[j] = (int32_t)j;
    memcpy(s->p + 3 * j, p + 3 * j, 24);
    memcpy(s->a + 3 * j, a + 3 * j, 24);
    if (res_type == TLOAM_RES_LINE) memcpy(s->b + 3 * j, b + 3 * j, 24); else memset(s->b + 3 * j, 0, 24);
    s->d[j] = (res_type == TLOAM_RES_PLANE) ? d[j] : 0.0;
    s->w[j] = w[j];
    s->cost[j] = 0.0;
  }
  return TLOAM_OK;
}
int orc_accumulate(orc_ctx* c, const double se3[6], double H[36], double g[6], double* cost) {
  if (!c || !se3) return TLOAM_E_INVALID;
  orc_normal nrm;
  evaluate(c, se3, 1, &nrm);
  if (H) memcpy(H, nrm.H, sizeof(double) * 36);
  if (g) memcpy(g, nrm.g, sizeof(double) * 6);
  if (cost) *cost = nrm.cost;
  return TLOAM_OK;
}
/* H = sum rho' J^T J, g = sum rho' J^T r and the cost at the accepted iterate of the last Solve (what the
 * minimiser held when it returned) -- parity probe of the per-outer-iteration linear system */
int orc_get_normal_equations(orc_ctx* c, double H[36], double g[6], double* cost) {
  if (!c) return TLOAM_E_INVALID;
  if (H) memcpy(H, c->last_H, sizeof(c->last_H));
  if (g) memcpy(g, c->last_g, sizeof(c->last_g));
  if (cost) *cost = c->last_cost;
  return TLOAM_OK;
}

int orc_get_costs(orc_ctx* c, int res_type, size_t capacity, size_t* n, double* cost) {
  if (!c || res_type < 0 || res_type >= 3 || !n) return TLOAM_E_INVALID;
  int kind = (res_type == TLOAM_RES_PLANE) ? TLOAM_KIND_PLANAR : (res_type == TLOAM_RES_LINE) ? TLOAM_KIND_EDGE : TLOAM_KIND_SPHERE;
  orc_rset* s = &c->set[kind];
  *n = (size_t)s->n;
  if ((size_t)s->n > capacity) return TLOAM_E_INVALID;
  if (cost) memcpy(cost, s->cost, sizeof(double) * (size_t)s->n);
  return TLOAM_OK;
}
int orc_solve(orc_ctx* c, double se3[6], tloam_stats* stats) {
  if (!c || !se3) return TLOAM_E_INVALID;
  tloam_stats st;
  memset(&st, 0, sizeof(st));
  c->ntrace = 0; c->nsolve = 0;
  ceres_solve(c, se3, &st);
  memcpy(st.se3, se3, sizeof(double) * 6);
  if (stats) *stats = st;
  return TLOAM_OK;
}


/* ==========================================================================
 *  PCA feature extraction -- TEST INFRASTRUCTURE like the rest of this file.
 *  featureExtract::calculatePCAInfo     feature_extract.cpp:47-122
 *  featureExtract::extractPlanarSphere  feature_extract.cpp:133-197
 *  "parity unpinned": no reference test covers it; Eigen::SelfAdjointEigenSolver is restated by the same
 *  cyclic Jacobi as above (ascending eigenvalues; the sign of normal_dir is immaterial, only |z| is used).
 * ========================================================================== */
#define ORC_FEAT_MAXK 20
int orc_pca_info(const tloam_feature_config* cfg, const double* xyz, size_t n, double* flatness, double* cvr,
                 double* sphericity, double* normal, int32_t* num_sum, int32_t* neigh) {
  if (!cfg || cfg->K < 3 || cfg->K > ORC_FEAT_MAXK || !(cfg->radius >= 0.0)) return TLOAM_E_INVALID; /* assert :55 */
  const int K = cfg->K;
  orc_grid g;
  memset(&g, 0, sizeof(g));
  if (n > 0 && cfg->radius > 0.0) grid_build(&g, xyz, (int)n, cfg->radius);
  for (size_t i = 0; i < n; ++i) {
    /* value-initialised PCAInfo (pca_info_.resize, :59) */
    if (flatness) flatness[i] = 0.0;
    if (cvr) cvr[i] = 0.0;
    if (sphericity) sphericity[i] = 0.0;
    if (normal) normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0.0;
    if (num_sum) num_sum[i] = 0;
    if (neigh) for (int m = 0; m < K; ++m) neigh[i * (size_t)K + m] = -1;
    int idx[ORC_FEAT_MAXK];
    double d2[ORC_FEAT_MAXK];
    const int cnt = (cfg->radius > 0.0) ? grid_knn_hybrid(&g, xyz + 3 * i, cfg->radius, K, idx, d2) : 0; /* :71 */
    if (cnt <= 0) continue;
    if (cnt <= cfg->min_neigh) continue; /* :72 */
    double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < cnt; ++m) { /* :80-91, neighbours in ascending distance */
      const double* p = xyz + 3 * (size_t)idx[m];
      cum[0] += p[0]; cum[1] += p[1]; cum[2] += p[2];
      cum[3] += p[0] * p[0]; cum[4] += p[0] * p[1]; cum[5] += p[0] * p[2];
      cum[6] += p[1] * p[1]; cum[7] += p[1] * p[2]; cum[8] += p[2] * p[2];
    }
    for (int m = 0; m < 9; ++m) cum[m] /= (double)cnt; /* :92 */
    double cov[9];
    cov[0] = cum[3] - cum[0] * cum[0];
    cov[4] = cum[6] - cum[1] * cum[1];
    cov[8] = cum[8] - cum[2] * cum[2];
    cov[1] = cov[3] = cum[4] - cum[0] * cum[1];
    cov[2] = cov[6] = cum[5] - cum[0] * cum[2];
    cov[5] = cov[7] = cum[7] - cum[1] * cum[2];
    double ev[3], V[9];
    orc_eig3_sym(cov, ev, V);
    const double sum = (ev[0] + ev[1]) + ev[2]; /* Eigen sum(): left to right for a 3-vector */
    if (normal) { normal[3 * i] = V[0]; normal[3 * i + 1] = V[3]; normal[3 * i + 2] = V[6]; } /* eigenvectors().col(0) */
    if (cvr) cvr[i] = (sum == 0.0) ? 0.0 : ev[0] / sum;                 /* :109-114 */
    if (flatness) flatness[i] = (ev[1] - ev[0]) / ev[2];                /* :116 */
    if (sphericity) sphericity[i] = ev[0] / ev[2];                      /* :117 */
    if (num_sum) num_sum[i] = cnt;
    if (neigh) for (int m = 0; m < cnt; ++m) neigh[i * (size_t)K + m] = idx[m];
  }
  grid_free(&g);
  return TLOAM_OK;
}

typedef struct { double f; int32_t idx; } feat_pair;
static int feat_cmp(const void* a, const void* b) { /* descending flatness; ties: ascending index (stable order) */
  const feat_pair* x = (const feat_pair*)a; const feat_pair* y = (const feat_pair*)b;
  if (x->f > y->f) return -1;
  if (x->f < y->f) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}
int orc_extract_planar_sphere(const tloam_feature_config* cfg, const double* xyz, size_t n, int32_t* planar_scan,
                              size_t* n_ps, int32_t* planar_submap, size_t* n_pm, int32_t* sphere_scan, size_t* n_ss,
                              int32_t* sphere_submap, size_t* n_sm) {
  if (!cfg || !n_ps || !n_pm || !n_ss || !n_sm) return TLOAM_E_INVALID;
  *n_ps = *n_pm = *n_ss = *n_sm = 0;
  if (n == 0) return TLOAM_OK; /* "cloud_in_ does not contain points" -> calculatePCAInfo false, lists stay empty */
  const int K = cfg->K;
  double* fl = (double*)malloc(sizeof(double) * n);
  double* cv = (double*)malloc(sizeof(double) * n);
  double* nr = (double*)malloc(sizeof(double) * 3 * n);
  int32_t* ns = (int32_t*)malloc(sizeof(int32_t) * n);
  int32_t* ng = (int32_t*)malloc(sizeof(int32_t) * n * (size_t)(K > 0 ? K : 1));
  int rc = orc_pca_info(cfg, xyz, n, fl, cv, NULL, nr, ns, ng);
  if (rc != TLOAM_OK) { free(fl); free(cv); free(nr); free(ns); free(ng); return rc; }
  feat_pair* planar = (feat_pair*)malloc(sizeof(feat_pair) * n);
  feat_pair* sphere = (feat_pair*)malloc(sizeof(feat_pair) * n);
  size_t np = 0, nsph = 0;
  for (size_t id = 0; id < n; ++id) { /* :149-165; cur_index of a skipped point is 0, but such a point is never selected */
    if (fl[id] > cfg->planar_submap_thres && fabs(nr[3 * id + 2]) < cfg->planar_vertic_thres) {
      planar[np].f = fl[id]; planar[np].idx = (int32_t)id; ++np;
    } else if (cv[id] > cfg->cvr_submap) {
      int max_uniform = 1;
      for (int m = 0; m < ns[id]; ++m)
        if (cv[id] < cv[ng[id * (size_t)K + m]]) { max_uniform = 0; break; }
      if (max_uniform) { sphere[nsph].f = fl[id]; sphere[nsph].idx = (int32_t)id; ++nsph; } /* :162 pairs FLATNESS */
    }
  }
  qsort(planar, np, sizeof(feat_pair), feat_cmp);   /* :168-174 */
  qsort(sphere, nsph, sizeof(feat_pair), feat_cmp);
  for (size_t id = 0; id < np; ++id) {              /* :178-183 */
    if (id < (size_t)cfg->planar_num || planar[id].f > cfg->planar_scan_thres) planar_scan[(*n_ps)++] = planar[id].idx;
    planar_submap[(*n_pm)++] = planar[id].idx;
  }
  for (size_t id = 0; id < nsph; ++id) {            /* :185-190: the RANK id is stored, not the point index */
    if (id < (size_t)cfg->sphere_num || sphere[id].f > cfg->cvr_scan) sphere_scan[(*n_ss)++] = (int32_t)id;
    sphere_submap[(*n_sm)++] = (int32_t)id;
  }
  free(planar); free(sphere); free(fl); free(cv); free(nr); free(ns); free(ng);
  return TLOAM_OK;
}
