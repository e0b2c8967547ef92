// tl_se3.hpp -- SE(3) arithmetic of the hot path, host + device (gfx950).
//
// The reference leans on ~150 lines of vendored Sophus: SE3d::exp
// (sophus/se3.hpp:761-785), SE3d::log (:223-256), SE3(Matrix4) (:497-504), the point action
// (:321-324 over so3.hpp:358-367), the group product (se3.hpp:304-309, so3.hpp:325-340) and
// SO3::hat (so3.hpp:671-680).  They are restated here as small value types so that the device
// kernels (pose update inside the GN step kernel) and the host driver use one definition.
// Branch thresholds follow Sophus: Constants<double>::epsilon() = 1e-10 (common.hpp:93-95).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define TL_HD __host__ __device__ __forceinline__

namespace tl {

constexpr double kSophusEps = 1e-10;
constexpr double kPi = 3.14159265358979323846;

struct Vec3 {
  double x, y, z;
};
TL_HD Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
TL_HD Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
TL_HD Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
TL_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TL_HD Vec3 cross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// unit quaternion (w, v) + translation: the storage Sophus::SE3d uses
struct Pose {
  double qw, qx, qy, qz;
  double tx, ty, tz;
};

// so3.hpp:358-367 then se3.hpp:321-324:  p + w*uv + v x uv + t,  uv = 2 (v x p)
TL_HD Vec3 act(const Pose& T, Vec3 p) {
  const Vec3 v{T.qx, T.qy, T.qz};
  Vec3 uv = cross(v, p);
  uv = uv + uv;
  const Vec3 c2 = cross(v, uv);
  return {(p.x + T.qw * uv.x + c2.x) + T.tx, (p.y + T.qw * uv.y + c2.y) + T.ty,
          (p.z + T.qw * uv.z + c2.z) + T.tz};
}
TL_HD Vec3 rotate(const Pose& T, Vec3 p) {
  const Vec3 v{T.qx, T.qy, T.qz};
  Vec3 uv = cross(v, p);
  uv = uv + uv;
  const Vec3 c2 = cross(v, uv);
  return {p.x + T.qw * uv.x + c2.x, p.y + T.qw * uv.y + c2.y, p.z + T.qw * uv.z + c2.z};
}

// Eigen Quaternion::toRotationMatrix (what SO3::matrix() returns); row-major R[9]
TL_HD void rotation_matrix(const Pose& T, double R[9]) {
  const double tx = 2.0 * T.qx, ty = 2.0 * T.qy, tz = 2.0 * T.qz;
  const double twx = tx * T.qw, twy = ty * T.qw, twz = tz * T.qw;
  const double txx = tx * T.qx, txy = ty * T.qx, txz = tz * T.qx;
  const double tyy = ty * T.qy, tyz = tz * T.qy, tzz = tz * T.qz;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// rotation matrix (row-major) + translation form of a Pose
struct Rt {
  double r[9];
  double t[3];
};
TL_HD Rt to_rt(const Pose& T) {
  Rt o;
  rotation_matrix(T, o.r);
  o.t[0] = T.tx; o.t[1] = T.ty; o.t[2] = T.tz;
  return o;
}

// so3.hpp:583-619 expAndTheta + se3.hpp:761-785
TL_HD Pose se3_exp(const double a[6]) {
  const double ox = a[3], oy = a[4], oz = a[5];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  double theta, imag, real;
  if (theta_sq < kSophusEps * kSophusEps) {
    theta = 0.0;
    const double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    theta = sqrt(theta_sq);
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  Pose T;
  T.qw = real; T.qx = imag * ox; T.qy = imag * oy; T.qz = imag * oz;
  // V * upsilon with V = I + c1 Om + c2 Om^2 ;  Om u = om x u ; Om^2 u = om x (om x u)
  const Vec3 om{ox, oy, oz}, u{a[0], a[1], a[2]};
  Vec3 t;
  if (theta < kSophusEps) {
    t = rotate(T, u);  // "V = so3.matrix()"
  } else {
    const double c1 = (1.0 - cos(theta)) / theta_sq;
    const double c2 = (theta - sin(theta)) / (theta_sq * theta);
    const Vec3 w1 = cross(om, u);
    const Vec3 w2 = cross(om, w1);
    t = u + c1 * w1 + c2 * w2;
  }
  T.tx = t.x; T.ty = t.y; T.tz = t.z;
  return T;
}

// so3.hpp:247-290 logAndTheta + se3.hpp:223-256
TL_HD void se3_log(const Pose& T, double a[6]) {
  const double squared_n = T.qx * T.qx + T.qy * T.qy + T.qz * T.qz;
  const double w = T.qw;
  double f, theta;
  if (squared_n < kSophusEps * kSophusEps) {
    const double squared_w = w * w;
    f = 2.0 / w - (2.0 / 3.0) * squared_n / (w * squared_w);
    theta = 2.0 * squared_n / w;
  } else {
    const double n = sqrt(squared_n);
    if (fabs(w) < kSophusEps) {
      f = (w > 0.0) ? kPi / n : -kPi / n;
    } else {
      f = 2.0 * atan(n / w) / n;
    }
    theta = f * n;
  }
  const Vec3 om{f * T.qx, f * T.qy, f * T.qz};
  double c2;
  if (fabs(theta) < kSophusEps) {
    c2 = 1.0 / 12.0;
  } else {
    const double half = 0.5 * theta;
    c2 = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
  }
  const Vec3 t{T.tx, T.ty, T.tz};
  const Vec3 w1 = cross(om, t);
  const Vec3 w2 = cross(om, w1);
  const Vec3 ups = t + (-0.5) * w1 + c2 * w2;
  a[0] = ups.x; a[1] = ups.y; a[2] = ups.z;
  a[3] = om.x;  a[4] = om.y;  a[5] = om.z;
}

// se3.hpp:304-309 with so3.hpp:325-340; the SO3(quaternion) ctor re-normalises (so3.hpp:481-487)
TL_HD Pose compose(const Pose& A, const Pose& B) {
  Pose C;
  C.qw = A.qw * B.qw - A.qx * B.qx - A.qy * B.qy - A.qz * B.qz;
  C.qx = A.qw * B.qx + A.qx * B.qw + A.qy * B.qz - A.qz * B.qy;
  C.qy = A.qw * B.qy + A.qy * B.qw + A.qz * B.qx - A.qx * B.qz;
  C.qz = A.qw * B.qz + A.qz * B.qw + A.qx * B.qy - A.qy * B.qx;
  const double len = sqrt(C.qw * C.qw + C.qx * C.qx + C.qy * C.qy + C.qz * C.qz);
  C.qw /= len; C.qx /= len; C.qy /= len; C.qz /= len;
  const Vec3 rt = rotate(A, Vec3{B.tx, B.ty, B.tz});
  C.tx = A.tx + rt.x; C.ty = A.ty + rt.y; C.tz = A.tz + rt.z;
  return C;
}

// PoseSE3Parameterization::Plus, registration.cpp:162-173: log(exp(delta) * exp(x))
TL_HD void se3_plus(const double x[6], const double delta[6], double out[6]) {
  const Pose Tx = se3_exp(x);
  const Pose Td = se3_exp(delta);
  se3_log(compose(Td, Tx), out);
}

// se3.hpp:497-504 SE3(Matrix4): false where Sophus would SOPHUS_ENSURE-abort.
// M is column-major 4x4 (Eigen::Isometry3d::matrix()).
inline bool pose_from_matrix(const double M[16], Pose* out) {
  auto m = [&](int r, int c) { return M[c * 4 + r]; };
  // (a non-finite translation passes Sophus' checks -- they look at the rotation and the last row -- and the reference then
  //  computes on NaNs; here it is a bad pose like the others: a status code instead of a NaN result)
  auto finite = [](double v) { return v - v == 0.0; };   // false for NaN and +-Inf
  if (!(finite(m(0, 3)) && finite(m(1, 3)) && finite(m(2, 3)))) return false;
  const double last = m(3, 0) * m(3, 0) + m(3, 1) * m(3, 1) + m(3, 2) * m(3, 2) +
                      (m(3, 3) - 1.0) * (m(3, 3) - 1.0);
  if (!(last < kSophusEps)) return false;
  double fro = 0.0;  // rotation_matrix.hpp:17-27 isOrthogonal
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = (i == j) ? -1.0 : 0.0;
      for (int k = 0; k < 3; ++k) s += m(i, k) * m(j, k);
      fro += s * s;
    }
  if (!(sqrt(fro) < kSophusEps)) return false;
  const double det = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) -
                     m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
                     m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
  if (!(det > 0.0)) return false;
  double q[4];  // w,x,y,z -- Eigen's quaternion-from-matrix
  const double tr = m(0, 0) + m(1, 1) + m(2, 2);
  if (tr > 0.0) {
    double s = sqrt(tr + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (m(2, 1) - m(1, 2)) * s;
    q[2] = (m(0, 2) - m(2, 0)) * s;
    q[3] = (m(1, 0) - m(0, 1)) * s;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (m(k, j) - m(j, k)) * s;
    q[1 + j] = (m(j, i) + m(i, j)) * s;
    q[1 + k] = (m(k, i) + m(i, k)) * s;
  }
  out->qw = q[0]; out->qx = q[1]; out->qy = q[2]; out->qz = q[3];
  out->tx = m(0, 3); out->ty = m(1, 3); out->tz = m(2, 3);
  return true;
}

inline void pose_to_matrix(const Pose& T, double M[16]) {
  double R[9];
  rotation_matrix(T, R);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) M[c * 4 + r] = R[r * 3 + c];
  M[3] = M[7] = M[11] = 0.0;
  M[12] = T.tx; M[13] = T.ty; M[14] = T.tz; M[15] = 1.0;
}

}  // namespace tl
