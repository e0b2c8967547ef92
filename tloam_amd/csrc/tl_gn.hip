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
#include <stdlib.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "tl_common.hpp"
#include "tl_finish.hpp"

namespace tl {

// ================================================================================================
//  K3
// ================================================================================================
// 28 running sums per lane: v[0..20] upper-triangular H, v[21..26] g.  The cost 0.5*sum log(1+s) is
// carried as a running PRODUCT of (1+s) in (mantissa, exponent) form -- two VALU ops per
// correspondence instead of a ~50-instruction fp64 log -- and turned into a log once per lane.
struct Acc {
  double v[27];
  double pm;  // product mantissa in [0.5, 1)
  int pe;     // product exponent
};

// upper-triangular index of (i,j), i<=j, row-major: 0..20
__host__ __device__ constexpr int ut(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

__device__ __forceinline__ double fast_rcp(double x) {  // v_rcp_f64 + one Newton step (<= 1 ulp)
  double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {  // v_rsq_f64 + one Newton step
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x * y;
  return __builtin_fma(y, __builtin_fma(-h, y, 0.5), y);
}

// This translation unit is compiled with -ffp-contract=off: the residual code below is inlined into several kernels (the
// streaming sweep, the one-wave-per-chunk sweep, the one-launch Solve) and has to round alike in all of them -- the tests
// compare those paths bit for bit, and with contraction left to the optimiser (`fast`, and in practice `on` as well) the
// same source line was fused differently from kernel to kernel.  Every fused multiply-add is therefore spelled out.
__device__ __forceinline__ double fdot(Vec3 a, Vec3 b) { return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ double fdot_add(Vec3 a, Vec3 b, double c) {
  return __builtin_fma(a.x, b.x, __builtin_fma(a.y, b.y, __builtin_fma(a.z, b.z, c)));
}
__device__ __forceinline__ Vec3 fcross(Vec3 a, Vec3 b) {
  return {__builtin_fma(a.y, b.z, -(a.z * b.y)), __builtin_fma(a.z, b.x, -(a.x * b.z)), __builtin_fma(a.x, b.y, -(a.y * b.x))};
}
// T * p with T as a row-major rotation matrix + translation (what SE3::operator* computes through the
// quaternion sandwich, sophus so3.hpp:358-367; same value to rounding, 9 FMAs instead of ~21 ops)
__device__ __forceinline__ Vec3 act_rt(const Rt& T, Vec3 p) {
  return {__builtin_fma(T.r[0], p.x, __builtin_fma(T.r[1], p.y, __builtin_fma(T.r[2], p.z, T.t[0]))),
          __builtin_fma(T.r[3], p.x, __builtin_fma(T.r[4], p.y, __builtin_fma(T.r[5], p.z, T.t[1]))),
          __builtin_fma(T.r[6], p.x, __builtin_fma(T.r[7], p.y, __builtin_fma(T.r[8], p.z, T.t[2])))};
}

// CauchyLoss(1.0): rho' = 1/(1+s) (rho'' < 0 -> Ceres' Corrector takes its clamped branch: r and J
// are only scaled by sqrt(rho')), block cost 0.5*log(1+s).  Returns rho'.
__device__ __forceinline__ double cauchy(Acc& a, double s) {
  const double sum = 1.0 + s;
  a.pm *= sum;
  a.pe += __builtin_amdgcn_frexp_exp(a.pm);
  a.pm = __builtin_amdgcn_frexp_mant(a.pm);
  return fast_rcp(sum);
}

// one residual row: adds rho*J^T J (upper triangle) and rho*J^T r
__device__ __forceinline__ void acc_row(Acc& a, const double J[6], double r, double rho) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double ji = rho * J[i];
    a.v[21 + i] = __builtin_fma(ji, r, a.v[21 + i]);
#pragma unroll
    for (int j = i; j < 6; ++j) a.v[ut(i, j)] = __builtin_fma(ji, J[j], a.v[ut(i, j)]);
  }
}

// PointToPlaneErr::Evaluate (registration.cpp:96-117) through ResidualBlock::Evaluate + CauchyLoss(1)
__device__ __forceinline__ double eval_plane(const Rt& T, Vec3 p, Vec3 n, double d, double w, Acc& a) {
  const Vec3 pw = act_rt(T, p);
  const double r = fdot_add(n, pw, d);       // :100 (unweighted)
  const Vec3 c = fcross(pw, n);              // n^T (-hat(pw)) = (pw x n)^T   :110,:112
  const double J[6] = {n.x * w, n.y * w, n.z * w, c.x * w, c.y * w, c.z * w};
  const double s = r * r;                    // squared norm of the block; also *cost = r^2  :101
  acc_row(a, J, r, cauchy(a, s));
  return s;
}

// PointToLineErr::Evaluate (registration.cpp:55-88)
__device__ __forceinline__ double eval_line(const Rt& T, Vec3 p, Vec3 la, Vec3 lb, double w, Acc& a) {
  const Vec3 pw = act_rt(T, p);
  const Vec3 nu = fcross(pw - la, pw - lb);  // :62
  const Vec3 e = lb - la;                    // :80  (|la - lb| = |e|, :63)
  // (a HOLE of a direct set -- tl_common.hpp DirectSet -- is a = b = 0: 1 / |e| taken as 0 makes the row add exact zeros and its
  //  side-channel cost exactly 0; a real line has |e| = 0.2, :483-484, and takes the same value as ever)
  const double e2 = fdot(e, e);
  const double k = e2 > 0.0 ? w * fast_rsqrt(e2) : 0.0;
  const double r0 = nu.x * k, r1 = nu.y * k, r2 = nu.z * k;    // :65-67  nu / |de| * w
  const double rs = (r0 + r1) + r2;
  // J = hat(e) [I w, -hat(pw) w] / |de|  :77-83 ; hat(a) hat(b) = b a^T - (a.b) I  =>
  // -hat(e) hat(pw) = (e.pw) I - pw e^T
  const double ep = fdot(e, pw);
  const double ex = e.x * k, ey = e.y * k, ez = e.z * k, epk = ep * k;
  const double J0[6] = {0.0, -ez, ey, __builtin_fma(-pw.x, ex, epk), -(pw.x * ey), -(pw.x * ez)};
  const double J1[6] = {ez, 0.0, -ex, -(pw.y * ex), __builtin_fma(-pw.y, ey, epk), -(pw.y * ez)};
  const double J2[6] = {-ey, ex, 0.0, -(pw.z * ex), -(pw.z * ey), __builtin_fma(-pw.z, ez, epk)};
  const double rho1 = cauchy(a, __builtin_fma(r2, r2, __builtin_fma(r1, r1, r0 * r0)));
  acc_row(a, J0, r0, rho1);
  acc_row(a, J1, r1, rho1);
  acc_row(a, J2, r2, rho1);
  return rs * rs;                            // :69 side channel (r0+r1+r2)^2
}

// PointToPointErr::Evaluate (registration.cpp:19-47)
__device__ __forceinline__ double eval_point(const Rt& T, Vec3 p, Vec3 q, double w_in, Acc& a) {
  const Vec3 pw = act_rt(T, p);
  // (a HOLE of a direct set is q = (NaN, 0, 0): weight and q.x taken as 0 -- exact zeros, side-channel cost exactly 0)
  const bool hole = q.x != q.x;
  const double w = hole ? 0.0 : w_in;
  q.x = hole ? 0.0 : q.x;
  const double r0 = (q.x - pw.x) * w, r1 = (q.y - pw.y) * w, r2 = (q.z - pw.z) * w;  // :26-30
  const double rs = (r0 + r1) + r2;
  const double wx = pw.x * w, wy = pw.y * w, wz = pw.z * w;
  const double J0[6] = {-w, 0.0, 0.0, 0.0, -wz, wy};   // [-I w, hat(pw) w]  :39-40
  const double J1[6] = {0.0, -w, 0.0, wz, 0.0, -wx};
  const double J2[6] = {0.0, 0.0, -w, -wy, wx, 0.0};
  const double rho1 = cauchy(a, __builtin_fma(r2, r2, __builtin_fma(r1, r1, r0 * r0)));
  acc_row(a, J0, r0, rho1);
  acc_row(a, J1, r1, rho1);
  acc_row(a, J2, r2, rho1);
  return rs * rs;                            // :32
}

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }

// wave-level reduce-scatter of 32 per-lane values: after the five exchange steps (xor 32,16,8,4,2)
// every lane holds ONE component (index lane>>1) summed over its 32-lane class, the last xor-1 step
// completes the 64-lane sum.  32 cross-lane exchanges instead of 28 x 6.
template <int N, int D>
__device__ __forceinline__ void rs_step(double (&v)[32], int lane) {
  const bool hi = (lane & D) != 0;
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const double keep = hi ? v[i + N / 2] : v[i];
    const double send = hi ? v[i] : v[i + N / 2];
    v[i] = keep + __shfl_xor(send, D, 64);
  }
}
// The two widest steps (xor 32, xor 16: 24 of the 32 exchanges) on gfx950's v_permlane32_swap /
// v_permlane16_swap: swapping the odd rows of A = v[i] with the even rows of B = v[i + N/2] and adding
// gives every lane "own kept value + partner's sent value" in two VALU ops per 32-bit half -- no select,
// no address arithmetic and no trip through the LDS crossbar.
template <int N, int D>
__device__ __forceinline__ void rs_step_swap(double (&v)[32]) {
  static_assert(D == 32 || D == 16, "row swaps exist for 32- and 16-lane rows");
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const unsigned alo = (unsigned)__double2loint(v[i]), ahi = (unsigned)__double2hiint(v[i]);
    const unsigned blo = (unsigned)__double2loint(v[i + N / 2]), bhi = (unsigned)__double2hiint(v[i + N / 2]);
    double x, y;
    if (D == 32) {
      const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto hh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      x = __hiloint2double((int)hh[0], (int)lo[0]);
      y = __hiloint2double((int)hh[1], (int)lo[1]);
    } else {
      const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto hh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      x = __hiloint2double((int)hh[0], (int)lo[0]);
      y = __hiloint2double((int)hh[1], (int)lo[1]);
    }
    v[i] = x + y;
  }
}

// One wave-chunk in flight: RES-dependent number of 16-byte streams per lane (plane 8, line 10,
// point 7).  fetch() is branch-free -- every lane always issues every load (the segments are padded
// by one chunk, so a tail lane reads valid memory and is masked in consume()) -- which lets the
// compiler count outstanding loads exactly and wait with vmcnt(N) for the OLDER chunk only.
struct ChunkData {
  double2 px, py, pz, ax, ay, az, bx, by, bz, d, w;
};
template <int RES>
using ChunkBuf = ChunkData;   // (one type: which streams are filled / used is the RES of the function handling it)
// uniform base pointer (SGPR pair) + 32-bit per-lane byte offset: lets the compiler use the
// `global_load_dwordx4 v, v_off, s[base]` addressing form (one offset VGPR for all streams)
template <bool NT>
__device__ __forceinline__ double2 ld2o_t(const double* base, unsigned byte_off) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d* p = reinterpret_cast<const v2d*>(reinterpret_cast<const char*>(base) + byte_off);
  const v2d v = NT ? __builtin_nontemporal_load(p) : *p;
  return double2{v.x, v.y};
}
// Cache policy of the streams (measured on the 74.88 MB set, scripts/k3_stats.py, same box, p50 of 40 batches of 20
// launches): default loads + default stores 14.1-14.2 us; streaming (nt) stores only 14.6; nt loads only 13.0-13.4;
// nt loads AND nt stores 12.7-13.0 (0.72-0.74 of 8 TB/s).  Every byte is used exactly once per launch, so the
// lines should not displace each other in L2 on their way through; the set itself (and the cost slots the weight
// kernel reads afterwards) still lives in the 256 MB Infinity Cache across the launches of a Solve.
#ifndef TLOAM_K3_NT
#define TLOAM_K3_NT true
#endif
// NT: streaming policy (the 1 M-class sets); false for the KITTI-size sets, which live in L2 across the launches of
// a frame (442 KB: with streaming loads/stores the weight kernel and the next sweep fetch them from further away)
template <int RES, bool NT>
__device__ __forceinline__ void fetch(const CorrSeg& seg, int j, ChunkBuf<RES>& b) {
#define ld2o ld2o_t<NT>
  const unsigned o = (unsigned)j * 8u;
  b.px = ld2o(seg.px, o); b.py = ld2o(seg.py, o); b.pz = ld2o(seg.pz, o);
  b.ax = ld2o(seg.ax, o); b.ay = ld2o(seg.ay, o); b.az = ld2o(seg.az, o);
  b.w = ld2o(seg.w, o);
  if (RES == TLOAM_RES_PLANE) b.d = ld2o(seg.d, o);
  if (RES == TLOAM_RES_LINE) { b.bx = ld2o(seg.bx, o); b.by = ld2o(seg.by, o); b.bz = ld2o(seg.bz, o); }
#undef ld2o
}
// the planar segment's chunk straight from (base, stride): the kernel's first arguments, preloaded into SGPRs
// w2: the weights come from the segment's SECOND weight stream (a direct set's odd outer iterations, tl_common.hpp DirectSet) --
// known to the launcher, handed over in a preloaded argument like the base itself
template <bool NT>
__device__ __forceinline__ void fetch_spec(const double* base, int stride, int j, ChunkBuf<TLOAM_RES_PLANE>& b, bool w2 = false) {
#define ld2o ld2o_t<NT>
  const unsigned o = (unsigned)j * 8u;
  const size_t st = (size_t)stride;
  b.px = ld2o(base + SS_PX * st, o); b.py = ld2o(base + SS_PY * st, o); b.pz = ld2o(base + SS_PZ * st, o);
  b.ax = ld2o(base + SS_AX * st, o); b.ay = ld2o(base + SS_AY * st, o); b.az = ld2o(base + SS_AZ * st, o);
  b.w = ld2o(base + (w2 ? SS_W2 : SS_W) * st, o);
  b.d = ld2o(base + SS_D * st, o);
#undef ld2o
}
// cost_out: the lane's two costs, for the wave that keeps its correspondences across Solves and ends the outer iteration
// itself (k_solve_all); untouched (the caller's zeros) where the lane has no factor.
template <int RES, bool NT>
__device__ __forceinline__ void consume(const Rt& T, const CorrSeg& seg, int j, int n, const ChunkBuf<RES>& b, Acc& a,
                                        double2* cost_out = nullptr) {
  const int rem = n - j;  // >= 2: both correspondences of this lane, 1: the first only, <= 0: none
  if (rem <= 0) return;
  double c0, c1 = 0.0;
  if (RES == TLOAM_RES_PLANE) {
    c0 = eval_plane(T, Vec3{b.px.x, b.py.x, b.pz.x}, Vec3{b.ax.x, b.ay.x, b.az.x}, b.d.x, b.w.x, a);
    if (rem > 1) c1 = eval_plane(T, Vec3{b.px.y, b.py.y, b.pz.y}, Vec3{b.ax.y, b.ay.y, b.az.y}, b.d.y, b.w.y, a);
  } else if (RES == TLOAM_RES_LINE) {
    c0 = eval_line(T, Vec3{b.px.x, b.py.x, b.pz.x}, Vec3{b.ax.x, b.ay.x, b.az.x}, Vec3{b.bx.x, b.by.x, b.bz.x}, b.w.x, a);
    if (rem > 1)
      c1 = eval_line(T, Vec3{b.px.y, b.py.y, b.pz.y}, Vec3{b.ax.y, b.ay.y, b.az.y}, Vec3{b.bx.y, b.by.y, b.bz.y}, b.w.y, a);
  } else {
    c0 = eval_point(T, Vec3{b.px.x, b.py.x, b.pz.x}, Vec3{b.ax.x, b.ay.x, b.az.x}, b.w.x, a);
    if (rem > 1) c1 = eval_point(T, Vec3{b.px.y, b.py.y, b.pz.y}, Vec3{b.ax.y, b.ay.y, b.az.y}, b.w.y, a);
  }
  // the `mutable double* cost` side channel (registration.hpp:51,76,96): written on EVERY sweep
  if (NT) {
    // streaming store: the slots are not read again before the weight kernel, keep them out of the way of the loads
    // and out of the end-of-kernel write-back
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (rem > 1) __builtin_nontemporal_store(v2d{c0, c1}, reinterpret_cast<v2d*>(seg.cost + j));
    else __builtin_nontemporal_store(c0, seg.cost + j);
  } else {
    if (rem > 1) *reinterpret_cast<double2*>(seg.cost + j) = double2{c0, c1};
    else seg.cost[j] = c0;
  }
  if (cost_out) *cost_out = double2{c0, c1};
}

// All chunks i0, i0+W, i0+2W, ... (< nchunks) of one segment.  DEPTH 2: software-pipelined, the loads of
// the next chunk are in flight while the current one is evaluated (planes: 76 % of the bytes).  DEPTH 1:
// plain load-then-evaluate (lines/points: keeps the register budget at 3 waves/SIMD).  Control flow is
// wave-uniform.
template <int RES, int DEPTH, bool NT>
__device__ __forceinline__ void sweep_segment(const Rt& T, const CorrSeg& seg, int n, int nchunks, int i0, int W,
                                              int lane, Acc& a, const ChunkBuf<RES>& pre, bool use_pre) {
  if (i0 >= nchunks) return;
  const int m = (nchunks - i0 + W - 1) / W;  // chunks owned by this wave
  const int l2 = lane * 2;
#define TL_J(t) ((i0 + (t) * W) * kChunk + l2)
  if (DEPTH == 1) {
    for (int t = 0; t < m; ++t) {
      ChunkBuf<RES> b;
      if (t == 0 && use_pre) b = pre;  // requested before the state's scalar loads came back
      else fetch<RES, NT>(seg, TL_J(t), b);
      consume<RES, NT>(T, seg, TL_J(t), n, b, a);
    }
  } else if (DEPTH == 2) {
    ChunkBuf<RES> b0, b1;
    if (use_pre) b0 = pre;  // chunk i0 was requested before the kernel even knew the segment sizes
    else fetch<RES, NT>(seg, TL_J(0), b0);
    for (int t = 1;; t += 2) {
      if (t >= m) { consume<RES, NT>(T, seg, TL_J(t - 1), n, b0, a); break; }
      fetch<RES, NT>(seg, TL_J(t), b1);
      consume<RES, NT>(T, seg, TL_J(t - 1), n, b0, a);
      if (t + 1 >= m) { consume<RES, NT>(T, seg, TL_J(t), n, b1, a); break; }
      fetch<RES, NT>(seg, TL_J(t + 1), b0);
      consume<RES, NT>(T, seg, TL_J(t), n, b1, a);
    }
  }
#undef TL_J
}

#ifndef TLOAM_K3_WAVES
#define TLOAM_K3_WAVES 4  // register budget of 128: the one-chunk-deep sweep needs 118
#endif
#ifndef TLOAM_K3_PLANE_DEPTH
#define TLOAM_K3_PLANE_DEPTH 1
#endif
#ifndef TLOAM_K3_LINE_DEPTH
#define TLOAM_K3_LINE_DEPTH 1
#endif
// one full sweep of this wave's share of the four segments
__device__ __forceinline__ void sweep_all(const CorrView& cv, const int* __restrict__ seg_n, const Rt& T, int gw, int W,
                                          int lane, Acc& a, const ChunkBuf<TLOAM_RES_PLANE>& pre0, bool use_pre0) {
  const ChunkBuf<TLOAM_RES_LINE> no_line{};
  const ChunkBuf<TLOAM_RES_POINT> no_point{};
#pragma unroll
  for (int i = 0; i < 27; ++i) a.v[i] = 0.0;
  a.pm = 0.5;
  a.pe = 1;  // 0.5 * 2^1 = 1
  // chunk g of the concatenated (planar | ground | edge | sphere) chunk list belongs to wave g % W:
  // within segment k the wave starts at i0 = (gw - first_k) mod W -- balanced across segments
  int first = 0;
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    const int n = seg_n[k];
    const int nchunks = (n + kChunk - 1) / kChunk;
    int i0 = (gw - first) % W;
    if (i0 < 0) i0 += W;
    if (k <= TLOAM_KIND_GROUND)
      sweep_segment<TLOAM_RES_PLANE, TLOAM_K3_PLANE_DEPTH, TLOAM_K3_NT>(T, cv.k[k], n, nchunks, i0, W, lane, a, pre0, use_pre0 && k == 0);
    else if (k == TLOAM_KIND_EDGE) sweep_segment<TLOAM_RES_LINE, TLOAM_K3_LINE_DEPTH, TLOAM_K3_NT>(T, cv.k[k], n, nchunks, i0, W, lane, a, no_line, false);
    else sweep_segment<TLOAM_RES_POINT, TLOAM_K3_LINE_DEPTH, TLOAM_K3_NT>(T, cv.k[k], n, nchunks, i0, W, lane, a, no_point, false);
    first = (first + nchunks) % W;
  }
}
// Small correspondence sets (KITTI caps: <= 5.9 k factors): the grid has one wave per chunk of the
// concatenated (planar | ground | edge | sphere) list, so wave gw owns exactly chunk gw -- one fetch, one
// evaluation, no chunk loop and no modulo distribution.  A line or point correspondence costs ~3x the instructions of a
// plane (three residual rows), and the launch ends when its slowest wave does: the chunks of those kinds are 64
// correspondences (one per lane) instead of 128, which evens the waves out (single_chunk_of).
#ifndef TLOAM_SMALL_LINE_CHUNK
#define TLOAM_SMALL_LINE_CHUNK 64
#endif
__host__ __device__ constexpr int single_chunk_of(int kind) { return kind <= TLOAM_KIND_GROUND ? kChunk : TLOAM_SMALL_LINE_CHUNK; }
// one correspondence per lane (the 64-chunks): the streams as 8-byte loads into the first slot of the chunk buffer
template <int RES>
__device__ __forceinline__ void fetch_one(const CorrSeg& seg, int j, ChunkData& b) {
  b.px.x = seg.px[j]; b.py.x = seg.py[j]; b.pz.x = seg.pz[j];
  b.ax.x = seg.ax[j]; b.ay.x = seg.ay[j]; b.az.x = seg.az[j];
  b.w.x = seg.w[j];
  if (RES == TLOAM_RES_PLANE) b.d.x = seg.d[j];
  if (RES == TLOAM_RES_LINE) { b.bx.x = seg.bx[j]; b.by.x = seg.by[j]; b.bz.x = seg.bz[j]; }
}
// The chunk of wave gw: the kinds' chunk ranges follow each other in CAPACITY order (capacities are kernel arguments:
// known before any device memory has been read), so every wave can request its streams in its first instructions --
// the planar segment straight from the preloaded (base, stride, capacity), the others as soon as the kernel-argument
// block has been read -- while the minimiser state (done flag, pose) and the segment sizes are still on their way.
// (Mapping by the actual sizes made every wave but the first planar ones wait for state -> sizes -> data: three
// dependent trips, ~0.8 us on the launch's slowest waves.)  A chunk beyond its segment's size does nothing.
struct SingleWork {
  int kind;   // -1: no chunk (grid padding)
  int j;      // first correspondence of this lane
};
__device__ __forceinline__ SingleWork single_work_of(const CorrView& cv, int cap0, int gw, int lane) {
  SingleWork wk{-1, 0};
  // wave 0 of the grid holds no chunk (in the one-launch Solve the lead's stepper has the state, the set sizes and the
  // result slots to look after), and the one-launch-per-iteration kernels map the same way so that all of them add the
  // same waves up in the same rows
  int g = gw - 1;
  if (g < 0) return wk;
  const int n0 = cap0 / kChunk;   // (preloaded: the planar waves need nothing else)
  if (g < n0) { wk.kind = 0; wk.j = g * kChunk + lane * 2; return wk; }
  g -= n0;
#pragma unroll
  for (int k = 1; k < kKinds; ++k) {
    const int c = single_chunk_of(k), nk = cv.k[k].cap / c;
    if (wk.kind < 0 && g < nk) { wk.kind = k; wk.j = (c == kChunk) ? g * kChunk + lane * 2 : g * 64 + lane; }
    g -= nk;
  }
  if (wk.kind >= 0 && g >= 0) {}  // (g has gone negative once a kind matched)
  return wk;
}
__device__ __forceinline__ void single_fetch(const CorrView& cv, const double* __restrict__ seg0, int stride0, const SingleWork& wk,
                                             ChunkData& b, bool w2 = false) {
  if (wk.kind == TLOAM_KIND_PLANAR) fetch_spec<false>(seg0, stride0, wk.j, b, w2);
  else if (wk.kind == TLOAM_KIND_GROUND) fetch<TLOAM_RES_PLANE, false>(cv.k[TLOAM_KIND_GROUND], wk.j, b);
  else if (wk.kind == TLOAM_KIND_EDGE) {
    if (single_chunk_of(TLOAM_KIND_EDGE) == kChunk) fetch<TLOAM_RES_LINE, false>(cv.k[TLOAM_KIND_EDGE], wk.j, b);
    else fetch_one<TLOAM_RES_LINE>(cv.k[TLOAM_KIND_EDGE], wk.j, b);
  } else if (wk.kind == TLOAM_KIND_SPHERE) {
    if (single_chunk_of(TLOAM_KIND_SPHERE) == kChunk) fetch<TLOAM_RES_POINT, false>(cv.k[TLOAM_KIND_SPHERE], wk.j, b);
    else fetch_one<TLOAM_RES_POINT>(cv.k[TLOAM_KIND_SPHERE], wk.j, b);
  }
}
__device__ __forceinline__ void sweep_single_n(const CorrView& cv, int n, const Rt& T, const SingleWork& wk,
                                               const ChunkData& b, Acc& a, double2* cost_out = nullptr) {
#pragma unroll
  for (int i = 0; i < 27; ++i) a.v[i] = 0.0;
  a.pm = 0.5;
  a.pe = 1;
  if (wk.kind < 0) return;
  if (wk.kind <= TLOAM_KIND_GROUND) {
    if (wk.kind == TLOAM_KIND_PLANAR) consume<TLOAM_RES_PLANE, false>(T, cv.k[TLOAM_KIND_PLANAR], wk.j, n, b, a, cost_out);
    else consume<TLOAM_RES_PLANE, false>(T, cv.k[TLOAM_KIND_GROUND], wk.j, n, b, a, cost_out);
  } else if (wk.kind == TLOAM_KIND_EDGE) {
    const int ne = single_chunk_of(TLOAM_KIND_EDGE) == kChunk ? n : (n < wk.j + 1 ? n : wk.j + 1);   // 64-chunks: this lane's one
    consume<TLOAM_RES_LINE, false>(T, cv.k[TLOAM_KIND_EDGE], wk.j, ne, b, a, cost_out);
  } else {
    const int ne = single_chunk_of(TLOAM_KIND_SPHERE) == kChunk ? n : (n < wk.j + 1 ? n : wk.j + 1);
    consume<TLOAM_RES_POINT, false>(T, cv.k[TLOAM_KIND_SPHERE], wk.j, ne, b, a, cost_out);
  }
}
__device__ __forceinline__ void sweep_single(const CorrView& cv, const int* __restrict__ seg_n, const Rt& T, const SingleWork& wk,
                                             const ChunkData& b, Acc& a) {
  sweep_single_n(cv, wk.kind >= 0 ? seg_n[wk.kind] : 0, T, wk, b, a);
}
// wave total of the 28 sums: component c ends up (complete) in lanes 2c and 2c+1
__device__ __forceinline__ double wave_reduce_acc(const Acc& a, int lane) {
  double v[32];
#pragma unroll
  for (int i = 0; i < 27; ++i) v[i] = a.v[i];
  v[27] = 0.5 * (log(a.pm) + (double)a.pe * 0.6931471805599453094);  // 0.5 * sum log(1+s)
  v[28] = v[29] = v[30] = v[31] = 0.0;
  rs_step_swap<32, 32>(v);
  rs_step_swap<16, 16>(v);
  rs_step<8, 8>(v, lane);
  rs_step<4, 4>(v, lane);
  rs_step<2, 2>(v, lane);
  return v[0] + __shfl_xor(v[0], 1, 64);
}

// SINGLE = false: the streaming variant (grid = the resident chip, every wave loops over its chunks with
// software-pipelined loads).  SINGLE = true: small sets, one wave per chunk (sweep_single).

// ---- sharded contexts: the last block of the sweep folds the rows and hands the totals on -------------------
// system-scope accesses to the (uncached, fine-grained) mailbox memory
__device__ __forceinline__ void mbox_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double mbox_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// rank `mb.rank` posts vals[0 .. count) (count <= kMboxSlot - 1) as exchange `id` into every rank's buffer.
// Called by (at least) the first 64 threads of a block; `vals` in LDS or registers of lane c.
__device__ __forceinline__ void mbox_post(const MboxView& mb, unsigned long long id, const double* vals, int count, int tid) {
  const size_t slot = ((size_t)(id & 1ull) * kMaxRanks + (size_t)mb.rank) * kMboxSlot;
  if (tid < 64) {
    for (int r = 0; r < mb.nranks; ++r)
      if (tid < count) mbox_store(mb.peer[r] + slot + tid, vals[tid]);
    __threadfence_system();  // the values before the id, on every link
    if (tid < mb.nranks)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peer[tid] + slot + (kMboxSlot - 1)), id, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// waits (bounded: ~2 s of the 100 MHz wall clock) for exchange `id` from every rank, then out[c] = sum over ranks
// in rank order for c < count.  One wave; returns false on time-out (a peer died / never posted).
__device__ __forceinline__ bool mbox_gather(const MboxView& mb, unsigned long long id, double* out /* LDS */, int count, int lane) {
  const double* mine = mb.peer[mb.rank] + (size_t)(id & 1ull) * kMaxRanks * kMboxSlot;
  const unsigned long long t0 = wall_clock64();
  bool ok = true;
  if (lane < mb.nranks) {
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(mine + (size_t)lane * kMboxSlot + (kMboxSlot - 1));
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != id) {
      if (wall_clock64() - t0 > 200000000ull) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  ok = __all(ok ? 1 : 0) != 0;
  __threadfence_system();
  if (lane < count) {
    double t = 0.0;
    for (int r = 0; r < mb.nranks; ++r) t += mbox_load(mine + (size_t)r * kMboxSlot + lane);
    out[lane] = t;
  }
  return ok;
}

// The block's row is handed over with DEVICE-scope stores (written through to memory: the eight XCDs' L2s do not snoop
// each other) followed by a wait for their completion and the ticket; the last block reads the rows with device-scope
// loads.  No cache-wide release / acquire (a `__threadfence()` here writes back and invalidates the whole L2 of the
// XCD: that is what made round 1's fused sweep + step cost what the launch boundary it removed did).
// the block's row from its waves' totals: four waves in order; a WIDE block (eight waves, launch_k3) adds its two halves
template <int W>
__device__ __forceinline__ double block_row(const double (*red)[32], int c) {
  double t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
#pragma unroll
  for (int q = 4; q < W; q += 4) t += ((red[q][c] + red[q + 1][c]) + red[q + 2][c]) + red[q + 3][c];
  return t;
}
template <int W = 4>
__device__ __forceinline__ bool k3_take_ticket(double* __restrict__ partials, const double (*red)[32], int* ticket) {
  __shared__ int s_last;
  if (threadIdx.x < kAccStride)
    __hip_atomic_store(partials + (size_t)blockIdx.x * kAccStride + threadIdx.x, block_row<W>(red, threadIdx.x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // The row must have REACHED the coherence point before the ticket is taken: row store and ticket RMW travel through
  // different L2 channels and are not ordered with each other.  A workgroup-scope release emits no vmcnt wait on gfx950
  // (non-tgsplit), an agent-scope release would write back the XCD's whole L2 -- so wait for the wave's own write-through
  // (sc1) stores explicitly, then meet the other waves.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                          // ... in every wave of the block
  if (threadIdx.x == 0)
    s_last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  return s_last != 0;
}
// Fixed-order fold of the per-block rows by 256 threads -> tot[0 .. kReduceBuf) in LDS: 8 row groups x 32 columns,
// sixteen independent loads in flight per thread (a 489-row fold is four rounds), one tree -- the SAME tree whether the
// rows are folded by the last block of the sweep (AGENT: device-scope loads, the rows come from other XCDs), by the
// separate reduce kernel or by the step kernel, so the paths agree bit for bit.
template <bool AGENT, int kP = 1>
__device__ __forceinline__ void fold_rows(const double* __restrict__ partials, int rows, double* s_grp /*[8*33]*/, double* s_tot) {
  const int comp = threadIdx.x & 31, grp = threadIdx.x >> 5;
  constexpr int kU = 16;
  double v[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) v[u] = 0.0;
  // (a wide block -- eight waves, k3_plan -- folds with its first 256 threads; the others only meet the barriers)
  const int rows_mine = threadIdx.x < 256 ? rows : 0;
  // FOUR passes of sixteen loads requested before the first is added (a full-chip grid of 512 rows is four passes: four dependent
  // trips to memory became one -- round 6: the 1 M GN iteration 21.8 -> 21.5 us, three interleaved pairs on one box); the additions
  // keep the order of the one-pass-at-a-time form, so the sums keep their bits
  // (kP = 4 where the kernel has the registers -- the step kernels, one wave per SIMD; the sweeps that fold in their last block run
  //  at 128 registers per wave and keep one pass in flight: 64 more doubles there are scratch)
  for (int b00 = grp; b00 < rows_mine; b00 += kP * kU * 8) {
    double x[kP][kU];
#pragma unroll
    for (int q = 0; q < kP; ++q)
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const double* p = partials + (size_t)min(b00 + (q * kU + u) * 8, rows - 1) * kAccStride + comp;
        x[q][u] = AGENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
      }
#pragma unroll
    for (int q = 0; q < kP; ++q) {
      if (b00 + q * kU * 8 >= rows) break;   // (a pass the one-pass form would not have run: nothing is added, not even zeros)
#pragma unroll
      for (int u = 0; u < kU; ++u) v[u] += (b00 + (q * kU + u) * 8 < rows) ? x[q][u] : 0.0;
    }
  }
#pragma unroll
  for (int w = kU / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) v[u] += v[u + w];
  if (threadIdx.x < 256) s_grp[grp * 33 + comp] = v[0];
  __syncthreads();
  if (threadIdx.x < kReduceBuf) {
    double t = 0.0;
    if (threadIdx.x < kAccN)
#pragma unroll
      for (int g = 0; g < 8; ++g) t += s_grp[g * 33 + threadIdx.x];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
}
// ---- KITTI-size grids (<= kTaggedRows blocks): the rows ARE the flag ---------------------------------------------
// Ticket hand-over above = row stores -> wait for their completion -> returning atomic -> (last block) row loads: three
// dependent trips to the coherence point, ~2.2 us per GN iteration of a KITTI-size frame (phase stamps: 0.85 us from the
// end of the sweep to the ticket, 1.35 us for the fold).  For a grid of a dozen blocks a FIXED consumer (block 0) that
// polls the rows themselves needs one: every block stores its row as four 64-byte segments of (seven sums, check word)
// with ONE write-through store instruction, check = tag XOR the seven sums, tag = a launch counter every block reads at
// its start and the consumer advances at its end; the consumer re-reads all rows (device-scope loads) until every
// segment checks, then folds them in the order fold_rows uses.  No ordering between any two words is relied on: a
// segment that has only partly arrived -- or still holds an older launch's sums -- does not check.  All blocks of such a
// grid are resident at once (a dozen blocks on 256 CUs); the spin is bounded all the same (~1 s, then the Solve is
// stopped with GnState::comm_error).
constexpr int kTaggedRows = 16;
// (The ROW segments below enter their check word by a plain XOR, not through seg_word: their payload is seven fp64 SUMS that
//  change from one GN iteration to the next.  A segment that mixes old and new words passes only if the words that are still old
//  changed by the same 64-bit pattern: one such word -> it did not change, the segment reads as what it is; two -> two sums whose
//  old and new bit patterns XOR alike, which for sums of squares and products of residuals means both unchanged or both a pure
//  sign flip of a zero -- values that add to the same totals either way.  The poll sits on the dependent chain of every GN
//  iteration of a KITTI-size frame; seven 64-bit multiplies per look are not spent there.  Segments with integers in them --
//  post_ext_segment, the host's slots -- do go through seg_word.)
__device__ __forceinline__ unsigned long long xor8(unsigned long long x) {  // XOR over aligned groups of eight lanes
  x ^= __shfl_xor(x, 1, 64);
  x ^= __shfl_xor(x, 2, 64);
  x ^= __shfl_xor(x, 4, 64);
  return x;
}
__device__ __forceinline__ void k3_post_row_tagged(double* __restrict__ partials, const double (*red)[32], unsigned long long tag) {
  if (threadIdx.x < kAccStride) {
    const int t = threadIdx.x, seg = t >> 3, pos = t & 7, c = seg * 7 + pos;   // word t of the row carries sum c (pos < 7)
    const double val = pos < 7 ? ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c] : 0.0;
    unsigned long long w = (unsigned long long)__double_as_longlong(val);
    const unsigned long long x = xor8(w);
    if (pos == 7) w = check_mix(tag) ^ x;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(partials) + (size_t)blockIdx.x * kAccStride + t, w, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ONE wave of the consumer block; rows <= kTaggedRows = 16, i.e. at most 64 segments: lane = (row, segment).  Every lane
// re-reads the eight words of its segment (device-scope loads) and XORs them in registers -- no cross-lane exchange, no LDS,
// no barrier per poll -- until every segment of every row checks; the sums then go through LDS once and are added in the
// order fold_rows uses (so the fused iteration and sweep + separate reduce agree bit for bit).
// false: timed out (a block of the grid never posted).  s_rows: LDS, kTaggedRows * 28 doubles.
__device__ __forceinline__ bool poll_fold_tagged(const double* __restrict__ partials, int rows, unsigned long long tag,
                                                 double* s_rows, double* s_tot, int lane) {
  const int r = lane >> 2, sgm = lane & 3;
  const bool have = r < rows;
  const unsigned long long mtag = check_mix(tag);
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(partials) + (size_t)(have ? r : 0) * kAccStride + sgm * 8;
  unsigned long long w[8];
  const unsigned long long t0 = wall_clock64();
  bool ok_all;
  // (read - check - sleep: one look per round trip to the coherence point.  Two looks in flight -- the next one issued before
  //  this one is checked -- measured 1 % SLOWER on the headline frame, three interleaved pairs on one box: 0.1749 against
  //  0.1729 ms; the extra reads of sixteen pollers compete with the row stores they wait for.  Nor does it pay to hold the FIRST
  //  look back until the late blocks have posted: 4 / 8 / 16 / 32 s_sleep units in front of it measured 0.1735-0.1747 /
  //  0.1731-0.1748 / 0.1743-0.1774 / 0.1790-0.1792 against 0.1711-0.1721 ms, three interleaved rounds.  Round 4, not kept.)
  for (unsigned spins = 1;; ++spins) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long x = w[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) x ^= w[i];
    ok_all = __all((!have || x == mtag) ? 1 : 0) != 0;
    if (ok_all) break;
    if ((spins & 63u) == 0 && wall_clock64() - t0 > 100000000ull) break;   // ~1 s of the 100 MHz wall clock
    __builtin_amdgcn_s_sleep(1);
  }
  if (have) {
#pragma unroll
    for (int i = 0; i < 7; ++i) s_rows[r * 28 + sgm * 7 + i] = __longlong_as_double((long long)w[i]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its LDS stores are in order; this also stops the compiler
  if (lane < kReduceBuf) {
    double t = 0.0;
    if (lane < kAccN) {
      // fold_rows: group g holds rows g and g + 8 (of sixteen values per thread fourteen are zero here), the groups are
      // added in order -- literally, zeros included
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const double a = g < rows ? s_rows[g * 28 + lane] : 0.0;
        const double b = g + 8 < rows ? s_rows[(g + 8) * 28 + lane] : 0.0;
        // (fold_rows adds fourteen zeros to these two: x + 0.0 is x except for x = -0.0, and a -0.0 -- or +0.0 -- added to
        //  an accumulator that started at +0.0 leaves it as it is: the same bits without the fourteen additions)
        t += a + b;
      }
    }
    s_tot[lane] = t;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return ok_all;
}
template <int W = 4>
__device__ __forceinline__ void k3_last_block_reduce(double* __restrict__ partials, const double (*red)[32], const K3Fuse& fuse) {
  __shared__ double s_grp[8 * 33];
  __shared__ double s_tot[kReduceBuf];
  if (!k3_take_ticket<W>(partials, red, fuse.ticket)) return;
  fold_rows<true>(partials, (int)gridDim.x, s_grp, s_tot);
  if (threadIdx.x < kReduceBuf) fuse.out48[threadIdx.x] = s_tot[threadIdx.x];
  if (threadIdx.x == 0) __hip_atomic_store(fuse.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
  if (fuse.mb.nranks > 0) {
    const unsigned long long id = __hip_atomic_load(fuse.mb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    mbox_post(fuse.mb, id, s_tot, kReduceBuf, threadIdx.x);
  }
}

// Argument order: the first twelve dwords -- the planar segment as (base, stride, capacity), the flag, the state,
// the segment sizes and the output rows -- are preloaded into SGPRs by the command processor
// (-mllvm -amdgpu-kernarg-preload-count=12, see build.py), so the wave's first chunk AND the state's scalar loads
// are requested in the first instructions, before the kernel-argument segment itself has been read.
// W: waves per block.  4, two blocks per CU; 8 (WIDE, streaming form on a full-chip grid): one block per CU, the same waves over the
// same chunks, HALF the rows for the one block that folds them afterwards.
template <bool SINGLE, bool FUSE, int W = 4>
__global__ __launch_bounds__(64 * W, TLOAM_K3_WAVES) void k3_accumulate(const double* __restrict__ seg0, int stride0, int cap0,
                                                        int force, GnState* __restrict__ st,
                                                        const int* __restrict__ seg_n, double* __restrict__ partials,
                                                        CorrView cv, K3Fuse fuse) {
  __shared__ double red[W][32];
  // the wave index is wave-uniform: tell the compiler (readfirstlane) so that chunk -> segment
  // pointers are scalar (SGPR) work instead of per-lane loads of the kernel-argument table
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * W + wave;
  // Speculative first fetch: the wave's first chunk of the planar segment is requested straight from the
  // preloaded arguments, BEFORE the dependent scalar loads of the state (done flag, pose, segment sizes)
  // come back -- their latency overlaps the first HBM round trip.  The capacity bound keeps it in range.
#ifdef TLOAM_K3_PROFILE
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
  const unsigned long long wc0 = wall_clock64();  // 100 MHz, one base for the whole device
#endif
  ChunkData pre;
  SingleWork wk{-1, 0};
  const bool spec = !SINGLE && (TLOAM_K3_PLANE_DEPTH <= 2) && (gw + 1) * kChunk <= cap0;
  const bool w2 = (force & 2) != 0;   // (bit 1 of the flag word: the planar weights are in the second stream)
  force &= 1;
  if (SINGLE) {
    wk = single_work_of(cv, cap0, gw, lane);
    single_fetch(cv, seg0, stride0, wk, pre, w2);
  } else if (spec) {
    fetch_spec<TLOAM_K3_NT>(seg0, stride0, gw * kChunk + lane * 2, pre, w2);
  }
  if (!force && st->done) return;  // after a tolerance exit the remaining launches are no-ops
  const Rt T = st->Rt_eval;        // exp(point), hoisted out of the per-block Evaluate (:22,:58,:98)
  Acc a;
#ifdef TLOAM_K3_PROFILE
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  if (SINGLE) sweep_single(cv, seg_n, T, wk, pre, a);
  else sweep_all(cv, seg_n, T, gw, gridDim.x * W, lane, a, pre, spec);
#ifdef TLOAM_K3_PROFILE
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  const double tot = wave_reduce_acc(a, lane);
  if ((lane & 1) == 0) red[wave][lane >> 1] = tot;
  __syncthreads();
  if (FUSE) {
    k3_last_block_reduce<W>(partials, red, fuse);
  } else if (threadIdx.x < kAccStride) {
    partials[(size_t)blockIdx.x * kAccStride + threadIdx.x] = block_row<W>(red, threadIdx.x);
  }
#ifdef TLOAM_K3_PROFILE
  if (threadIdx.x == 0) {  // development aid (scripts/k3_profile.py): wave 0's timeline in the spare columns
    const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
    double* row = partials + (size_t)blockIdx.x * kAccStride;
    row[28] = (double)(wc0 & 0xffffffffull); row[29] = (double)(ts1 - ts0); row[30] = (double)(ts2 - ts1);
    row[31] = (double)(ts3 - ts2) + 65536.0 * (double)(wall_clock64() - wc0);
  }
#endif
}


// blocks of the streaming sweep the device holds at once: two per CU.  The CU count is the context's (tloam_ctx::device_cus,
// hipDeviceAttributeMultiprocessorCount); a device that reports none is taken for the full part
static int k3_resident_blocks(int device_cus) { return (device_cus > 0 ? device_cus : 256) * 2; }
int k3_grid_for(int total_cap, int device_cus) {
  if (const char* e = getenv("TLOAM_K3_BLOCKS")) {  // tuning aid
    const int b = atoi(e);
    if (b > 0) return b;
  }
  // one wave per 128-correspondence chunk up to a full-chip resident grid (the device's CUs x 2 blocks: 512 on an MI355X)
  int waves = (total_cap + kChunk - 1) / kChunk;
  int blocks = (waves + 3) / 4;
  if (blocks < 1) blocks = 1;

  const int resident = k3_resident_blocks(device_cus);  // two blocks of 4 waves per CU (2 waves per SIMD): more resident waves only lengthen the dispatch ramp
  if (blocks > resident) {
    // balance: every wave gets the same number of chunks
    const int per_wave = (waves + resident * 4 - 1) / (resident * 4);
    const int need_waves = (waves + per_wave - 1) / per_wave;
    blocks = (need_waves + 7) / 8 * 2;   // an even count: a full-chip grid goes out as wide blocks of eight waves (k3_plan)
  }
  return blocks;
}
// The sweep of a set with segment capacities cap[k] (multiples of kChunk): one wave per chunk (sweep_single: chunks of
// single_chunk_of(kind) correspondences) while that fits the resident chip, the streaming variant otherwise.
// wide: the streaming sweep of a grid of at least 1.5 four-wave blocks per CU goes out as HALF as many blocks of EIGHT waves -- the
// same waves over the same chunks with the same registers, one block per CU instead of two, and half the rows for the one block
// that folds them afterwards (k_reduce_and_step, the last block of the fused forms): the fold is a stream of 256 B per row through
// ONE CU, ~13 cycles per row -- round 6, 1 M frame: fold 7500 -> 4100 cycles, GN iteration 22.6 -> 21.0 us, three interleaved pairs;
// blocks of sixteen waves leave half the CUs idle: K3 12.3 -> 17.3 us).  *grid = blocks LAUNCHED = rows written.
void k3_plan(const int cap[kKinds], int device_cus, int* grid, bool* single, bool* wide) {
  long long waves = 0, total = 0;
  *wide = false;
  for (int k = 0; k < kKinds; ++k) {
    waves += (cap[k] + single_chunk_of(k) - 1) / single_chunk_of(k);
    total += cap[k];
  }
  waves += 1;   // wave 0 of the grid holds no chunk (single_work_of)
  if (waves <= (long long)k3_resident_blocks(device_cus) * 4 && !getenv("TLOAM_K3_BLOCKS")) {
    *single = true;
    *grid = (int)((waves + 3) / 4) < 1 ? 1 : (int)((waves + 3) / 4);
    return;
  }
  *grid = k3_grid_for((int)total, device_cus);
  *single = ((total + kChunk - 1) / kChunk <= (long long)*grid * 4) && TLOAM_SMALL_LINE_CHUNK == kChunk;
  static const bool no_wide = getenv("TLOAM_K3_NARROW") != nullptr;   // tuning aid / A-B
  if (!*single && !no_wide && *grid % 2 == 0 && 2 * *grid >= 3 * (device_cus > 0 ? device_cus : 256)) {
    *wide = true;
    *grid /= 2;
  }
}
// bit 1 of the sweep kernels' flag word: the planar segment's weight pointer is the segment's SECOND weight stream (the planar
// streams are addressed from the preloaded (base, stride), not through the view -- see fetch_spec)
static int w2_flag(const CorrView& cv) {
  return (cv.k[0].w != nullptr && cv.k[0].w == cv.k[0].px + (size_t)(SS_W2 - SS_PX) * (size_t)cv.k[0].stride) ? 2 : 0;
}
void launch_k3(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, bool force, hipStream_t s,
               hipEvent_t ev_start, hipEvent_t ev_stop) {
  auto kern = single ? k3_accumulate<true, false> : wide ? k3_accumulate<false, false, 8> : k3_accumulate<false, false>;
  const int threads = wide ? 512 : 256;
  K3Fuse none;
  memset(&none, 0, sizeof(none));
  if (ev_start && ev_stop) {
    // HIP events bound to THIS dispatch (start/stop taken from the kernel's own dispatch packet):
    // their elapsed time is the kernel duration itself, the number rocprofv3 --kernel-trace reports
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, ev_start, ev_stop, 0, (const double*)cv.k[0].px, cv.k[0].stride,
                          cv.k[0].cap, (force ? 1 : 0) | w2_flag(cv), st, cv.seg_n, partials, cv, none);
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, (const double*)cv.k[0].px, cv.k[0].stride, cv.k[0].cap, (force ? 1 : 0) | w2_flag(cv), st,
                       cv.seg_n, partials, cv, none);
  }
}
void launch_k3_fused(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, bool force,
                     const K3Fuse& fuse, hipStream_t s) {
  auto kern = single ? k3_accumulate<true, true> : wide ? k3_accumulate<false, true, 8> : k3_accumulate<false, true>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(wide ? 512 : 256), 0, s, (const double*)cv.k[0].px, cv.k[0].stride, cv.k[0].cap, (force ? 1 : 0) | w2_flag(cv), st,
                     cv.seg_n, partials, cv, fuse);
}

// ================================================================================================
//  fixed-order reduction of the per-block rows (one block of 256 threads: 8 row groups x 32 columns)
// ================================================================================================
constexpr int kRedThreads = 256;  // one wave per SIMD: the step's wave may use the whole register file (no scratch)
__global__ __launch_bounds__(kRedThreads) void k_reduce(const double* __restrict__ partials, int rows,
                                                        const GnState* __restrict__ st, double* __restrict__ out48) {
  __shared__ double lds[8 * 33];
  __shared__ double tot[kReduceBuf];
  (void)st;
  fold_rows<false, 4>(partials, rows, lds, tot);
  if (threadIdx.x < kReduceBuf) out48[threadIdx.x] = (threadIdx.x < kAccN) ? tot[threadIdx.x] : 0.0;
}
void launch_reduce(const double* partials, int grid, GnState* st, double* out48, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kRedThreads), 0, s, partials, grid, st, out48);
}

}  // namespace tl
#include "tl_step.hpp"   // K5: the minimiser step (gn_consume)
namespace tl {

// ---- the period of a GN iteration, measured on the device (bench aid: tloam_gn_iter_timer; span may be null) ---------------
// a GN iteration = one sweep + reduction (+ exchange) + 6x6 step + pose update.  Every kernel that ends one -- the
// step kernels of the launch-per-iteration forms, the lead's stepper inside the one-launch Solve -- stamps the device's 100 MHz
// wall clock when its step is done: span[0] = that stamp, span[1] += stamp - span[0] for every iteration that FOLLOWS another one
// of the same Solve (the first has no stamp to start from: the kernel in front of it is a search or a compaction), span[2] += 1.
// The period therefore holds everything between two poses: sweep, launch boundaries, fold, exchange, step.  One thread.
__device__ __forceinline__ void iter_span_note(unsigned long long* span, bool first_of_solve) {
  if (!span) return;
  const unsigned long long now = wall_clock64();
  if (!first_of_solve) { span[1] += now - span[0]; span[2] += 1ull; }
  span[0] = now;
}

__global__ void k_solve_init(GnState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->T_cur = se3_exp(st->x);
  arm_solver(*st);
}
void launch_solve_init(GnState* st, hipStream_t s) { hipLaunchKernelGGL(k_solve_init, dim3(1), dim3(64), 0, s, st); }

// Start of a scan_match as a launch of its own (frame_init_body, tl_common.hpp); normally the first launch of the grid
// build carries it (k_grid_count_all)
__global__ __launch_bounds__(256) void k_frame_init(FrameInit fi, FrameInitBufs b) {
  frame_init_body(fi, b, (int)blockIdx.x, (int)gridDim.x);
}
void launch_frame_init(const FrameInit& fi, const FrameInitBufs& b, hipStream_t s) {
  const int n = fi.slot_off[kKinds];
  int blocks = (n + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_frame_init, dim3(blocks), dim3(256), 0, s, fi, b);
}

// a copy of the state whose builder pose is that of the set's last build: the rebuild of a stale direct set (tloam_ctx::set_stale)
__global__ void k_pose_from_x_build(GnState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->T_cur = se3_exp(st->x_build);
}
void launch_pose_from_x_build(GnState* st, hipStream_t s) { hipLaunchKernelGGL(k_pose_from_x_build, dim3(1), dim3(64), 0, s, st); }

__global__ void k_set_eval(GnState* st, const double* se3) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[6];
  for (int i = 0; i < 6; ++i) a[i] = se3[i];
  st->T_eval = se3_exp(a);
  st->Rt_eval = to_rt(st->T_eval);
}
void launch_set_eval(GnState* st, const double* se3_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_set_eval, dim3(1), dim3(64), 0, s, st, se3_dev);
}

// one wave: the state into LDS (three coalesced rounds)
__device__ __forceinline__ void load_state_lds(const GnState* st, GnState* sm, int lane) {
  constexpr int kWords = (int)(sizeof(GnState) / 8);
  for (int i = lane; i < kWords; i += 64)
    reinterpret_cast<unsigned long long*>(sm)[i] = reinterpret_cast<const unsigned long long*>(st)[i];
}
__global__ __launch_bounds__(64) void k_gn_step(GnState* st, const double* __restrict__ in48, unsigned long long* iter_span) {
  __shared__ double tot[kReduceBuf];
  __shared__ double scr[32];
  __shared__ GnState s_in;
  if (threadIdx.x < kReduceBuf) tot[threadIdx.x] = in48[threadIdx.x];
  load_state_lds(st, &s_in, threadIdx.x);
  __syncthreads();
  if (s_in.done) return;
  const bool first = s_in.phase == PH_ITER0;
  gn_consume(st, tot, threadIdx.x, &s_in, scr);
  if (threadIdx.x == 0) iter_span_note(iter_span, first);
}
void launch_gn_step(GnState* st, const double* in48, hipStream_t s, unsigned long long* iter_span) {
  hipLaunchKernelGGL(k_gn_step, dim3(1), dim3(64), 0, s, st, in48, iter_span);
}
// mailbox contexts: wait for every rank's totals of this sweep (posted by the last block of its K3), add them in
// rank order and advance the minimiser -- identically on every rank
__global__ __launch_bounds__(64) void k_gn_step_mbox(GnState* st, MboxView mb, unsigned long long* iter_span) {
  __shared__ double tot[kMboxSlot];
  __shared__ double scr[32];
  __shared__ GnState s_in;
  load_state_lds(st, &s_in, threadIdx.x);
  __syncthreads();
  if (s_in.done) return;  // (the sweep of this launch was a no-op on every rank: nothing was posted)
  const unsigned long long id = __hip_atomic_load(mb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  const bool ok = mbox_gather(mb, id, tot, kReduceBuf, threadIdx.x);
  if (threadIdx.x == 0) {
    mb.ctr[0] = id;
    if (!ok) mb.ctr[1] = 1ull;
  }
  __syncthreads();
  if (!ok) {  // a peer never posted: stop the minimiser, the host reports TLOAM_E_RCCL
    if (threadIdx.x == 0) { st->done = 1; st->comm_error = 1; }
    return;
  }
  const bool first = s_in.phase == PH_ITER0;
  gn_consume(st, tot, threadIdx.x, &s_in, scr);
  if (threadIdx.x == 0) iter_span_note(iter_span, first);
}
void launch_gn_step_mbox(GnState* st, const MboxView& mb, hipStream_t s, unsigned long long* iter_span) {
  hipLaunchKernelGGL(k_gn_step_mbox, dim3(1), dim3(64), 0, s, st, mb, iter_span);
}
// timing aid (tloam_time_sharded_sweep): the gather half of the step without the minimiser
__global__ __launch_bounds__(64) void k_mbox_gather_only(double* out48, MboxView mb) {
  __shared__ double tot[kMboxSlot];
  const unsigned long long id = __hip_atomic_load(mb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  const bool ok = mbox_gather(mb, id, tot, kReduceBuf, threadIdx.x);
  if (threadIdx.x < kReduceBuf) out48[threadIdx.x] = tot[threadIdx.x];
  if (threadIdx.x == 0) {
    mb.ctr[0] = id;
    if (!ok) mb.ctr[1] = 1ull;
  }
}
void launch_mbox_gather_only(double* out48, const MboxView& mb, hipStream_t s) {
  hipLaunchKernelGGL(k_mbox_gather_only, dim3(1), dim3(64), 0, s, out48, mb);
}
// the small side exchanges of a sharded frame (cap prefix counts, cost sums) through the same mailbox: one launch
__global__ __launch_bounds__(64) void k_mbox_allreduce(double* buf, int count, MboxView mb) {
  __shared__ double vals[kMboxSlot];
  const int t = threadIdx.x;
  if (t < count) vals[t] = buf[t];
  __syncthreads();
  const unsigned long long id = __hip_atomic_load(mb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  mbox_post(mb, id, vals, count, t);
  const bool ok = mbox_gather(mb, id, vals, count, t);
  __syncthreads();
  if (t < count) buf[t] = vals[t];
  if (t == 0) {
    mb.ctr[0] = id;
    if (!ok) mb.ctr[1] = 1ull;
  }
}
void launch_mbox_allreduce(double* buf, int count, const MboxView& mb, hipStream_t s) {
  hipLaunchKernelGGL(k_mbox_allreduce, dim3(1), dim3(64), 0, s, buf, count, mb);
}

// single-GPU fast path: reduce the block rows and advance the minimiser in ONE launch
__global__ __launch_bounds__(kRedThreads) void k_reduce_and_step(const double* __restrict__ partials, int rows,
                                                                 GnState* __restrict__ st, unsigned long long* iter_span) {
  __shared__ double lds[8 * 33];
  __shared__ double tot[kReduceBuf];
  __shared__ GnState s_in;
#ifdef TLOAM_STEP_PROFILE
  if (threadIdx.x == 0) st->dbg[0] = (double)__builtin_readcyclecounter();
#endif
  // the minimiser state comes in as ONE coalesced load into LDS, in flight together with the partial rows (the
  // step's ~100 scattered field loads were a second memory round trip in front of the serial fp64 chain); the
  // `done` flag of a finished Solve is read from that copy -- no round trip of its own in front of the loads
  {
    constexpr int kWords = (int)(sizeof(GnState) / 8);
    static_assert(kWords <= kRedThreads, "one word per thread");
    const unsigned long long w = threadIdx.x < kWords ? reinterpret_cast<const unsigned long long*>(st)[threadIdx.x] : 0ull;
    if (threadIdx.x < kWords) reinterpret_cast<unsigned long long*>(&s_in)[threadIdx.x] = w;
  }
#ifdef TLOAM_STEP_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) st->dbg[7] = (double)__builtin_readcyclecounter();   // the state has arrived
#endif
  fold_rows<false, 4>(partials, rows, lds, tot);  // (its barriers also publish s_in)
#ifdef TLOAM_STEP_PROFILE
  if (threadIdx.x == 0) st->dbg[8] = (double)__builtin_readcyclecounter();   // the rows are folded
#endif
  if (s_in.done) return;  // after a tolerance exit the remaining launches are no-ops
  const bool first = s_in.phase == PH_ITER0;
  if (threadIdx.x < 64) gn_consume(st, tot, threadIdx.x, &s_in, lds /* free again: the fold is over */);
  if (threadIdx.x == 0) iter_span_note(iter_span, first);
#ifdef TLOAM_STEP_PROFILE
  if (threadIdx.x == 0) st->dbg[6] = (double)__builtin_readcyclecounter();
#endif
}
// ---- one GN iteration of a LARGE set in ONE launch (round 4) -------------------------------------------------------
// The streaming sweep whose LAST block (ticket) folds the rows and -- on its first wave -- advances the minimiser:
// registration.cpp:1036-1047 is one Solve, and a GN iteration of it is one launch here whatever the size of the set.
// Against k3_accumulate + k_reduce_and_step this removes, per GN iteration, a kernel boundary, the dispatch of a one-block
// kernel and its row / state loads from the dependent chain (the state is requested into LDS by every block in its first
// instructions: which block will be last is not known).  Sharded contexts with a mailbox: the last block posts the folded
// row to every rank, then GATHERS the ranks' rows itself and runs the step -- sweep + exchange + step is one launch there
// too (RCCL / callback contexts keep k3_accumulate<*, true> + collective + k_gn_step: the collective is a host-enqueued
// operation between two launches).
// Registers: the grid is two blocks per CU (k3_grid_for), i.e. two waves per SIMD, which leaves each wave 256 of the
// SIMD's 512 registers -- the sweep needs ~118, the step ~285 when unconstrained: the step's overflow goes to scratch,
// touched by one wave of one block per launch.
// span (development / bench aid, may be null): [0] block 0 stores the wall clock (100 MHz) in its first instructions, the
// last block adds (its clock at the ticket - [0]) to [1] and one to [2]: the STREAMING span of the launch, first wave in to
// last row out, without the serial tail -- what the roofline fraction of the sweep inside a frame is computed from, now that
// the dispatch duration includes fold + step.
struct K3Step {
  int* ticket;                 // zero between launches
  unsigned long long* span;    // [4] or null
  unsigned long long* iter_span;   // iter_span_note, or null
  MboxView mb;                 // mb.nranks == 0: one rank
};
template <bool SINGLE, int W = 4>
__global__ __launch_bounds__(64 * W, 2) void k3_sweep_step(const double* __restrict__ seg0, int stride0, int cap0, int unused,
                                                        GnState* __restrict__ st, const int* __restrict__ seg_n,
                                                        double* __restrict__ partials, CorrView cv, K3Step fs) {
  __shared__ double red[W][32];
  __shared__ double s_grp[8 * 33];
  __shared__ double tot[kMboxSlot];
  __shared__ GnState s_in;
  const bool w2 = (unused & 2) != 0;   // (as k3_accumulate's flag word)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * W + wave;
  ChunkData pre;
  SingleWork wk{-1, 0};
  const bool spec = !SINGLE && (gw + 1) * kChunk <= cap0;
  if (SINGLE) {
    wk = single_work_of(cv, cap0, gw, lane);
    single_fetch(cv, seg0, stride0, wk, pre, w2);
  } else if (spec) {
    fetch_spec<TLOAM_K3_NT>(seg0, stride0, gw * kChunk + lane * 2, pre, w2);
  }
  if (fs.span && blockIdx.x == 0 && threadIdx.x == 0)   // (device-scope: read by the last block, on another XCD)
    __hip_atomic_store(fs.span, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    constexpr int kWords = (int)(sizeof(GnState) / 8);
    static_assert(kWords <= 256, "one word per thread");   // (the first 256 threads of a wide block)
    if (threadIdx.x < kWords)
      reinterpret_cast<unsigned long long*>(&s_in)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(st)[threadIdx.x];
  }
  if (st->done) return;            // after a tolerance exit the remaining launches are no-ops (uniform over the grid and the ranks)
  const Rt T = st->Rt_eval;
  Acc a;
  if (SINGLE) sweep_single(cv, seg_n, T, wk, pre, a);
  else sweep_all(cv, seg_n, T, gw, gridDim.x * W, lane, a, pre, spec);
  const double wtot = wave_reduce_acc(a, lane);
  if ((lane & 1) == 0) red[wave][lane >> 1] = wtot;
  __syncthreads();                 // (also publishes s_in)
  if (!k3_take_ticket<W>(partials, red, fs.ticket)) return;
  // ---- the last block: every row has reached the coherence point
  if (fs.span && threadIdx.x == 0) {
    const unsigned long long t1 = wall_clock64();
    const unsigned long long t0 = __hip_atomic_load(fs.span, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fs.span[1] += t1 - t0;
    fs.span[2] += 1ull;
  }
  fold_rows<true>(partials, (int)gridDim.x, s_grp, tot);
  if (threadIdx.x == 0) __hip_atomic_store(fs.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  if (threadIdx.x >= 64) return;
  if (fs.mb.nranks > 0) {
    // exchange: this rank's totals to every rank, every rank's totals from the local mailbox, added in rank order
    const unsigned long long id = __hip_atomic_load(fs.mb.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    mbox_post(fs.mb, id, tot, kReduceBuf, (int)threadIdx.x);
    const bool ok = mbox_gather(fs.mb, id, tot, kReduceBuf, (int)threadIdx.x);
    if (threadIdx.x == 0) {
      fs.mb.ctr[0] = id;
      if (!ok) fs.mb.ctr[1] = 1ull;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its LDS stores are in order
    if (!ok) {  // a peer never posted: stop the minimiser, the host reports TLOAM_E_RCCL
      if (threadIdx.x == 0) { st->done = 1; st->comm_error = 1; }
      return;
    }
  }
  const bool first = s_in.phase == PH_ITER0;
  gn_consume(st, tot, (int)threadIdx.x, &s_in, s_grp /* free again: the fold is over */);
  if (threadIdx.x == 0) iter_span_note(fs.iter_span, first);
}
void launch_k3_step(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, int* ticket, unsigned long long* span,
                    const MboxView* mb_or_null, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned long long* iter_span) {
  auto kern = single ? k3_sweep_step<true> : wide ? k3_sweep_step<false, 8> : k3_sweep_step<false>;
  K3Step fs;
  memset(&fs, 0, sizeof(fs));
  fs.ticket = ticket;
  fs.span = span;
  fs.iter_span = iter_span;
  if (mb_or_null) fs.mb = *mb_or_null;
  if (ev_start && ev_stop) {
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(wide ? 512 : 256), 0, s, ev_start, ev_stop, 0, (const double*)cv.k[0].px, cv.k[0].stride,
                          cv.k[0].cap, w2_flag(cv), st, cv.seg_n, partials, cv, fs);
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wide ? 512 : 256), 0, s, (const double*)cv.k[0].px, cv.k[0].stride, cv.k[0].cap, w2_flag(cv), st, cv.seg_n,
                       partials, cv, fs);
  }
}

// ---- one GN iteration of a KITTI-size set in ONE launch ----------------------------------------------------------
// The sweep (one wave per chunk, as k3_accumulate<true, *>) and the minimiser step: every block hands its row over
// with device-scope stores and takes a ticket (k3_take_ticket: no cache-wide fence), the LAST block folds the rows
// and runs the step on its first wave.  Every block requests the minimiser state into LDS in its first instructions
// (which block will be last is not known), so the step starts without a memory round trip of its own.  Against
// sweep + step as two launches this removes a kernel boundary, the step's dispatch and its row / state loads from the
// dependent chain of every GN iteration (measured: the hand-over adds 1.8 us to the 3.65 us sweep of a KITTI-cap set).
__global__ __launch_bounds__(256, 1) void k_sweep_step_small(const double* __restrict__ seg0, int stride0, int cap0, int tagged,
                                                             GnState* __restrict__ st, const int* __restrict__ seg_n,
                                                             double* __restrict__ partials, int* __restrict__ ticket,
                                                             CorrView cv, unsigned long long* iter_span) {
  __shared__ double red[4][32];
  __shared__ double s_grp[8 * 33];
  __shared__ double s_rows[kTaggedRows * 28];
  __shared__ double tot[kReduceBuf];
  __shared__ GnState s_in;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
#ifdef TLOAM_STEP_PROFILE
  const unsigned long long c_entry = __builtin_readcyclecounter();
#endif
  ChunkData pre;
  const SingleWork wk = single_work_of(cv, cap0, gw, lane);
  const bool w2 = (tagged & 2) != 0;   // (as k3_accumulate's flag word)
  tagged &= 1;
  single_fetch(cv, seg0, stride0, wk, pre, w2);
  {
    constexpr int kWords = (int)(sizeof(GnState) / 8);
    static_assert(kWords <= 256, "one word per thread");
    if (threadIdx.x < kWords)
      reinterpret_cast<unsigned long long*>(&s_in)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(st)[threadIdx.x];
  }
  // launch counter of the tagged hand-over (words 2..3 of the ticket buffer): read by every block before any block can
  // have advanced it -- the consumer does so only after every block's row, which carries the tag, has arrived
  unsigned long long* const epoch = reinterpret_cast<unsigned long long*>(ticket + 2);
  const unsigned long long epoch0 = *epoch;
  if (st->done) return;            // after a tolerance exit the remaining launches are no-ops (uniform over the grid)
  const Rt T = st->Rt_eval;
#ifdef TLOAM_STEP_PROFILE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long c_pose = __builtin_readcyclecounter();
#endif
  Acc a;
  sweep_single(cv, seg_n, T, wk, pre, a);
#ifdef TLOAM_STEP_PROFILE
  const unsigned long long c_eval = __builtin_readcyclecounter();
#endif
  const double wtot = wave_reduce_acc(a, lane);
  if ((lane & 1) == 0) red[wave][lane >> 1] = wtot;
  __syncthreads();
#ifdef TLOAM_STEP_PROFILE
  const unsigned long long c_sweep = __builtin_readcyclecounter();
#endif
  if (tagged) {
    // a dozen blocks: the rows are the flag, block 0 is the consumer (see k3_post_row_tagged)
    k3_post_row_tagged(partials, red, epoch0 + 1ull);
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;   // the consumer is ONE wave: poll, fold, step
#ifdef TLOAM_STEP_PROFILE
    if (threadIdx.x == 0) {
      st->dbg[7] = (double)c_sweep; st->dbg[0] = (double)__builtin_readcyclecounter();
      st->dbg[8] = (double)c_entry; st->dbg[9] = (double)c_pose; st->dbg[10] = (double)c_eval;
    }
#endif
    const bool ok = poll_fold_tagged(partials, (int)gridDim.x, epoch0 + 1ull, s_rows, tot, (int)threadIdx.x);
    if (threadIdx.x == 0) *epoch = epoch0 + 1ull;   // (plain store: read by the next launch)
    if (!ok) {  // a block of the grid never posted: stop the Solve; the finish kernel reports OS_COMM_ERROR
      if (threadIdx.x == 0) { st->done = 1; st->comm_error = 1; }
      return;
    }
  } else {
    if (!k3_take_ticket(partials, red, ticket)) return;
#ifdef TLOAM_STEP_PROFILE
    if (threadIdx.x == 0) { st->dbg[7] = (double)c_sweep; st->dbg[0] = (double)__builtin_readcyclecounter(); }
#endif
    fold_rows<true>(partials, (int)gridDim.x, s_grp, tot);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  }
  const bool first = s_in.phase == PH_ITER0;
  if (threadIdx.x < 64) gn_consume(st, tot, threadIdx.x, &s_in, s_grp /* free again: the fold is over */);
  if (threadIdx.x == 0) iter_span_note(iter_span, first);
#ifdef TLOAM_STEP_PROFILE
  if (threadIdx.x == 0) st->dbg[6] = (double)__builtin_readcyclecounter();
#endif
}
void launch_sweep_step_small(const CorrView& cv, GnState* st, double* partials, int* ticket, int grid, hipStream_t s,
                             unsigned long long* iter_span) {
  const int tagged = grid <= kTaggedRows ? 1 : 0;
  hipLaunchKernelGGL(k_sweep_step_small, dim3(grid), dim3(256), 0, s, (const double*)cv.k[0].px, cv.k[0].stride, cv.k[0].cap, tagged | w2_flag(cv), st,
                     cv.seg_n, partials, ticket, cv, iter_span);
}
#include "tl_prep.hpp"   // SolvePrep: the Solve's own caps / compaction / refresh (kind_set_of, self_compact, self_refresh)

// ---- the sums of an outer iteration's finish, carried by the Solve itself -------------------------------------------------
// updateWeight (registration.cpp:858-876) and the cost sums (:1091-1094) look at the side-channel costs of the LAST
// evaluation of the Solve -- which the wave that evaluated them still holds.  Every sweep, every wave adds up the costs of
// its chunk (lane order, one shuffle tree) and counts the factors whose new weight would leave [0, 1] (the reference's
// assert, :871) and posts the two numbers as a 64-byte segment of its own -- AFTER its block's row, off the path the next
// pose waits on; the consumer, once the minimiser says "done", adds the waves' segments per kind in wave order.  By then
// they have long arrived: no pass over the costs, no hand-over of its own.
// Only a factor whose cost is within 1e-9 (relative) of an end of the band (th2, th1) can round out of [0, 1]: with
// c = th1 (1 - d) the weight is ~ mu d / 2 against an error of a few 1e-16 mu, at the other end 1 - w ~ (mu + 1) d / 2
// against a few 1e-16 (mu + 1); for those few (and for every factor in the band once the band is narrower than that) the
// reference's expression is evaluated as it stands.
__device__ __forceinline__ bool weight_out_of_range(const WeightParams& wp, int kind, double c) {
  if (!wp.active[kind]) return false;
  if (c == 0) return false;                      // :862
  if (c >= wp.th1 || c <= wp.th2) return false;  // :865 / :867
  if (c < wp.th1 * (1.0 - 1e-9) && c > wp.th2 * (1.0 + 1e-9)) return false;
  const double w = sqrt(wp.noise_bound_sq * wp.mu * (wp.mu + 1) / c) - wp.mu;  // :870
  return !(w >= 0.0 && w <= 1.0);
}
constexpr int kExtRowBase = 512;   // the waves' extra segments: words kExtRowBase + 8 wave .. of the row buffer (wave = 4 block + wave in block)
// lanes 0..7 of the wave: (cost sum, count, kind, 0, 0, 0, 0, tag ^ xor) -- cs / bad: the wave totals (lane 0's are used)
__device__ __forceinline__ void post_ext_segment(double* __restrict__ partials, int gw, double cs, double bad, int kind, unsigned long long tag,
                                                 int lane, int base_word = kExtRowBase) {
  if (lane >= 8) return;
  unsigned long long wd = 0ull;
  const double cs0 = rdlane(cs, 0), bad0 = rdlane(bad, 0);
  if (lane == 0) wd = (unsigned long long)__double_as_longlong(cs0);
  else if (lane == 1) wd = (unsigned long long)__double_as_longlong(bad0);
  else if (lane == 2) wd = (unsigned long long)(long long)kind;
  // (position-dependent, non-linear entry of the payload into the check word, as the segments the host reads: this one mixes a
  //  double with small integers -- a count, a kind -- the class of payload where two stale words can XOR like two new ones)
  const unsigned long long x = xor8(lane < 7 ? seg_word(wd, lane) : 0ull);
  if (lane == 7) wd = check_mix(tag) ^ x;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(partials) + base_word + (size_t)gw * 8 + lane, wd, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// the consumer wave: waits for the extra segment of every other wave of the grid (stored right behind the rows it folded a
// step ago) and adds them per kind in wave order -> fin[0..4] in LDS.  false: timed out.  nwaves <= 64.
__device__ __forceinline__ bool poll_fold_ext(const double* __restrict__ partials, int nwaves, unsigned long long tag, double* s_ext /*[64*3]*/,
                                              double* fin /*[8]*/, int lane, int base_word = kExtRowBase) {
  const bool have = lane >= 1 && lane < nwaves;   // (wave 0 of the grid holds no chunk)
  const unsigned long long mtag = check_mix(tag);
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(partials) + base_word + (size_t)(have ? lane : 1) * 8;
  unsigned long long w[8];
  const unsigned long long t0 = wall_clock64();
  bool ok_all;
  for (unsigned spins = 1;; ++spins) {
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long x = w[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x ^= seg_word(w[i], i);
    ok_all = __all((!have || x == mtag) ? 1 : 0) != 0;
    if (ok_all) break;
    if ((spins & 63u) == 0 && wall_clock64() - t0 > 100000000ull) break;
    __builtin_amdgcn_s_sleep(1);
  }
  // per kind, one fixed shuffle tree over the waves (lane = wave): any fixed order will do -- every path that ends an
  // outer iteration of a one-launch Solve takes its sums from here
  const double cs = have ? __longlong_as_double((long long)w[0]) : 0.0;
  const double bad = have ? __longlong_as_double((long long)w[1]) : 0.0;
  const int kind = have ? (int)(long long)w[2] : -1;
  double t[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = kind == k ? cs : 0.0;
  t[4] = bad;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] += __shfl_down(t[k], off, 64);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) fin[k] = t[k];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  (void)s_ext;
  return ok_all;
}

// ---- the consumer wave's end of an outer iteration (SolveFinish): sums -> state, loop decisions, re-arm, result slot -------
// fin[0..3]: the kinds' cost sums of the Solve's last evaluation, fin[4]: weights out of range (poll_fold_ext); nseg: the set.
// Returns 3: the loop goes on with an unchanged pose -- this launch runs the next outer iteration too | 4: leave (the loop
// has ended, or the pose moved and the correspondence search has to run first).  One wave.
// lead == false (k_solve_all: every block runs the consumer on its own image of the state): nothing leaves the block.
__device__ __forceinline__ int finish_by_consumer(GnState* st, GnState* sm /* LDS image, current */, const SolveFinish& F, int oi,
                                                  const double* fin, const int nseg[kKinds], double* sh /* LDS [16] */, int lane,
                                                  bool lead = true) {
  constexpr int kWords = (int)(offsetof(GnState, dbg) / 8);
  const OuterCtl ctl{F.cost_threshold, 1, oi == F.n_iter - 1 ? 1 : 0};
  int next = 4;
  if (!sm->done) {   // the Solve ran out of this launch's evaluation budget (finish_gate: 2): the host tops it up
    if (lane == 0) {
      sm->incomplete = OS_INCOMPLETE;
      sm->stop = 2;
      sm->run_build = sm->run_refresh = 0;
    }
  } else {
    if (lane < 16) {
      double v = 0.0;
      if (lane < 4) v = fin[lane];
      else if (lane < 8) v = (double)(lane == 4 ? nseg[0] : (lane == 5 ? nseg[1] : (lane == 6 ? nseg[2] : nseg[3])));   // (no indexed array: registers)
      else if (lane == 8) v = fin[4];
      sh[lane] = v;
      if (lead) F.sums16[lane] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    publish_and_rearm(sh, sm, lane, ctl);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    next = (sm->stop == 0 && sm->run_build == 0 && sm->run_refresh != 0 && oi + 1 < F.n_iter) ? 3 : 4;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // the image into the device state (every word up to the development stamps), the result slot, and -- once the loop has
  // ended -- the slots of the iterations that will not run (what their gated-off finish kernels would have written)
  if (!lead) return next;
  for (int w = lane; w < kWords; w += 64) reinterpret_cast<unsigned long long*>(st)[w] = reinterpret_cast<const unsigned long long*>(sm)[w];
  // (the slot says whether the host has something to do: the loop goes on, but not inside this launch)
  mirror_wave(sm, F.hm[oi], lane, (sm->stop == 0 && next == 4) ? (int)(sm->incomplete | OS_NEEDS_HOST) : -1);
  if (sm->stop != 0)
    for (int j = oi + 1; j < F.n_iter; ++j) mirror_wave(sm, F.hm[j], lane, (int)OS_SKIPPED);
  return next;
}

// ---- small Solves in ONE launch, every block its own consumer -----------------------------------------------------------------
// A KITTI-size Solve is a chain sweep -> rows -> fold -> 6x6 step -> next pose, one to five times; as one launch per link
// it pays a launch boundary (~1.5 us of dispatch plus the ramp of a dozen blocks, plus every wave re-reading its chunk) per
// GN iteration, and the host must guess how many links to enqueue (planned_sweeps_for: learned budgets; a frame that breaks the
// guess costs a round trip and a second pass over the frame's launch list).  For a grid of <= kTaggedRows blocks -- all
// resident at once on a 256-CU part -- the chain runs inside one launch instead: every wave keeps its chunk of
// correspondences in registers; per GN iteration every block posts its row (k3_post_row_tagged), wave 0 of EVERY block polls
// all the rows, folds them in the same order and runs the same minimiser step (gn_consume) on its own LDS image of the state
// -- bit-identical in all blocks, the inputs are -- and hands the candidate pose to its block's other waves through LDS and
// a barrier.  One cross-CU exchange per GN iteration (round 3's form, one consumer wave for the whole grid publishing the pose
// to the others, paid two: 8.2-8.5 against 6.8-7.7 us per iteration; removed in round 5); the step is
// executed sixteen times side by side, which costs nothing (those CUs would be waiting for it anyway).  Block 0 ("lead")
// alone writes the device state, the compact set's sizes and the result slots.  Tags carry the launch counter and the
// hand-over number, so nothing has to be reset between launches.  The same sweep arithmetic per wave, the same fold order,
// the same step as the one-launch-per-iteration kernels.
// max_sweeps: evaluations a Solve of this launch may run (the stepwise API's budgets and the development knobs keep their meaning).
// Rows are double-buffered by the parity of the hand-over number h: block A may post row h + 1 while block B is still reading
// the rows of h; it cannot post h + 2 before B has posted h + 1, which B does only after it has read every row of h.  The waves'
// finish-sum segments likewise.  Layout of `partials` (64-bit words): rows [2][kTaggedRows][32] at 0, segments [2][64][8] at 1024.
// Wave 0 of block 0 holds no chunk, wave 0 of block b >= 1 holds one like the other waves (single_work_of).
constexpr int kAllRowsParity = kTaggedRows * kAccStride;   // 512 words
constexpr int kAllExtBase = 2 * kAllRowsParity;            // 1024
constexpr int kAllExtParity = 64 * 8;                      // 512 words
// (the stepper reads its own LDS message back: same wave, stores waited for)
__device__ __forceinline__ bool next_is_failure(const double* s_msg) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return (int)s_msg[12] == 5;
}
__global__ __launch_bounds__(256, 1) void k_solve_all(const double* __restrict__ seg0, int stride0, int cap0, int max_sweeps,
                                                      GnState* __restrict__ st, const int* __restrict__ seg_n,
                                                      double* __restrict__ partials, int* __restrict__ ticket, CorrView cv, SolvePrep prep,
                                                      int* __restrict__ seg_n_out, SolveFinish F) {
  __shared__ double red[4][32];
  __shared__ double s_scr[32];
  __shared__ double s_rows[kTaggedRows * 28];
  __shared__ double s_ext[64 * 3];
  __shared__ double s_fin[8];
  __shared__ double s_sh[16];
  __shared__ double tot[kReduceBuf];
  __shared__ double s_msg[16];    // [0..8] R, [9..11] t of the pose to evaluate next, [12] the verdict (as an integer value)
  __shared__ GnState s_in;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  const bool stepper = wave == 0;            // the block's consumer: polls the rows, folds them, takes the minimiser's step
  const bool lead = blockIdx.x == 0;         // the block whose image of the state is the one written out
  const bool self_prep = prep.sv.flagb != nullptr;
  // test hook (TLOAM_DEBUG_FAIL_HANDOVER, bit 16 of the argument): the steppers wait for rows that nobody posts, i.e. the launch
  // behaves as if one of its blocks had never been scheduled -- the bounded wait, OS_COMM_ERROR and the host's fallback to one
  // launch per GN iteration are exercised by tests/test_gpu_parity.py
  const unsigned long long sabotage = ((max_sweeps >> 16) & 1) ? (1ull << 40) : 0ull;
  max_sweeps &= 0xffff;
  unsigned long long* const epoch = reinterpret_cast<unsigned long long*>(ticket + 2);
#ifdef TLOAM_STEP_PROFILE
  // development aid (scripts/solve_profile2.py): wall-clock (100 MHz) stamps of the lead's stepper and of one producer wave
  // (block gridDim/2) per GN iteration, in the spare part of the row buffer
  unsigned long long* const prof = reinterpret_cast<unsigned long long*>(partials) + 2048;
  const bool prof_c = lead && stepper && lane == 0;
  const bool prof_p = (int)blockIdx.x == (int)gridDim.x / 2 && wave == 1 && lane == 0;
#define TL_PROF(cond, slot) if (cond) prof[(slot)] = wall_clock64();
#else
#define TL_PROF(cond, slot)
#endif
  // ---- this wave's chunk, requested in its first instructions (with SolvePrep: speculatively -- right if the set is kept or refreshed; a set that is rebuilt is fetched from the slots)
  ChunkData pre;
  const SingleWork wk = single_work_of(cv, cap0, gw, lane);
  single_fetch(cv, seg0, stride0, wk, pre);
  FlagBytes fbytes;
  if (self_prep && wk.kind >= 0) load_flag_bytes(prep.sv.flagb, wk.kind, lane, fbytes);
  if (stepper) {
    constexpr int kWords = (int)(sizeof(GnState) / 8);
    for (int w = lane; w < kWords; w += 64)
      reinterpret_cast<unsigned long long*>(&s_in)[w] = reinterpret_cast<const unsigned long long*>(st)[w];
  }
  // (every block reads the launch counter, the loop control and the gates before any block can change them: the lead writes
  //  them only after it has seen a row of every block, and a block posts its first row after these reads)
  const unsigned long long tag0 = (*epoch + 1ull) << 8;
  int oi = F.first_iter;
  if (F.enabled && (st->stop != 0 || st->next_outer != oi)) return;
  const bool build = self_prep && (!prep.run_build || *prep.run_build != 0);
  const bool refresh = self_prep && !build && prep.run_refresh && *prep.run_refresh != 0;
  const bool done_at_entry = st->done != 0;
  Rt T = st->Rt_eval;
  // ---- the factor set of this outer iteration (SolvePrep)
  int n_mine = 0;
  int slot[2] = {0, 0};
  int nseg[kKinds] = {0, 0, 0, 0};   // (the lead's stepper: the sizes of the whole set, for the state and the host)
  if (lead && stepper) {
    if (build) {
      FlagBytes fb[kKinds];
#pragma unroll
      for (int k = 0; k < kKinds; ++k) load_flag_bytes(prep.sv.flagb, k, lane, fb[k]);
#pragma unroll
      for (int k = 0; k < kKinds; ++k) nseg[k] = kind_set_of(prep, cv, k, lane, fb[k]).total;
      if (lane < kKinds) seg_n_out[lane] = lane == 0 ? nseg[0] : (lane == 1 ? nseg[1] : (lane == 2 ? nseg[2] : nseg[3]));
    } else {
#pragma unroll
      for (int k = 0; k < kKinds; ++k) nseg[k] = seg_n[k];
    }
  }
  if (wk.kind >= 0) {
    if (build) {
      n_mine = self_compact(prep, cv, wk, lane, fbytes, pre, slot);
    } else {
      n_mine = seg_n[wk.kind];
      if (self_prep) {
        slots_of_chunk(prep, cv, wk, n_mine, slot);
        if (refresh) self_refresh(prep, cv, wk, n_mine, slot, pre);
      }
    }
  }
  if (stepper) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the image is in LDS
    if (build && lane < 6) {   // the set is built at this pose (k_prepare_small's x_build = x)
      s_in.x_build[lane] = s_in.x[lane];
      if (lead) st->x_build[lane] = s_in.x[lane];
    }
  }
  if (done_at_entry) return;          // a Solve that has already ended (uniform over the grid)
  TL_PROF(prof_c, 0)
  unsigned long long h = 0;           // hand-over number: rows of GN evaluation h carry tag0 | h, parity h & 1
  for (;;) {   // outer iterations run by this launch (exactly one unless F.enabled)
    int verdict = 2;
    double2 last_cost = double2{0.0, 0.0};
    for (int it = 0;; ++it) {
      Acc a;
      last_cost = double2{0.0, 0.0};
      sweep_single_n(cv, n_mine, T, wk, pre, a, &last_cost);
      TL_PROF(prof_p, 64 + it * 8 + 1)
      const double wtot = wave_reduce_acc(a, lane);
      if ((lane & 1) == 0) red[wave][lane >> 1] = wtot;
      __syncthreads();
      double* const rows = partials + (size_t)(h & 1ull) * kAllRowsParity;
      const int ext_base = kAllExtBase + (int)(h & 1ull) * kAllExtParity;
      k3_post_row_tagged(rows, red, tag0 | h);     // (the first 32 threads of the block)
      TL_PROF(prof_c, 8 + it * 8 + 0)
      if (F.have_wp) {   // this chunk's part of the finish sums, behind the row (see post_ext_segment)
        double cs = last_cost.x + last_cost.y;   // (a lane without a second / any factor holds 0 there)
        double bad = 0.0;
        if (wk.kind >= 0) {
          const bool two = single_chunk_of(wk.kind) == kChunk;
          if (wk.j < n_mine && weight_out_of_range(F.wp[oi], wk.kind, last_cost.x)) bad += 1.0;
          if (two && wk.j + 1 < n_mine && weight_out_of_range(F.wp[oi], wk.kind, last_cost.y)) bad += 1.0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          cs += __shfl_down(cs, off, 64);
          bad += __shfl_down(bad, off, 64);
        }
        post_ext_segment(partials, gw, cs, bad, wk.kind, tag0 | h, lane, ext_base);
      }
      TL_PROF(prof_p, 64 + it * 8 + 2)
      if (stepper) {
        const bool ok = poll_fold_tagged(rows, (int)gridDim.x, (tag0 | h) ^ sabotage, s_rows, tot, lane);
        TL_PROF(prof_c, 8 + it * 8 + 1)
        int vd = 5;
        if (ok) {
          gn_consume(st, tot, lane, &s_in, s_scr, /*WRITE_GLOBAL=*/false);
          vd = (s_in.done == 0 && it + 1 < max_sweeps) ? 1 : 2;
          if (lead && lane == 0) iter_span_note(F.iter_span, it == 0);
        }
        TL_PROF(prof_c, 8 + it * 8 + 2)
        if (lane < 9) s_msg[lane] = s_in.Rt_eval.r[lane];
        else if (lane < 12) s_msg[lane] = s_in.Rt_eval.t[lane - 9];
        else if (lane == 12) s_msg[12] = (double)vd;
      }
      __syncthreads();
      verdict = (int)s_msg[12];
      // (the lead's image of the state goes out to the device state now, beside the other waves' evaluation.  Holding it back until
      //  the lead's next row is out -- round 6 -- changed nothing: 5.42 against 5.42 us per GN iteration)
      if (lead && stepper && verdict != 5) gn_write_back(st, &s_in, lane);
#pragma unroll
      for (int v = 0; v < 9; ++v) T.r[v] = s_msg[v];
#pragma unroll
      for (int v = 0; v < 3; ++v) T.t[v] = s_msg[9 + v];
      TL_PROF(prof_c, 8 + it * 8 + 3)
      TL_PROF(prof_p, 64 + (it + 1) * 8 + 0)
      h += 1ull;
      if (verdict != 1) break;
    }
    if (verdict == 5) {   // a block of the grid never posted (every block finds out in the same poll): stop, the lead reports
      if (lead && stepper) {
        if (lane == 0) {
          st->done = 1; st->comm_error = 1;
          s_in.done = 1; s_in.comm_error = 1;
          if (F.enabled) {
            s_in.incomplete = OS_COMM_ERROR;
            st->incomplete = OS_COMM_ERROR;
            st->stop = 1; st->run_build = 0; st->run_refresh = 0;
          }
          *epoch = tag0 >> 8;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (F.enabled) {
          mirror_wave(&s_in, F.hm[oi], lane, (int)OS_COMM_ERROR);
          for (int j = oi + 1; j < F.n_iter; ++j) mirror_wave(&s_in, F.hm[j], lane, (int)OS_SKIPPED);
        }
      }
      return;
    }
    // ---- the Solve is over.  The finish sums of its last evaluation (hand-over h - 1): every stepper collects the waves'
    //      segments; meanwhile every wave works out the new GNC weights of the factors it holds (updateWeight, :858-876)
    double2 w_new = pre.w;
    if (F.enabled && wk.kind >= 0) {
      const bool two = single_chunk_of(wk.kind) == kChunk;
      if (wk.j < n_mine) w_new.x = refreshed_weight(F.wp[oi], wk.kind, last_cost.x, pre.w.x, prep.sv.w_src, slot[0]);
      if (two && wk.j + 1 < n_mine) w_new.y = refreshed_weight(F.wp[oi], wk.kind, last_cost.y, pre.w.y, prep.sv.w_src, slot[1]);
    }
    if (stepper) {
      int next = 4;
      bool ok = true;
      if (F.have_wp)
        ok = poll_fold_ext(partials, (int)gridDim.x * 4, tag0 | (h - 1ull), s_ext, s_fin, lane, kAllExtBase + (int)((h - 1ull) & 1ull) * kAllExtParity);
      if (!ok) {
        next = 5;
      } else if (!F.enabled) {
        // the launch ends with the Solve; a finish kernel that follows takes its sums from the state (fin_valid)
        if (lead) {
          if (F.have_wp && lane < 5) {
            const double v = s_fin[lane];
            if (lane < 4) st->fin_sum[lane] = v; else st->fin_bad = v;
          }
          if (lane == 0) { st->fin_valid = F.have_wp ? 1 : 0; *epoch = tag0 >> 8; }   // (plain stores: read by the next launch)
        }
      } else {
        next = finish_by_consumer(st, &s_in, F, oi, s_fin, nseg, s_sh, lane, lead);
      }
      if (lane < 9) s_msg[lane] = s_in.Rt_eval.r[lane];          // (the pose the re-armed minimiser starts from: exp(x))
      else if (lane < 12) s_msg[lane] = s_in.Rt_eval.t[lane - 9];
      else if (lane == 12) s_msg[12] = (double)next;
    }
    if (!F.enabled) {
      // (a Solve outside the device-driven loop, F.have_wp set: if the waves' finish-sum segments never arrived the lead still
      //  has to say so -- st->fin_valid keeps its old value and the launch counter must advance, or the next launch would take
      //  this one's stale rows and segments for fresh ones)
      if (lead && stepper && next_is_failure(s_msg) && lane == 0) {
        st->done = 1; st->comm_error = 1; s_in.done = 1; s_in.comm_error = 1;
        st->fin_valid = 0;
        *epoch = tag0 >> 8;
      }
      return;
    }
    __syncthreads();
    const int next = (int)s_msg[12];
#pragma unroll
    for (int v = 0; v < 9; ++v) T.r[v] = s_msg[v];
#pragma unroll
    for (int v = 0; v < 3; ++v) T.t[v] = s_msg[9 + v];
    if (next == 5) {   // the waves' segments never arrived: as above
      if (lead && stepper) {
        if (lane == 0) {
          st->done = 1; st->comm_error = 1; s_in.done = 1; s_in.comm_error = 1;
          s_in.incomplete = OS_COMM_ERROR; st->incomplete = OS_COMM_ERROR;
          st->stop = 1; st->run_build = 0; st->run_refresh = 0;
          *epoch = tag0 >> 8;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mirror_wave(&s_in, F.hm[oi], lane, (int)OS_COMM_ERROR);
        for (int j = oi + 1; j < F.n_iter; ++j) mirror_wave(&s_in, F.hm[j], lane, (int)OS_SKIPPED);
      }
      return;
    }
    if (next != 3) {
      if (lead && stepper && lane == 0) *epoch = tag0 >> 8;
      return;
    }
    // ---- the next outer iteration on the same correspondences: new captured weights, zeroed side-channel slots (k_refresh)
    if (wk.kind >= 0) {
      const CorrSeg& seg = cv.k[wk.kind];
      const bool two = single_chunk_of(wk.kind) == kChunk;
      pre.w = w_new;
      if (wk.j < n_mine) { seg.w[wk.j] = w_new.x; seg.cost[wk.j] = 0.0; }
      if (two && wk.j + 1 < n_mine) { seg.w[wk.j + 1] = w_new.y; seg.cost[wk.j + 1] = 0.0; }
    }
    oi += 1;
  }
}
void launch_solve_small(const CorrView& cv, GnState* st, double* partials, int* ticket, int grid, int max_sweeps,
                        const SolvePrep* prep_or_null, int* seg_n, const SolveFinish* finish_or_null, hipStream_t s) {
  SolvePrep P;
  if (prep_or_null) P = *prep_or_null;
  else memset(&P, 0, sizeof(P));
  SolveFinish F;
  if (finish_or_null) F = *finish_or_null;
  else memset(&F, 0, sizeof(F));
  hipLaunchKernelGGL(k_solve_all, dim3(grid), dim3(256), 0, s, (const double*)cv.k[0].px, cv.k[0].stride, cv.k[0].cap, max_sweeps, st,
                     cv.seg_n, partials, ticket, cv, P, seg_n, F);
}
// The launch spin-waits between its blocks, so all of them must be resident at once: one block per CU (the consumer's step
// wants ~400 registers: one wave per SIMD), i.e. grid <= the device's CU count -- 256 on an MI355X, fewer on a CU-masked or
// partitioned device (then the one-launch-per-iteration kernels run instead).
bool solve_small_fits(int grid, int device_cus) { return grid <= kTaggedRows && grid <= device_cus; }
void launch_reduce_and_step(const double* partials, int grid, GnState* st, hipStream_t s, unsigned long long* iter_span) {
  hipLaunchKernelGGL(k_reduce_and_step, dim3(1), dim3(kRedThreads), 0, s, partials, grid, st, iter_span);
}

// ---- test aid: the DEVICE SE(3) arithmetic of the minimiser step, exposed one operation at a time --------------------
// out (per item, 26 doubles): [0..6] exp(delta) as (qw qx qy qz tx ty tz), [7..12] log(exp(x)), [13..18] Plus(x, delta) =
// log(exp(delta) * exp(x)) exactly as gn_consume forms it (exp_fast2 / compose_fast2 / log_fast2), [19..25] exp(x) through the
// host/device-shared se3_exp (what k_solve_init uses).  One thread per item.
__global__ void k_debug_se3(const double* __restrict__ x, const double* __restrict__ delta, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double xv[6], dv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { xv[k] = x[6 * i + k]; dv[k] = delta[6 * i + k]; }
  const Pose E = exp_fast2(dv);
  const Pose X = exp_fast2(xv);
  double lx[6], pl[6];
  log_fast2(X, lx);
  log_fast2(compose_fast2(E, X), pl);
  const Pose X2 = se3_exp(xv);
  double* o = out + 26 * (size_t)i;
  o[0] = E.qw; o[1] = E.qx; o[2] = E.qy; o[3] = E.qz; o[4] = E.tx; o[5] = E.ty; o[6] = E.tz;
#pragma unroll
  for (int k = 0; k < 6; ++k) { o[7 + k] = lx[k]; o[13 + k] = pl[k]; }
  o[19] = X2.qw; o[20] = X2.qx; o[21] = X2.qy; o[22] = X2.qz; o[23] = X2.tx; o[24] = X2.ty; o[25] = X2.tz;
}
void launch_debug_se3(const double* x, const double* delta, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_se3, dim3((n + 63) / 64), dim3(64), 0, s, x, delta, n, out);
}

// ================================================================================================
//  K4: GNC-TLS weights (registration.cpp:858-876) + per-kind side-channel sums (:1091-1094)
// ================================================================================================
// Gate: the host enqueues only as many sweeps as the Solve is expected to need; if the minimiser has not
// terminated yet (st->done == 0) the weight update and the finish kernel do nothing, the finish kernel raises
// st->incomplete, and the host tops the Solve up and runs them again.
__global__ __launch_bounds__(256) void k_weights(WeightArgs A, double* __restrict__ partial, const GnState* __restrict__ st) {
  __shared__ double red[4][8];
  if (!st->done || st->stop) return;  // (stop: the device-driven loop has ended, the set's weights are final)
  double sum[kKinds] = {0, 0, 0, 0};
  double bad = 0.0;
  const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    const int n = A.cv.seg_n[k];
    const CorrSeg& seg = A.cv.k[k];
    // four strided elements per trip, their (cost, index) pairs requested together (the loop is a chain of memory
    // round trips: 14 per thread on a 1 M set); consumed in the same order as one by one, so the sums keep their bits
    constexpr int kU = 4;
    for (int i0 = tid; i0 < n; i0 += kU * stride) {
      double cu[kU];
      int iu[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * stride;
        cu[u] = (i < n) ? seg.cost[i] : 0.0;
        iu[u] = (i < n) ? seg.idx[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (i0 + u * stride >= n) break;
        const double c = cu[u];
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
        const int slot = A.sv.slot_off[k] + (iu[u] - A.sv.src_lo[k]);
        A.sv.w_src[slot] = w;
      }
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
                    const GnState* st, hipStream_t s) {
  WeightArgs A;
  A.cv = cv;
  A.sv = sv;
  A.wp = wp;
  hipLaunchKernelGGL(k_weights, dim3(blocks), dim3(256), 0, s, A, partial, st);
}

__global__ __launch_bounds__(64) void k_outer_finish(const double* __restrict__ partial, int blocks,
                                                     const int* __restrict__ seg_n, double* __restrict__ sums16,
                                                     GnState* st_or_null, GnState* gate, HostMirror hm, OuterCtl ctl) {
  __shared__ double sh[16];
  const int t = threadIdx.x;
  {
    const int g = finish_gate(gate, ctl, t);  // 2: the Solve is still running, nothing to finish yet (see k_weights)
    if (g != 0) {
      if (st_or_null) mirror_to_host(gate, hm, t, 64, g == 1 ? (int)OS_SKIPPED : -1);
      return;
    }
  }
  double v[5] = {0, 0, 0, 0, 0};
  for (int b = t; b < blocks; b += 64) {  // blocks <= 256: at most 4 rows per lane, fixed order
#pragma unroll
    for (int c = 0; c < 5; ++c) v[c] += partial[b * 8 + c];
  }
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
  if (t < 16) sh[t] = 0.0;
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[c] = v[c];
    sh[8] = v[4];
  }
  if (t >= 4 && t < 8) sh[t] = (double)seg_n[t - 4];
  __syncthreads();
  if (t < 16) sums16[t] = sh[t];
  if (st_or_null) {  // single rank: no exchange in between
    publish_and_rearm(sh, st_or_null, t, ctl);
    mirror_to_host(st_or_null, hm, t, 64);
  }
}
void launch_outer_finish(const double* partial, int blocks, const int* seg_n, GnState* st_or_null, GnState* gate,
                         double* sums16, HostMirror hm, OuterCtl ctl, hipStream_t s) {
  hipLaunchKernelGGL(k_outer_finish, dim3(1), dim3(64), 0, s, partial, blocks, seg_n, sums16, st_or_null, gate, hm, ctl);
}
__global__ void k_outer_publish(const double* __restrict__ sums16, GnState* st, HostMirror hm,
                                const unsigned long long* __restrict__ comm_err) {
  if (st->done) publish_and_rearm(sums16, st, threadIdx.x, OuterCtl{0.0, 0, 0});  // gated like k_outer_finish (which raised st->incomplete)
  if (threadIdx.x == 0 && (st->comm_error || (comm_err && *comm_err))) st->incomplete = 3;  // exchange timed out
  mirror_to_host(st, hm, threadIdx.x, 64);
}
void launch_outer_publish(const double* sums16, GnState* st, HostMirror hm, const unsigned long long* comm_err,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_outer_publish, dim3(1), dim3(64), 0, s, sums16, st, hm, comm_err);
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
