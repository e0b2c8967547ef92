// tl_knn.hpp -- device-side building blocks shared by the translation units that search the uniform grid
// (tl_nn.hip: K1/K2 of scanMatching; tl_feature.hip: PCA feature extraction): cell indexing, the sorted
// top-K list with exact (distance, index) order, the 27-cell walk, fitBestPlane and the 3x3 symmetric eigen
// solve.  Include only from TUs compiled with -ffp-contract=off (the gates that consume these results are
// discontinuous and must see the oracle's un-fused fp64 operation order).
#pragma once

#include "tl_common.hpp"

namespace tl {

__device__ __forceinline__ int cell_coord(double v, double org, double inv_cell, int dim) {
  // build and queries use this SAME function, so a point and a query always agree on cell boundaries;
  // the 1e-6 slack of the cell size over the radius covers the rounding of the multiply
  double f = floor((v - org) * inv_cell);
  f = fmax(f, -2.0);
  f = fmin(f, (double)dim + 1.0);
  return (int)f;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ================================================================================================
//  K1: exact k-NN over the 27-cell neighbourhood, top-K kept sorted in registers.
//  Candidates come from the cell-sorted packed records in HBM (PtsGlobal); the L1/L2 reuse comes from the
//  tile-sorted query order (an LDS-staged tile variant was measured and rejected).
// ================================================================================================
struct PtsGlobal {
  const double4* p;
  __device__ __forceinline__ double X(int j) const { return p[j].x; }
  __device__ __forceinline__ double Y(int j) const { return p[j].y; }
  __device__ __forceinline__ double Z(int j) const { return p[j].z; }
  __device__ __forceinline__ int I(int j) const { return (int)__double_as_longlong(p[j].w); }
};

template <int K>
struct TopK {
  double d[K];
  int j[K];  // position in the point source
};

// nanoflann L2_Simple_Adaptor: sum of squared differences accumulated in dimension order
__device__ __forceinline__ double sqdist(double qx, double qy, double qz, double x, double y, double z) {
  const double d0 = qx - x, d1 = qy - y, d2 = qz - z;
  double r = d0 * d0;
  r += d1 * d1;
  r += d2 * d2;
  return r;
}

// Branch-free sorted insertion: the candidate bubbles down the ascending list with one compare and
// four selects per level, so a wave pays the same ~5 instructions per level whether or not any of its
// lanes inserts (the branchy version made every wave run the long insert path on almost every
// candidate).  Strict total order (d, original index): exact ties go to the lower target index; the
// index is only fetched when two distances are bit-equal.
template <int K, class P>
__device__ __forceinline__ void topk_insert(TopK<K>& tk, const P& pts, double d, int j) {
#pragma unroll
  for (int m = 0; m < K; ++m) {
    bool before = d < tk.d[m];
    if (d == tk.d[m]) before = (tk.j[m] < 0) || (pts.I(j) < pts.I(tk.j[m]));  // rare: exact tie
    const double dm = tk.d[m];
    const int jm = tk.j[m];
    tk.d[m] = before ? d : dm;
    tk.j[m] = before ? j : jm;
    d = before ? dm : d;   // the displaced (larger) element carries on down the list
    j = before ? jm : j;
  }
}
// ------------------------------------------------------------------------------------------------
// Packed-key list (the hot loops of K1): the candidate's position j in the cell-sorted records replaces the low
// `bits` mantissa bits of its squared distance, so ONE double carries (distance, position) and a sorted
// insertion is a v_min_f64 / v_max_f64 pair per level -- 2 VALU ops where the (distance, index) list needs 8.
// The key orders by the distance TRUNCATED to 52 - bits mantissa bits, then by position: that is the exact
// (distance, original index) order unless two of the kept entries (K + 1 are kept: the boundary entry too)
// share a truncated distance -- detected afterwards on adjacent pairs, and such a query is redone with the
// exact insertion.  +inf = empty slot / masked candidate (min/max leave it at the tail).
// ------------------------------------------------------------------------------------------------
template <int N>
struct KeyList {
  double k[N];
};
template <int N>
__device__ __forceinline__ void keys_clear(KeyList<N>& L) {
#pragma unroll
  for (int m = 0; m < N; ++m) L.k[m] = __builtin_inf();
}
__device__ __forceinline__ int key_bits_for(int n) {  // positions 0 .. n-1
  return n > 1 ? 32 - __clz(n - 1) : 1;
}
// valid == false (masked lane), d == +inf or d == NaN (a NaN coordinate: never a neighbour, as under the exact
// comparisons) all give the empty key
__device__ __forceinline__ double key_pack(double d, int j, unsigned keep_mask, bool valid = true) {
  const unsigned lo = ((unsigned)__double2loint(d) & keep_mask) | (unsigned)j;
  const double key = __hiloint2double(__double2hiint(d), (int)lo);
  return (valid && d < __builtin_inf()) ? key : __builtin_inf();
}
template <int N>
__device__ __forceinline__ void key_insert(KeyList<N>& L, double key) {
#pragma unroll
  for (int m = 0; m < N; ++m) {
    const double lo = __builtin_fmin(L.k[m], key);
    key = __builtin_fmax(L.k[m], key);
    L.k[m] = lo;
  }
}
// true: two kept entries share a truncated distance (or the list cannot be trusted): redo exactly
template <int N>
__device__ __forceinline__ bool keys_ambiguous(const KeyList<N>& L, unsigned keep_mask) {
  bool amb = false;
#pragma unroll
  for (int m = 0; m + 1 < N; ++m) {
    const bool both = L.k[m + 1] < __builtin_inf();
    const bool same = __double2hiint(L.k[m]) == __double2hiint(L.k[m + 1]) &&
                      (((unsigned)__double2loint(L.k[m]) ^ (unsigned)__double2loint(L.k[m + 1])) & keep_mask) == 0u;
    amb |= both && same;
  }
  return amb;
}
// The coordinates of a list's K entries, kept from the records the list was unpacked from (knn_rows hands them to the per-query
// fits of K2, which used to fetch them again: 15 of the ~90 scattered loads of a five-neighbour query).  Entry m is defined where
// tk.j[m] >= 0.
template <int K>
struct NbrXyz {
  double x[K], y[K], z[K];
};
// the first K entries as a (distance, position) list: the distance is recomputed from the record, exactly as
// the scan computed it before truncation (xyz: the records' coordinates are kept as well)
template <int K, int N>
__device__ __forceinline__ void keys_unpack(const KeyList<N>& L, const PtsGlobal& pts, double qx, double qy, double qz,
                                            unsigned keep_mask, TopK<K>& tk, NbrXyz<K>* xyz = nullptr) {
#pragma unroll
  for (int m = 0; m < K; ++m) {
    const bool have = L.k[m] < __builtin_inf();
    const int j = have ? (int)((unsigned)__double2loint(L.k[m]) & ~keep_mask) : 0;
    const double4 p = pts.p[j];
    tk.d[m] = have ? sqdist(qx, qy, qz, p.x, p.y, p.z) : __builtin_inf();
    tk.j[m] = have ? j : -1;
    if (xyz) { xyz->x[m] = p.x; xyz->y[m] = p.y; xyz->z[m] = p.z; }
  }
}
// the same from the point source (lists that were NOT unpacked from keys: the exact redo of an ambiguous query)
template <int K>
__device__ __forceinline__ void nbr_fetch(const PtsGlobal& pts, const TopK<K>& tk, NbrXyz<K>* xyz) {
#pragma unroll
  for (int m = 0; m < K; ++m) {
    const double4 p = pts.p[tk.j[m] >= 0 ? tk.j[m] : 0];
    xyz->x[m] = p.x; xyz->y[m] = p.y; xyz->z[m] = p.z;
  }
}

template <int K>
__device__ __forceinline__ void topk_clear(TopK<K>& tk) {
#pragma unroll
  for (int m = 0; m < K; ++m) {
    tk.d[m] = __builtin_inf();
    tk.j[m] = -1;
  }
}
template <int K, class P>
__device__ __forceinline__ void scan_range(const P& pts, int s, int e, double qx, double qy, double qz, TopK<K>& tk) {
  for (int j = s; j < e; ++j) topk_insert<K, P>(tk, pts, sqdist(qx, qy, qz, pts.X(j), pts.Y(j), pts.Z(j)), j);
}
// the same over the packed HBM records, two candidates per trip so that two 32-byte loads are in flight
template <int K>
__device__ __forceinline__ void scan_range(const PtsGlobal& pts, int s, int e, double qx, double qy, double qz,
                                           TopK<K>& tk) {
  int j = s;
  for (; j + 1 < e; j += 2) {
    const double4 a = pts.p[j], b = pts.p[j + 1];
    topk_insert<K, PtsGlobal>(tk, pts, sqdist(qx, qy, qz, a.x, a.y, a.z), j);
    topk_insert<K, PtsGlobal>(tk, pts, sqdist(qx, qy, qz, b.x, b.y, b.z), j + 1);
  }
  if (j < e) {
    const double4 a = pts.p[j];
    topk_insert<K, PtsGlobal>(tk, pts, sqdist(qx, qy, qz, a.x, a.y, a.z), j);
  }
}

template <int K>
__device__ __forceinline__ void knn_grid(const GridView& g, double qx, double qy, double qz, TopK<K>& tk) {
  topk_clear<K>(tk);
  if (g.n <= 0) return;
  const PtsGlobal pts{g.gp};
  const int cx = cell_coord(qx, g.org[0], g.inv_cell, g.dim[0]);
  const int cy = cell_coord(qy, g.org[1], g.inv_cell, g.dim[1]);
  const int cz = cell_coord(qz, g.org[2], g.inv_cell, g.dim[2]);
  int x0 = cx - 1, x1 = cx + 1;
  if (x0 < 0) x0 = 0;
  if (x1 >= g.dim[0]) x1 = g.dim[0] - 1;
  if (x0 > x1) return;
  for (int z = cz - 1; z <= cz + 1; ++z) {
    if (z < 0 || z >= g.dim[2]) continue;
    for (int y = cy - 1; y <= cy + 1; ++y) {
      if (y < 0 || y >= g.dim[1]) continue;
      const size_t base = ((size_t)z * g.dim[1] + y) * g.dim[0];
      scan_range<K>(pts, g.cell_start[base + x0], g.cell_start[base + x1 + 1], qx, qy, qz, tk);
    }
  }
}

// the same search over the (2 reach + 1)^3 cells around the query: any radius, not only radius <= cell
// (getFitnessScore takes whatever fitness_thres the configuration holds).  The (distance, original index) order
// of the list is total, so the result does not depend on the walk order.
template <int K>
__device__ __forceinline__ void knn_grid_reach(const GridView& g, double qx, double qy, double qz, int reach, TopK<K>& tk) {
  topk_clear<K>(tk);
  if (g.n <= 0) return;
  const PtsGlobal pts{g.gp};
  const int cx = cell_coord(qx, g.org[0], g.inv_cell, g.dim[0]);
  const int cy = cell_coord(qy, g.org[1], g.inv_cell, g.dim[1]);
  const int cz = cell_coord(qz, g.org[2], g.inv_cell, g.dim[2]);
  const int x0 = max(cx - reach, 0), x1 = min(cx + reach, g.dim[0] - 1);
  if (x0 > x1) return;
  for (int z = max(cz - reach, 0); z <= min(cz + reach, g.dim[2] - 1); ++z)
    for (int y = max(cy - reach, 0); y <= min(cy + reach, g.dim[1] - 1); ++y) {
      const size_t base = ((size_t)z * g.dim[1] + y) * g.dim[0];
      scan_range<K>(pts, g.cell_start[base + x0], g.cell_start[base + x1 + 1], qx, qy, qz, tk);
    }
}

// knn_grid with the packed-key list in the walk (2 VALU ops per list level instead of ~9: K = 20 in the PCA
// feature extraction); a query whose kept entries are ambiguous under the truncated order is redone exactly.
template <int K>
__device__ __forceinline__ void knn_grid_fast(const GridView& g, double qx, double qy, double qz, TopK<K>& tk) {
  topk_clear<K>(tk);
  if (g.n <= 0) return;
  const PtsGlobal pts{g.gp};
  const unsigned keep_mask = ~((1u << key_bits_for(g.n)) - 1u);
  KeyList<K + 1> L;
  keys_clear<K + 1>(L);
  const int cx = cell_coord(qx, g.org[0], g.inv_cell, g.dim[0]);
  const int cy = cell_coord(qy, g.org[1], g.inv_cell, g.dim[1]);
  const int cz = cell_coord(qz, g.org[2], g.inv_cell, g.dim[2]);
  int x0 = cx - 1, x1 = cx + 1;
  if (x0 < 0) x0 = 0;
  if (x1 >= g.dim[0]) x1 = g.dim[0] - 1;
  if (x0 > x1) return;
  for (int z = cz - 1; z <= cz + 1; ++z) {
    if (z < 0 || z >= g.dim[2]) continue;
    for (int y = cy - 1; y <= cy + 1; ++y) {
      if (y < 0 || y >= g.dim[1]) continue;
      const size_t base = ((size_t)z * g.dim[1] + y) * g.dim[0];
      const int s = g.cell_start[base + x0], e = g.cell_start[base + x1 + 1];
      for (int j = s; j < e; j += 2) {  // two 32-byte records in flight
        const bool vb = j + 1 < e;
        const double4 a = pts.p[j], b = pts.p[vb ? j + 1 : j];
        key_insert<K + 1>(L, key_pack(sqdist(qx, qy, qz, a.x, a.y, a.z), j, keep_mask));
        key_insert<K + 1>(L, key_pack(sqdist(qx, qy, qz, b.x, b.y, b.z), j + 1, keep_mask, vb));
      }
    }
  }
  if (!keys_ambiguous<K + 1>(L, keep_mask)) keys_unpack<K, K + 1>(L, pts, qx, qy, qz, keep_mask, tk);
  else knn_grid<K>(g, qx, qy, qz, tk);
}

template <int K>
__device__ __forceinline__ int radius_cut(const TopK<K>& tk, double radius) {
  const double r2 = radius * radius;
  int cnt = 0;
#pragma unroll
  for (int m = 0; m < K; ++m) cnt += (tk.d[m] < r2) ? 1 : 0;  // sorted ascending => prefix
  return cnt;
}

// ================================================================================================
//  K2 helpers: fitBestPlane (registration.cpp:303-368) and the 3x3 symmetric eigen solve that stands
//  in for Eigen::SelfAdjointEigenSolver (registration.cpp:476-479) -- cyclic Jacobi, all indices
//  compile-time so the 3x3 work stays in VGPRs.
// ================================================================================================
__device__ __forceinline__ void fit_best_plane5(const double px[5], const double py[5], const double pz[5],
                                                double plane[4]) {
  const double total = 5.0;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
  for (int i = 0; i < 5; ++i) { c0 += px[i]; c1 += py[i]; c2 += pz[i]; }
  c0 /= total; c1 /= total; c2 /= total;
  double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double a = px[i] - c0, b = py[i] - c1, g = pz[i] - c2;
    xx += a * a; xy += a * b; xz += a * g; yy += b * b; yz += b * g; zz += g * g;
  }
  xx /= total; xy /= total; xz /= total; yy /= total; yz /= total; zz /= total;
  double w0 = 0, w1 = 0, w2 = 0;
  {
    const double det_x = yy * zz - yz * yz;
    const double a0 = det_x, a1 = xz * yz - xy * zz, a2 = xy * yz - xz * yy;
    double w = det_x * det_x;
    if (w0 * a0 + w1 * a1 + w2 * a2 < 0.0) w = -w;
    w0 += a0 * w; w1 += a1 * w; w2 += a2 * w;
  }
  {
    const double det_y = xx * zz - xz * xz;
    const double a0 = xz * yz - xy * zz, a1 = det_y, a2 = xy * xz - yz * xx;
    double w = det_y * det_y;
    if (w0 * a0 + w1 * a1 + w2 * a2 < 0.0) w = -w;
    w0 += a0 * w; w1 += a1 * w; w2 += a2 * w;
  }
  {
    const double det_z = xx * yy - xy * xy;
    const double a0 = xy * yz - xz * yy, a1 = xy * xz - yz * xx, a2 = det_z;
    double w = det_z * det_z;
    if (w0 * a0 + w1 * a1 + w2 * a2 < 0.0) w = -w;
    w0 += a0 * w; w1 += a1 * w; w2 += a2 * w;
  }
  const double norm = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  if (norm == 0.0) { plane[0] = plane[1] = plane[2] = plane[3] = 0.0; return; }
  w0 /= norm; w1 /= norm; w2 /= norm;
  plane[0] = w0; plane[1] = w1; plane[2] = w2;
  plane[3] = -(w0 * c0 + w1 * c1 + w2 * c2);
}

struct Sym3 {
  double a[3][3];
  double v[3][3];
};
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(Sym3& m) {
  const double apq = m.a[P][Q];
  if (apq == 0.0) return;
  const double app = m.a[P][P], aqq = m.a[Q][Q];
  // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), c = 1 / sqrt(1 + t^2), s = t c without tau and t (oracle: orc_eig3_sym, the
  // same operations in the same order): the dependent chain is sqrt -> sqrt -> div instead of div -> sqrt -> div -> sqrt -> div,
  // and the per-query fit of an edge is ~20 such rotations one after the other on a lane that has nothing else to do
  double d = aqq - app, b = 2.0 * apq;
  double h = sqrt(d * d + b * b);
  double u = fabs(d) + h;
  double r = sqrt(u * u + b * b);
  if (__builtin_expect(!(r > 0.0 && r < __builtin_inf()), 0)) {
    // d*d + b*b under- or overflowed (|d|, |b| < ~1e-154 or > ~1e154; b != 0 here): the rotation only depends on d : b,
    // so take them relative to the larger one.  Never taken for covariances of metre-scale clouds (the sweep loop stops at
    // off-diagonal mass < 1e-36 of the diagonal's long before); the oracle has the same branch (orc_eig3_sym).
    const double sc = fmax(fabs(d), fabs(b));
    d /= sc; b /= sc;
    h = sqrt(d * d + b * b);
    u = fabs(d) + h;
    r = sqrt(u * u + b * b);
  }
  const double cs = u / r;
  double sn = fabs(b) / r;
  if (!(d == 0.0 || (d > 0.0) == (b > 0.0))) sn = -sn;   // sgn(tau), tau = +-0 counting as positive
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double akp = m.a[k][P], akq = m.a[k][Q];
    m.a[k][P] = cs * akp - sn * akq;
    m.a[k][Q] = sn * akp + cs * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double apk = m.a[P][k], aqk = m.a[Q][k];
    m.a[P][k] = cs * apk - sn * aqk;
    m.a[Q][k] = sn * apk + cs * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double vkp = m.v[k][P], vkq = m.v[k][Q];
    m.v[k][P] = cs * vkp - sn * vkq;
    m.v[k][Q] = sn * vkp + cs * vkq;
  }
}
template <int J>
__device__ __forceinline__ void sort_swap(double ev[3], Sym3& m) {
  if (ev[J] > ev[J + 1]) {
    const double t = ev[J]; ev[J] = ev[J + 1]; ev[J + 1] = t;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double tv = m.v[k][J]; m.v[k][J] = m.v[k][J + 1]; m.v[k][J + 1] = tv; }
  }
}
// eigenvalues ascending in ev, unit eigenvectors in the columns of m.v
__device__ __forceinline__ void eig3_sym(Sym3& m, double ev[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) m.v[i][j] = (i == j) ? 1.0 : 0.0;
  // Entries whose SQUARES leave the fp64 range (covariances below ~1e-120 or above ~1e120 -- not of metre-scale clouds) are
  // brought to order 1 by a power of two first: exact, so the rotations and the eigenvectors are the same ones and the
  // eigenvalues are scaled back exactly; in the normal range nothing is touched (orc_eig3_sym has the same branch).
  int rescale = 0;
  {
    double mx = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) mx = fmax(mx, fabs(m.a[i][j]));
    if (__builtin_expect(mx != 0.0 && mx < __builtin_inf() && (mx < 1e-120 || mx > 1e120), 0)) {
      (void)frexp(mx, &rescale);
      const double f = ldexp(1.0, -rescale);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m.a[i][j] *= f;
    }
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = m.a[0][1] * m.a[0][1] + m.a[0][2] * m.a[0][2] + m.a[1][2] * m.a[1][2];
    const double dia = m.a[0][0] * m.a[0][0] + m.a[1][1] * m.a[1][1] + m.a[2][2] * m.a[2][2];
    if (off <= 1e-36 * dia) break;  // off-diagonal mass below fp64 resolution of the eigenvalues
    jacobi_rotate<0, 1>(m);
    jacobi_rotate<0, 2>(m);
    jacobi_rotate<1, 2>(m);
  }
  ev[0] = m.a[0][0]; ev[1] = m.a[1][1]; ev[2] = m.a[2][2];
  if (__builtin_expect(rescale != 0, 0)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) ev[i] = ldexp(ev[i], rescale);
  }
  sort_swap<0>(ev, m);
  sort_swap<1>(ev, m);
  sort_swap<0>(ev, m);
}

}  // namespace tl
