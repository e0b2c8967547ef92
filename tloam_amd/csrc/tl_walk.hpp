// tl_walk.hpp -- knn_rows: one query against the HBM grid of its kind, by LPQ cooperating lanes (1, 4 or 16): the walk of
// K1 (tl_nn.hip: the builders of the registration path, K = 1 / 5) and, since round 4, of the PCA pass of the feature extraction
// (tl_feature.hip, K = 20, LPQ = 1).  Included where TL_K1_STAMP is defined (a development aid of tl_nn.hip; empty elsewhere).
//   LPQ = 1: the lane resolves the nine (z,y) rows of the 27-cell neighbourhood (one 16-byte cell-table
//            request per row, all in flight) and walks them as one candidate stream, four records per trip.
//   LPQ = 4 / 16: smaller frames are latency-bound (a ~35-candidate dependent chain per query), so 4 (16)
//            adjacent lanes split the nine rows -- three (at most one) each --, walk them in parallel and merge
//            their packed-key lists with two (four) xor-shuffle rounds; lane 0 of the group finishes the fit.
#pragma once

#include "tl_common.hpp"
#include "tl_knn.hpp"

#ifndef TL_K1_STAMP
#define TL_K1_STAMP(i)
#endif

namespace tl {

template <int K, int LPQ>
__device__ __forceinline__ void knn_rows(const GridView& g, const PtsGlobal& pts, Vec3 pw, int sub, TopK<K>& tk,
                                         int2* __restrict__ lds_rows, double radius, NbrXyz<K>* xyz = nullptr) {
  TL_K1_STAMP(1)
  const int cx = cell_coord(pw.x, g.org[0], g.inv_cell, g.dim[0]);
  const int cy = cell_coord(pw.y, g.org[1], g.inv_cell, g.dim[1]);
  const int cz = cell_coord(pw.z, g.org[2], g.inv_cell, g.dim[2]);
  int x0 = cx - 1;
  if (x0 < 0) x0 = 0;
  constexpr int NR = (9 + LPQ - 1) / LPQ;  // rows per lane
  int rs[NR], re[NR];
  // CLIPPED walk: only the cells the search ball can reach.  A neighbour is kept only below the radius (radius_cut), so
  // a cell whose nearest point is farther than radius * (1 + 1e-6) from the query holds nothing that can be kept: per
  // axis the distance to the lower / upper neighbour slab follows from the query's position inside its own cell, a row
  // (dy, dz) is walked only if dy^2 + dz^2 <= reach^2 and its outer cells only if the x term still fits.  With
  // cell = radius, on average 20.6 of the 27 cells survive (the volume of cube (+) ball): a quarter of the candidate
  // records is never fetched.  The margin (2e-6 relative on the squares) dwarfs the rounding of these few products.
  const double cl = g.cell, reach = radius * (1.0 + 1e-6), c2 = reach * reach;
  const double fx = (pw.x - g.org[0]) * g.inv_cell - (double)cx, fy = (pw.y - g.org[1]) * g.inv_cell - (double)cy,
               fz = (pw.z - g.org[2]) * g.inv_cell - (double)cz;
  const double dxl = fx * cl, dxr = (1.0 - fx) * cl, dyl = fy * cl, dyr = (1.0 - fy) * cl, dzl = fz * cl, dzr = (1.0 - fz) * cl;
  const double sxl = dxl * dxl, sxr = dxr * dxr, syl = dyl * dyl, syr = dyr * dyr, szl = dzl * dzl, szr = dzr * dzr;
  // The row's table entries (start of its first cell, end of its last) are at most three ints apart: ONE 16-byte
  // request per row instead of two 4-byte ones.  (Reads up to 12 bytes past the last entry of the table: inside the
  // allocation slack of DBuf.)
  typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = sub + i * LPQ;
    const int rz = r / 3, ry = r % 3;
    const int z = cz - 1 + rz, y = cy - 1 + ry;
    const double s2 = (rz == 0 ? szl : (rz == 2 ? szr : 0.0)) + (ry == 0 ? syl : (ry == 2 ? syr : 0.0));
    int xa = cx - ((s2 + sxl <= c2) ? 1 : 0), xb = cx + ((s2 + sxr <= c2) ? 1 : 0);
    if (xa < 0) xa = 0;
    if (xb >= g.dim[0]) xb = g.dim[0] - 1;
    const bool in = (r < 9) && (s2 <= c2) && (xa <= xb) && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
    const size_t base = in ? ((size_t)z * g.dim[1] + y) * g.dim[0] + x0 : 0;
    const int4u t = *reinterpret_cast<const int4u*>(g.cell_start + base);
    const int ia = xa - x0, ib = xb + 1 - x0;  // 0..1, 1..3
    rs[i] = in ? (ia == 0 ? t.x : t.y) : 0;
    re[i] = in ? (ib == 3 ? t.w : (ib == 2 ? t.z : t.y)) : 0;
  }
  topk_clear<K>(tk);
  TL_K1_STAMP(2)
  if (LPQ == 1) {
    // FLATTENED walk: the lane's non-empty rows are queued in LDS ([row][lane], conflict-free) and consumed
    // as ONE candidate stream, two candidates per trip.  The wave then runs max_lanes(total candidates)/2
    // trips instead of sum_rows max_lanes(row length)/2 -- half the trips on the 1 M frame, where the lanes
    // of a wave (one 4x4x4-cell tile) see very different row lengths.
    const int lane = threadIdx.x & 63;
    int nr = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int len = re[i] - rs[i];
      if (len > 0) {
        lds_rows[nr * 64 + lane] = int2{rs[i], re[i]};
        ++nr;
        total += len;
      }
    }
    int j = 0, e = 0, r = 0;
    int2 nx = (nr > 0) ? lds_rows[lane] : int2{0, 0};  // next row, pre-loaded
    // (ballot, not a shuffle reduction: the kinds of a wave's lanes may differ, and only a ballot is
    //  well-defined under the divergent kind branch)
    const unsigned keep_mask = ~((1u << key_bits_for(g.n)) - 1u);
    KeyList<K + 1> L;
    keys_clear<K + 1>(L);
    // next position of this lane's candidate stream (index 0 = a harmless in-range dummy when exhausted)
    auto next = [&](int& jx, bool& vx) {
      if (j >= e && r < nr) { j = nx.x; e = nx.y; ++r; nx = lds_rows[(r < nr ? r : 0) * 64 + lane]; }
      vx = j < e;
      jx = vx ? j : 0;
      j += vx ? 1 : 0;
    };
    // kCpt candidates per trip, and the records of the NEXT trip requested before this trip's insertions: with
    // the two-instruction key insertion the walk is bound by the record round trips (PMC: 53 % of the wave
    // cycles waiting on memory at two records in flight), so the trip carries as many independent loads as the
    // register budget allows.
#ifndef TLOAM_K1_CPT
#define TLOAM_K1_CPT 4
#endif
    constexpr int kCpt = TLOAM_K1_CPT;
    int jc[kCpt];
    bool vc[kCpt];
    double4 rc[kCpt];
#pragma unroll
    for (int u = 0; u < kCpt; ++u) next(jc[u], vc[u]);
#pragma unroll
    for (int u = 0; u < kCpt; ++u) rc[u] = pts.p[jc[u]];
#ifdef TLOAM_K1_DBG_NOWALK  // timing experiment only: row resolution and the first trip, no candidate loop
    total = 0;
#endif
#ifdef TLOAM_K1_DBG_MAXTRIPS  // timing experiment only: the walk cut off after a fixed number of trips
    int trip_no = 0;
    for (int left = total; __any(left > 0) && trip_no < TLOAM_K1_DBG_MAXTRIPS; left -= kCpt, ++trip_no) {
#else
    for (int left = total; __any(left > 0); left -= kCpt) {
#endif
      int jn[kCpt];
      bool vn[kCpt];
      double4 rn[kCpt];
#pragma unroll
      for (int u = 0; u < kCpt; ++u) next(jn[u], vn[u]);
#pragma unroll
      for (int u = 0; u < kCpt; ++u) rn[u] = pts.p[jn[u]];
#pragma unroll
      for (int u = 0; u < kCpt; ++u) {
        const double du = sqdist(pw.x, pw.y, pw.z, rc[u].x, rc[u].y, rc[u].z);
        key_insert<K + 1>(L, key_pack(du, jc[u], keep_mask, vc[u]));
      }
#pragma unroll
      for (int u = 0; u < kCpt; ++u) { jc[u] = jn[u]; vc[u] = vn[u]; rc[u] = rn[u]; }
    }
    if (!keys_ambiguous<K + 1>(L, keep_mask)) {
      keys_unpack<K, K + 1>(L, pts, pw.x, pw.y, pw.z, keep_mask, tk, xyz);
    } else {  // two kept distances agree in every mantissa bit the key keeps: redo with the exact (d, original index) order
      topk_clear<K>(tk);
      for (int q = 0; q < nr; ++q) {
        const int2 v = lds_rows[q * 64 + lane];
        scan_range<K>(pts, v.x, v.y, pw.x, pw.y, pw.z, tk);
      }
      if (xyz) nbr_fetch<K>(pts, tk, xyz);
    }
  } else {
    const unsigned keep_mask = ~((1u << key_bits_for(g.n)) - 1u);
    KeyList<K + 1> L;
    keys_clear<K + 1>(L);
    int len = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) len = max(len, re[i] - rs[i]);
    // kU steps per trip: candidates s .. s + kU - 1 of each of this lane's rows are requested together (the walk of a
    // small frame is a latency chain: one record round trip per trip, so the trip carries as many as the rows allow)
#ifndef TLOAM_K1_WIDE_KU
#define TLOAM_K1_WIDE_KU 4
#endif
    constexpr int kU = (NR == 1) ? TLOAM_K1_WIDE_KU : 2;
#ifdef TLOAM_K1_DBG_NOWALK  // timing experiment only
    len = 0;
#endif
    for (int s = 0; s < len; s += kU) {
      double4 c[kU][NR];
#pragma unroll
      for (int u = 0; u < kU; ++u)
#pragma unroll
        for (int i = 0; i < NR; ++i) c[u][i] = pts.p[(rs[i] + s + u < re[i]) ? rs[i] + s + u : 0];
#pragma unroll
      for (int u = 0; u < kU; ++u)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const bool v = rs[i] + s + u < re[i];
          key_insert<K + 1>(L, key_pack(sqdist(pw.x, pw.y, pw.z, c[u][i].x, c[u][i].y, c[u][i].z), rs[i] + s + u, keep_mask, v));
        }
    }
    TL_K1_STAMP(3)
    // merge across the quad: after xor-1 and xor-2 every lane holds the global list (the lanes' candidate
    // sets are disjoint, +inf entries fall through)
#ifndef TLOAM_K1_DBG_NOMERGE  // timing experiment only: without the cross-lane merge
#pragma unroll
    for (int x = 1; x < LPQ; x <<= 1) {
      double ok[K + 1];
#pragma unroll
      for (int m = 0; m < K + 1; ++m) ok[m] = __shfl_xor(L.k[m], x, 64);
#pragma unroll
      for (int m = 0; m < K + 1; ++m) key_insert<K + 1>(L, ok[m]);
    }
#endif
    TL_K1_STAMP(4)
    if (!keys_ambiguous<K + 1>(L, keep_mask)) {  // (the same verdict on all lanes of the quad)
      keys_unpack<K, K + 1>(L, pts, pw.x, pw.y, pw.z, keep_mask, tk, xyz);
      TL_K1_STAMP(5)
    } else {  // redo with the exact (d, original index) order
      for (int s = 0; s < len; ++s) {
        double4 c[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) c[i] = pts.p[(rs[i] + s < re[i]) ? rs[i] + s : 0];
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (rs[i] + s < re[i])
            topk_insert<K, PtsGlobal>(tk, pts, sqdist(pw.x, pw.y, pw.z, c[i].x, c[i].y, c[i].z), rs[i] + s);
      }
#pragma unroll
      for (int x = 1; x < LPQ; x <<= 1) {
        double od[K];
        int oj[K];
#pragma unroll
        for (int m = 0; m < K; ++m) { od[m] = __shfl_xor(tk.d[m], x, 64); oj[m] = __shfl_xor(tk.j[m], x, 64); }
#pragma unroll
        for (int m = 0; m < K; ++m)
          if (oj[m] >= 0) topk_insert<K, PtsGlobal>(tk, pts, od[m], oj[m]);
      }
      if (xyz) nbr_fetch<K>(pts, tk, xyz);
    }
  }
}

}  // namespace tl
