// tl_feature.hip -- PCA feature extraction on the device: the step right before the
// registration path, and the other OpenMP hot loop of the odometry nodelet.
//
// Replaces
//   featureExtract::calculatePCAInfo     src/models/feature_extraction/feature_extract.cpp:47-122
//   featureExtract::extractPlanarSphere  src/models/feature_extraction/feature_extract.cpp:133-197 (selection +
//                                        ranking; the final rank/threshold loops :178-190 run on the host)
// with the machinery of K1/K2: the uniform grid (cell >= radius), the exact (distance, index)-ordered top-K
// list -- K = 20 here, walked with packed keys (tl_knn.hpp) -- and the cyclic-Jacobi 3x3 eigen solve.  Compiled with -ffp-contract=off.
#include "tl_common.hpp"
#include "tl_knn.hpp"
#include "tl_walk.hpp"

namespace tl {

namespace {
constexpr int kFeatK = 20;  // feature.yaml: K

// calculatePCAInfo (:61-119), one thread per point.  The points are taken in the GRID's order (cell by cell: thread j owns
// the j-th record of the cell-sorted array, which carries the point's own index): the 64 queries of a wave then sit in a few
// neighbouring cells, walk the same rows of cells and share the candidates' cache lines -- what the query sort does for the
// 1 M-query search of the registration path, here for free, the queries being the grid's own points.  Every point's result is
// the same whichever thread computes it.  (TLOAM_PCA_INDEX_ORDER: thread i owns point i, as until round 4.)
__global__ __launch_bounds__(64) void k_pca_info(FeatArgs A) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= A.n) return;
  // value-initialised PCAInfo (pca_info_.resize, :59)
  double flat = 0.0, cvr = 0.0, sph = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
  int num = 0;
#ifdef TLOAM_PCA_INDEX_ORDER
  const int i = j;
  const double qx = A.x[i], qy = A.y[i], qz = A.z[i];
#else
  const double4 self = A.g.gp[j];
  const int i = (int)__double_as_longlong(self.w);
  const double qx = self.x, qy = self.y, qz = self.z;
#endif
  TopK<kFeatK> tk;
#ifdef TLOAM_PCA_PLAIN_WALK
  knn_grid_fast<kFeatK>(A.g, qx, qy, qz, tk);
#else
  // the walk of the registration path's 1 M-query search (tl_walk.hpp, one lane per query): only the cells the search ball can
  // reach, the lane's rows as ONE candidate stream, four records per trip with the next trip's records requested ahead.  Entries
  // at or beyond the radius may differ from the unclipped walk's; they are cut below (SearchHybrid: K nearest, THEN the radius)
  __shared__ int2 s_rows[9 * 64];
  knn_rows<kFeatK, 1>(A.g, PtsGlobal{A.g.gp}, Vec3{qx, qy, qz}, 0, tk, s_rows, A.radius);
#endif
  // SearchHybrid(cur_pt, r, K): the K nearest (K <= 20: a prefix of the sorted list), then the radius cut (:71)
  const double r2 = A.radius * A.radius;
  int cnt = 0;
#pragma unroll
  for (int m = 0; m < kFeatK; ++m) cnt += (m < A.K && tk.d[m] < r2) ? 1 : 0;
  const PtsGlobal pts{A.g.gp};
  int nb[kFeatK];
#pragma unroll
  for (int m = 0; m < kFeatK; ++m) nb[m] = -1;
  if (cnt > A.min_neigh) {  // :72 `neigh_index.size() <= min_neigh -> continue`
    double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < kFeatK; ++m) {
      if (m < cnt) {  // :80-91 neighbours in ascending distance
        const double4 p = pts.p[tk.j[m]];   // (ONE record load, two 16-byte accesses: X / Y / Z / I apart were four -- the L1 is asked per lane access)
        const double x = p.x, y = p.y, z = p.z;
        nb[m] = (int)__double_as_longlong(p.w);
        cum[0] += x; cum[1] += y; cum[2] += z;
        cum[3] += x * x; cum[4] += x * y; cum[5] += x * z;
        cum[6] += y * y; cum[7] += y * z; cum[8] += z * z;
      }
    }
    const double dn = (double)cnt;
#pragma unroll
    for (int m = 0; m < 9; ++m) cum[m] /= dn;  // :92
    Sym3 M;
    M.a[0][0] = cum[3] - cum[0] * cum[0];
    M.a[1][1] = cum[6] - cum[1] * cum[1];
    M.a[2][2] = cum[8] - cum[2] * cum[2];
    M.a[0][1] = M.a[1][0] = cum[4] - cum[0] * cum[1];
    M.a[0][2] = M.a[2][0] = cum[5] - cum[0] * cum[2];
    M.a[1][2] = M.a[2][1] = cum[7] - cum[1] * cum[2];
    double ev[3];
    eig3_sym(M, ev);  // ascending (SelfAdjointEigenSolver, :104-107)
    nx = M.v[0][0]; ny = M.v[1][0]; nz = M.v[2][0];  // eigenvectors().col(0)
    const double sum = (ev[0] + ev[1]) + ev[2];
    cvr = (sum == 0.0) ? 0.0 : ev[0] / sum;  // :109-114
    flat = (ev[1] - ev[0]) / ev[2];          // :116
    sph = ev[0] / ev[2];                     // :117
    num = cnt;
  }
  A.flatness[i] = flat; A.cvr[i] = cvr; A.sphericity[i] = sph;
  A.normal[3 * (size_t)i] = nx; A.normal[3 * (size_t)i + 1] = ny; A.normal[3 * (size_t)i + 2] = nz;
  A.num_sum[i] = num;
  if (A.K == kFeatK) {   // (the configured K: a point's twenty indices as five 16-byte stores instead of twenty scattered 4-byte ones)
    typedef int int4s __attribute__((ext_vector_type(4), aligned(16)));
    int4s* __restrict__ o = reinterpret_cast<int4s*>(A.neigh + (size_t)i * kFeatK);
#pragma unroll
    for (int m = 0; m < kFeatK; m += 4) o[m / 4] = int4s{nb[m], nb[m + 1], nb[m + 2], nb[m + 3]};
    return;
  }
#pragma unroll
  for (int m = 0; m < kFeatK; ++m)
    if (m < A.K) A.neigh[(size_t)i * A.K + m] = nb[m];
}

// extractPlanarSphere :149-165 -- candidate flags, packed (planar << 32 | sphere) for one scan
__global__ __launch_bounds__(256) void k_feat_select(FeatArgs A, FeatSelect S, unsigned long long* __restrict__ flags) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id > A.n) return;
  if (id == A.n) { flags[id] = 0ull; return; }  // scan terminator
  unsigned long long f = 0ull;
  const double fl = A.flatness[id], cv = A.cvr[id];
  if (fl > S.planar_submap_thres && fabs(A.normal[3 * (size_t)id + 2]) < S.planar_vertic_thres) {
    f = 1ull << 32;
  } else if (cv > S.cvr_submap) {
    bool max_uniform = true;
    const int m = A.num_sum[id];
    for (int q = 0; q < m; ++q)
      if (cv < A.cvr[A.neigh[(size_t)id * A.K + q]]) { max_uniform = false; break; }
    if (max_uniform) f = 1ull;
  }
  flags[id] = f;
}
// ---- std::sort descending by flatness (:168-174), made total: ties by ascending index ------------------------------------
// Round 3 ranked by counting over the WHOLE list (O(m^2): 59 + 98 us for the 10 k + 20 k candidates of a 100 k-point scan,
// more than the PCA pass).  Now the counting is confined to a neighbourhood of the element's own value:
//   k_feat_compact  also takes the smallest / largest flatness of either list per block (order-preserving keys); k_rank_range
//                   folds the blocks' rows and empties the histograms
//   k_rank_hist     bucket of every candidate -- kRankBuckets equal slices of [min, max], bucket 0 the LARGEST values -- and its
//                   place inside the bucket (the histogram's returning atomic: arbitrary, only used to group)
//   k_rank_scan     bucket starts (one block per list)
//   k_rank_group    candidates grouped by bucket
//   k_rank_final    block b ranks the grouped candidates [256 b, 256 b + 256): counting, as before, but only over the buckets
//                   those candidates lie in -- a few hundred entries instead of the list; the bucket is a monotone function of
//                   the value, so everything in an earlier bucket sorts earlier and the count is exact.  Writes the sorted
//                   lists packed end to end (FeatRankOut) for ONE device-to-host copy.
// A list whose values are all equal degenerates to the old cost (one bucket), never to a wrong order.
constexpr int kRankBuckets = 4096;
__device__ __forceinline__ unsigned long long fkey(double f) {   // order-preserving bits of a double
  const unsigned long long u = (unsigned long long)__double_as_longlong(f);
  return (u >> 63) ? ~u : (u | (1ull << 63));
}
__device__ __forceinline__ double fkey_inv(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
  return __longlong_as_double((long long)u);
}
struct RankRange { double lo, scale; };
__device__ __forceinline__ RankRange rank_range(const FeatRankCtl* ctl, int list) {
  const double lo = fkey_inv(ctl->kmin[list]), hi = fkey_inv(ctl->kmax[list]);
  return RankRange{lo, hi > lo ? (double)kRankBuckets / (hi - lo) : 0.0};
}
__device__ __forceinline__ int rank_bucket(double f, const RankRange& r) {   // monotone non-increasing in f
  const double t = (f - r.lo) * r.scale;
  int b = (int)t;
  b = b < 0 ? 0 : (b > kRankBuckets - 1 ? kRankBuckets - 1 : b);
  return kRankBuckets - 1 - b;
}
__global__ __launch_bounds__(256) void k_feat_compact(FeatArgs A, const unsigned long long* __restrict__ flags,
                                                      const unsigned long long* __restrict__ scan, double* __restrict__ pf,
                                                      int* __restrict__ pidx, double* __restrict__ sf, int* __restrict__ sidx,
                                                      unsigned long long* __restrict__ part /*[blocks][4]: lo0 hi0 lo1 hi1*/) {
  __shared__ unsigned long long s_red[4][4];
  const int id = blockIdx.x * 256 + threadIdx.x;
  unsigned long long v[4] = {~0ull, 0ull, ~0ull, 0ull};   // keys of the smallest / largest flatness this thread adds to list 0 / 1
  if (id < A.n) {
    const unsigned long long f = flags[id], s = scan[id];
    const double fl = A.flatness[id];
    const unsigned long long k = fkey(fl);
    if (f >> 32) { pf[s >> 32] = fl; pidx[s >> 32] = id; v[0] = v[1] = k; }
    if (f & 0xffffffffull) { sf[s & 0xffffffffull] = fl; sidx[s & 0xffffffffull] = id; v[2] = v[3] = k; }  // :162 FLATNESS
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(v[a], off, 64);
      v[a] = (a & 1) ? (o > v[a] ? o : v[a]) : (o < v[a] ? o : v[a]);
    }
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][a] = v[a];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int a = threadIdx.x;
    unsigned long long r = s_red[0][a];
    for (int w = 1; w < 4; ++w) r = (a & 1) ? (s_red[w][a] > r ? s_red[w][a] : r) : (s_red[w][a] < r ? s_red[w][a] : r);
    part[(size_t)blockIdx.x * 4 + a] = r;
  }
}
// the lists' value ranges from the blocks' partials (block 0; min / max are exact in any order); every block of the launch also
// empties its share of the two histograms
__global__ __launch_bounds__(256) void k_rank_range(const unsigned long long* __restrict__ part, int blocks, FeatRankCtl* __restrict__ ctl) {
  __shared__ unsigned long long s_red[4][4];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * (kRankBuckets + 1); i += gridDim.x * 256) (&ctl->hist[0][0])[i] = 0;
  if (blockIdx.x != 0) return;
  unsigned long long v[4] = {~0ull, 0ull, ~0ull, 0ull};
  for (int b = threadIdx.x; b < blocks; b += 256) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned long long o = part[(size_t)b * 4 + a];
      v[a] = (a & 1) ? (o > v[a] ? o : v[a]) : (o < v[a] ? o : v[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(v[a], off, 64);
      v[a] = (a & 1) ? (o > v[a] ? o : v[a]) : (o < v[a] ? o : v[a]);
    }
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][a] = v[a];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int a = threadIdx.x;
    unsigned long long r = s_red[0][a];
    for (int w = 1; w < 4; ++w) r = (a & 1) ? (s_red[w][a] > r ? s_red[w][a] : r) : (s_red[w][a] < r ? s_red[w][a] : r);
    if (a & 1) ctl->kmax[a >> 1] = r; else ctl->kmin[a >> 1] = r;
  }
}
// blockIdx.y = list (0 planar, 1 sphere); f / idx / bkt / pos: the two lists at [0, n) and [n, 2n)
__global__ __launch_bounds__(256) void k_rank_hist(const double* __restrict__ f, const unsigned long long* __restrict__ total, int n,
                                                   FeatRankCtl* __restrict__ ctl, int* __restrict__ bkt, int* __restrict__ pos) {
  const int l = blockIdx.y;
  const int m = (int)((*total >> (l ? 0 : 32)) & 0xffffffffull);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int b = rank_bucket(f[(size_t)l * n + i], rank_range(ctl, l));
  bkt[(size_t)l * n + i] = b;
  pos[(size_t)l * n + i] = atomicAdd(&ctl->hist[l][b], 1);
}
__global__ __launch_bounds__(1024) void k_rank_scan(FeatRankCtl* __restrict__ ctl) {   // one block per list, 4 buckets per thread
  __shared__ int wsum[16];
  int* h = ctl->hist[blockIdx.x];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[4], run = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { v[u] = h[4 * t + u]; run += v[u]; }
  int incl = run;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int pre = incl - run;
  for (int w = 0; w < wave; ++w) pre += wsum[w];
#pragma unroll
  for (int u = 0; u < 4; ++u) { h[4 * t + u] = pre; pre += v[u]; }
  if (t == 1023) h[kRankBuckets] = pre;
}
static_assert(kRankBuckets == 4096, "k_rank_scan: 1024 threads x 4 buckets");
__global__ __launch_bounds__(256) void k_rank_group(const double* __restrict__ f, const int* __restrict__ idx,
                                                    const unsigned long long* __restrict__ total, int n, const FeatRankCtl* __restrict__ ctl,
                                                    const int* __restrict__ bkt, const int* __restrict__ pos, double* __restrict__ gf,
                                                    int* __restrict__ gi) {
  const int l = blockIdx.y;
  const int m = (int)((*total >> (l ? 0 : 32)) & 0xffffffffull);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const size_t o = (size_t)l * n;
  const int p = ctl->hist[l][bkt[o + i]] + pos[o + i];
  gf[o + p] = f[o + i];
  gi[o + p] = idx[o + i];
}
__global__ __launch_bounds__(256) void k_rank_final(const double* __restrict__ gf, const int* __restrict__ gi,
                                                    const unsigned long long* __restrict__ total, int n, const FeatRankCtl* __restrict__ ctl,
                                                    double* __restrict__ out) {
  __shared__ double tf[256];
  __shared__ int ti[256];
  __shared__ int s_rng[2];
  const int l = blockIdx.y;
  const int np = (int)(*total >> 32), ns = (int)(*total & 0xffffffffull);
  const int m = l ? ns : np;
  if ((int)(blockIdx.x * 256) >= m) return;  // block-uniform
  const size_t o = (size_t)l * n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < m;
  const double fi = live ? gf[o + i] : 0.0;
  const int ii = live ? gi[o + i] : 0;
  // the buckets this block's candidates lie in: grouped order = bucket order, so the first and the last live candidate bound them
  const RankRange rr = rank_range(ctl, l);
  const int last = min(m, (int)(blockIdx.x * 256) + 256) - 1;
  if (threadIdx.x == 0) s_rng[0] = ctl->hist[l][rank_bucket(fi, rr)];
  if (i == last) s_rng[1] = ctl->hist[l][rank_bucket(fi, rr) + 1];
  // a candidate's rank = start of its bucket + the entries of that bucket that sort before it.  Small buckets (the rule): every
  // thread walks its own bucket; a block that meets a crowded one (many equal values) counts over all its buckets through LDS
  const int bi = rank_bucket(fi, rr);
  const int b0 = live ? ctl->hist[l][bi] : 0, b1 = live ? ctl->hist[l][bi + 1] : 0;
  const bool crowded = __syncthreads_or((b1 - b0 > 48) ? 1 : 0) != 0;   // (also the barrier behind the s_rng stores)
  if (!crowded) {
    if (!live) return;
    int rank = b0;
    for (int j = b0; j < b1; ++j) {
      const double fj = gf[o + j];
      rank += (fj > fi || (fj == fi && gi[o + j] < ii)) ? 1 : 0;
    }
    int* oi = reinterpret_cast<int*>(out + (size_t)np + (size_t)ns);
    if (l == 0) { out[rank] = fi; oi[rank] = ii; }
    else { out[(size_t)np + rank] = fi; oi[(size_t)np + rank] = ii; }
    return;
  }
  const int j0 = s_rng[0], j1 = s_rng[1];
  int rank = j0;   // everything in front of the range sorts earlier
  for (int t0 = j0; t0 < j1; t0 += 256) {
    const int j = t0 + threadIdx.x;
    tf[threadIdx.x] = j < j1 ? gf[o + j] : 0.0;
    ti[threadIdx.x] = j < j1 ? gi[o + j] : 0;
    __syncthreads();
    const int lim = min(256, j1 - t0);
    for (int q = 0; q < lim; ++q) {
      const double fj = tf[q];
      rank += (fj > fi || (fj == fi && ti[q] < ii)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (!live) return;
  // FeatRankOut: [planar flatness np | sphere flatness ns] doubles, then [planar index np | sphere index ns] ints
  int* oi = reinterpret_cast<int*>(out + (size_t)np + (size_t)ns);
  if (l == 0) { out[rank] = fi; oi[rank] = ii; }
  else { out[(size_t)np + rank] = fi; oi[(size_t)np + rank] = ii; }
}
}  // namespace

void launch_pca_info(const FeatArgs& A, hipStream_t s) {
  if (A.n <= 0) return;
  hipLaunchKernelGGL(k_pca_info, dim3((A.n + 63) / 64), dim3(64), 0, s, A);
}
// flags/scan: n + 1 entries; candidate lists pf | sf and pidx | sidx: 2n entries each (the sphere list at [n, 2n)), likewise the
// work arrays bkt / pos / gf / gi; total = scan[n] (planar << 32 | sphere); out: 3n doubles (see k_rank_final)
void launch_feat_select(const FeatArgs& A, const FeatSelect& S, unsigned long long* flags, unsigned long long* scan,
                        unsigned long long* scan_tmp, double* f2, int* idx2, FeatRankCtl* ctl, int* bkt, int* pos, double* gf, int* gi,
                        double* out, hipStream_t s) {
  const int n = A.n;
  hipLaunchKernelGGL(k_feat_select, dim3((n + 1 + 255) / 256), dim3(256), 0, s, A, S, flags);
  launch_exclusive_scan_u64(flags, scan, (size_t)n + 1, scan_tmp, s);
  if (n <= 0) return;
  // (the blocks' value ranges go through the front of the work array `pos`: 4 x 8 bytes per block of 256 points, read by
  //  k_rank_range before k_rank_hist writes the array)
  unsigned long long* part = reinterpret_cast<unsigned long long*>(pos);
  const int cblocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_feat_compact, dim3(cblocks), dim3(256), 0, s, A, flags, scan, f2, idx2, f2 + n, idx2 + n, part);
  hipLaunchKernelGGL(k_rank_range, dim3(8), dim3(256), 0, s, part, cblocks, ctl);
  // the candidate counts are only known on the device: the rank kernels cover n and bound themselves
  const dim3 g2((n + 255) / 256, 2);
  hipLaunchKernelGGL(k_rank_hist, g2, dim3(256), 0, s, f2, scan + n, n, ctl, bkt, pos);
  hipLaunchKernelGGL(k_rank_scan, dim3(2), dim3(1024), 0, s, ctl);
  hipLaunchKernelGGL(k_rank_group, g2, dim3(256), 0, s, f2, idx2, scan + n, n, ctl, bkt, pos, gf, gi);
  hipLaunchKernelGGL(k_rank_final, g2, dim3(256), 0, s, gf, gi, scan + n, n, ctl, out);
}

}  // namespace tl
