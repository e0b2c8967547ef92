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

namespace tl {

namespace {
constexpr int kFeatK = 20;  // feature.yaml: K

// calculatePCAInfo (:61-119), one thread per point
__global__ __launch_bounds__(64) void k_pca_info(FeatArgs A) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= A.n) return;
  // value-initialised PCAInfo (pca_info_.resize, :59)
  double flat = 0.0, cvr = 0.0, sph = 0.0, nx = 0.0, ny = 0.0, nz = 0.0;
  int num = 0;
  const double qx = A.x[i], qy = A.y[i], qz = A.z[i];
  TopK<kFeatK> tk;
  knn_grid_fast<kFeatK>(A.g, qx, qy, qz, tk);
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
        const int j = tk.j[m];
        const double x = pts.X(j), y = pts.Y(j), z = pts.Z(j);
        nb[m] = pts.I(j);
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
__global__ __launch_bounds__(256) void k_feat_compact(FeatArgs A, const unsigned long long* __restrict__ flags,
                                                      const unsigned long long* __restrict__ scan, double* __restrict__ pf,
                                                      int* __restrict__ pidx, double* __restrict__ sf, int* __restrict__ sidx) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= A.n) return;
  const unsigned long long f = flags[id], s = scan[id];
  if (f >> 32) { pf[s >> 32] = A.flatness[id]; pidx[s >> 32] = id; }
  if (f & 0xffffffffull) { sf[s & 0xffffffffull] = A.flatness[id]; sidx[s & 0xffffffffull] = id; }  // :162 FLATNESS
}
// std::sort descending by flatness (:168-174), made total: ties by ascending index.  Rank by counting --
// O(m^2) over at most a few 10^4 candidates, spread over a 2-D grid: block (bx, by) counts, for its 256
// candidates, the entries of the by-th slice of the list that sort before them (slice staged through LDS in
// tiles of 256), and adds the partial count to rank[] -- integer atomics, so the result does not depend on
// the order of arrival.  A second kernel scatters by rank.
constexpr int kRankSlices = 64;
__global__ __launch_bounds__(256) void k_feat_rank(const double* __restrict__ f, const int* __restrict__ idx,
                                                   const unsigned long long* __restrict__ total, int shift,
                                                   int* __restrict__ rank_out) {
  __shared__ double tf[256];
  __shared__ int ti[256];
  const int m = (int)((*total >> shift) & 0xffffffffull);
  if ((int)(blockIdx.x * 256) >= m) return;  // block-uniform
  const int per = ((m + kRankSlices - 1) / kRankSlices + 255) / 256 * 256;  // slice length, a multiple of the tile
  const int j0 = blockIdx.y * per, j1 = min(m, j0 + per);
  if (j0 >= j1) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < m;
  const double fi = live ? f[i] : 0.0;
  const int ii = live ? idx[i] : 0;
  int rank = 0;
  for (int t0 = j0; t0 < j1; t0 += 256) {
    const int j = t0 + threadIdx.x;
    tf[threadIdx.x] = j < j1 ? f[j] : 0.0;
    ti[threadIdx.x] = j < j1 ? idx[j] : 0;
    __syncthreads();
    const int lim = min(256, j1 - t0);
    for (int q = 0; q < lim; ++q) {
      const double fj = tf[q];
      rank += (fj > fi || (fj == fi && ti[q] < ii)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (live && rank) atomicAdd(&rank_out[i], rank);
}
__global__ __launch_bounds__(256) void k_feat_scatter(const double* __restrict__ f, const int* __restrict__ idx,
                                                      const unsigned long long* __restrict__ total, int shift,
                                                      const int* __restrict__ rank, double* __restrict__ of,
                                                      int* __restrict__ oidx) {
  const int m = (int)((*total >> shift) & 0xffffffffull);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  of[rank[i]] = f[i];
  oidx[rank[i]] = idx[i];
}
}  // namespace

void launch_pca_info(const FeatArgs& A, hipStream_t s) {
  if (A.n <= 0) return;
  hipLaunchKernelGGL(k_pca_info, dim3((A.n + 63) / 64), dim3(64), 0, s, A);
}
// flags/scan: n + 1 entries; candidate lists: n entries each; total = scan[n] (planar << 32 | sphere)
void launch_feat_select(const FeatArgs& A, const FeatSelect& S, unsigned long long* flags, unsigned long long* scan,
                        unsigned long long* scan_tmp, double* pf, int* pidx, double* sf, int* sidx, double* pf_sorted,
                        int* pidx_sorted, double* sf_sorted, int* sidx_sorted, int* rank /*[2n]*/, hipStream_t s) {
  const int n = A.n;
  hipLaunchKernelGGL(k_feat_select, dim3((n + 1 + 255) / 256), dim3(256), 0, s, A, S, flags);
  launch_exclusive_scan_u64(flags, scan, (size_t)n + 1, scan_tmp, s);
  if (n <= 0) return;
  hipLaunchKernelGGL(k_feat_compact, dim3((n + 255) / 256), dim3(256), 0, s, A, flags, scan, pf, pidx, sf, sidx);
  // the candidate counts are only known on the device: the rank kernels cover n and bound themselves
  (void)hipMemsetAsync(rank, 0, sizeof(int) * 2 * (size_t)n, s);
  const dim3 g2((n + 255) / 256, kRankSlices), g1((n + 255) / 256);
  hipLaunchKernelGGL(k_feat_rank, g2, dim3(256), 0, s, pf, pidx, scan + n, 32, rank);
  hipLaunchKernelGGL(k_feat_rank, g2, dim3(256), 0, s, sf, sidx, scan + n, 0, rank + n);
  hipLaunchKernelGGL(k_feat_scatter, g1, dim3(256), 0, s, pf, pidx, scan + n, 32, rank, pf_sorted, pidx_sorted);
  hipLaunchKernelGGL(k_feat_scatter, g1, dim3(256), 0, s, sf, sidx, scan + n, 0, rank + n, sf_sorted, sidx_sorted);
}

}  // namespace tl
