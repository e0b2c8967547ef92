// tl_submap.hip -- device-resident submap maintenance: the target side of the path.
//
// Replaces, for the four submap clouds handed to setInputTarget, the CPU chain of FrontEnd::updateSubmap
// (front_end.cpp:201-275):
//     cloud->Transform(pose)                       PointCloud2.cpp:71-75   (Open3D TransformPoints)
//     *submap += cloud                             PointCloud2.cpp:96-132  (append, index order kept)
//     submap->Crop(box)                            PointCloud2.cpp:551-559 (inclusive bounds, index order kept)
//           ->VoxelDownSample(voxel)               PointCloud2.cpp:358-403 (per-voxel mean, accumulated in
//                                                                            index order; AccumulatedPoint :246-291)
// so that the submap never leaves HBM between frames: the result is written straight into the SoA target
// arrays the search grids are built from.
//
// VoxelDownSample emits its voxels in std::unordered_map iteration order, which is implementation-defined;
// this restatement (device AND oracle) emits them in order of FIRST OCCURRENCE (ascending smallest member
// index).  The scan-matching result does not depend on the order of the target points (exact k-NN, ties by
// index are measure-zero after averaging).
//
// Voxel membership uses an open-addressing hash table keyed by the packed voxel coordinates
// (ix | iy << 21 | iz << 42); per-voxel member lists are ordered by index with a rank-by-counting pass so the
// floating-point accumulation order is the reference's (ascending index).  Compiled with -ffp-contract=off.
#include <algorithm>

#include "tl_common.hpp"

namespace tl {

namespace {
constexpr unsigned long long kEmpty = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

struct Mat16 { double m[16]; };  // column-major 4x4

// Open3D TransformPoints: new = T * (x, y, z, 1); point = new.head<3>() / new(3).  Each row is accumulated
// left to right (assumed upstream behaviour: Open3D is not under /root/reference).
__global__ void k_transform_to_soa(const double* __restrict__ aos, size_t n, Mat16 M, double* __restrict__ ox,
                                   double* __restrict__ oy, double* __restrict__ oz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = aos[3 * i], y = aos[3 * i + 1], z = aos[3 * i + 2];
  double r[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) r[a] = ((M.m[a] * x + M.m[4 + a] * y) + M.m[8 + a] * z) + M.m[12 + a] * 1.0;
  ox[i] = r[0] / r[3];
  oy[i] = r[1] / r[3];
  oz[i] = r[2] / r[3];
}

// the same transformed cloud written to TWO outputs (the sphere submap is rebuilt from the planar buffer too)
__global__ void k_transform_to_soa2(const double* __restrict__ aos, size_t n, Mat16 M, double* __restrict__ ax,
                                    double* __restrict__ ay, double* __restrict__ az, double* __restrict__ bx,
                                    double* __restrict__ by, double* __restrict__ bz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = aos[3 * i], y = aos[3 * i + 1], z = aos[3 * i + 2];
  double r[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) r[a] = ((M.m[a] * x + M.m[4 + a] * y) + M.m[8 + a] * z) + M.m[12 + a] * 1.0;
  const double px = r[0] / r[3], py = r[1] / r[3], pz = r[2] / r[3];
  ax[i] = px; ay[i] = py; az[i] = pz;
  bx[i] = px; by[i] = py; bz[i] = pz;
}

// the frames of the planar ring buffer (front_end.cpp:220-243) in ONE launch: blockIdx.y = frame
constexpr int kRingMax = 8;
struct RingArgs {
  const double* aos[kRingMax];
  size_t n[kRingMax], off[kRingMax];
  Mat16 M[kRingMax];
};
__global__ void k_transform_ring(RingArgs R, double* __restrict__ ax, double* __restrict__ ay, double* __restrict__ az,
                                 double* __restrict__ bx, double* __restrict__ by, double* __restrict__ bz) {
  const int f = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R.n[f]) return;
  const double* __restrict__ aos = R.aos[f];
  const Mat16& M = R.M[f];
  const double x = aos[3 * i], y = aos[3 * i + 1], z = aos[3 * i + 2];
  double r[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) r[a] = ((M.m[a] * x + M.m[4 + a] * y) + M.m[8 + a] * z) + M.m[12 + a] * 1.0;
  const double px = r[0] / r[3], py = r[1] / r[3], pz = r[2] / r[3];
  const size_t o = R.off[f] + i;
  ax[o] = px; ay[o] = py; az[o] = pz;
  bx[o] = px; by[o] = py; bz[o] = pz;
}

__global__ void k_copy3(const double* __restrict__ ax, const double* __restrict__ ay, const double* __restrict__ az,
                        size_t n, double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ox[i] = ax[i]; oy[i] = ay[i]; oz[i] = az[i];
}

__global__ void k_soa_to_aos(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                             size_t n, double* __restrict__ aos) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aos[3 * i] = x[i]; aos[3 * i + 1] = y[i]; aos[3 * i + 2] = z[i];
}

__device__ __forceinline__ bool in_box(const VoxelJob& J, int seg, double x, double y, double z) {
  // AxisAlignedBoundingBox::GetPointIndicesWithinBoundingBox: inclusive on both sides
  return x >= J.lo[seg][0] && x <= J.hi[seg][0] && y >= J.lo[seg][1] && y <= J.hi[seg][1] && z >= J.lo[seg][2] &&
         z <= J.hi[seg][2];
}

// GetMinBound() of the cropped clouds (one per segment), block partials
__global__ __launch_bounds__(256) void k_vox_min(VoxelJob J, double* __restrict__ partial /*[blocks][6]*/,
                                                 unsigned long long* __restrict__ keys,
                                                 unsigned long long* __restrict__ cnt, int* __restrict__ overflow) {
  __shared__ double sm[6][256];
  if (blockIdx.x == 0 && threadIdx.x == 0) *overflow = 0;  // (set by k_vox_insert, the launch after the next)
  // the same launch empties the hash table of this job (keys = empty, cnt = 0 incl. the scan terminator)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i <= J.mask + 1; i += (size_t)gridDim.x * 256) {
    if (i <= J.mask) keys[i] = kEmpty;
    cnt[i] = 0ull;
  }
  double m[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) m[a] = __builtin_inf();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (size_t)gridDim.x * 256) {
    const double x = J.x[i], y = J.y[i], z = J.z[i];
    const int seg = i >= J.n0 ? 1 : 0;
    if (in_box(J, seg, x, y, z)) {
      if (seg == 0) { m[0] = fmin(m[0], x); m[1] = fmin(m[1], y); m[2] = fmin(m[2], z); }
      else { m[3] = fmin(m[3], x); m[4] = fmin(m[4], y); m[5] = fmin(m[5], z); }
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) sm[a][threadIdx.x] = m[a];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int a = 0; a < 6; ++a) sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = sm[threadIdx.x][0];
}
// voxel_min_bound = GetMinBound() - voxel_size * 0.5 (:366); an empty cloud has min bound (0, 0, 0)
__global__ __launch_bounds__(384) void k_vox_min_final(const double* __restrict__ partial, int blocks, double voxel0,
                                                       double voxel1, double* __restrict__ vmin) {
  // one wave per (segment, axis), the rows spread over its lanes (min is exact in any order); a 3-thread serial
  // loop over the 256 rows was a 26 us chain of dependent loads -- a quarter of the submap update
  const int a = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double m = __builtin_inf();
  for (int b = lane; b < blocks; b += 64) m = fmin(m, partial[b * 6 + a]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
  if (!(m < __builtin_inf())) m = 0.0;
  if (lane == 0) vmin[a] = m - (a < 3 ? voxel0 : voxel1) * 0.5;
}

// voxel of every in-box point -> hash slot; the counting atomic also hands out an (arbitrary) member rank
__global__ __launch_bounds__(256) void k_vox_insert(VoxelJob J, const double* __restrict__ vmin,
                                                    unsigned long long* __restrict__ keys,
                                                    unsigned long long* __restrict__ cnt, int* __restrict__ slot_of_pt,
                                                    int* __restrict__ urank, int* __restrict__ overflow) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= J.n) return;
  const double x = J.x[i], y = J.y[i], z = J.z[i];
  const int seg = i >= J.n0 ? 1 : 0;
  if (!in_box(J, seg, x, y, z)) { slot_of_pt[i] = -1; return; }
  // ref_coord = (p - voxel_min_bound) / voxel_size; index = int(floor(ref_coord))   (:380-383)
  const double voxel = J.voxel[seg];
  const long long ix = (long long)floor((x - vmin[3 * seg + 0]) / voxel);
  const long long iy = (long long)floor((y - vmin[3 * seg + 1]) / voxel);
  const long long iz = (long long)floor((z - vmin[3 * seg + 2]) / voxel);
  if (ix < 0 || iy < 0 || iz < 0 || ix >= (1ll << 21) || iy >= (1ll << 21) || iz >= (1ll << 21)) {
    *overflow = 1;  // "[VoxelDownSample] voxel_size is too small." (:370-372)
    slot_of_pt[i] = -1;
    return;
  }
  // 3 x 21 bits of voxel coordinates, bit 63 = segment.  The all-ones key (segment 1, all three indices 2^21 - 1)
  // is the empty marker: reported like an index out of range
  const unsigned long long key = (unsigned long long)ix | ((unsigned long long)iy << 21) | ((unsigned long long)iz << 42) |
                                 ((unsigned long long)seg << 63);
  if (key == kEmpty) {
    *overflow = 1;
    slot_of_pt[i] = -1;
    return;
  }
  unsigned long long h = mix64(key) & J.mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&keys[h], kEmpty, key);
    if (prev == kEmpty || prev == key) break;
    h = (h + 1) & J.mask;
  }
  slot_of_pt[i] = (int)h;
  urank[i] = (int)atomicAdd(&cnt[h], 1ull);
}

__global__ __launch_bounds__(256) void k_vox_scatter(size_t n, const int* __restrict__ slot_of_pt,
                                                     const int* __restrict__ urank,
                                                     const unsigned long long* __restrict__ off, int* __restrict__ members) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int h = slot_of_pt[i];
  if (h < 0) return;
  members[off[h] + (unsigned long long)urank[i]] = (int)i;
}

// rank of every point among its voxel's members in INDEX order (the order AddPoint is called in, :379-385);
// the first member is the voxel's leader
__global__ __launch_bounds__(256) void k_vox_order(size_t n, const int* __restrict__ slot_of_pt,
                                                   const unsigned long long* __restrict__ off,
                                                   const unsigned long long* __restrict__ cnt,
                                                   const int* __restrict__ members, int* __restrict__ sorted,
                                                   unsigned long long* __restrict__ leader) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  if (i == n) { leader[n] = 0ull; return; }  // scan terminator
  const int h = slot_of_pt[i];
  if (h < 0) { leader[i] = 0ull; return; }
  const unsigned long long o = off[h];
  const int m = (int)cnt[h];
  int r = 0;
  for (int q = 0; q < m; ++q) r += (members[o + q] < (int)i) ? 1 : 0;
  sorted[o + r] = (int)i;
  leader[i] = (r == 0) ? 1ull : 0ull;
}

// AccumulatedPoint: point_ += p in index order, GetAveragePoint = point_ / double(num) (:253-272); the voxels of
// segment 0 come first in first-occurrence order, so segment 1's positions are rebased by their count
__global__ __launch_bounds__(256) void k_vox_accumulate(VoxelJob J, VoxelWork W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long base1 = W.leader_scan[J.n0];
  if (i == 0) {  // numbers of voxels = sizes of the down-sampled clouds
    W.n_out[0] = base1;
    W.n_out[1] = W.leader_scan[J.n] - base1;
  }
  if (W.host_seg && i < 8) {  // ... and straight to the host (stream order: everything the host waits for precedes this kernel)
    unsigned long long w = i == 0 ? base1 : i == 1 ? W.leader_scan[J.n] - base1 : i == 2 ? (unsigned long long)W.overflow[0] : 0ull;
    // word 7 = sequence number XOR the payload words (tlh::wait_segment: a torn segment reads as "not there yet")
    unsigned long long x = w;
    x ^= __shfl_xor(x, 1, 64);
    x ^= __shfl_xor(x, 2, 64);
    x ^= __shfl_xor(x, 4, 64);
    if (i == 7) w = check_mix(W.host_seq) ^ x;
    __hip_atomic_store(&W.host_seg[i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (i >= J.n) return;
  const int h = W.slot_of_pt[i];
  if (h < 0) return;
  const unsigned long long pos = W.leader_scan[i];
  if (W.leader_scan[i + 1] == pos) return;  // not a leader
  const int seg = i >= J.n0 ? 1 : 0;
  const unsigned long long o = W.off[h];
  const int m = (int)W.cnt[h];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int q = 0; q < m; ++q) {
    const int j = W.sorted[o + q];
    sx += J.x[j]; sy += J.y[j]; sz += J.z[j];
  }
  const double dn = (double)m;
  const unsigned long long p = pos - (seg ? base1 : 0ull);
  W.out[seg][0][p] = sx / dn; W.out[seg][1][p] = sy / dn; W.out[seg][2][p] = sz / dn;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
}  // namespace

void launch_transform_to_soa(const double* aos, size_t n, const double M[16], double* ox, double* oy, double* oz,
                             hipStream_t s) {
  if (n == 0) return;
  Mat16 m;
  for (int i = 0; i < 16; ++i) m.m[i] = M[i];
  hipLaunchKernelGGL(k_transform_to_soa, dim3(blocks_for(n)), dim3(256), 0, s, aos, n, m, ox, oy, oz);
}
void launch_transform_to_soa2(const double* aos, size_t n, const double M[16], double* ax, double* ay, double* az,
                              double* bx, double* by, double* bz, hipStream_t s) {
  if (n == 0) return;
  Mat16 m;
  for (int i = 0; i < 16; ++i) m.m[i] = M[i];
  hipLaunchKernelGGL(k_transform_to_soa2, dim3(blocks_for(n)), dim3(256), 0, s, aos, n, m, ax, ay, az, bx, by, bz);
}
// Input assembly of the two-segment job in ONE launch (blockIdx.y = segment): [old submap | Transform(new scan)]
// per segment, written back to back into (wx, wy, wz)
__global__ void k_assemble(AssembleArgs A, double* __restrict__ wx, double* __restrict__ wy, double* __restrict__ wz) {
  const int s = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n_old = A.n_old[s], n_new = A.n_new[s];
  if (i >= n_old + n_new) return;
  const size_t o = A.base[s] + i;
  if (i < n_old) {  // *submap (kept as is)
    wx[o] = A.ox[s][i]; wy[o] = A.oy[s][i]; wz[o] = A.oz[s][i];
    return;
  }
  const size_t k = i - n_old;  // += scan->Transform(pose)
  const double* __restrict__ aos = A.aos[s];
  const double x = aos[3 * k], y = aos[3 * k + 1], z = aos[3 * k + 2];
  double r[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) r[a] = ((A.M[a] * x + A.M[4 + a] * y) + A.M[8 + a] * z) + A.M[12 + a] * 1.0;
  wx[o] = r[0] / r[3]; wy[o] = r[1] / r[3]; wz[o] = r[2] / r[3];
}
void launch_assemble(const AssembleArgs& A, double* wx, double* wy, double* wz, hipStream_t s) {
  const size_t nmax = std::max(A.n_old[0] + A.n_new[0], A.n_old[1] + A.n_new[1]);
  if (nmax == 0) return;
  hipLaunchKernelGGL(k_assemble, dim3(blocks_for(nmax), 2), dim3(256), 0, s, A, wx, wy, wz);
}
int transform_ring_max() { return kRingMax; }
void launch_transform_ring(int count, const double* const aos[], const size_t n[], const double* const poses[],
                           double* ax, double* ay, double* az, double* bx, double* by, double* bz, hipStream_t s) {
  RingArgs R;
  size_t off = 0, nmax = 0;
  for (int f = 0; f < count; ++f) {
    R.aos[f] = aos[f];
    R.n[f] = n[f];
    R.off[f] = off;
    for (int i = 0; i < 16; ++i) R.M[f].m[i] = poses[f][i];
    off += n[f];
    nmax = std::max(nmax, n[f]);
  }
  if (count == 0 || nmax == 0) return;
  hipLaunchKernelGGL(k_transform_ring, dim3(blocks_for(nmax), count), dim3(256), 0, s, R, ax, ay, az, bx, by, bz);
}
void launch_copy3(const double* ax, const double* ay, const double* az, size_t n, double* ox, double* oy, double* oz,
                  hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_copy3, dim3(blocks_for(n)), dim3(256), 0, s, ax, ay, az, n, ox, oy, oz);
}
void launch_soa_to_aos(const double* x, const double* y, const double* z, size_t n, double* aos, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n)), dim3(256), 0, s, x, y, z, n, aos);
}

size_t voxel_table_size(size_t n) {
  size_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

// Crop(box) -> VoxelDownSample(voxel) of the one or two SoA clouds in J, written to W.out; the output sizes land in
// W.n_out[0..1] (device).  No host synchronisation.
void launch_crop_voxel(const VoxelJob& J, const VoxelWork& W, hipStream_t s) {
  const size_t n = J.n;
  constexpr int kMinBlocks = 256;
  hipLaunchKernelGGL(k_vox_min, dim3(kMinBlocks), dim3(256), 0, s, J, W.min_partial, W.keys, W.cnt, W.overflow);
  hipLaunchKernelGGL(k_vox_min_final, dim3(1), dim3(384), 0, s, W.min_partial, kMinBlocks, J.voxel[0], J.voxel[1], W.vmin);
  const size_t cap = (size_t)J.mask + 1;
  if (n > 0)
    hipLaunchKernelGGL(k_vox_insert, dim3(blocks_for(n)), dim3(256), 0, s, J, W.vmin, W.keys, W.cnt, W.slot_of_pt,
                       W.urank, W.overflow);
  launch_exclusive_scan_u64(W.cnt, W.off, cap + 1, W.scan_tmp, s);
  if (n > 0)
    hipLaunchKernelGGL(k_vox_scatter, dim3(blocks_for(n)), dim3(256), 0, s, n, W.slot_of_pt, W.urank, W.off, W.members);
  hipLaunchKernelGGL(k_vox_order, dim3(blocks_for(n + 1)), dim3(256), 0, s, n, W.slot_of_pt, W.off, W.cnt, W.members,
                     W.sorted, W.leader);
  launch_exclusive_scan_u64(W.leader, W.leader_scan, n + 1, W.scan_tmp, s);
  hipLaunchKernelGGL(k_vox_accumulate, dim3(blocks_for(n + 1)), dim3(256), 0, s, J, W);
}

}  // namespace tl
