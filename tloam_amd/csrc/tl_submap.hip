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
#include <atomic>
#include <algorithm>
#include <string.h>

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
  int copy_frame;      // k_submap_front: the frame whose cloud is read from pinned HOST memory and kept in ...
  double* copy_dst;    // ... this device buffer for the updates to come (-1 / null: every frame is on the device already)
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

// ---- Crop + VoxelDownSample in THREE launches (round 4; eleven before: min, min-final, insert, a three-launch scan of the hash
// table, scatter, order, a three-launch scan of the leader flags, accumulate -- a chain of launch-latency-bound kernels) --------
//   k_vox_min2     GetMinBound() of the cropped clouds: block partials, the LAST block (ticket) finishes voxel_min_bound; the
//                  same launch empties the hash table and the control words of the job
//   k_vox_insert2  voxel of every in-box point -> hash slot, and the point is pushed on its voxel's member LIST (one atomic
//                  exchange on the slot's head) -- no counting, no offsets, hence no scan over the table
//   k_vox_emit     per point: walk the voxel's list -- the smallest index is the voxel's leader; the leader orders the members
//                  by index (in LDS: the order AddPoint is called in, :379-385), accumulates and averages; its output position
//                  -- the number of leaders in front, i.e. first-occurrence order -- comes from a single-pass scan over the
//                  blocks inside the same launch (a block publishes its count, then looks back over its predecessors' words)
// Same voxels, same member order, same sums, same output order as the eleven-launch form: tests/test_gpu_submap.py compares with
// the oracle bit for bit.
constexpr unsigned long long kNoHead = ~0ull;
constexpr int kVoxLocal = 32;           // members a leader orders in LDS; a fuller voxel is ordered in global scratch (heap sort)
constexpr unsigned long long kCntMask = (1ull << 30) - 1ull;   // look-back word: [0..29] leaders of segment 0, [32..61] of segment 1, [62..63] status
// VoxelWork::leader_scan, the control words of a job: [0] blocks of k_vox_emit started so far, [1] cursor of the sort scratch
// VoxelWork::overflow: [0] voxel index out of range, [1] ticket of k_vox_min2 (zero between launches)
// VoxelWork::cnt[h] = head of slot h's member list, VoxelWork::urank[i] = next member after point i (-1: none)
// VoxelWork::leader[b] = look-back word of block b of k_vox_emit
__global__ __launch_bounds__(256) void k_vox_min2(VoxelJob J, VoxelWork W, int emit_blocks) {
  __shared__ double sm[6][256];
  __shared__ int s_last;
  if (blockIdx.x == 0 && threadIdx.x == 0) { W.overflow[0] = 0; W.leader_scan[0] = 0ull; W.leader_scan[1] = 0ull; }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i <= J.mask; i += (size_t)gridDim.x * 256) {
    W.keys[i] = kEmpty;
    W.cnt[i] = kNoHead;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i <= (size_t)emit_blocks; i += (size_t)gridDim.x * 256) W.leader[i] = 0ull;
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
  // the row is handed over with device-scope stores, their completion is waited for, then the ticket (k3_take_ticket, tl_gn.hip)
  if (threadIdx.x < 6)
    __hip_atomic_store(W.min_partial + blockIdx.x * 6 + threadIdx.x, sm[threadIdx.x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = (__hip_atomic_fetch_add(W.overflow + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  // voxel_min_bound = GetMinBound() - voxel_size * 0.5 (:366); an empty cloud has min bound (0, 0, 0).  One wave per
  // (segment, axis): min is exact in any order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int a = wave; a < 6; a += 4) {
    double v = __builtin_inf();
    for (int b = lane; b < (int)gridDim.x; b += 64)
      v = fmin(v, __hip_atomic_load(W.min_partial + b * 6 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    if (!(v < __builtin_inf())) v = 0.0;
    if (lane == 0) W.vmin[a] = v - J.voxel[a < 3 ? 0 : 1] * 0.5;
  }
  if (threadIdx.x == 0) __hip_atomic_store(W.overflow + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
}

__global__ __launch_bounds__(256) void k_vox_insert2(VoxelJob J, VoxelWork W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= J.n) return;
  const double x = J.x[i], y = J.y[i], z = J.z[i];
  const int seg = i >= J.n0 ? 1 : 0;
  if (!in_box(J, seg, x, y, z)) { W.slot_of_pt[i] = -1; return; }
  // ref_coord = (p - voxel_min_bound) / voxel_size; index = int(floor(ref_coord))   (:380-383)
  const double voxel = J.voxel[seg];
  const long long ix = (long long)floor((x - W.vmin[3 * seg + 0]) / voxel);
  const long long iy = (long long)floor((y - W.vmin[3 * seg + 1]) / voxel);
  const long long iz = (long long)floor((z - W.vmin[3 * seg + 2]) / voxel);
  if (ix < 0 || iy < 0 || iz < 0 || ix >= (1ll << 21) || iy >= (1ll << 21) || iz >= (1ll << 21)) {
    W.overflow[0] = 1;  // "[VoxelDownSample] voxel_size is too small." (:370-372)
    W.slot_of_pt[i] = -1;
    return;
  }
  // 3 x 21 bits of voxel coordinates, bit 63 = segment.  The all-ones key (segment 1, all three indices 2^21 - 1)
  // is the empty marker: reported like an index out of range
  const unsigned long long key = (unsigned long long)ix | ((unsigned long long)iy << 21) | ((unsigned long long)iz << 42) |
                                 ((unsigned long long)seg << 63);
  if (key == kEmpty) {
    W.overflow[0] = 1;
    W.slot_of_pt[i] = -1;
    return;
  }
  unsigned long long h = mix64(key) & J.mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&W.keys[h], kEmpty, key);
    if (prev == kEmpty || prev == key) break;
    h = (h + 1) & J.mask;
  }
  W.slot_of_pt[i] = (int)h;
  W.urank[i] = (int)atomicExch(&W.cnt[h], (unsigned long long)i);   // next[i] = the old head (-1: none), head = i
}

__device__ __forceinline__ void heap_sift(int* a, int start, int end) {   // max-heap on a[0 .. end]
  int root = start;
  for (;;) {
    int child = 2 * root + 1;
    if (child > end) break;
    if (child + 1 <= end && a[child] < a[child + 1]) ++child;
    if (a[root] >= a[child]) break;
    const int t = a[root]; a[root] = a[child]; a[child] = t;
    root = child;
  }
}
__global__ __launch_bounds__(256) void k_vox_emit(VoxelJob J, VoxelWork W, int nblocks) {
  __shared__ int s_mem[kVoxLocal * 256];    // s_mem[k * 256 + t]: member k of thread t's voxel (conflict-free columns)
  __shared__ unsigned long long s_wave[4];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_bid;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A block only ever waits for blocks of LOWER place.  While the whole grid is resident on the device (the host checks:
  // vox_emit_resident_blocks) the place is the block index -- every lower block is running.  (A ticket taken at the start of
  // every block -- ~500 returning atomics on one word, served one after the other -- was a quarter of the launch: 0.0913-0.0921
  // against 0.0881-0.0886 ms per update, round 4.)  A larger grid, a partitioned or CU-masked device: places in the order the
  // blocks START (W.use_ticket), so that a block only ever waits for blocks that started before it.  The wait is bounded either way.
  int bid = (int)blockIdx.x;
  if (W.use_ticket) {
    if (tid == 0) s_bid = (int)atomicAdd(&W.leader_scan[0], 1ull);
    __syncthreads();
    bid = s_bid;
  }
  const size_t i = (size_t)bid * 256 + tid;
  const int h = i < J.n ? W.slot_of_pt[i] : -1;
  const int seg = i >= J.n0 ? 1 : 0;
  int m = 0;
  bool leader = false;
  if (h >= 0) {
    int first = 0x7fffffff;
    for (int j = (int)W.cnt[h]; j >= 0; j = W.urank[j]) {
      if (m < kVoxLocal) s_mem[m * 256 + tid] = j;
      first = j < first ? j : first;
      ++m;
    }
    leader = first == (int)i;
  }
  // AccumulatedPoint: point_ += p in index order, GetAveragePoint = point_ / double(num) (:253-272)
  double sx = 0.0, sy = 0.0, sz = 0.0;
  if (leader) {
    if (m <= kVoxLocal) {
      for (int a = 1; a < m; ++a) {   // insertion sort of this thread's column
        const int key = s_mem[a * 256 + tid];
        int b = a - 1;
        while (b >= 0 && s_mem[b * 256 + tid] > key) { s_mem[(b + 1) * 256 + tid] = s_mem[b * 256 + tid]; --b; }
        s_mem[(b + 1) * 256 + tid] = key;
      }
      for (int q = 0; q < m; ++q) {
        const int j = s_mem[q * 256 + tid];
        sx += J.x[j]; sy += J.y[j]; sz += J.z[j];
      }
    } else {
      // a crowded voxel: its members into a piece of the global scratch (only this thread ever touches it), heap sort, sum
      int* buf = W.members + atomicAdd(&W.leader_scan[1], (unsigned long long)m);
      int k = 0;
      for (int j = (int)W.cnt[h]; j >= 0; j = W.urank[j]) buf[k++] = j;
      for (int st = (m - 2) / 2; st >= 0; --st) heap_sift(buf, st, m - 1);
      for (int end = m - 1; end > 0; --end) {
        const int t = buf[0]; buf[0] = buf[end]; buf[end] = t;
        heap_sift(buf, 0, end - 1);
      }
      for (int q = 0; q < m; ++q) {
        const int j = buf[q];
        sx += J.x[j]; sy += J.y[j]; sz += J.z[j];
      }
    }
    const double dn = (double)m;
    sx /= dn; sy /= dn; sz /= dn;
  }
  // ---- the leader's output position: leaders of its segment in front of it.  Block-exclusive scan of the packed flags ...
  const unsigned long long flag = leader ? (seg ? (1ull << 32) : 1ull) : 0ull;
  unsigned long long incl = flag;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long wave_base = 0ull, block_total = 0ull;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) wave_base += s_wave[w];
    block_total += s_wave[w];
  }
  // ... and the blocks in front: publish this block's count, then look back (status 1: the block's own count, 2: the count of
  // everything up to and including the block)
  if (tid == 0) {
    unsigned long long prefix = 0ull;
    if (bid == 0) {
      __hip_atomic_store(&W.leader[0], (2ull << 62) | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(&W.leader[bid], (1ull << 62) | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      for (int p = bid - 1;;) {
        const unsigned long long w = __hip_atomic_load(&W.leader[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned st = (unsigned)(w >> 62);
        if (st == 0u) {
          if ((++spins & 63u) == 0 && wall_clock64() - t0 > 100000000ull) {   // ~1 s: a block in front never started
            if (W.fault) { __hip_atomic_store(W.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
            break;   // (the update's result is discarded by the host, tl_api_submap.hip)
          }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        prefix += w & ~(3ull << 62);
        if (st == 2u) break;
        --p;
      }
      __hip_atomic_store(&W.leader[bid], (2ull << 62) | (prefix + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_prefix = prefix;
  }
  __syncthreads();
  const unsigned long long before = s_prefix + wave_base + (incl - flag);
  if (leader) {
    const unsigned long long p = seg ? ((before >> 32) & kCntMask) : (before & kCntMask);
    W.out[seg][0][p] = sx; W.out[seg][1][p] = sy; W.out[seg][2][p] = sz;
  }
  if (bid == nblocks - 1 && tid < 8) {   // the block that holds the last point: the totals = the sizes of the down-sampled clouds
    const unsigned long long total = s_prefix + block_total;
    const unsigned long long n0_out = total & kCntMask, n1_out = (total >> 32) & kCntMask;
    if (tid == 0) { W.n_out[0] = n0_out; W.n_out[1] = n1_out; }
    if (W.host_seg) {  // ... and straight to the host: every leader's stores precede this block's in no particular order, so the
                       // host reads the clouds only through the stream (it waits for this word, then enqueues behind the launch)
      unsigned long long w = tid == 0 ? n0_out : tid == 1 ? n1_out : tid == 2 ? (unsigned long long)W.overflow[0] : 0ull;
      // word 7 = check_mix(sequence number) XOR seg_word of the payload words (tlh::wait_segment: a torn segment reads as "not there yet")
      unsigned long long x = tid < 7 ? seg_word(w, tid) : 0ull;
      x ^= __shfl_xor(x, 1, 64);
      x ^= __shfl_xor(x, 2, 64);
      x ^= __shfl_xor(x, 4, 64);
      if (tid == 7) w = check_mix(W.host_seq) ^ x;
      __hip_atomic_store(&W.host_seg[tid], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
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
// doubles [d0, d1) of `src` -> LDS, dst[d - (d0 & ~1)] = src[d]: one coalesced 16-byte load per lane and pass (src 16-byte
// aligned; up to one double in front of d0 and one behind d1 come along).  The new scan's clouds are read where the host left
// them -- pinned staging, across PCIe -- so every byte has to cross once: three 8-byte loads per lane at a 24-byte stride would
// fetch every line three times from memory the GPU does not cache.
__device__ __forceinline__ void stage_doubles(const double* __restrict__ src, size_t d0, size_t d1, double* __restrict__ dst) {
  const size_t p0 = d0 >> 1, p1 = (d1 + 1) >> 1;
  for (size_t p = p0 + threadIdx.x; p < p1; p += 256) {
    const double2 v = *reinterpret_cast<const double2*>(src + 2 * p);
    dst[2 * (p - p0)] = v.x;
    dst[2 * (p - p0) + 1] = v.y;
  }
}
// ---- the front of a submap update in ONE launch: the planar ring (blockIdx.y < ring frames: k_transform_ring's work), the
// assembly of [old submap | Transform(new scan)] of the edge and the ground cloud (the next two rows of blocks: k_assemble's
// work) and, by those same blocks on the points they have just written, what k_vox_min2 would do next: the hash table of the
// crop + voxel job emptied, the min bound of the cropped clouds as block partials, voxel_min_bound finished by the last block
// (ticket).  launch_crop_voxel then starts at the insert pass: an update is front | insert | emit.
__global__ __launch_bounds__(256) void k_submap_front(RingArgs R, int ring_count, AssembleArgs A, VoxelJob J, VoxelWork W, int emit_blocks,
                                                      double* __restrict__ px, double* __restrict__ py, double* __restrict__ pz,
                                                      double* __restrict__ qx, double* __restrict__ qy, double* __restrict__ qz,
                                                      double* __restrict__ wx, double* __restrict__ wy, double* __restrict__ wz) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  __shared__ double s_stage[3 * 256 + 4];
  if ((int)blockIdx.y < ring_count) {   // ---- planar / sphere submaps from the ring buffer (front_end.cpp:220-243)
    const int f = blockIdx.y;
    const size_t b0 = (size_t)blockIdx.x * 256, nf = R.n[f];
    if (b0 >= nf) return;   // (block-uniform)
    const size_t b1 = b0 + 256 < nf ? b0 + 256 : nf;
    stage_doubles(R.aos[f], 3 * b0, 3 * b1, s_stage);   // (3 * b0 is even: the block's first double is s_stage[0])
    __syncthreads();
    if (f == R.copy_frame) {   // the newest frame arrives in host memory: its cloud stays on the device for the frames it is buffered
      const size_t nd = 3 * (b1 - b0);
      for (size_t p = threadIdx.x; 2 * p < nd; p += 256)
        *reinterpret_cast<double2*>(R.copy_dst + 3 * b0 + 2 * p) = double2{s_stage[2 * p], s_stage[2 * p + 1]};
    }
    if (i >= nf) return;
    const Mat16& M = R.M[f];
    const double x = s_stage[3 * threadIdx.x], y = s_stage[3 * threadIdx.x + 1], z = s_stage[3 * threadIdx.x + 2];
    double r[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) r[a] = ((M.m[a] * x + M.m[4 + a] * y) + M.m[8 + a] * z) + M.m[12 + a] * 1.0;
    const double vx = r[0] / r[3], vy = r[1] / r[3], vz = r[2] / r[3];
    const size_t o = R.off[f] + i;
    px[o] = vx; py[o] = vy; pz[o] = vz;
    qx[o] = vx; qy[o] = vy; qz[o] = vz;
    return;
  }
  __shared__ double sm[3][256];
  __shared__ int s_last;
  const int s = (int)blockIdx.y - ring_count;            // segment: 0 edge, 1 ground
  const int lb = s * (int)gridDim.x + (int)blockIdx.x, nlb = 2 * (int)gridDim.x;   // this block among the assembling ones
  if (lb == 0 && threadIdx.x == 0) { W.overflow[0] = 0; W.leader_scan[0] = 0ull; W.leader_scan[1] = 0ull; }
  for (size_t t = (size_t)lb * 256 + threadIdx.x; t <= J.mask; t += (size_t)nlb * 256) {
    W.keys[t] = kEmpty;
    W.cnt[t] = kNoHead;
  }
  for (size_t t = (size_t)lb * 256 + threadIdx.x; t <= (size_t)emit_blocks; t += (size_t)nlb * 256) W.leader[t] = 0ull;
  double m[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
  const size_t n_old = A.n_old[s], n_new = A.n_new[s];
  // the points of the new scan this block transforms: [k0, k1) of the segment's scan cloud, through LDS
  const size_t blk0 = (size_t)blockIdx.x * 256, blk1 = blk0 + 256 < n_old + n_new ? blk0 + 256 : n_old + n_new;
  const size_t k0 = blk0 > n_old ? blk0 - n_old : 0, k1 = blk1 > n_old ? blk1 - n_old : 0;
  if (k1 > k0) stage_doubles(A.aos[s], 3 * k0, 3 * k1, s_stage);
  __syncthreads();
  if (i < n_old + n_new) {
    const size_t o = A.base[s] + i;
    double x, y, z;
    if (i < n_old) {  // *submap (kept as is)
      x = A.ox[s][i]; y = A.oy[s][i]; z = A.oz[s][i];
    } else {          // += scan->Transform(pose)
      const size_t k = i - n_old;
      const double* a = s_stage + (3 * k - ((3 * k0) & ~(size_t)1));
      const double ax = a[0], ay = a[1], az = a[2];
      double r[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) r[a] = ((A.M[a] * ax + A.M[4 + a] * ay) + A.M[8 + a] * az) + A.M[12 + a] * 1.0;
      x = r[0] / r[3]; y = r[1] / r[3]; z = r[2] / r[3];
    }
    wx[o] = x; wy[o] = y; wz[o] = z;
    if (in_box(J, s, x, y, z)) { m[0] = x; m[1] = y; m[2] = z; }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) sm[a][threadIdx.x] = m[a];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int a = 0; a < 3; ++a) sm[a][threadIdx.x] = fmin(sm[a][threadIdx.x], sm[a][threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x < 6) {   // this block's row: its segment's three columns, +inf in the other segment's
    const int a = threadIdx.x;
    const double v = (a / 3 == s) ? sm[a % 3][0] : __builtin_inf();
    __hip_atomic_store(W.min_partial + (size_t)lb * 6 + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = (__hip_atomic_fetch_add(W.overflow + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nlb - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int a = wave; a < 6; a += 4) {   // voxel_min_bound = GetMinBound() - voxel_size * 0.5 (:366); empty cloud: (0, 0, 0)
    double v = __builtin_inf();
    for (int b = lane; b < nlb; b += 64)
      v = fmin(v, __hip_atomic_load(W.min_partial + (size_t)b * 6 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    if (!(v < __builtin_inf())) v = 0.0;
    if (lane == 0) W.vmin[a] = v - J.voxel[a < 3 ? 0 : 1] * 0.5;
  }
  if (threadIdx.x == 0) __hip_atomic_store(W.overflow + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
}
// rows of VoxelWork::min_partial the front launch needs (6 doubles each)
size_t submap_front_rows(size_t n_ring_max, size_t n_seg_max) { return 2 * (size_t)blocks_for(std::max<size_t>(std::max(n_ring_max, n_seg_max), 1)); }
void launch_submap_front(int count, const double* const aos[], const size_t n[], const double* const poses[], const AssembleArgs& A,
                         const VoxelJob& J, const VoxelWork& W, double* px, double* py, double* pz, double* qx, double* qy, double* qz,
                         double* wx, double* wy, double* wz, hipStream_t s, int copy_frame, double* copy_dst) {
  RingArgs R;
  memset(&R, 0, sizeof(R));
  R.copy_frame = copy_dst ? copy_frame : -1;
  R.copy_dst = copy_dst;
  size_t off = 0, nmax = 1;
  for (int f = 0; f < count; ++f) {
    R.aos[f] = aos[f];
    R.n[f] = n[f];
    R.off[f] = off;
    for (int i = 0; i < 16; ++i) R.M[f].m[i] = poses[f][i];
    off += n[f];
    nmax = std::max(nmax, n[f]);
  }
  nmax = std::max(nmax, std::max(A.n_old[0] + A.n_new[0], A.n_old[1] + A.n_new[1]));
  const int emit_blocks = (int)blocks_for(J.n + 1);
  hipLaunchKernelGGL(k_submap_front, dim3(blocks_for(nmax), count + 2), dim3(256), 0, s, R, count, A, J, W, emit_blocks, px, py, pz, qx, qy,
                     qz, wx, wy, wz);
}
int transform_ring_max() { return kRingMax; }
void launch_transform_ring(int count, const double* const aos[], const size_t n[], const double* const poses[],
                           double* ax, double* ay, double* az, double* bx, double* by, double* bz, hipStream_t s) {
  RingArgs R;
  memset(&R, 0, sizeof(R));
  R.copy_frame = -1;
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
// n doubles (n even, both ends 16-byte aligned) from pinned host memory the device can address to device memory: what a
// hipMemcpyAsync would do, as a kernel -- for a few hundred KB the copy command costs the calling thread and the copy engine
// more than a launch that reads across PCIe with every load in flight at once
__global__ __launch_bounds__(256) void k_blit_pairs(const double2* __restrict__ src, double2* __restrict__ dst, size_t pairs) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) dst[p] = src[p];
}
void launch_blit_doubles(const double* src_host_view, double* dst, size_t n, hipStream_t s) {
  const size_t pairs = (n + 1) / 2;
  if (pairs == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((pairs + 255) / 256, 1024);
  hipLaunchKernelGGL(k_blit_pairs, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const double2*>(src_host_view),
                     reinterpret_cast<double2*>(dst), pairs);
}
void launch_soa_to_aos(const double* x, const double* y, const double* z, size_t n, double* aos, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_soa_to_aos, dim3(blocks_for(n)), dim3(256), 0, s, x, y, z, n, aos);
}

int vox_emit_resident_blocks(int device_cus) {
  // (one kernel, one architecture: the same for every gfx950 device of the process; contexts are created from several host
  //  threads -- an atomic, and two threads that both find it unset both store the same value)
  static std::atomic<int> per_cu_cache{-1};
  int per_cu = per_cu_cache.load(std::memory_order_relaxed);
  if (per_cu < 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_vox_emit, 256, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
    per_cu = occ;
    per_cu_cache.store(occ, std::memory_order_relaxed);
  }
  const long long all = (long long)device_cus * per_cu;
  return (int)(all - all / 16);   // with room to spare for whatever else is on the device
}
size_t voxel_table_size(size_t n) {
  size_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

// Crop(box) -> VoxelDownSample(voxel) of the one or two SoA clouds in J, written to W.out; the output sizes land in
// W.n_out[0..1] (device).  No host synchronisation.
// front_done: the table has been emptied and voxel_min_bound finished by k_submap_front
void launch_crop_voxel(const VoxelJob& J, const VoxelWork& W, hipStream_t s, bool front_done) {
  const size_t n = J.n;
  constexpr int kMinBlocks = 256;
  const int emit_blocks = (int)blocks_for(n + 1);   // (n + 1: an empty job still has a block that reports sizes of 0)
  if (!front_done) hipLaunchKernelGGL(k_vox_min2, dim3(kMinBlocks), dim3(256), 0, s, J, W, emit_blocks);
  if (n > 0) hipLaunchKernelGGL(k_vox_insert2, dim3(blocks_for(n)), dim3(256), 0, s, J, W);
  hipLaunchKernelGGL(k_vox_emit, dim3(emit_blocks), dim3(256), 0, s, J, W, emit_blocks);
}

}  // namespace tl
