// tl_nn.hip -- K1 (exact hybrid nearest-neighbour search) + K2 (correspondence builders) for gfx950.
//
// Replaces, on the device:
//   KDTreeFlann::SetGeometry x4            registration.cpp:889-915  -> uniform-grid build
//   KDTreeFlann::SearchHybrid call sites   registration.cpp:444 (edge r=1.0 k=5), :535 (sphere r=0.5
//                                          k=1), :588 (planar r=0.5 k=5), :731 (ground), :272 (fitness)
//   addEdgeCostFactor :427-505, addSphereCostFactor :517-559, addSurfCostFactor :571-635,
//   addGroundCostFactor :714-778, fitBestPlane :303-368
//
// Search structure: a dense uniform grid with cell >= radius*(1+1e-6), built per frame by
// histogram -> exclusive scan -> scatter.  Every target with squared distance < radius^2 lies in the
// 27-cell neighbourhood of the query's cell, so "k nearest, then cut at radius^2" (Open3D 0.12
// SearchHybrid) over that neighbourhood is EXACT.  Ties in distance are broken
// towards the lower original target index, which makes the result independent of the (atomic)
// scatter order inside a cell.
//
// This translation unit is compiled with -ffp-contract=off: the discontinuous gates (radius cut,
// eig[2] > 3 eig[1], |dir.z| > 0.85, plane validity > 0.2, knn_dist > 0.2) are evaluated with
// the same un-fused fp64 operation order as the CPU oracle, so a correspondence flips in/out on
// the device only where it flips on the host.
#include <string.h>
#include <algorithm>
#include <atomic>

#include "tl_common.hpp"
#include "tl_knn.hpp"
#include "tl_finish.hpp"

namespace tl {

// ================================================================================================
//  small utility kernels
// ================================================================================================
__global__ void k_aos_to_soa(const double* __restrict__ aos, size_t n, double* __restrict__ x,
                             double* __restrict__ y, double* __restrict__ z) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    x[i] = aos[3 * i];
    y[i] = aos[3 * i + 1];
    z[i] = aos[3 * i + 2];
  }
}
void launch_aos_to_soa(const double* aos, size_t n, double* x, double* y, double* z, hipStream_t s) {
  if (n == 0) return;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_aos_to_soa, dim3(blocks), dim3(256), 0, s, aos, n, x, y, z);
}

// ================================================================================================
//  exclusive scan of u64 (block-local scan + recursive scan of block totals + add)
// ================================================================================================
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;   // 2048 elements per block, 512 per wave
constexpr int kScanPer = 4;                            // consecutive elements per lane within a row
constexpr int kScanRowElems = 64 * kScanPer;           // 256 elements = 2 KiB per wave row
constexpr int kScanRows = kScanTile / (kScanThreads / 64) / kScanRowElems;   // rows per wave (2)

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2), aligned(8)));   // (8-byte aligned: any element of a u64 array)
// Wave-level exclusive scan of ROWS x 256 consecutive elements starting at `wbase`: lane l holds the four
// elements 4l .. 4l+3 of each row, so a wave instruction covers 2 KiB (16 cache lines) and one shuffle scan
// serves 256 elements.  The fully blocked layout (8-16 consecutive elements per thread) made every wave
// instruction touch 32-64 different lines, serialised by the texture addresser: 11.9 us for a 9.4 k-element
// scan.  Returns the wave total; out[row i, 4l + j] = ex[i] + a[i][0] + .. + a[i][j-1].
template <int ROWS>
__device__ __forceinline__ unsigned long long wave_scan_rows(const unsigned long long* __restrict__ in, size_t n,
                                                             size_t wbase, int lane, unsigned long long (&ex)[ROWS],
                                                             unsigned long long (&a)[ROWS][kScanPer]) {
  static_assert(kScanPer == 4, "a lane's elements of a row are loaded as two 16-byte halves");
  if (wbase + (size_t)ROWS * kScanRowElems <= n) {
    // a wave whose rows all lie inside the array (every wave but the last): a lane's four elements as TWO 16-byte loads -- what the
    // CU's L1 is asked for is accesses per lane, not bytes (64 eight-byte loads per thread of a 16 Ki tile were half of what the
    // single-pass scans waited for)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const u64x2* __restrict__ p = reinterpret_cast<const u64x2*>(in + wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer);
      const u64x2 lo = p[0], hi = p[1];
      a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = hi.x; a[i][3] = hi.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
      for (int j = 0; j < kScanPer; ++j) {  // unconditional (clamped) loads: all in flight together
        const size_t idx = wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer + j;
        const unsigned long long x = in[idx < n ? idx : n - 1];
        a[i][j] = idx < n ? x : 0ull;
      }
  }
  unsigned long long carry = 0;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) s += a[i][j];
    unsigned long long inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    ex[i] = carry + inc - s;
    carry += __shfl(inc, 63, 64);
  }
  return carry;
}
template <int ROWS>
__device__ __forceinline__ void wave_store_rows(unsigned long long* __restrict__ out, size_t n, size_t wbase, int lane,
                                                unsigned long long add, const unsigned long long (&ex)[ROWS],
                                                const unsigned long long (&a)[ROWS][kScanPer]) {
  if (wbase + (size_t)ROWS * kScanRowElems <= n) {   // (as the loads: two 16-byte stores per lane and row)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const unsigned long long r0 = add + ex[i], r1 = r0 + a[i][0], r2 = r1 + a[i][1], r3 = r2 + a[i][2];
      u64x2* __restrict__ p = reinterpret_cast<u64x2*>(out + wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer);
      p[0] = u64x2{r0, r1};
      p[1] = u64x2{r2, r3};
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    unsigned long long run = add + ex[i];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
      const size_t idx = wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer + j;
      if (idx < n) out[idx] = run;
      run += a[i][j];
    }
  }
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile(const unsigned long long* __restrict__ in,
                                                            unsigned long long* __restrict__ out, size_t n,
                                                            unsigned long long* __restrict__ tile_total,
                                                            const int* __restrict__ gate) {
  __shared__ unsigned long long wave_tot[kScanThreads / 64];
  if (gate && *gate == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wbase = (size_t)blockIdx.x * kScanTile + (size_t)wave * (kScanRows * kScanRowElems);
  unsigned long long ex[kScanRows], a[kScanRows][kScanPer];
  const unsigned long long tot = wave_scan_rows<kScanRows>(in, n, wbase, lane, ex, a);
  if (lane == 0) wave_tot[wave] = tot;
  __syncthreads();
  unsigned long long wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  wave_store_rows<kScanRows>(out, n, wbase, lane, wave_off, ex, a);
  if (threadIdx.x == kScanThreads - 1) tile_total[blockIdx.x] = wave_off + tot;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_add(unsigned long long* __restrict__ out, size_t n,
                                                           const unsigned long long* __restrict__ tile_off,
                                                           const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const unsigned long long add = tile_off[blockIdx.x];
  const size_t base = (size_t)blockIdx.x * kScanTile + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + (size_t)i * kScanThreads < n) out[base + (size_t)i * kScanThreads] += add;
}
// small inputs (<= 16 Ki elements, e.g. the flag scan of a KITTI-size frame): ONE block of 1024 threads, a
// single pass with every load in flight at once
constexpr int kSmallThreads = 1024;
constexpr int kSmallRows = 4;                                   // 1024 elements per wave
constexpr int kSmallTile = (kSmallThreads / 64) * kSmallRows * kScanRowElems;  // 16384
__global__ __launch_bounds__(kSmallThreads) void k_scan_small(const unsigned long long* __restrict__ in,
                                                              unsigned long long* __restrict__ out, size_t n,
                                                              const int* __restrict__ gate) {
  __shared__ unsigned long long wave_tot[kSmallThreads / 64];
  __shared__ unsigned long long carry_s;
  if (gate && *gate == 0) return;
  if (threadIdx.x == 0) carry_s = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t tile0 = 0; tile0 < n; tile0 += kSmallTile) {
    const size_t wbase = tile0 + (size_t)wave * (kSmallRows * kScanRowElems);
    unsigned long long ex[kSmallRows], a[kSmallRows][kScanPer];
    const unsigned long long tot = wave_scan_rows<kSmallRows>(in, n, wbase, lane, ex, a);
    if (lane == 0) wave_tot[wave] = tot;
    __syncthreads();
    unsigned long long wave_off = carry_s;
    for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
    wave_store_rows<kSmallRows>(out, n, wbase, lane, wave_off, ex, a);
    __syncthreads();
    if (threadIdx.x == kSmallThreads - 1) carry_s = wave_off + tot;
    __syncthreads();
  }
}
// medium inputs: every block sums the totals of the tiles before it itself (<= 1024 loads) -- two
// launches instead of three
__global__ __launch_bounds__(kScanThreads) void k_scan_add_direct(unsigned long long* __restrict__ out, size_t n,
                                                                  const unsigned long long* __restrict__ tile_total,
                                                                  const int* __restrict__ gate) {
  __shared__ unsigned long long red[kScanThreads / 64];
  if (gate && *gate == 0) return;
  unsigned long long acc = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += kScanThreads) acc += tile_total[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  unsigned long long add = 0;
  for (int w = 0; w < kScanThreads / 64; ++w) add += red[w];
  const size_t base = (size_t)blockIdx.x * kScanTile + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + (size_t)i * kScanThreads < n) out[base + (size_t)i * kScanThreads] += add;
}
size_t scan_tmp_elems(size_t n) {
  size_t total = 0;
  while (n > 1) {
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    total += 2 * tiles;
    if (tiles == 1) break;
    n = tiles;
  }
  return total + 16;
}
void launch_exclusive_scan_u64(const unsigned long long* in, unsigned long long* out, size_t n,
                               unsigned long long* tmp, hipStream_t s, const int* gate) {
  if (n == 0) return;
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  if (n <= (size_t)kSmallTile) {
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(kSmallThreads), 0, s, in, out, n, gate);
    return;
  }
  unsigned long long* totals = tmp;
  unsigned long long* totals_scan = tmp + tiles;
  hipLaunchKernelGGL(k_scan_tile, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, out, n, totals, gate);
  if (tiles <= 1024) {
    hipLaunchKernelGGL(k_scan_add_direct, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, out, n, totals, gate);
  } else {
    launch_exclusive_scan_u64(totals, totals_scan, tiles, tmp + 2 * tiles, s, gate);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, out, n, totals_scan, gate);
  }
}

// ---- single-pass scan of large COUNT arrays (round 4) --------------------------------------------------------------------------
// The 1 M-class frames scan two histograms of ~3 M entries per frame (cells of the target grids, bins of the query sort): tile
// scan + scan of the tile totals + add = three launches and two passes over the array.  Here ONE launch, one pass: a block
// scans its tile, publishes the tile's total in a status word, looks back over the words of the blocks in front of it (a wave
// reads 64 of them at a time) until it meets one that already carries its inclusive prefix, publishes its own, and emits its
// elements with the prefix added.  At most one block per CU OF THIS DEVICE (scan_1p_applies looks at the context's CU count: a
// partitioned or CU-masked part takes the multi-launch scan), so every block is resident at once and a block only ever waits
// for blocks that are running; the wait is bounded all the same (~1 s of the wall clock, then the block raises Scan1p::fault --
// a word in pinned host memory -- and goes on with what it has: the host discards the frame's result, switches the context to
// the multi-launch scans and runs the frame again, tl_api.hip check_device_faults).  Status word: [63:62] 0 nothing |
// 1 the tile's own total | 2 inclusive prefix, [61:32] the launch's epoch -- a word of an earlier launch reads as "nothing",
// so the array is never cleared (it is zeroed when it is allocated) --, [31:0] the value: totals below 2^32, i.e. counts.
// Tiles of 16 Ki elements (64 per thread, in registers): ~200 blocks for 3.2 M entries, all resident at once, so that the
// look-back is three or four reads of 64 status words -- with 2 Ki tiles (1563 blocks, not all resident) the prefix crept from
// block to block at a coherence-point round trip per hop and the launch took 52-57 us, longer than the three launches it replaced.
constexpr int kRows1p = 16;                                                  // rows of 256 elements per wave
constexpr int kTile1p = (kScanThreads / 64) * kRows1p * kScanRowElems;       // 16384 elements per block
struct Scan1p {
  unsigned long long* status;   // [tiles]
  unsigned epoch;               // 1 .. 2^30 - 1, different from the previous launch's on the same status array
  unsigned* fault;              // pinned host word raised when the look-back timed out (null: nobody to tell)
};
// returns the block's place (tile index); ex / a as wave_scan_rows leaves them, *base = sum of everything in front of this wave
__device__ __forceinline__ int scan1p_tile(const unsigned long long* __restrict__ in, size_t n, const Scan1p& C,
                                           unsigned long long (&ex)[kRows1p], unsigned long long (&a)[kRows1p][kScanPer],
                                           unsigned long long* base, size_t* wbase_out) {
  __shared__ unsigned long long wave_tot[kScanThreads / 64];
  __shared__ unsigned long long s_prefix;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The block's place is its index: the launcher only uses this kernel when the grid is at most one block per CU of the
  // device in use (scan_1p_applies), so every block is resident as soon as the device has room for it and nobody waits for a
  // block that cannot start.  (Places handed out by an atomic counter -- the usual guard -- cost ~10 us here: ~200 returning atomics on
  // one address are served one after the other at the memory side, ~50 ns each.)
  const int bid = (int)blockIdx.x;
  const size_t wbase = (size_t)bid * kTile1p + (size_t)wave * (kRows1p * kScanRowElems);
  const unsigned long long tot = wave_scan_rows<kRows1p>(in, n, wbase, lane, ex, a);
  if (lane == 0) wave_tot[wave] = tot;
  __syncthreads();
  unsigned long long wave_off = 0, block_total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    if (w < wave) wave_off += wave_tot[w];
    block_total += wave_tot[w];
  }
  const unsigned long long tag = (unsigned long long)C.epoch << 32;
  if (wave == 0) {
    unsigned long long prefix = 0ull;
    if (bid == 0) {
      if (lane == 0) __hip_atomic_store(&C.status[0], (2ull << 62) | tag | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&C.status[bid], (1ull << 62) | tag | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      for (int top = bid - 1;;) {   // lanes look at blocks top, top - 1, ..., top - 63
        const int p = top - lane;
        unsigned long long w = 0ull;
        if (p >= 0) w = __hip_atomic_load(&C.status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool mine = p >= 0 && ((w >> 32) & 0x3fffffffull) == (unsigned long long)C.epoch;
        const unsigned st = mine ? (unsigned)(w >> 62) : 0u;
        const unsigned long long have = __ballot(p < 0 || st != 0u);          // (lanes past block 0 count as ready, value 0)
        const unsigned long long full = __ballot(p >= 0 && st == 2u);
        // the nearest block that carries an inclusive prefix, provided every block nearer than it has published its total
        const int first_full = full ? __ffsll((long long)full) - 1 : 64;
        const unsigned long long need = first_full >= 63 ? ~0ull : ((2ull << first_full) - 1ull);
        if ((have & need) != need) {   // somebody in the window is not there yet
          if ((++spins & 63u) == 0 && wall_clock64() - t0 > 100000000ull) {   // ~1 s: a block in front never started
            if (lane == 0 && C.fault) { __hip_atomic_store(C.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        unsigned long long v = (p >= 0 && lane <= first_full) ? (w & 0xffffffffull) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        prefix += __shfl(v, 0, 64);
        if (first_full < 64 || top - 64 < 0) break;
        top -= 64;
      }
      if (lane == 0)
        __hip_atomic_store(&C.status[bid], (2ull << 62) | tag | (prefix + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = prefix;
  }
  __syncthreads();
  *base = s_prefix + wave_off;
  *wbase_out = wbase;
  return bid;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_1p(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                          size_t n, Scan1p C) {
  unsigned long long ex[kRows1p], a[kRows1p][kScanPer], base;
  size_t wbase;
  (void)scan1p_tile(in, n, C, ex, a, &base, &wbase);
  wave_store_rows<kRows1p>(out, n, wbase, threadIdx.x & 63, base, ex, a);
}
// the scan of the concatenated cell histogram AND what k_grid_finalize_all does with it: cell_start of every kind (relative to the
// kind's own point block, with its terminator), the histogram left empty for the next build.  No cell_scan array is written.
__global__ __launch_bounds__(kScanThreads) void k_grid_scan_finalize_1p(GridSet gs, unsigned long long* __restrict__ cell_cnt, size_t n,
                                                                        int* __restrict__ cell_start, Scan1p C) {
  unsigned long long ex[kRows1p], a[kRows1p][kScanPer], base;
  size_t wbase;
  const int bid = scan1p_tile(cell_cnt, n, C, ex, a, &base, &wbase);
  const int lane = threadIdx.x & 63;
  if (bid == 0 && threadIdx.x < kKinds && gs.ncell[threadIdx.x] == 0)   // a kind without a grid: just its terminator
    cell_start[gs.cell_base[threadIdx.x] + threadIdx.x] = gs.n[threadIdx.x];
  typedef int int4s __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int i = 0; i < kRows1p; ++i) {
    unsigned long long run = base + ex[i];
    {
      // the common case -- the lane's four cells lie inside ONE kind's table and its last cell is not among them: one 16-byte store
      // of the four starts, two of the zeroes (the general form below is four conditional stores per cell)
      const long long e0 = (long long)(wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer);
      int kf = -1;
#pragma unroll
      for (int k = 0; k < kKinds; ++k)
        if (e0 >= gs.cell_base[k] && e0 + (kScanPer - 1) < gs.cell_base[k] + gs.ncell[k] - 1) kf = k;
      if (kf >= 0 && (size_t)e0 + kScanPer <= n) {
        const unsigned long long off = (unsigned long long)gs.tgt_off[kf];
        const unsigned long long r0 = run - off, r1 = r0 + a[i][0], r2 = r1 + a[i][1], r3 = r2 + a[i][2];
        *reinterpret_cast<int4s*>(cell_start + e0 + kf) = int4s{(int)r0, (int)r1, (int)r2, (int)r3};
        u64x2* __restrict__ z = reinterpret_cast<u64x2*>(cell_cnt + e0);
        z[0] = u64x2{0ull, 0ull};
        z[1] = u64x2{0ull, 0ull};
        continue;
      }
    }
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
      const long long e = (long long)(wbase + (size_t)i * kScanRowElems + (size_t)lane * kScanPer + j);
      if ((size_t)e < n) {
#pragma unroll
        for (int k = 0; k < kKinds; ++k) {
          const long long c = e - gs.cell_base[k];
          if (c >= 0 && c < gs.ncell[k]) {
            cell_start[e + k] = (int)(run - (unsigned long long)gs.tgt_off[k]);
            if (c == gs.ncell[k] - 1) cell_start[e + k + 1] = gs.n[k];
          }
        }
        cell_cnt[e] = 0ull;
      }
      run += a[i][j];
    }
  }
}
// scatter of the one-pass build: the cell's first position comes from cell_start (what the scan wrote), not from a scan array
__global__ void k_grid_scatter_start_all(GridSet gs, const int* __restrict__ cell_of_pt, const int* __restrict__ cell_start,
                                         const int* __restrict__ rank_of_pt, double4* __restrict__ gp) {
  const int k = blockIdx.y;
  const int n = gs.n[k];
  const long long base = gs.cell_base[k] + k;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = cell_of_pt[gs.tgt_off[k] + i];
    const int pos = cell_start[base + c] + rank_of_pt[gs.tgt_off[k] + i];
    gp[gs.tgt_off[k] + pos] =
        double4{gs.tx[k][i], gs.ty[k][i], gs.tz[k][i], __longlong_as_double((long long)i)};
  }
}
static unsigned next_scan_epoch() {   // process-wide (contexts on different host threads share it): any two launches differ
  static std::atomic<unsigned> e{0};
  return e.fetch_add(1u, std::memory_order_relaxed) % 0x3ffffffeu + 1u;
}
size_t scan_1p_ctl_elems(size_t n) { return (n + kTile1p - 1) / kTile1p + 8; }
// arrays of more than 1024 small tiles (the 1 M-class frames; smaller ones use the fused / two-launch forms) whose 16 Ki tiles
// number at most the device's CUs: k_scan_1p holds its 64 elements per thread in registers (256 VGPRs + 76 AGPRs: one wave per
// SIMD, i.e. ONE block per CU), so a grid of up to one block per CU is resident at once on an otherwise idle device -- and on a
// busy one a block still only waits for blocks of lower index, which the dispatcher started earlier (a CPX partition or a
// CU-masked device reports fewer CUs and gets the multi-launch scan; round 4 stopped at 240 tiles).  The wait is bounded either way.
bool scan_1p_applies(size_t n, int device_cus) {
  // (a sixteenth of the CUs to spare for whatever else runs on the device -- another context's stream, another rank: the look-back
  //  waits for blocks that have to be running; as vox_emit_resident_blocks)
  return (n + kScanTile - 1) / kScanTile > 1024 && (long long)((n + kTile1p - 1) / kTile1p) <= (long long)(device_cus - device_cus / 16);
}
void launch_scan_counts_1p(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* ctl, unsigned* fault,
                           hipStream_t s) {
  const size_t tiles = (n + kTile1p - 1) / kTile1p;
  Scan1p C{ctl, next_scan_epoch(), fault};
  hipLaunchKernelGGL(k_scan_1p, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, out, n, C);
}

// first half of the medium-size scan only: tile-local exclusive scans + per-tile totals (the consumer adds the tiles'
// offsets itself, see k_grid_finalize_scatter_all).  Returns the number of tiles; 0 = not applicable (use the full scan).
int scan_tiles_only(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* totals, hipStream_t s,
                    const int* gate) {
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  if (n <= (size_t)kSmallTile || tiles > 1024) return 0;
  hipLaunchKernelGGL(k_scan_tile, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, in, out, n, totals, gate);
  return (int)tiles;
}

// ================================================================================================
//  grid build
// ================================================================================================
// per-block min/max of every kind's cloud (blockIdx.y = kind); the host finishes over 64 rows of 6 (`out` may
// be pinned host memory: the rows then need no copy kernel)
__global__ __launch_bounds__(256) void k_bbox_all(GridSet gs, double* __restrict__ out) {
  __shared__ double red[4][6];
  const int k = blockIdx.y;
  const double* __restrict__ x = gs.tx[k];
  const double* __restrict__ y = gs.ty[k];
  const double* __restrict__ z = gs.tz[k];
  const int n = gs.n[k];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    // only FINITE coordinates extend the box: fmin / fmax already pass over a NaN, an infinite one would make the box -- and
    // with it the cell size -- infinite.  Such a point still gets a (boundary) cell and is simply never anybody's neighbour:
    // its distance to every query is inf or NaN, as in the reference's kd-tree (tests/test_gpu_parity.py).
    double a = x[i], b = y[i], c = z[i];
    if (!(fabs(a) < __builtin_inf())) a = __builtin_nan("");
    if (!(fabs(b) < __builtin_inf())) b = __builtin_nan("");
    if (!(fabs(c) < __builtin_inf())) c = __builtin_nan("");
    lo[0] = fmin(lo[0], a); hi[0] = fmax(hi[0], a);
    lo[1] = fmin(lo[1], b); hi[1] = fmax(hi[1], b);
    lo[2] = fmin(lo[2], c); hi[2] = fmax(hi[2], c);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v = (threadIdx.x < 3) ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
    out[((size_t)k * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = v;
  }
}
// setInputTarget of a whole Frame: AoS -> SoA of the four clouds AND their bounding-box rows in ONE launch (four conversion
// launches and k_bbox_all otherwise).  Same rows as k_bbox_all: 64 per kind, finite coordinates only.
__global__ __launch_bounds__(256) void k_ingest_targets(IngestArgs A, double* __restrict__ out) {
  __shared__ double red[4][6];
  const int k = blockIdx.y;
  const double* __restrict__ aos = A.aos[k];
  double* __restrict__ x = A.x[k];
  double* __restrict__ y = A.y[k];
  double* __restrict__ z = A.z[k];
  const int n = A.n[k];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double a = aos[3 * (size_t)i], b = aos[3 * (size_t)i + 1], c = aos[3 * (size_t)i + 2];
    x[i] = a; y[i] = b; z[i] = c;
    if (!(fabs(a) < __builtin_inf())) a = __builtin_nan("");
    if (!(fabs(b) < __builtin_inf())) b = __builtin_nan("");
    if (!(fabs(c) < __builtin_inf())) c = __builtin_nan("");
    lo[0] = fmin(lo[0], a); hi[0] = fmax(hi[0], a);
    lo[1] = fmin(lo[1], b); hi[1] = fmax(hi[1], b);
    lo[2] = fmin(lo[2], c); hi[2] = fmax(hi[2], c);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v = (threadIdx.x < 3) ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
    out[((size_t)k * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = v;
  }
}
void launch_ingest_targets(const IngestArgs& A, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_ingest_targets, dim3(64, kKinds), dim3(256), 0, s, A, out);
}
void launch_bbox_all(const GridSet& gs, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_bbox_all, dim3(64, kKinds), dim3(256), 0, s, gs, out);
}


// (y == kKinds: the rows of blocks that carry the start of the scan_match, see FrameInitHook)
__global__ __launch_bounds__(256) void k_grid_count_all(GridSet gs, unsigned long long* __restrict__ cell_cnt,
                                                        int* __restrict__ cell_of_pt, int* __restrict__ rank_of_pt,
                                                        FrameInit fi, FrameInitBufs fb) {
  const int k = blockIdx.y;
  if (k == kKinds) {
    frame_init_body(fi, fb, (int)blockIdx.x, (int)gridDim.x);
    return;
  }
  const int n = gs.n[k];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int cx = clampi(cell_coord(gs.tx[k][i], gs.org[k][0], gs.inv_cell[k], gs.dim[k][0]), 0, gs.dim[k][0] - 1);
    const int cy = clampi(cell_coord(gs.ty[k][i], gs.org[k][1], gs.inv_cell[k], gs.dim[k][1]), 0, gs.dim[k][1] - 1);
    const int cz = clampi(cell_coord(gs.tz[k][i], gs.org[k][2], gs.inv_cell[k], gs.dim[k][2]), 0, gs.dim[k][2] - 1);
    const int c = (cz * gs.dim[k][1] + cy) * gs.dim[k][0] + cx;
    cell_of_pt[gs.tgt_off[k] + i] = c;
    // the one atomic of the build: the histogram count doubles as the point's rank inside its cell
    rank_of_pt[gs.tgt_off[k] + i] = (int)atomicAdd(&cell_cnt[gs.cell_base[k] + c], 1ull);
  }
}
static int max_n(const GridSet& gs) {
  int m = 1;
  for (int k = 0; k < kKinds; ++k) m = std::max(m, gs.n[k]);
  return m;
}
void launch_grid_count_all(const GridSet& gs, unsigned long long* cell_cnt, int* cell_of_pt, int* rank_of_pt,
                           hipStream_t s, const FrameInitHook* frame) {
  int blocks = (max_n(gs) + 255) / 256;
  if (blocks > 4096) blocks = 4096;   // (1 M points: one point per thread -- every atomic / scattered store of the pass in flight at once)
  FrameInit fi;
  FrameInitBufs fb;
  memset(&fi, 0, sizeof(fi));
  memset(&fb, 0, sizeof(fb));
  if (frame) { fi = frame->fi; fb = frame->b; }
  hipLaunchKernelGGL(k_grid_count_all, dim3(blocks, kKinds + (frame ? 1 : 0)), dim3(256), 0, s, gs, cell_cnt, cell_of_pt,
                     rank_of_pt, fi, fb);
}
// cell_start of kind k lives at cell_start[cell_base[k] + k ...] (ncell + 1 entries per kind), relative to the
// kind's own point block
__global__ void k_grid_finalize_all(GridSet gs, const unsigned long long* __restrict__ cell_scan,
                                    int* __restrict__ cell_start, unsigned long long* __restrict__ cell_cnt) {
  const int k = blockIdx.y;
  const long long ncell = gs.ncell[k], base = gs.cell_base[k];
  const unsigned long long first = cell_scan[base];
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i <= ncell; i += stride) {
    cell_start[base + k + i] = (i < ncell) ? (int)(cell_scan[base + i] - first) : gs.n[k];
    cell_cnt[base + i] = 0ull;  // the histogram has been scanned: leave it empty for the next build (no memset per frame)
  }
}
void launch_grid_finalize_all(const GridSet& gs, const unsigned long long* cell_scan, int* cell_start,
                              unsigned long long* cell_cnt, hipStream_t s) {
  long long m = 1;
  for (int k = 0; k < kKinds; ++k) m = std::max(m, gs.ncell[k] + 1);
  int blocks = (int)std::min<long long>((m + 255) / 256, 2048);
  hipLaunchKernelGGL(k_grid_finalize_all, dim3(blocks, kKinds), dim3(256), 0, s, gs, cell_scan, cell_start, cell_cnt);
}
__global__ void k_grid_scatter_all(GridSet gs, const int* __restrict__ cell_of_pt,
                                   const unsigned long long* __restrict__ cell_scan,
                                   const int* __restrict__ rank_of_pt, double4* __restrict__ gp) {
  const int k = blockIdx.y;
  const int n = gs.n[k];
  const long long base = gs.cell_base[k];
  const unsigned long long first = cell_scan[base];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = cell_of_pt[gs.tgt_off[k] + i];
    const int pos = (int)(cell_scan[base + c] - first) + rank_of_pt[gs.tgt_off[k] + i];
    gp[gs.tgt_off[k] + pos] =
        double4{gs.tx[k][i], gs.ty[k][i], gs.tz[k][i], __longlong_as_double((long long)i)};
  }
}
// finalize + scatter in ONE launch, straight from the TILE-LOCAL scan: every block first turns the (<= 1024) tile totals
// into tile offsets in LDS (what k_scan_add_direct would add in place in a launch of its own), then blocks
// [0, fin_blocks) of every kind write the cell table and re-zero the histogram, the others scatter the points.
__global__ __launch_bounds__(256) void k_grid_finalize_scatter_all(GridSet gs, const unsigned long long* __restrict__ cell_scan,
                                                                   const unsigned long long* __restrict__ totals, int tiles,
                                                                   int fin_blocks, int* __restrict__ cell_start,
                                                                   unsigned long long* __restrict__ cell_cnt,
                                                                   const int* __restrict__ cell_of_pt,
                                                                   const int* __restrict__ rank_of_pt, double4* __restrict__ gp) {
  __shared__ unsigned long long toff[1024];
  __shared__ unsigned long long wtot[4];
  {  // exclusive prefix of the tile totals: 4 consecutive tiles per thread
    const int t0 = threadIdx.x * 4;
    unsigned long long v[4], run = 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = (t0 + u < tiles) ? totals[t0 + u] : 0ull; run += v[u]; }
    unsigned long long incl = run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long pre = incl - run;
    for (int w = 0; w < wave; ++w) pre += wtot[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) { toff[t0 + u] = pre; pre += v[u]; }
    __syncthreads();
  }
  const int k = blockIdx.y;
  const long long ncell = gs.ncell[k], base = gs.cell_base[k];
  const unsigned long long first = cell_scan[base] + toff[base / kScanTile];
  if ((int)blockIdx.x < fin_blocks) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)fin_blocks * 256;
    for (; i <= ncell; i += stride) {
      cell_start[base + k + i] = (i < ncell) ? (int)(cell_scan[base + i] + toff[(base + i) / kScanTile] - first) : gs.n[k];
      cell_cnt[base + i] = 0ull;  // the histogram has been scanned: leave it empty for the next build (no memset per frame)
    }
  } else {
    const int n = gs.n[k];
    const int nb = (int)gridDim.x - fin_blocks;
    for (int i = ((int)blockIdx.x - fin_blocks) * 256 + threadIdx.x; i < n; i += nb * 256) {
      const int c = cell_of_pt[gs.tgt_off[k] + i];
      const int pos = (int)(cell_scan[base + c] + toff[(base + c) / kScanTile] - first) + rank_of_pt[gs.tgt_off[k] + i];
      gp[gs.tgt_off[k] + pos] = double4{gs.tx[k][i], gs.ty[k][i], gs.tz[k][i], __longlong_as_double((long long)i)};
    }
  }
}
void launch_grid_finalize_scatter_all(const GridSet& gs, const unsigned long long* cell_scan, const unsigned long long* totals,
                                      int tiles, int* cell_start, unsigned long long* cell_cnt, const int* cell_of_pt,
                                      const int* rank_of_pt, double4* gp, hipStream_t s) {
  long long m = 1;
  for (int k = 0; k < kKinds; ++k) m = std::max(m, gs.ncell[k] + 1);
  const int fin_blocks = (int)std::min<long long>((m + 255) / 256, 2048);
  int sc_blocks = (max_n(gs) + 255) / 256;
  if (sc_blocks > 4096) sc_blocks = 4096;
  hipLaunchKernelGGL(k_grid_finalize_scatter_all, dim3(fin_blocks + sc_blocks, kKinds), dim3(256), 0, s, gs, cell_scan, totals,
                     tiles, fin_blocks, cell_start, cell_cnt, cell_of_pt, rank_of_pt, gp);
}
void launch_grid_scatter_all(const GridSet& gs, const int* cell_of_pt, const unsigned long long* cell_scan,
                             const int* rank_of_pt, double4* gp, hipStream_t s) {
  int blocks = (max_n(gs) + 255) / 256;
  if (blocks > 4096) blocks = 4096;   // (1 M points: one point per thread -- every atomic / scattered store of the pass in flight at once)
  hipLaunchKernelGGL(k_grid_scatter_all, dim3(blocks, kKinds), dim3(256), 0, s, gs, cell_of_pt, cell_scan, rank_of_pt, gp);
}

// ================================================================================================
//  K2: the per-query part of the four builders, given the k nearest neighbours (ascending distance)
// ================================================================================================
struct RawRec {
  double a[3], b[3], d;
  unsigned long long flag;  // low 32: counted, high 32: valid
};

// addSphereCostFactor registration.cpp:517-559
// (the neighbour's coordinates: from the point source, or -- knn_rows' NbrXyz -- from the record its list entry was unpacked from)
__device__ __forceinline__ void finish_sphere_xyz(const TopK<1>& tk, double x, double y, double z, double radius, RawRec& r) {
  const int cnt = radius_cut<1>(tk, radius);
  const bool found = cnt > 0;
  const bool skip = found && (tk.d[0] > 0.2);  // :536 squared distance vs 0.2 -> `continue`
  const bool valid = found && !skip;
  const bool counted = !skip;                    // :551 sphere_sum++ for every non-`continue`d point
  if (valid) { r.a[0] = x; r.a[1] = y; r.a[2] = z; }
  r.flag = ((unsigned long long)(valid ? 1 : 0) << 32) | (unsigned long long)(counted ? 1 : 0);
}
// addEdgeCostFactor :427-505 (kind == edge) / addSurfCostFactor :571-635, addGroundCostFactor :714-778
// (cnt neighbours inside the radius, their coordinates in ascending (distance, index) order, 0 beyond cnt)
__device__ __forceinline__ void finish_knn5_xyz(int kind, int cnt, const double (&nx)[5], const double (&ny)[5], const double (&nz)[5],
                                                double edge_dir_thres, RawRec& r) {
  bool valid = false;
  if (kind == TLOAM_KIND_EDGE) {
    if (cnt > 3) {  // :445
      double cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        if (m < cnt) {  // :453-464, neighbours in ascending distance
          const double x = nx[m], y = ny[m], z = nz[m];
          cum[0] += x; cum[1] += y; cum[2] += z;
          cum[3] += x * x; cum[4] += x * y; cum[5] += x * z;
          cum[6] += y * y; cum[7] += y * z; cum[8] += z * z;
        }
      }
      const double nn = (double)cnt;
#pragma unroll
      for (int m = 0; m < 9; ++m) cum[m] /= nn;  // :465
      Sym3 M;
      M.a[0][0] = cum[3] - cum[0] * cum[0];
      M.a[1][1] = cum[6] - cum[1] * cum[1];
      M.a[2][2] = cum[8] - cum[2] * cum[2];
      M.a[0][1] = M.a[1][0] = cum[4] - cum[0] * cum[1];
      M.a[0][2] = M.a[2][0] = cum[5] - cum[0] * cum[2];
      M.a[1][2] = M.a[2][1] = cum[7] - cum[1] * cum[2];
      double ev[3];
      eig3_sym(M, ev);
      const double dx = M.v[0][2], dy = M.v[1][2], dz = M.v[2][2];  // eigenvectors().col(2) :479
      if (ev[2] > 3 * ev[1] && fabs(dz) > edge_dir_thres) {          // :481
        r.a[0] = 0.1 * dx + cum[0];  r.a[1] = 0.1 * dy + cum[1];  r.a[2] = 0.1 * dz + cum[2];   // :483
        r.b[0] = -0.1 * dx + cum[0]; r.b[1] = -0.1 * dy + cum[1]; r.b[2] = -0.1 * dz + cum[2];  // :484
        valid = true;
      }
    }
  } else {
    if (cnt > 4) {  // :589 / :732  (all five neighbours inside the radius)
      double plane[4];
      fit_best_plane5(nx, ny, nz, plane);
      bool ok = true;
#pragma unroll
      for (int m = 0; m < 5; ++m) {  // :605-613 signed, no fabs
        const double dis = plane[0] * nx[m] + plane[1] * ny[m] + plane[2] * nz[m] + plane[3];
        if (dis > 0.2) ok = false;
      }
      if (ok) { r.a[0] = plane[0]; r.a[1] = plane[1]; r.a[2] = plane[2]; r.d = plane[3]; valid = true; }
    }
  }
  // the cap tests (:448 / :592 / :735) only matter once num >= maxnum, and num counts ADDED factors:
  // "added iff valid && #valid before < maxnum" (see launch_compact).
  r.flag = valid ? ((1ull << 32) | 1ull) : 0ull;
}
// ... with the coordinates knn_rows kept from the records the list was unpacked from (NbrXyz)
__device__ __forceinline__ void finish_knn5_kept(int kind, const TopK<5>& tk, const NbrXyz<5>& c, double radius, double edge_dir_thres,
                                                 RawRec& r) {
  const int cnt = radius_cut<5>(tk, radius);
  double nx[5], ny[5], nz[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const bool in = m < cnt;
    nx[m] = in ? c.x[m] : 0.0; ny[m] = in ? c.y[m] : 0.0; nz[m] = in ? c.z[m] : 0.0;
  }
  finish_knn5_xyz(kind, cnt, nx, ny, nz, edge_dir_thres, r);
}

// the (counted | valid << 32) flag of a slot, and its one-byte twin for the self-compacting Solve (SlotView::flagb)
__device__ __forceinline__ void store_flag(const SlotView& sv, int slot, unsigned long long flag) {
  sv.flags[slot] = flag;
  if (sv.flagb) {
    int kind = 0;
#pragma unroll
    for (int k = 1; k < kKinds; ++k) kind += (slot >= sv.slot_off[k]) ? 1 : 0;
    sv.flagb[kind * kFlagbStride + (slot - sv.slot_off[kind])] =
        (unsigned char)(((flag >> 32) != 0ull ? 1 : 0) | ((flag & 0xffffffffull) != 0ull ? 2 : 0));
  }
}
__device__ __forceinline__ void store_raw(const SlotView& sv, int slot, const RawRec& r) {
  store_flag(sv, slot, r.flag);
  if ((r.flag >> 32) == 0ull) return;  // no factor: the compaction never looks at the record
  double2* q = reinterpret_cast<double2*>(sv.raw + (size_t)slot * 8);
  q[0] = double2{r.a[0], r.a[1]};
  q[1] = double2{r.a[2], r.b[0]};
  q[2] = double2{r.b[1], r.b[2]};
  q[3] = double2{r.d, 0.0};
}

// ================================================================================================
//  K1+K2 driver.  The target grid of every kind is cut into TILES of 4x4x4 cells; the source points are
//  bucketed by the tile their (transformed) cell falls into (a counting sort, once per frame) and the
//  queries are then processed in tile order, so the lanes of a wave walk the same few cells and share the
//  packed candidate records through L1/L2.  Every query still searches its own exact 27 cells.
//  Replaces the kd-tree walks of registration.cpp:444/:535/:588/:731.
// ================================================================================================
constexpr int kTile = 4;        // cells per tile edge
#ifndef TLOAM_K1_QUAD_LIMIT
#define TLOAM_K1_QUAD_LIMIT 131072
#endif
#ifndef TLOAM_K1_WIDE_LIMIT
#define TLOAM_K1_WIDE_LIMIT 16384
#endif
constexpr int kQuadLimit = TLOAM_K1_QUAD_LIMIT;  // at most this many queries: four lanes per query (latency-bound regime)
constexpr int kWideLimit = TLOAM_K1_WIDE_LIMIT;   // at most this many: sixteen lanes per query (one row of the 27 cells per lane)

struct TileMeta {
  int tdim[kKinds][3];
  int bin_base[kKinds + 1];   // concatenated bin index space over the 4 kinds: a kind's tiles x its bins per tile
  int sub[kKinds];            // bins per tile of the kind: 1 (queries grouped by tile) or 64 (by cell inside the tile, see bin_sub)
};
// Thread-per-query frames (> kQuadLimit queries) sort their queries by CELL, tile-major: the 64 lanes of a wave then
// cover ~25 neighbouring cells instead of 64 scattered ones of a tile, lanes of one cell walk the same nine rows in
// the same order, and a gather instruction touches 2-3x fewer distinct cache lines -- the walk is bound by the
// lines the CU's address unit retires, not by bytes.  Smaller frames (several lanes per query) keep the coarse sort:
// their scan over the bins would cost more than it saves.
// Per KIND since round 5: a kind with few queries (the 40 k sphere queries of the 1 M frame against 28 k tiles) gains nothing from
// 64 bins per tile and only lengthens the histogram every frame zeroes, counts into and scans (4.7 M -> 2.9 M bins there).
static int bin_sub(int n_slots, int n_kind) { return (n_slots > kQuadLimit && n_kind >= 65536) ? kTile * kTile * kTile : 1; }
// the tile / bin metadata of the sorted query order; returns the number of bins
static int tile_meta(const GridView grids[kKinds], const int slot_off[kKinds + 1], TileMeta* tm) {
  int base = 0;
  for (int k = 0; k < kKinds; ++k) {
    tm->bin_base[k] = base;
    for (int a = 0; a < 3; ++a) tm->tdim[k][a] = (std::max(grids[k].dim[a], 1) + kTile - 1) / kTile;
    tm->sub[k] = bin_sub(slot_off[kKinds], slot_off[k + 1] - slot_off[k]);
    base += tm->tdim[k][0] * tm->tdim[k][1] * tm->tdim[k][2] * tm->sub[k];
  }
  tm->bin_base[kKinds] = base;
  return base;
}
struct BuildArgs {
  SlotView sv;
  GridView grid[kKinds];
  BuildParams bp;
  TileMeta tm;
  int identity_n;   // > 0: no query sort -- query i is source slot i, identity_n of them
  DirectSet ds;     // ds.on: the search writes the factor set itself, row = sorted position (tl_common.hpp DirectSet) ...
  CorrView cv;      // ... into these segments
};
bool direct_set_size(int n_slots) { return n_slots > kQuadLimit; }

__device__ __forceinline__ int slot_kind(const SlotView& sv, int slot) {
  int kind = 0;
#pragma unroll
  for (int k = 1; k < kKinds; ++k) kind += (slot >= sv.slot_off[k]) ? 1 : 0;
  return kind;
}
// The factor of the query at sorted position `pos` straight into its ROW of the direct set (tl_common.hpp DirectSet); q = the
// query record (source point in the sensor frame, slot).  Consecutive lanes are consecutive rows: every stream is written coalesced.
__device__ __forceinline__ void store_direct(const BuildArgs& A, int kind, int pos, const double4& q, int slot, const RawRec& r) {
  // (every one-wave block of the search also leaves its factor counts per kind: build_sorted_block)
  const CorrSeg& seg = A.cv.k[kind];
  const int row = pos - A.sv.slot_off[kind];
  if (row < 0 || row >= seg.cap) return;   // (cannot happen while every kind is searched -- the host checks -- but never out of bounds)
  const bool valid = (r.flag >> 32) != 0ull;
  const int id = slot - A.sv.slot_off[kind] + A.sv.src_lo[kind];
  seg.idx[row] = valid ? id : ~id;
  if (A.ds.first) {
    seg.px[row] = q.x; seg.py[row] = q.y; seg.pz[row] = q.z;
    seg.w[row] = 1.0;   // registration.cpp:931-949: every weight starts at 1 (weight stream 0: what the first Solve reads)
  }
  if (kind == TLOAM_KIND_SPHERE) {
    seg.ax[row] = valid ? r.a[0] : __builtin_nan("");
    seg.ay[row] = valid ? r.a[1] : 0.0;
    seg.az[row] = valid ? r.a[2] : 0.0;
  } else {
    seg.ax[row] = valid ? r.a[0] : 0.0; seg.ay[row] = valid ? r.a[1] : 0.0; seg.az[row] = valid ? r.a[2] : 0.0;
    if (kind == TLOAM_KIND_EDGE) { seg.bx[row] = valid ? r.b[0] : 0.0; seg.by[row] = valid ? r.b[1] : 0.0; seg.bz[row] = valid ? r.b[2] : 0.0; }
    else seg.d[row] = valid ? r.d : 0.0;
  }
}

// pass 1: tile of every source slot under the current pose; histogram of queries per tile
__device__ __forceinline__ void query_bin_slot(const BuildArgs& A, const GnState* __restrict__ st, int* __restrict__ tile_of_slot,
                                               int* __restrict__ rank_in_tile, unsigned long long* __restrict__ tile_cnt, int slot) {
  if (slot >= A.sv.slot_off[kKinds]) return;
  const int kind = slot_kind(A.sv, slot);
  const GridView& g = A.grid[kind];
  if (!A.bp.active[kind] || g.n <= 0) {  // inactive kind / empty target: no factor, nothing counted...
    // ...except the sphere builder, whose counter also advances for points WITHOUT a neighbour (:551)
    store_flag(A.sv, slot, (A.bp.active[kind] && kind == TLOAM_KIND_SPHERE) ? 1ull : 0ull);
    tile_of_slot[slot] = -1;
    return;
  }
  const Pose T = st->T_cur;  // exp(se3_pose_)  registration.cpp:434/:524/:578/:721
  const Vec3 pw = act(T, Vec3{A.sv.sx[slot], A.sv.sy[slot], A.sv.sz[slot]});
  // bucket by the cell clamped INTO the grid (queries outside the grid sort with its border tiles)
  const int cx = clampi(cell_coord(pw.x, g.org[0], g.inv_cell, g.dim[0]), 0, g.dim[0] - 1);
  const int cy = clampi(cell_coord(pw.y, g.org[1], g.inv_cell, g.dim[1]), 0, g.dim[1] - 1);
  const int cz = clampi(cell_coord(pw.z, g.org[2], g.inv_cell, g.dim[2]), 0, g.dim[2] - 1);
  const int sub = A.tm.sub[kind];
  int t = A.tm.bin_base[kind] + (((cz / kTile) * A.tm.tdim[kind][1] + (cy / kTile)) * A.tm.tdim[kind][0] + (cx / kTile)) * sub;
  if (sub > 1) t += ((cz % kTile) * kTile + (cy % kTile)) * kTile + (cx % kTile);
  tile_of_slot[slot] = t;
  rank_in_tile[slot] = (int)atomicAdd(&tile_cnt[t], 1ull);  // the one atomic of the sort: count AND rank
}
__global__ __launch_bounds__(256) void k_query_bin(BuildArgs A, const GnState* __restrict__ st,
                                                   int* __restrict__ tile_of_slot, int* __restrict__ rank_in_tile,
                                                   unsigned long long* __restrict__ tile_cnt) {
  query_bin_slot(A, st, tile_of_slot, rank_in_tile, tile_cnt, blockIdx.x * 256 + threadIdx.x);
}
// the scatter of the one-pass grid build (k_grid_scatter_start_all) with the query binning riding on it: rows blockIdx.y >= kKinds
// of the launch are k_query_bin's blocks
__global__ __launch_bounds__(256) void k_grid_scatter_qbin(GridSet gs, const int* __restrict__ cell_of_pt, const int* __restrict__ cell_start,
                                                           const int* __restrict__ rank_of_pt, double4* __restrict__ gp, BuildArgs A,
                                                           const GnState* __restrict__ st, int* __restrict__ tile_of_slot,
                                                           int* __restrict__ rank_in_tile, unsigned long long* __restrict__ tile_cnt, int rows) {
  // ONE row of blocks, the roles interleaved (block L: role L % roles, place L / roles): dispatched in index order, scatter and
  // binning blocks then run side by side from the first to the last -- as whole rows (binning behind the scatter) the launch
  // took 78 us, the two kernels apart 37 + 51
  const int roles = kKinds + rows, role = (int)blockIdx.x % roles, bx = (int)blockIdx.x / roles, gx = (int)gridDim.x / roles;
  if (role >= kKinds) {
    const int n = A.sv.slot_off[kKinds];
    const int rider = (role - kKinds) * gx + bx, nriders = gx * rows;
    for (int slot = rider * 256 + (int)threadIdx.x; slot < n; slot += nriders * 256) query_bin_slot(A, st, tile_of_slot, rank_in_tile, tile_cnt, slot);
    return;
  }
  const int k = role;
  const int n = gs.n[k];
  const long long base = gs.cell_base[k] + k;
  for (int i = bx * (int)blockDim.x + (int)threadIdx.x; i < n; i += gx * (int)blockDim.x) {
    const int c = cell_of_pt[gs.tgt_off[k] + i];
    const int pos = cell_start[base + c] + rank_of_pt[gs.tgt_off[k] + i];
    gp[gs.tgt_off[k] + pos] = double4{gs.tx[k][i], gs.ty[k][i], gs.tz[k][i], __longlong_as_double((long long)i)};
  }
}
// pass 2: slots grouped by tile (order inside a tile is irrelevant: every result goes to its own slot)
// The sorted entry is a 32-byte record (x, y, z, slot): K1 then reads its queries coalesced instead of
// chasing slot -> three scattered 8-byte loads.
__global__ __launch_bounds__(256) void k_query_scatter(SlotView sv, const int* __restrict__ tile_of_slot,
                                                       const unsigned long long* __restrict__ tile_scan,
                                                       const int* __restrict__ rank_in_tile, double4* __restrict__ qrec) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= sv.slot_off[kKinds]) return;
  const int t = tile_of_slot[slot];
  if (t < 0) return;
  qrec[(int)tile_scan[t] + rank_in_tile[slot]] =
      double4{sv.sx[slot], sv.sy[slot], sv.sz[slot], __longlong_as_double((long long)slot)};
}

// One query against the HBM grid of its kind, by LPQ cooperating lanes (1, 4 or 16).
//   LPQ = 1: the lane resolves the nine (z,y) rows of the 27-cell neighbourhood (one 16-byte cell-table
//            request per row, all in flight) and walks them as one candidate stream, four records per trip.
//   LPQ = 4 / 16: smaller frames are latency-bound (a ~35-candidate dependent chain per query), so 4 (16)
//            adjacent lanes split the nine rows -- three (at most one) each --, walk them in parallel and merge
//            their packed-key lists with two (four) xor-shuffle rounds; lane 0 of the group finishes the fit.
#ifdef TLOAM_K1_PROF   // development aid (scripts/k1_prof.py): wall-clock stamps of one wave of the sixteen-lane search
__device__ unsigned long long g_k1_prof[64];
extern "C" void tloam_debug_k1_prof(unsigned long long* out64) { (void)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_k1_prof), sizeof(g_k1_prof)); }
__device__ __forceinline__ int k1_prof_which() {   // four sampled one-wave workgroups: first, a third, two thirds, late
  const int b = (int)blockIdx.x;
  return b == 0 ? 0 : (b == 800 ? 1 : (b == 1600 ? 2 : (b == 2300 ? 3 : -1)));
}
#define TL_K1_STAMP(i) if (LPQ == 16 && (threadIdx.x & 63) == 0 && k1_prof_which() >= 0) g_k1_prof[k1_prof_which() * 16 + (i)] = wall_clock64();
#else
#define TL_K1_STAMP(i)
#endif
}  // namespace tl
#include "tl_walk.hpp"   // knn_rows (shared with the PCA pass of tl_feature.hip)
namespace tl {

template <int LPQ>
__device__ __forceinline__ bool query_one(const BuildArgs& A, int kind, const Pose& T, const double4& q, int slot,
                                          int sub, int2* __restrict__ lds_rows, int pos = 0) {
  const GridView& g = A.grid[kind];
  if (!A.bp.active[kind] || g.n <= 0) {   // (never reached in sorted order: k_query_bin leaves these slots out, same flags)
    if (sub == 0) store_flag(A.sv, slot, (A.bp.active[kind] && kind == TLOAM_KIND_SPHERE) ? 1ull : 0ull);
    return false;
  }
  TL_K1_STAMP(0)
  const Vec3 pw = act(T, Vec3{q.x, q.y, q.z});
  const PtsGlobal pts{g.gp};
  RawRec rec;
  rec.a[0] = rec.a[1] = rec.a[2] = rec.b[0] = rec.b[1] = rec.b[2] = rec.d = 0.0;
  rec.flag = 0ull;
  const double radius = A.bp.radius[kind];
  // (the neighbours' coordinates stay in registers from the records the walk's list was unpacked from: the fits do not fetch them again)
  if (kind == TLOAM_KIND_SPHERE) {
    TopK<1> tk;
    NbrXyz<1> c;
    knn_rows<1, LPQ>(g, pts, pw, sub, tk, lds_rows, radius, &c);
    if (sub == 0) finish_sphere_xyz(tk, c.x[0], c.y[0], c.z[0], radius, rec);
  } else {
    TopK<5> tk;
    NbrXyz<5> c;
    knn_rows<5, LPQ>(g, pts, pw, sub, tk, lds_rows, radius, &c);
#ifdef TLOAM_K1_DBG_NOFIT   // timing experiment only: the walk without the per-query fit
    if (sub == 0) { rec.a[0] = tk.d[0] + tk.d[4]; rec.flag = (tk.j[4] >= 0) ? ((1ull << 32) | 1ull) : 0ull; }
#else
    if (sub == 0) finish_knn5_kept(kind, tk, c, radius, A.bp.edge_dir_thres, rec);
#endif
  }
  TL_K1_STAMP(6)
  if (sub == 0) {
    if (LPQ == 1 && A.ds.on) store_direct(A, kind, pos, q, slot, rec);
    else store_raw(A.sv, slot, rec);
  }
  TL_K1_STAMP(7)
  return (rec.flag >> 32) != 0ull;
}

// pass 3: the queries in TILE-SORTED order: the lanes of a wave query the same few cells, so the packed
// candidate records they touch are shared through L1/L2 instead of being re-fetched from HBM per query
// (the unsorted SoA version moved ~30x the algorithmic bytes as 64-byte sectors).
// the work of (physical) block `blk` of the search launch
template <int LPQ>
__device__ __forceinline__ void build_sorted_block(const BuildArgs& A, const GnState* __restrict__ st,
                                                   const unsigned long long* __restrict__ n_sorted,
                                                   const double4* __restrict__ qrec, int blk, int2* __restrict__ lds_rows) {
  // XCD-aware order: the dispatcher deals blocks round-robin to the 8 XCDs (each with a private L2), so
  // physical block b is given logical position (b % 8) * (blocks / 8) + b / 8 -- every XCD then walks one
  // CONTIGUOUS eighth of the tile-sorted queries and neighbouring tiles share target records in its L2
  // ... in CHUNKS: a contiguous eighth per XCD would hand whole kinds to single XCDs (planar to XCDs 0-3, the
  // edge kind with its 3x3 eigen solves to XCD 6-7) and the slowest XCD sets the kernel time; instead XCD x takes
  // the chunks x, x + 8, x + 16, ... of kXcdChunk consecutive blocks (1024 tile-sorted queries: still local).
  constexpr int kXcdChunk = 16;
  const int xcd = blk & 7, in_xcd = blk >> 3;
  const int lb0 = ((in_xcd / kXcdChunk) * 8 + xcd) * kXcdChunk + in_xcd % kXcdChunk;  // grid: a multiple of 8 chunks
  // ... and BACK TO FRONT: the sorted list ends with the edge kind (most candidates, eigen solve per query); the
  // expensive blocks are dispatched first so that the cheap ones fill the tail (longest-processing-time first)
  const int ns = A.identity_n ? A.identity_n : (int)*n_sorted;
  const int nblk = (int)(((long long)ns * LPQ + 63) / 64);
  if (lb0 >= nblk) return;  // whole block past the end (grid rounded up)
  const int lb = nblk - 1 - lb0;
  const int t = lb * 64 + threadIdx.x;
  const int i = t / LPQ, sub = t % LPQ;
  // slots without a tile (inactive kinds) are not in qrec.  LPQ = 1 keeps the tail lanes alive (they take
  // part in the wave-wide trip count) on a harmless duplicate of the last query; LPQ = 4 exits quad-uniformly.
  if (LPQ != 1 && i >= ns) return;
  if (ns <= 0) return;
  const bool live = i < ns;
  const int qi = live ? i : ns - 1;
  const double4 q = A.identity_n ? double4{A.sv.sx[qi], A.sv.sy[qi], A.sv.sz[qi], __longlong_as_double((long long)qi)} : qrec[qi];
  const int slot = (int)__double_as_longlong(q.w);
  const int kind = slot_kind(A.sv, slot);
  int row_pos = qi;
  if (LPQ == 1 && A.ds.on && A.ds.row_of_pos) {   // the row of this sorted position: its bin's queries in slot order (DirectSet)
    if (A.ds.first) {
      if (live) {
        const int t = A.ds.tile_of_slot[slot];
        const unsigned long long s0 = A.ds.tile_scan[t], e0 = A.ds.tile_scan[t + 1];
        if (e0 - s0 > 1ull && e0 - s0 <= (unsigned long long)kDirectBinMax) {
          int rank = 0;
          for (unsigned long long j = s0; j < e0; ++j) rank += ((int)__double_as_longlong(qrec[j].w) < slot) ? 1 : 0;
          row_pos = (int)s0 + rank;
        }
        A.ds.row_of_pos[qi] = row_pos;
      }
    } else {
      row_pos = A.ds.row_of_pos[qi];
    }
  }
  const bool valid = query_one<LPQ>(A, kind, st->T_cur, q, slot, live ? sub : 1, lds_rows, row_pos);
  if (LPQ == 1 && A.ds.on && A.ds.blk_cnt) {   // this block's factors per kind (a wave can straddle a kind boundary), by LOGICAL block
    const bool f = live && valid;
#pragma unroll
    for (int k = 0; k < kKinds; ++k) {
      const int cnt = __popcll(__ballot(f && kind == k));
      if (threadIdx.x == 0) A.ds.blk_cnt[(size_t)lb * kKinds + k] = cnt;
    }
  }
}
template <int LPQ>
__global__ __launch_bounds__(64) void k_build_sorted(BuildArgs A, const GnState* __restrict__ st,
                                                     const unsigned long long* __restrict__ n_sorted,
                                                     const double4* __restrict__ qrec, const int* __restrict__ gate, GnState* st_w) {
  if (gate && *gate == 0) return;  // device-driven outer loop: the pose did not move, the set is only refreshed
#ifdef TLOAM_K1_PROF
  if (LPQ == 16 && threadIdx.x == 0 && blockIdx.x == 0) g_k1_prof[15] = wall_clock64();   // (about when the launch starts)
#endif
  __shared__ int2 lds_rows[LPQ == 1 ? 9 * 64 : 1];
  // a direct set is built at the pose this search runs at (what the compaction records for a compact set)
  if (LPQ == 1 && A.ds.on && A.ds.set_x_build && blockIdx.x == 0 && threadIdx.x < 6) st_w->x_build[threadIdx.x] = st->x[threadIdx.x];
  build_sorted_block<LPQ>(A, st, n_sorted, qrec, (int)blockIdx.x, lds_rows);
}
// ---- large sets, device-driven loop: the finish of outer iteration k-1 (k_weights + k_outer_finish as one-wave blocks,
//      weights_finish_large_ride) at the head of the thread-per-query search of iteration k, which runs on the
//      minimiser's own "ended somewhere else than x_build" (spec_build) -- see k_build_finish_small
__global__ __launch_bounds__(64) void k_build_finish_large(BuildArgs A, GnState* st, const unsigned long long* __restrict__ n_sorted,
                                                           const double4* __restrict__ qrec, const int* __restrict__ seg_n,
                                                           double* __restrict__ sums16, HostMirror hm, OuterCtl ctl, WeightArgs W,
                                                           FinishRideLarge R, FinishDirect D) {
  const int nfin = 4 * R.wblocks;
  if ((int)blockIdx.x < nfin) {
    if (A.ds.on) finish_direct_block(st, seg_n, sums16, hm, ctl, W, D, (int)blockIdx.x);
    else weights_finish_large_ride(st, seg_n, sums16, hm, ctl, W, R, (int)blockIdx.x);
    return;
  }
  if (st->spec_build == 0 || __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  __shared__ int2 lds_rows[9 * 64];
  build_sorted_block<1>(A, st, n_sorted, qrec, (int)blockIdx.x - nfin, lds_rows);
}
// the finish of a direct set as a launch of its own
__global__ __launch_bounds__(64) void k_finish_direct(GnState* st, const int* __restrict__ seg_n, double* __restrict__ sums16, HostMirror hm,
                                                      OuterCtl ctl, WeightArgs W, FinishDirect D) {
  finish_direct_block(st, seg_n, sums16, hm, ctl, W, D, (int)blockIdx.x);
}
void launch_finish_direct(const FinishLargeArgs& fin, GnState* st, hipStream_t s) {
  WeightArgs W;
  memset(&W, 0, sizeof(W));
  W.cv = *fin.cv;
  W.wp = *fin.wp;
  FinishDirect D{fin.rows, fin.ticket, 4 * fin.wblocks, {fin.w_next[0], fin.w_next[1], fin.w_next[2], fin.w_next[3]}, fin.blk_cnt, fin.nblk, fin.built};
  hipLaunchKernelGGL(k_finish_direct, dim3(4 * fin.wblocks), dim3(64), 0, s, st, fin.seg_n, fin.sums16, fin.hm, fin.ctl, W, D);
}
void launch_grid_scan_finalize_scatter_1p(const GridSet& gs, unsigned long long* cell_cnt, size_t ncells_plus_1, int* cell_start,
                                          unsigned long long* ctl, unsigned* fault, const int* cell_of_pt, const int* rank_of_pt, double4* gp,
                                          hipStream_t s, const QueryBinRide* qbin, const GridView* views) {
  const size_t tiles = (ncells_plus_1 + kTile1p - 1) / kTile1p;
  Scan1p C{ctl, next_scan_epoch(), fault};
  hipLaunchKernelGGL(k_grid_scan_finalize_1p, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, gs, cell_cnt, ncells_plus_1, cell_start, C);
  int blocks = (max_n(gs) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (qbin && views) {   // the query sort's first pass as extra rows of this launch (QueryBinRide)
    BuildArgs A;
    memset(&A, 0, sizeof(A));
    A.sv = qbin->sv;
    A.bp = qbin->bp;
    for (int k = 0; k < kKinds; ++k) A.grid[k] = views[k];
    (void)tile_meta(views, qbin->sv.slot_off, &A.tm);
    const int n = qbin->sv.slot_off[kKinds];
    const int rows = std::max(1, std::min(4, ((n + 255) / 256 + blocks - 1) / blocks));
    hipLaunchKernelGGL(k_grid_scatter_qbin, dim3(blocks * (kKinds + rows)), dim3(256), 0, s, gs, cell_of_pt, cell_start, rank_of_pt, gp, A, qbin->st,
                       qbin->tile_of_slot, qbin->rank_in_tile, qbin->tile_cnt, rows);
    return;
  }
  hipLaunchKernelGGL(k_grid_scatter_start_all, dim3(blocks, kKinds), dim3(256), 0, s, gs, cell_of_pt, cell_start, rank_of_pt, gp);
}
void launch_build(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                  int* tile_of_slot, unsigned long long* tile_cnt, unsigned long long* tile_scan, int* tile_fill,
                  double4* qrec, unsigned long long* scan_tmp, bool rebin, hipStream_t s, const int* gate,
                  unsigned long long* scan1p_ctl, unsigned* scan1p_fault, const CorrView* direct_cv, const DirectSet* ds, bool binned) {
  const int n = sv.slot_off[kKinds];
  if (n <= 0) return;
  BuildArgs A;
  A.sv = sv;
  A.bp = bp;
  memset(&A.ds, 0, sizeof(A.ds));
  memset(&A.cv, 0, sizeof(A.cv));
  if (direct_cv && ds && direct_set_size(n)) { A.ds = *ds; A.cv = *direct_cv; }
  for (int k = 0; k < kKinds; ++k) A.grid[k] = grids[k];
  const int ntiles = tile_meta(grids, sv.slot_off, &A.tm);   // bins of the counting sort
  // KITTI-size frames (sixteen lanes per query) are searched in SLOT order, unsorted: their target records (2-3 MB)
  // stay L2-resident whatever the order, and the sort's three launches (bin, scan, scatter: 13 us of a 270 us frame)
  // cost more than its locality saves -- measured with randomly ordered source clouds, the worst case: 0.273 -> 0.261 ms.
  A.identity_n = (n <= kWideLimit) ? n : 0;
  if (rebin && !A.identity_n) {
    // The processing ORDER only buys locality -- every query still searches its own exact 27 cells --
    // so the tile sort is done once per frame (first outer iteration, predicted pose) and reused while
    // the pose moves by centimetres.
    // (tile_cnt[0 .. ntiles] was zeroed by k_frame_init -- one launch less at the start of every frame)
    if (!binned) hipLaunchKernelGGL(k_query_bin, dim3((n + 255) / 256), dim3(256), 0, s, A, st, tile_of_slot, tile_fill, tile_cnt);
    if (scan1p_ctl) launch_scan_counts_1p(tile_cnt, tile_scan, (size_t)ntiles + 1, scan1p_ctl, scan1p_fault, s);   // (the caller has checked scan_1p_applies)
    else launch_exclusive_scan_u64(tile_cnt, tile_scan, (size_t)ntiles + 1, scan_tmp, s);
    hipLaunchKernelGGL(k_query_scatter, dim3((n + 255) / 256), dim3(256), 0, s, sv, tile_of_slot, tile_scan,
                       tile_fill, qrec);
  }
  // every slot with a tile is in qrec[0 .. n_binned); n_binned <= n is only known on the device, so the
  // launch covers n positions and the kernel bounds itself by the scanned total
  auto grid8 = [](long long threads) { return (unsigned)(((threads + 63) / 64 + 127) / 128 * 128); };  // 8 XCDs x kXcdChunk
  // lanes per query: the smaller the frame, the more the build is a latency chain per query and the more lanes
  // pay (KITTI-size 9.4 k queries: 0.371-0.378 ms per frame with 4, 0.36 with 8, 0.348-0.36 with 16)
  if (n <= kWideLimit)
    hipLaunchKernelGGL(k_build_sorted<16>, dim3(grid8(16LL * n)), dim3(64), 0, s, A, st, tile_scan + ntiles, qrec, gate, st);
  else if (n <= kQuadLimit)
    hipLaunchKernelGGL(k_build_sorted<4>, dim3(grid8(4LL * n)), dim3(64), 0, s, A, st, tile_scan + ntiles, qrec, gate, st);
  else
    hipLaunchKernelGGL(k_build_sorted<1>, dim3(grid8(n)), dim3(64), 0, s, A, st, tile_scan + ntiles, qrec, gate, st);
}
// ---- the small-set finish (tl_finish.hpp), stand-alone -----------------------------------------------------------
__global__ __launch_bounds__(1024) void k_weights_finish_small(GnState* st, const int* __restrict__ seg_n,
                                                               double* __restrict__ sums16, HostMirror hm, OuterCtl ctl,
                                                               WeightArgs A) {
  weights_finish_small_body(st, seg_n, sums16, hm, ctl, A);
}
void launch_weights_finish_small(const CorrView& cv, const SlotView& sv, const WeightParams& wp, const int* seg_n,
                                 double* sums16, GnState* st, HostMirror hm, OuterCtl ctl, hipStream_t s) {
  WeightArgs A;
  A.cv = cv;
  A.sv = sv;
  A.wp = wp;
  hipLaunchKernelGGL(k_weights_finish_small, dim3(1), dim3(1024), 0, s, st, seg_n, sums16, hm, ctl, A);
}
// ---- ... and riding on the correspondence search of the NEXT outer iteration ---------------------------------------
// The finish of iteration k-1 reads the old factor set's costs and writes slot weights, sums and the loop decisions;
// the search of iteration k reads the source slots, the target grids and the pose the Solve ended at, and writes raw
// records and flags: neither reads what the other writes, so they share a launch -- the first sixteen one-wave blocks
// are the finish (weights_finish_small_ride), the others search sixteen lanes per query in slot order exactly as
// k_build_sorted<16> does (one-wave blocks: the search is bound by what one CU's L1 looks up, so it wants to be spread
// over all of them -- 1024-thread blocks on 147 CUs took twice as long).  The search cannot wait for the finish's
// verdict (run_build), so it runs on the minimiser's own "ended somewhere else than x_build" (spec_build) unless an
// EARLIER iteration has stopped the loop; when this finish stops the loop (plateau, last iteration) or finds the Solve
// unfinished, the records written here are simply never compacted (k_prepare_small is gated on run_build).
constexpr int kFinishBlocks = 16;
__global__ __launch_bounds__(64) void k_build_finish_small(BuildArgs A, GnState* st, const int* __restrict__ seg_n,
                                                           double* __restrict__ sums16, HostMirror hm, OuterCtl ctl,
                                                           WeightArgs W, FinishRide R) {
  if (blockIdx.x < kFinishBlocks) {
    weights_finish_small_ride(st, seg_n, sums16, hm, ctl, W, R, (int)blockIdx.x);
    return;
  }
  // (`stop` may be written by this launch's own finish while it is read here: either value is fine -- see above -- but
  //  the access is made an atomic one so that it is not a data race)
  if (st->spec_build == 0 || __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  // block order as in k_build_sorted: chunks of kXcdChunk blocks dealt to the XCDs, back to front (edge kind first)
  constexpr int kXcdChunk = 16;
  const int b = (int)blockIdx.x - kFinishBlocks;
  const int xcd = b & 7, in_xcd = b >> 3;
  const int lb0 = ((in_xcd / kXcdChunk) * 8 + xcd) * kXcdChunk + in_xcd % kXcdChunk;
  const int n = A.identity_n;
  const int nblk = (int)(((long long)n * 16 + 63) / 64);
  if (lb0 >= nblk) return;
  const int t = (nblk - 1 - lb0) * 64 + (int)threadIdx.x;
  const int i = t >> 4, sub = t & 15;
  if (i >= n) return;
  const double4 q = double4{A.sv.sx[i], A.sv.sy[i], A.sv.sz[i], __longlong_as_double((long long)i)};
  query_one<16>(A, slot_kind(A.sv, i), st->T_cur, q, i, sub, nullptr);
}
bool build_finish_small_fits(const SlotView& sv) { return sv.slot_off[kKinds] > 0 && sv.slot_off[kKinds] <= kWideLimit; }
void launch_build_finish_small(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                               const FinishSmallArgs& fin, hipStream_t s) {
  const int n = sv.slot_off[kKinds];
  BuildArgs A;
  A.sv = sv;
  A.bp = bp;
  for (int k = 0; k < kKinds; ++k) A.grid[k] = grids[k];
  memset(&A.tm, 0, sizeof(A.tm));   // slot order: no tiles
  memset(&A.ds, 0, sizeof(A.ds));
  memset(&A.cv, 0, sizeof(A.cv));
  A.identity_n = n;
  WeightArgs W;
  W.cv = *fin.cv;
  W.sv = sv;
  W.wp = *fin.wp;
  const unsigned build_blocks = (unsigned)(((16LL * n + 63) / 64 + 127) / 128 * 128);  // 8 XCDs x kXcdChunk, as launch_build
  hipLaunchKernelGGL(k_build_finish_small, dim3(kFinishBlocks + build_blocks), dim3(64), 0, s, A, st, fin.seg_n, fin.sums16,
                     fin.hm, fin.ctl, W, FinishRide{fin.rows, fin.ticket});
}

bool build_finish_large_fits(const SlotView& sv) { return sv.slot_off[kKinds] > kQuadLimit; }
void launch_build_finish_large(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                               const unsigned long long* n_sorted, const double4* qrec, const FinishLargeArgs& fin, hipStream_t s,
                               const DirectSet* ds) {
  const int n = sv.slot_off[kKinds];
  BuildArgs A;
  A.sv = sv;
  A.bp = bp;
  for (int k = 0; k < kKinds; ++k) A.grid[k] = grids[k];
  (void)tile_meta(grids, sv.slot_off, &A.tm);   // (as launch_build: the tile metadata of the sorted query order)
  A.identity_n = 0;
  memset(&A.ds, 0, sizeof(A.ds));
  memset(&A.cv, 0, sizeof(A.cv));
  if (ds) { A.ds = *ds; A.cv = *fin.cv; }
  const FinishDirect D{fin.rows, fin.ticket, 4 * fin.wblocks, {fin.w_next[0], fin.w_next[1], fin.w_next[2], fin.w_next[3]}, fin.blk_cnt, fin.nblk, fin.built};
  WeightArgs W;
  W.cv = *fin.cv;
  W.sv = sv;
  W.wp = *fin.wp;
  const unsigned build_blocks = (unsigned)((((long long)n + 63) / 64 + 127) / 128 * 128);   // 8 XCDs x kXcdChunk, as launch_build
  hipLaunchKernelGGL(k_build_finish_large, dim3(4 * fin.wblocks + build_blocks), dim3(64), 0, s, A, st, n_sorted, qrec, fin.seg_n,
                     fin.sums16, fin.hm, fin.ctl, W, FinishRideLarge{fin.rows, fin.ticket, fin.wblocks}, D);
}

int build_tile_count(const GridView grids[kKinds], const int slot_off[kKinds + 1]) {
  TileMeta tm;
  return tile_meta(grids, slot_off, &tm);
}

// ================================================================================================
//  cap + compaction.  With C = exclusive prefix of `counted` within the kind (+ the counts of the
//  lower ranks when sharded) and V = exclusive prefix of `valid`:
//     added(i)  <=>  valid(i) && C(i) < maxnum           (registration.cpp:448/:538/:592/:735)
//  and because C is non-decreasing every valid j < i of an added i was added too, so the compact
//  position of i is simply V(i).
// ================================================================================================
struct CompactArgs {
  SlotView sv;
  CorrView cv;
  int maxnum[kKinds];
  int* seg_n;
  const double* rank_counts;  // [nranks*4] counted totals per rank (all-reduced), or null
  int rank, nranks;
  GnState* st;                // receives the pose the set was built at (x_build)
  const int* gate;
  const int* refresh_gate;    // non-null: the launch stands for BOTH alternatives of a device-gated iteration -- compaction
                              // if *gate, else the refresh of the unchanged set (k_refresh) if *refresh_gate
  const unsigned long long* totals;   // tiles > 0: `scan` holds TILE-LOCAL scans, the tiles' totals are here (the kernel adds
  int tiles;                          // the tile offsets itself: no k_scan_add_direct launch; <= 1024)
};
__global__ __launch_bounds__(256) void k_compact(CompactArgs A) {
  __shared__ unsigned long long toff[1024];
  __shared__ unsigned long long wtot[4];
  if (A.gate && *A.gate == 0) {
    if (A.refresh_gate && *A.refresh_gate != 0) {   // the set of the previous iteration: new captured weights, zeroed slots
      const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
      for (int k = 0; k < kKinds; ++k) {
        const int n = A.cv.seg_n[k];
        const CorrSeg& seg = A.cv.k[k];
        for (int i = tid; i < n; i += stride) {
          seg.w[i] = A.sv.w_src[A.sv.slot_off[k] + (seg.idx[i] - A.sv.src_lo[k])];
          seg.cost[i] = 0.0;
        }
      }
    }
    return;
  }
  if (A.tiles > 0) {  // exclusive prefix of the tile totals (packed pairs of 32-bit counts add without carries): 4 tiles per thread
    const int t0 = threadIdx.x * 4;
    unsigned long long v[4], run = 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = (t0 + u < A.tiles) ? A.totals[t0 + u] : 0ull; run += v[u]; }
    unsigned long long incl = run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned long long pre = incl - run;
    for (int w = 0; w < wave; ++w) pre += wtot[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) { toff[t0 + u] = pre; pre += v[u]; }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x < 6 && A.st) A.st->x_build[threadIdx.x] = A.st->x[threadIdx.x];
  const int slot = blockIdx.x * 256 + threadIdx.x;
  const int n_slots = A.sv.slot_off[kKinds];
  if (slot >= n_slots) return;
  int kind = 0;
#pragma unroll
  for (int k = 1; k < kKinds; ++k) kind += (slot >= A.sv.slot_off[k]) ? 1 : 0;
  const unsigned long long f = A.sv.flags[slot];
  auto scanned = [&](int i) { return A.sv.scan[i] + (A.tiles > 0 ? toff[i / kScanTile] : 0ull); };
  const unsigned long long sc = scanned(slot);
  const unsigned long long sn = scanned(slot + 1);  // the scan has n_slots + 1 entries
  const unsigned long long sb = scanned(A.sv.slot_off[kind]);
  long long C = (long long)((sc & 0xffffffffull) - (sb & 0xffffffffull));
  long long Cn = (long long)((sn & 0xffffffffull) - (sb & 0xffffffffull));
  const int V = (int)((sc >> 32) - (sb >> 32));
  const int Vn = (int)((sn >> 32) - (sb >> 32));
  if (A.rank_counts) {
    double off = 0.0;
    for (int r = 0; r < A.rank; ++r) off += A.rank_counts[r * kKinds + kind];
    C += (long long)off;
    Cn += (long long)off;
  }
  const long long maxnum = (long long)A.maxnum[kind];
  // number of factors of this kind = V just past the LAST slot whose C is still below the cap;
  // exactly one slot per kind satisfies this (no atomics)
  const bool last_slot = (slot + 1 == A.sv.slot_off[kind + 1]);
  if (C < maxnum && (last_slot || Cn >= maxnum)) A.seg_n[kind] = Vn;
  if ((f >> 32) == 0ull) return;  // not valid
  if (C >= maxnum) return;
  const CorrSeg& seg = A.cv.k[kind];
  if (V >= seg.cap) return;  // cannot happen (cap >= min(n, maxnum)); defensive
  const int pos = V;
  const int local = slot - A.sv.slot_off[kind];
  seg.idx[pos] = local + A.sv.src_lo[kind];
  seg.px[pos] = A.sv.sx[slot]; seg.py[pos] = A.sv.sy[slot]; seg.pz[pos] = A.sv.sz[slot];
  const double2* q = reinterpret_cast<const double2*>(A.sv.raw + (size_t)slot * 8);
  const double2 q0 = q[0], q1 = q[1];
  seg.ax[pos] = q0.x; seg.ay[pos] = q0.y; seg.az[pos] = q1.x;
  if (kind == TLOAM_KIND_EDGE) {
    const double2 q2 = q[2];
    seg.bx[pos] = q1.y; seg.by[pos] = q2.x; seg.bz[pos] = q2.y;
  }
  if (kind <= TLOAM_KIND_GROUND) seg.d[pos] = q[3].x;
  seg.w[pos] = A.sv.w_src[slot];  // weight captured by value at construction (registration.hpp:51,76,96)
  seg.cost[pos] = 0.0;            // fresh side-channel slot (registration.cpp:1118-1121)
}
// ---- small single-rank frames: ONE launch prepares the factor set of an outer iteration --------------------------
// A KITTI-size frame is a chain of launches that each cost more to start than to run, and the device-driven outer loop
// enqueues BOTH alternatives of every iteration behind device flags: builders + flag scan + compaction if the pose
// moved, refresh if it did not -- three launches that exit at once for every one that works.  Here the flag scan is
// folded into the compaction (every block belongs to one kind and sums that kind's flags in front of it itself: at most
// ~16 loads per thread, all in flight) and the refresh shares the launch: `run_build` -> scan + caps + compaction,
// else `run_refresh` -> refresh, else nothing.  Same factor lists, same records (the prefix sums are integers).
struct PrepareArgs {
  SlotView sv;
  CorrView cv;
  int maxnum[kKinds];
  int blk_off[kKinds + 1];   // first block of every kind (blocks of 256 slots, per kind)
  int* seg_n;
  GnState* st;
  const int* run_build;      // null: always compact
  const int* run_refresh;
};
__global__ __launch_bounds__(256) void k_prepare_small(PrepareArgs A) {
  __shared__ unsigned long long wsum[4];
  __shared__ unsigned long long wpre[4];
  const bool build = !A.run_build || *A.run_build != 0;
  if (!build) {
    if (A.run_refresh && *A.run_refresh != 0) {   // the set of the previous iteration, new captured weights, zeroed slots
      const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
      for (int k = 0; k < kKinds; ++k) {
        const int n = A.cv.seg_n[k];
        const CorrSeg& seg = A.cv.k[k];
        for (int i = tid; i < n; i += stride) {
          seg.w[i] = A.sv.w_src[A.sv.slot_off[k] + (seg.idx[i] - A.sv.src_lo[k])];
          seg.cost[i] = 0.0;
        }
      }
    }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x < 6 && A.st) A.st->x_build[threadIdx.x] = A.st->x[threadIdx.x];
  int kind = 0;
#pragma unroll
  for (int k = 1; k < kKinds; ++k) kind += ((int)blockIdx.x >= A.blk_off[k]) ? 1 : 0;
  const int chunk = (int)blockIdx.x - A.blk_off[kind];
  const int base = A.sv.slot_off[kind], nk = A.sv.slot_off[kind + 1] - base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // packed (counted | valid << 32) flags: own slot + this thread's column of every earlier chunk of the kind
  const int local = chunk * 256 + (int)threadIdx.x;
  const unsigned long long f = local < nk ? A.sv.flags[base + local] : 0ull;
  unsigned long long before = 0ull;
  for (int c0 = 0; c0 < chunk; c0 += 8) {
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (c0 + u < chunk) ? A.sv.flags[base + (c0 + u) * 256 + (int)threadIdx.x] : 0ull;
#pragma unroll
    for (int u = 0; u < 8; ++u) before += v[u];
  }
  // block totals of `before` (the kind's flags in front of this block) and the exclusive scan of the block's own flags
  unsigned long long incl = f;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wpre[wave] = before;
  __syncthreads();
  unsigned long long pre = (wpre[0] + wpre[1]) + (wpre[2] + wpre[3]);
  for (int w = 0; w < wave; ++w) pre += wsum[w];
  const unsigned long long sc = pre + incl - f, sn = pre + incl;
  if (local >= nk) return;
  const long long C = (long long)(sc & 0xffffffffull), Cn = (long long)(sn & 0xffffffffull);
  const int V = (int)(sc >> 32), Vn = (int)(sn >> 32);
  const long long maxnum = (long long)A.maxnum[kind];
  // number of factors of this kind = V just past the LAST slot whose C is still below the cap (see k_compact)
  const bool last_slot = (local + 1 == nk);
  if (C < maxnum && (last_slot || Cn >= maxnum)) A.seg_n[kind] = Vn;
  if ((f >> 32) == 0ull) return;  // not valid
  if (C >= maxnum) return;
  const CorrSeg& seg = A.cv.k[kind];
  if (V >= seg.cap) return;
  const int slot = base + local, pos = V;
  seg.idx[pos] = local + A.sv.src_lo[kind];
  seg.px[pos] = A.sv.sx[slot]; seg.py[pos] = A.sv.sy[slot]; seg.pz[pos] = A.sv.sz[slot];
  const double2* q = reinterpret_cast<const double2*>(A.sv.raw + (size_t)slot * 8);
  const double2 q0 = q[0], q1 = q[1];
  seg.ax[pos] = q0.x; seg.ay[pos] = q0.y; seg.az[pos] = q1.x;
  if (kind == TLOAM_KIND_EDGE) {
    const double2 q2 = q[2];
    seg.bx[pos] = q1.y; seg.by[pos] = q2.x; seg.bz[pos] = q2.y;
  }
  if (kind <= TLOAM_KIND_GROUND) seg.d[pos] = q[3].x;
  seg.w[pos] = A.sv.w_src[slot];  // weight captured by value at construction (registration.hpp:51,76,96)
  seg.cost[pos] = 0.0;            // fresh side-channel slot (registration.cpp:1118-1121)
}
bool prepare_small_fits(const SlotView& sv) { return sv.slot_off[kKinds] <= 16384; }
void launch_prepare_small(const SlotView& sv, const CorrView& cv, const BuildParams& bp, int* seg_n, GnState* st,
                          const int* run_build, const int* run_refresh, hipStream_t s) {
  PrepareArgs A;
  A.sv = sv;
  A.cv = cv;
  int blocks = 0;
  for (int k = 0; k < kKinds; ++k) {
    A.maxnum[k] = bp.maxnum[k];
    A.blk_off[k] = blocks;
    blocks += (sv.slot_off[k + 1] - sv.slot_off[k] + 255) / 256;
  }
  A.blk_off[kKinds] = blocks;
  A.seg_n = seg_n;
  A.st = st;
  A.run_build = run_build;
  A.run_refresh = run_refresh;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_prepare_small, dim3(blocks), dim3(256), 0, s, A);
}
void launch_compact(const SlotView& sv, const CorrView& cv, const BuildParams& bp, int* seg_n,
                    const double* rank_counts, int rank, int nranks, GnState* st, hipStream_t s, const int* gate,
                    const int* refresh_gate, const unsigned long long* totals, int tiles) {
  const int n = sv.slot_off[kKinds];
  if (n <= 0) return;
  CompactArgs A;
  A.refresh_gate = refresh_gate;
  A.totals = totals;
  A.tiles = tiles;
  A.sv = sv;
  A.cv = cv;
  for (int k = 0; k < kKinds; ++k) A.maxnum[k] = bp.maxnum[k];
  A.seg_n = seg_n;
  A.rank_counts = rank_counts;
  A.rank = rank;
  A.nranks = nranks;
  A.st = st;
  A.gate = gate;
  hipLaunchKernelGGL(k_compact, dim3((n + 255) / 256), dim3(256), 0, s, A);
}
// Outer iteration with an unchanged pose: the factor list is the previous one.  What a fresh
// AddResidualBlock pass would change is only the weight captured by value (registration.hpp:51,76,96) and
// the zeroed residual slot (registration.cpp:1118-1121).
__global__ __launch_bounds__(256) void k_refresh(SlotView sv, CorrView cv, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int tid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    const int n = cv.seg_n[k];
    const CorrSeg& seg = cv.k[k];
    for (int i = tid; i < n; i += stride) {
      seg.w[i] = sv.w_src[sv.slot_off[k] + (seg.idx[i] - sv.src_lo[k])];
      seg.cost[i] = 0.0;
    }
  }
}
void launch_refresh(const SlotView& sv, const CorrView& cv, hipStream_t s, const int* gate) {
  int cap = 0;
  for (int k = 0; k < kKinds; ++k) cap += cv.k[k].cap;
  int blocks = (cap + 255) / 256;
  blocks = std::max(1, std::min(blocks, 2048));
  hipLaunchKernelGGL(k_refresh, dim3(blocks), dim3(256), 0, s, sv, cv, gate);
}

// per-rank `counted` totals -> row `rank` of a zeroed [nranks*4] buffer (summed by the all-reduce)
__global__ void k_rank_counts(SlotView sv, double* rank_counts, int rank, int nranks) {
  const int t = threadIdx.x;
  if (t < nranks * kKinds) rank_counts[t] = 0.0;
  __syncthreads();
  if (t < kKinds) {
    const unsigned long long a = sv.scan[sv.slot_off[t]];
    const unsigned long long b = sv.scan[sv.slot_off[t + 1]];
    rank_counts[rank * kKinds + t] = (double)((b & 0xffffffffull) - (a & 0xffffffffull));
  }
}
void launch_rank_counts(const SlotView& sv, double* rank_counts, int rank, int nranks, hipStream_t s) {
  hipLaunchKernelGGL(k_rank_counts, dim3(1), dim3(64), 0, s, sv, rank_counts, rank, nranks);
}

// ================================================================================================
//  generic hybrid search (tloam_knn) and getFitnessScore (registration.cpp:257-296)
// ================================================================================================
template <int K>
__global__ __launch_bounds__(256) void k_knn(GridView g, const double* __restrict__ qx,
                                             const double* __restrict__ qy, const double* __restrict__ qz,
                                             int nq, double radius, int k, int* __restrict__ out_idx,
                                             double* __restrict__ out_d2, int* __restrict__ out_cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  TopK<K> tk;
  knn_grid<K>(g, qx[i], qy[i], qz[i], tk);
  const double r2 = radius * radius;
  int cnt = 0;
#pragma unroll
  for (int m = 0; m < K; ++m) {
    const bool in = (m < k) && (tk.d[m] < r2);
    cnt += in ? 1 : 0;
    if (m < k) {
      out_idx[(size_t)i * k + m] = in ? (int)__double_as_longlong(g.gp[tk.j[m]].w) : -1;
      out_d2[(size_t)i * k + m] = in ? tk.d[m] : 0.0;
    }
  }
  out_cnt[i] = cnt;
}
void launch_knn(const GridView& g, const double* qx, const double* qy, const double* qz, int nq, double radius,
                int k, int* out_idx, double* out_d2, int* out_cnt, hipStream_t s) {
  if (nq <= 0) return;
  const dim3 grid((nq + 255) / 256), block(256);
  // top-k of exactly k: instantiate the sizes the path uses plus the test sizes
  if (k == 1) hipLaunchKernelGGL(k_knn<1>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k <= 5 && k == 5) hipLaunchKernelGGL(k_knn<5>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k == 2) hipLaunchKernelGGL(k_knn<2>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k == 3) hipLaunchKernelGGL(k_knn<3>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k == 4) hipLaunchKernelGGL(k_knn<4>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k == 6) hipLaunchKernelGGL(k_knn<6>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else if (k == 7) hipLaunchKernelGGL(k_knn<7>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
  else hipLaunchKernelGGL(k_knn<8>, grid, block, 0, s, g, qx, qy, qz, nq, radius, k, out_idx, out_d2, out_cnt);
}

// fitness: per block (sum of squared distances of hits, number of hits); fixed grid, fixed tree
__global__ __launch_bounds__(256) void k_fitness(GridView g, const double* __restrict__ qx,
                                                 const double* __restrict__ qy, const double* __restrict__ qz,
                                                 int nq, double radius, int reach, double* __restrict__ partial) {
  __shared__ double red[4][2];
  double err = 0.0, hits = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
    TopK<1> tk;
    // :271-272 raw scan-frame point, k = 1; the structure was built for the builders' radius, a larger
    // fitness_thres widens the walk
    if (reach <= 1) knn_grid<1>(g, qx[i], qy[i], qz[i], tk);
    else knn_grid_reach<1>(g, qx[i], qy[i], qz[i], reach, tk);
    if (tk.d[0] < radius * radius) { err += tk.d[0]; hits += 1.0; }  // :273 adds the SQUARED distance
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    err += __shfl_down(err, off, 64);
    hits += __shfl_down(hits, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = err; red[wave][1] = hits; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v += red[w][threadIdx.x];
    partial[blockIdx.x * 2 + threadIdx.x] = v;
  }
}
void launch_fitness(const GridView& g, const double* qx, const double* qy, const double* qz, int nq,
                    double radius, double* partial, int blocks, hipStream_t s) {
  // every target within `radius` of a query lies within `reach` cells of the query's cell in each direction
  // (clamped: beyond 2^20 cells the walk covers the whole grid anyway)
  const double cells = radius * g.inv_cell;
  const int reach = cells < 1.0 ? 1 : (cells > 1048576.0 ? 1048576 : (int)cells + 1);
  hipLaunchKernelGGL(k_fitness, dim3(blocks), dim3(256), 0, s, g, qx, qy, qz, nq, radius, reach, partial);
}

}  // namespace tl
