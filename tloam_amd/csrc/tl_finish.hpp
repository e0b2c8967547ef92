// tl_finish.hpp -- end of an outer GNC iteration on the device: weight update (registration.cpp:858-876), cost sums
// (:1091-1094), the loop decisions of :1108-1121 and the result slot.  Device code shared by tl_gn.hip (the large-set
// finish kernels) and tl_nn.hip (the small-set finish, stand-alone and riding on the correspondence search of the next
// iteration -- both in ONE translation unit so that they round alike).
#pragma once

#include "tl_common.hpp"

namespace tl {

struct WeightArgs {
  CorrView cv;
  SlotView sv;
  WeightParams wp;
};
// sums16 = [kind_cost x4, n_corr x4 (as doubles), bad, 0...]; all-reduced by the host when sharded.
// One lane per partial row (blocks == 64), fixed shuffle tree.
// publish the outer iteration's sums into the state and re-arm the minimiser for the next ceres::Solve
// (the pose, hence T_cur, is already exp(x) after a Solve) -- saves the separate init launch
// Host mirror (HostMirror, tl_common.hpp).  Called by every thread of the (single) block once the block's own
// state writes are done: ONE store instruction of one wave writes the three 64-byte segments of the slot, each seven
// words of the host-visible prefix and, last, the sequence number XORed with those seven words.  No fence; the speed
// relies on "an aligned 64-byte segment written by one instruction leaves the GPU as one PCIe write" (a platform
// property, not a language guarantee), the CORRECTNESS does not: the host accepts a segment only when the XOR of its
// eight words equals the number it waits for (tlh::wait_segment), so a torn segment simply reads as "not there yet".
// status >= 0 replaces the `incomplete` word of the copy (a launch that was gated off reports OS_SKIPPED without
// touching the state itself).
__device__ __forceinline__ void mirror_wave(const GnState* st, const HostMirror& hm, int tid, int status = -1);
__device__ __forceinline__ void mirror_to_host(const GnState* st, const HostMirror& hm, int tid, int nthreads, int status = -1) {
  if (!hm.out) return;
  __syncthreads();
  mirror_wave(st, hm, tid, status);
}
// the store itself, by lanes 0..23 of ONE wave (tid = lane); st may be the state's image in LDS
__device__ __forceinline__ void mirror_wave(const GnState* st, const HostMirror& hm, int tid, int status) {
  if (!hm.out) return;
  if (tid >= 24) return;  // one store instruction of one wave: 3 segments x (7 words + sequence number), see MirrorSlot
  const int seg = tid >> 3, pos = tid & 7, word = seg * 7 + pos;
  unsigned long long w = 0ull;
  if (pos < 7) {
    w = word < kMirrorWords ? reinterpret_cast<const unsigned long long*>(st)[word] : 0ull;
    constexpr int kStatusWord = (int)(offsetof(GnState, incomplete) / 8);
    static_assert(offsetof(GnState, incomplete) % 8 == 4, "incomplete is the high half of its word");
    if (status >= 0 && word == kStatusWord) w = (w & 0xffffffffull) | ((unsigned long long)(unsigned)status << 32);
  }
  // last word of the segment = check_mix(sequence number) XOR seg_word of the segment's seven payload words: the host accepts a
  // segment only when its check word agrees with the payload it reads, so a torn segment reads as "not there yet"
  unsigned long long x = pos < 7 ? seg_word(w, pos) : 0ull;
  x ^= __shfl_xor(x, 1, 64);
  x ^= __shfl_xor(x, 2, 64);
  x ^= __shfl_xor(x, 4, 64);
  if (pos == 7) w = check_mix(hm.seq) ^ x;
  __hip_atomic_store(&hm.out->w[tid], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ctl.fast: the outer loop is driven from the device (every outer iteration of the frame is already enqueued) -- the
// plateau test of registration.cpp:1108 and the "did the pose move" comparison that selects build or refresh are made
// here, and the gates of the next iteration's launches are set accordingly; once the loop has ended `done` stays 1
// (sweeps and steps are no-ops) and both gates are 0.
__device__ __forceinline__ void publish_and_rearm(const double* sums16, GnState* st, int t, const OuterCtl& ctl) {
  if (t < 4) {
    st->kind_cost[t] = sums16[t];
    st->n_corr[t] = (int)sums16[4 + t];
  }
  if (t == 0) {
    st->bad_weights += (int)sums16[8];
    st->incomplete = OS_OK;
    st->fin_valid = 0;
    if (ctl.fast) st->next_outer += 1;
    if (!ctl.fast) {
      arm_solver(*st);
    } else {
      const double cur = sums16[TLOAM_KIND_PLANAR];
      if (fabs(cur - st->prev_planar) < ctl.cost_threshold) {   // :1108 (prev = +inf in the first iteration)
        st->incomplete = OS_PLATEAU;
        __hip_atomic_store(&st->stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st->run_build = st->run_refresh = 0;
      } else {
        st->prev_planar = cur;                                   // :1113-1116
        if (ctl.last) {
          __hip_atomic_store(&st->stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          st->run_build = st->run_refresh = 0;
        } else {
          bool moved = false;
#pragma unroll
          for (int i = 0; i < 6; ++i) moved = moved || (st->x[i] != st->x_build[i]);
          st->run_build = moved ? 1 : 0;
          st->run_refresh = moved ? 0 : 1;
          if (ctl.direct && moved) {   // the rebuild runs beside this finish (same launch) or behind it: it is built at this pose
#pragma unroll
            for (int i = 0; i < 6; ++i) st->x_build[i] = st->x[i];
          }
          arm_solver(*st);
        }
      }
      // direct set, the loop has just ended in a launch whose search ran on the Solve's own verdict (spec_build): the rows hold
      // the geometry of a set that will never be solved -- the host rebuilds them at x_build before anybody reads them
      if (ctl.direct == 2 && st->stop == 1 && st->spec_build) st->incomplete |= (int)OS_SET_STALE;
    }
    if (st->comm_error) {   // an in-launch hand-over timed out (tagged rows of the fused GN iteration / a mailbox exchange)
      st->incomplete = OS_COMM_ERROR;
      if (ctl.fast) {
        __hip_atomic_store(&st->stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st->run_build = st->run_refresh = 0;
      }
    }
  }
}
// entry of a finish kernel: 0 go on | 1 gated off (loop ended earlier) | 2 the Solve has not terminated
__device__ __forceinline__ int finish_gate(GnState* gate, const OuterCtl& ctl, int t) {
  const int stop0 = ctl.fast ? gate->stop : 0, done0 = gate->done;
  __syncthreads();  // every thread has read the flags before thread 0 changes them
  if (stop0) return 1;
  if (!done0) {
    if (t == 0) {
      gate->incomplete = OS_INCOMPLETE;
      if (ctl.fast) { gate->stop = 2; gate->run_build = gate->run_refresh = 0; }
    }
    return 2;
  }
  return 0;
}
// Small sets on one rank (KITTI caps: <= 5.9 k factors): weight update, cost sums, publish and re-arm in ONE
// launch of one 1024-thread block -- the frame is a chain of launch-latency-bound kernels, every boundary
// removed is ~4 us.  Same per-element arithmetic as k_weights; the sums are accumulated thread-strided and
// folded by a fixed tree (deterministic, though not the 64-block order of the two-kernel path).
// Argument order: state, sizes, output and mirror first (preloaded SGPRs, see k3_accumulate).  The kernel is one
// dependent chain of memory round trips on an otherwise idle GPU, so everything it will need is REQUESTED in its
// first instructions: the loop flags (done / stop), the segment sizes and -- speculatively, bounded by the segment
// capacities from the kernel arguments -- the first three (cost, index) pairs of every kind per thread (a KITTI-cap
// set is 2500 / 2000 / 1200 / 200: all of it); the gate is evaluated when they are all back.
// The arithmetic is that of 1024 logical threads in sixteen waves; two shapes run it and agree bit for bit:
//   * ONE 1024-thread block (k_weights_finish_small): the waves' sums meet in LDS;
//   * SIXTEEN 64-thread blocks at the head of k_build_finish_small (the finish riding on the correspondence search of
//     the next iteration): each block is one of the waves, hands its five sums over with device-scope stores and takes a
//     ticket, the last one adds the sixteen rows in the same order and publishes.  No block writes the state before
//     every block has read the loop flags (the ticket), and every block sees the same flags.
struct FinishRide {
  double* rows;   // [16][8] hand-over rows (device memory)
  int* ticket;    // zero between launches
};
// per-thread part: the gate (0 go on | 1 gated off: the loop ended earlier | 2 the Solve has not terminated), the
// weights of this thread's factors and -- on lane 0 of every wave -- the wave's five sums
__device__ __forceinline__ int finish_small_thread(const GnState* st, const int* __restrict__ seg_n, const OuterCtl& ctl,
                                                   const WeightArgs& A, int tid, double v[5], int nseg[kKinds]) {
  constexpr int kPre = 3;
  double pc[kKinds][kPre];
  int pi[kKinds][kPre];
#pragma unroll
  for (int k = 0; k < kKinds; ++k)
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int i = tid + u * 1024;
      const bool in = i < A.cv.k[k].cap;
      pc[k][u] = in ? A.cv.k[k].cost[i] : 0.0;
      pi[k][u] = in ? A.cv.k[k].idx[i] : 0;
    }
#pragma unroll
  for (int k = 0; k < kKinds; ++k) nseg[k] = seg_n[k];
  const int stop0 = ctl.fast ? st->stop : 0, done0 = st->done;
#pragma unroll
  for (int i = 0; i < 5; ++i) v[i] = 0.0;
  if (stop0) return 1;
  if (!done0) return 2;
  double sum[kKinds] = {0, 0, 0, 0};
  double bad = 0.0;
  // (the side-channel costs and the index lists are only read here, the slot weights only written: say so, or the
  //  possible aliasing serialises the thread's load -> load -> store chains)
  double* __restrict__ w_src = A.sv.w_src;
  auto one = [&](int k, double c, int id) {
    sum[k] += c;
    if (!A.wp.active[k]) return;
    if (c == 0) return;                            // :862
    double w;
    if (c >= A.wp.th1) w = 0.0;                    // :865
    else if (c <= A.wp.th2) w = 1.0;               // :867
    else {
      w = sqrt(A.wp.noise_bound_sq * A.wp.mu * (A.wp.mu + 1) / c) - A.wp.mu;  // :870
      if (!(w >= 0.0 && w <= 1.0)) bad += 1.0;     // the reference asserts here (:871)
    }
    w_src[A.sv.slot_off[k] - A.sv.src_lo[k] + id] = w;
  };
#pragma unroll
  for (int k = 0; k < kKinds; ++k) {
    const int n = nseg[k];
#pragma unroll
    for (int u = 0; u < kPre; ++u)   // same element order per thread as a plain strided loop
      if (tid + u * 1024 < n) one(k, pc[k][u], pi[k][u]);
    const double* __restrict__ cost = A.cv.k[k].cost;
    const int* __restrict__ idx = A.cv.k[k].idx;
    for (int i = tid + kPre * 1024; i < n; i += 1024) one(k, cost[i], idx[i]);
  }
  v[0] = sum[0]; v[1] = sum[1]; v[2] = sum[2]; v[3] = sum[3]; v[4] = bad;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
  return 0;
}
// The weight update of ONE factor by the wave that holds it (the one-launch Solve ending its outer iteration itself,
// SolveFinish): what `one()` above stores into the slot weight, from the factor's own side-channel cost.  Returns the weight the
// factor captures at the next (refresh) iteration (w_old where the reference leaves the weight alone).
__device__ __forceinline__ double refreshed_weight(const WeightParams& wp, int kind, double c, double w_old, double* __restrict__ w_src,
                                                   int slot) {
  if (!wp.active[kind]) return w_old;
  if (c == 0) return w_old;                        // :862
  double w;
  if (c >= wp.th1) w = 0.0;                        // :865
  else if (c <= wp.th2) w = 1.0;                   // :867
  else w = sqrt(wp.noise_bound_sq * wp.mu * (wp.mu + 1) / c) - wp.mu;  // :870
  w_src[slot] = w;
  return w;
}
// what a gated-off launch leaves behind (thread t of the block that publishes; every thread has read the flags)
__device__ __forceinline__ void finish_gate_writes(GnState* gate, const OuterCtl& ctl, int g, int t) {
  if (g == 2 && t == 0) {
    gate->incomplete = OS_INCOMPLETE;
    if (ctl.fast) {
      __hip_atomic_store(&gate->stop, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (read by the riding search)
      gate->run_build = gate->run_refresh = 0;
    }
  }
}
// the sixteen wave sums (red[wave][0..4]) -> sums16, the state, the result slot; t = thread of the publishing block
__device__ __forceinline__ void finish_small_publish(GnState* st, double* __restrict__ sums16, const HostMirror& hm,
                                                     const OuterCtl& ctl, const double (*red)[8], double* sh /*[16] LDS*/,
                                                     const int nseg[kKinds], int t, int nthreads) {
  if (t < 16) sh[t] = 0.0;
  __syncthreads();
  if (t < 5) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w][t];
    // a Solve that ran as ONE launch has added the costs of its last evaluation up itself, in ITS order (k_solve_all:
    // post_ext_row); its finish -- in that launch or here -- publishes those sums, so that the device-driven loop and the
    // stepwise API agree bit for bit
    if (st->fin_valid) s = t < 4 ? st->fin_sum[t] : st->fin_bad;
    sh[t < 4 ? t : 8] = s;
  }
  if (t >= 8 && t < 12) sh[t - 4] = (double)nseg[t - 8];
  __syncthreads();
  if (t < 16) sums16[t] = sh[t];
  if (t < 64) publish_and_rearm(sh, st, t, ctl);
  mirror_to_host(st, hm, t, nthreads);
}
__device__ __forceinline__ void weights_finish_small_body(GnState* st, const int* __restrict__ seg_n,
                                                          double* __restrict__ sums16, const HostMirror& hm,
                                                          const OuterCtl& ctl, const WeightArgs& A) {
  __shared__ double red[16][8];
  __shared__ double sh[16];
  double v[5];
  int nseg[kKinds];
  const int g = finish_small_thread(st, seg_n, ctl, A, (int)threadIdx.x, v, nseg);
  __syncthreads();  // every thread has read the flags before thread 0 changes them
  if (g != 0) {
    finish_gate_writes(st, ctl, g, (int)threadIdx.x);
    mirror_to_host(st, hm, threadIdx.x, 1024, g == 1 ? (int)OS_SKIPPED : -1);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) red[wave][i] = v[i];
  }
  finish_small_publish(st, sums16, hm, ctl, red, sh, nseg, (int)threadIdx.x, 1024);
}
// block `wave` (0..15, 64 threads) of the riding shape
__device__ __forceinline__ void weights_finish_small_ride(GnState* st, const int* __restrict__ seg_n,
                                                          double* __restrict__ sums16, const HostMirror& hm,
                                                          const OuterCtl& ctl, const WeightArgs& A, const FinishRide& R,
                                                          int wave) {
  __shared__ double red[16][8];
  __shared__ double sh[16];
  __shared__ int s_last;
  const int lane = threadIdx.x;
  double v[5];
  int nseg[kKinds];
  const int g = finish_small_thread(st, seg_n, ctl, A, wave * 64 + lane, v, nseg);
  if (lane == 0 && g == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) __hip_atomic_store(R.rows + wave * 8 + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the row has been acknowledged by the coherence point (and the flags have been read) before the ticket is taken: an
  // explicit wait for the wave's write-through stores -- a workgroup-scope release emits none on gfx950 (see k3_take_ticket)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) s_last = (__hip_atomic_fetch_add(R.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 15) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  if (lane == 0) __hip_atomic_store(R.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  if (g != 0) {
    finish_gate_writes(st, ctl, g, lane);
    mirror_to_host(st, hm, lane, 64, g == 1 ? (int)OS_SKIPPED : -1);
    return;
  }
  for (int i = lane; i < 16 * 8; i += 64)
    red[i >> 3][i & 7] = ((i & 7) < 5) ? __hip_atomic_load(R.rows + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  finish_small_publish(st, sums16, hm, ctl, red, sh, nseg, lane, 64);
}

// ---- large sets: the finish of outer iteration k-1 riding on the thread-per-query search of iteration k -------------------------
// k_weights (wblocks blocks of 256 threads) + k_outer_finish (one wave) as 4 * wblocks ONE-WAVE blocks at the head of the search
// launch: block b is wave b % 4 of k_weights' block b / 4 -- the same strided elements in the same order, the same shuffle
// tree --, hands its five sums over with device-scope stores and takes a ticket; the last one adds the four waves of every
// block in k_weights' order, the blocks' rows in k_outer_finish's order, and publishes.  Bit-identical to the two kernels
// (tests/test_gpu_parity.py: the device-driven loop against the stepwise API on the 1 M-class frame).
struct FinishRideLarge {
  double* rows;   // [4 * wblocks][8] hand-over rows (device memory)
  int* ticket;    // zero between launches
  int wblocks;    // what launch_weights would have been given (<= 256)
};
__device__ __forceinline__ void weights_finish_large_ride(GnState* st, const int* __restrict__ seg_n, double* __restrict__ sums16,
                                                          const HostMirror& hm, const OuterCtl& ctl, const WeightArgs& A,
                                                          const FinishRideLarge& R, int b) {
  __shared__ double sh[16];
  __shared__ int s_last;
  const int lane = threadIdx.x;
  // the loop flags first: every block reads them before the last one may change them (the ticket)
  const int stop0 = ctl.fast ? st->stop : 0, done0 = st->done;
  const int g = stop0 ? 1 : (!done0 ? 2 : 0);
  double v[5] = {0, 0, 0, 0, 0};
  if (g == 0) {
    double sum[kKinds] = {0, 0, 0, 0};
    double bad = 0.0;
    const int tid = (b >> 2) * 256 + (b & 3) * 64 + lane, stride = R.wblocks * 256;
    double* __restrict__ w_src = A.sv.w_src;
#pragma unroll
    for (int k = 0; k < kKinds; ++k) {
      const int n = seg_n[k];
      const CorrSeg& seg = A.cv.k[k];
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
          w_src[A.sv.slot_off[k] + (iu[u] - A.sv.src_lo[k])] = w;
        }
      }
    }
    v[0] = sum[0]; v[1] = sum[1]; v[2] = sum[2]; v[3] = sum[3]; v[4] = bad;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) __hip_atomic_store(R.rows + (size_t)b * 8 + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (row acknowledged, flags read -- see weights_finish_small_ride)
  if (lane == 0) s_last = (__hip_atomic_fetch_add(R.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 4 * R.wblocks - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  if (lane == 0) __hip_atomic_store(R.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  if (g != 0) {
    finish_gate_writes(st, ctl, g, lane);
    mirror_to_host(st, hm, lane, 64, g == 1 ? (int)OS_SKIPPED : -1);
    return;
  }
  // k_weights' block rows -- ((wave 0 + wave 1) + wave 2) + wave 3 -- added as k_outer_finish adds them: lane t takes rows
  // t, t + 64, ..., then one shuffle tree.  (In registers: static LDS here would be charged to every block of the search.)
  double t5[5] = {0, 0, 0, 0, 0};
  for (int wb = lane; wb < R.wblocks; wb += 64) {
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const double* r = R.rows + (size_t)wb * 32 + c;
      const double r0 = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   r1 = __hip_atomic_load(r + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   r2 = __hip_atomic_load(r + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   r3 = __hip_atomic_load(r + 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t5[c] += ((r0 + r1) + r2) + r3;
    }
  }
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t5[c] += __shfl_down(t5[c], off, 64);
  if (lane < 16) sh[lane] = 0.0;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[c] = t5[c];
    sh[8] = t5[4];
  }
  if (lane >= 4 && lane < 8) sh[lane] = (double)seg_n[lane - 4];
  __syncthreads();
  if (lane < 16) sums16[lane] = sh[lane];
  publish_and_rearm(sh, st, lane, ctl);
  mirror_to_host(st, hm, lane, 64);
}


// ---- the finish of a DIRECT set (tl_common.hpp DirectSet): nblocks one-wave blocks, riding on the search of the next iteration
//      (k_build_finish_large) or as a launch of their own (k_finish_direct) -- the same function, bit for bit -----------------------
// Per row: the side-channel cost into its kind's sum, idx >= 0 counted (the factors: holes carry ~index), updateWeight
// (registration.cpp:858-876) from the cost and the weight the Solve read, written to the OTHER weight stream -- every row, a row the
// reference leaves alone copied.  Hand-over as the riding finish of the compact sets: a row of sums per block with device-scope
// stores, a ticket, the last block adds the rows in a fixed order and publishes.
constexpr int kDirRow = 16;   // [0..3] cost sums, [4] weights out of [0, 1], [8..11] factors per kind
struct FinishDirect {
  double* rows;               // [nblocks][kDirRow], then [groups of 64 blocks][kDirRow]
  int* ticket;                // [1 + groups]: the top ticket, then one per group; zero between launches
  int nblocks;
  double* w_next[kKinds];     // the weight stream this finish writes (cv.k[].w is the one the Solve read)
  const int* blk_cnt;         // [nblk][4] factors per search block of the set this iteration solved (DirectSet::blk_cnt)
  int nblk;
  int built;                  // 1: this iteration built its set | 0: it solved the previous one | < 0: GnState::run_build tells
};
__device__ __forceinline__ void finish_direct_block(GnState* st, const int* __restrict__ seg_n, double* __restrict__ sums16,
                                                    const HostMirror& hm, const OuterCtl& ctl, const WeightArgs& A,
                                                    const FinishDirect& R, int b) {
  __shared__ double sh[16];
  __shared__ int s_last;
  const int lane = threadIdx.x;
  const int stop0 = ctl.fast ? st->stop : 0, done0 = st->done;   // every block reads the flags before the last one may change them
  const int built0 = R.built < 0 ? st->run_build : R.built;      // (likewise: as the finish of the iteration before left it)
  int ncorr0 = 0;
  if (lane < kKinds) ncorr0 = st->n_corr[lane];
  const int g = stop0 ? 1 : (!done0 ? 2 : 0);
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (g == 0) {
    const int tid = b * 64 + lane, stride = R.nblocks * 64;
    if (built0) {   // the factors of the set this iteration built: the search's per-block counts, added up (integers: any order)
      for (int j = tid; j < R.nblk; j += stride) {
        const int4 c4 = *reinterpret_cast<const int4*>(R.blk_cnt + (size_t)j * kKinds);
        v[5] += (double)c4.x; v[6] += (double)c4.y; v[7] += (double)c4.z; v[8] += (double)c4.w;
      }
    }
#pragma unroll
    for (int k = 0; k < kKinds; ++k) {
      const int n = seg_n[k];
      const CorrSeg& seg = A.cv.k[k];
      const double* __restrict__ cost = seg.cost;
      const double* __restrict__ w_cur = seg.w;
      double* __restrict__ w_out = R.w_next[k];
      constexpr int kU = 4;   // (eight at a time measured the same: 26.9 against 26.6 us for the launch of its own, 1 M rows)
      for (int i0 = tid; i0 < n; i0 += kU * stride) {
        double cu[kU], wu[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = i0 + u * stride;
          cu[u] = (i < n) ? cost[i] : 0.0;
          wu[u] = (i < n) ? w_cur[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = i0 + u * stride;
          if (i >= n) break;
          const double c = cu[u];
          v[k] += c;
          double w = wu[u];
          if (A.wp.active[k] && c != 0) {                // :862
            if (c >= A.wp.th1) w = 0.0;                  // :865
            else if (c <= A.wp.th2) w = 1.0;             // :867
            else {
              w = sqrt(A.wp.noise_bound_sq * A.wp.mu * (A.wp.mu + 1) / c) - A.wp.mu;  // :870
              if (!(w >= 0.0 && w <= 1.0)) v[4] += 1.0;  // the reference asserts here (:871)
            }
          }
          w_out[i] = w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) __hip_atomic_store(R.rows + (size_t)b * kDirRow + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 4; ++i) __hip_atomic_store(R.rows + (size_t)b * kDirRow + 8 + i, v[5 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- hand-over in TWO levels: a thousand arrivals on ONE counter are a thousand serialised atomics (~12 ns each: 13 us of the
  //      31 us this finish took as a launch of its own with one ticket) -- groups of 64 blocks take a ticket of their own, the
  //      last block of a group adds the group's rows (one per lane, one shuffle tree), stores a group row and takes the top
  //      ticket; the last group adds the group rows the same way.  Fixed orders throughout.
  const int group = b >> 6, ngroups = (R.nblocks + 63) >> 6, gsize = min(64, R.nblocks - (group << 6));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (row acknowledged, flags read -- see weights_finish_small_ride)
  if (lane == 0) s_last = (__hip_atomic_fetch_add(R.ticket + 1 + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  if (lane == 0) __hip_atomic_store(R.ticket + 1 + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  double t9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double* const grows = R.rows + (size_t)R.nblocks * kDirRow;   // [ngroups][kDirRow] behind the blocks' rows
  if (g == 0) {
    if (lane < gsize) {
      const double* r = R.rows + (size_t)((group << 6) + lane) * kDirRow;
#pragma unroll
      for (int c = 0; c < 5; ++c) t9[c] = __hip_atomic_load(r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int c = 0; c < 4; ++c) t9[5 + c] = __hip_atomic_load(r + 8 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t9[c] += __shfl_down(t9[c], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) __hip_atomic_store(grows + (size_t)group * kDirRow + c, t9[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int c = 0; c < 4; ++c) __hip_atomic_store(grows + (size_t)group * kDirRow + 8 + c, t9[5 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) s_last = (__hip_atomic_fetch_add(R.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  if (lane == 0) __hip_atomic_store(R.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (g != 0) {
    finish_gate_writes(st, ctl, g, lane);
    mirror_to_host(st, hm, lane, 64, g == 1 ? (int)OS_SKIPPED : -1);
    return;
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) t9[c] = 0.0;
  for (int rb = lane; rb < ngroups; rb += 64) {   // (at most 64 groups: one per lane)
    const double* r = grows + (size_t)rb * kDirRow;
#pragma unroll
    for (int c = 0; c < 5; ++c) t9[c] += __hip_atomic_load(r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int c = 0; c < 4; ++c) t9[5 + c] += __hip_atomic_load(r + 8 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t9[c] += __shfl_down(t9[c], off, 64);
  if (lane < 16) sh[lane] = 0.0;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[c] = t9[c];   // kind_cost
    sh[8] = t9[4];
  }
  // n_corr: the counts of the set's own search if this iteration built it, what the state holds otherwise (as doubles: exact)
  const double nc_built = __shfl(t9[5], 0, 64), nc1 = __shfl(t9[6], 0, 64), nc2 = __shfl(t9[7], 0, 64), nc3 = __shfl(t9[8], 0, 64);
  if (lane < kKinds) sh[4 + lane] = built0 ? (lane == 0 ? nc_built : (lane == 1 ? nc1 : (lane == 2 ? nc2 : nc3))) : (double)ncorr0;
  __syncthreads();
  if (lane < 16) sums16[lane] = sh[lane];
  publish_and_rearm(sh, st, lane, ctl);
  mirror_to_host(st, hm, lane, 64);
}

}  // namespace tl
