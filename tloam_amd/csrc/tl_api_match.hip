// tl_api_match.hip -- the host driver of LocalRegistration::scanMatching (registration.cpp:879-1133) behind the C ABI
// (include/tloam_hip.h): stepwise (tloam_sm_*) and device-driven (tloam_scan_match) outer GNC loop, getFitnessScore,
// introspection, pre-built correspondence sets, the bench's timing helpers.
//
// Host side mirrors the reference's control flow (outer GNC loop, mu schedule, plateau test);
// every per-point / per-correspondence computation is a HIP kernel (tl_nn.hip, tl_gn.hip).
#include <atomic>
#include <chrono>

#include "tl_ctx.hpp"

using namespace tl;

namespace {
int sync_stream(tloam_ctx* c) {
  HIPC(c, hipStreamSynchronize(c->stream));
  return TLOAM_OK;
}
}  // namespace

namespace tlh {
int reserve_seg(tloam_ctx* c, int k, size_t n) {
  KindData& K = c->kd[k];
  const size_t cap = round_up(std::max<size_t>(n, 1), kChunk) + kChunk;  // + one chunk: double2 tail reads
  HIPC(c, K.c_idx.reserve(cap));
  if (cap > K.c_stride) {  // (grow-only, like every DBuf; the contents are rewritten by the caller)
    const size_t stride = std::max(cap, K.c_stride + K.c_stride / 2);
    HIPC(c, K.c_buf.reserve(stride * kSegStreams));
    K.c_stride = stride;
  }
  K.c_cap = cap - kChunk;
  CorrSeg& s = c->cv.k[k];
  double* b = K.c_buf.p;
  const size_t st = K.c_stride;
  s.idx = K.c_idx.p;
  s.px = b + SS_PX * st; s.py = b + SS_PY * st; s.pz = b + SS_PZ * st;
  s.ax = b + SS_AX * st; s.ay = b + SS_AY * st; s.az = b + SS_AZ * st;
  s.bx = (k == TLOAM_KIND_EDGE) ? b + SS_BX * st : nullptr;
  s.by = (k == TLOAM_KIND_EDGE) ? b + SS_BY * st : nullptr;
  s.bz = (k == TLOAM_KIND_EDGE) ? b + SS_BZ * st : nullptr;
  s.d = (k <= TLOAM_KIND_GROUND) ? b + SS_D * st : nullptr;
  s.w = b + SS_W * st;
  s.cost = b + SS_COST * st;
  s.cap = (int)K.c_cap;
  s.stride = (int)st;
  return TLOAM_OK;
}
int ensure_common(tloam_ctx* c) {
  HIPC(c, c->state.reserve(1));
  HIPC(c, c->seg_n.reserve(8));
  HIPC(c, c->red48.reserve(kReduceBuf));
  HIPC(c, c->sums16.reserve(16));
  HIPC(c, c->wpart.reserve(256 * 8));
  HIPC(c, c->rank_counts.reserve((size_t)kMaxRanks * kKinds));
  HIPC(c, c->se3_dev.reserve(8));
  if (!c->k3_ticket.p) {
    HIPC(c, c->k3_ticket.reserve(4));
    HIPC(c, hipMemsetAsync(c->k3_ticket.p, 0, 4 * sizeof(int), c->stream));
    {
      // The launch counter of the tagged hand-overs (words 2..3 of the ticket buffer; tl_gn.hip k3_post_row_tagged) starts from a
      // number no other context of this process uses: a context's row buffer can be memory that a context destroyed a moment ago
      // wrote ITS rows into, with valid check words for ITS launch numbers -- which would be this context's first launch numbers
      // too if both counted from zero.  A stepper that looked before the fresh row landed then folded the dead context's sums,
      // the blocks' images of the minimiser disagreed and the launch ran into its bounded wait (round 5: seen as a one-in-two
      // TLOAM_E_HIP in the GPU suite when contexts solving DIFFERENT scenes followed each other through the stepwise API).
      static std::atomic<unsigned long long> serial{0};
      const unsigned long long base = (serial.fetch_add(1ull, std::memory_order_relaxed) + 1ull) << 36;   // 2^36 launches per context
      HIPC(c, hipMemcpyAsync(reinterpret_cast<unsigned long long*>(c->k3_ticket.p + 2), &base, sizeof(base), hipMemcpyHostToDevice, c->stream));
      HIPC(c, hipStreamSynchronize(c->stream));   // (`base` is a local)
    }
    HIPC(c, c->k3_span.reserve(4));     // K3Step::span
    HIPC(c, hipMemsetAsync(c->k3_span.p, 0, 4 * sizeof(unsigned long long), c->stream));
    HIPC(c, c->iter_span.reserve(4));   // iter_span_note
    HIPC(c, hipMemsetAsync(c->iter_span.p, 0, 4 * sizeof(unsigned long long), c->stream));
  }
  c->cv.seg_n = c->seg_n.p;
  return TLOAM_OK;
}
}  // namespace tlh

namespace {

// ---- the direct factor set (tl_common.hpp DirectSet) ------------------------------------------------------------------------
// Which frames take it: one rank, searched one thread per query, all four builders running, and no cap that could bind -- a
// kind's cap >= its source points ("first N valid in index order", registration.cpp:448 / :538 / :592 / :735, then selects
// everything valid, whatever the order).  Everything else compacts.
bool direct_applies(const tloam_ctx* c, size_t n_slots) {
  if (c->no_direct_set || !one_rank(c) || c->cfg.factor_num != 4 || !direct_set_size((int)std::min<size_t>(n_slots, (size_t)INT32_MAX))) return false;
  for (int k = 0; k < kKinds; ++k)
    if ((long long)c->kd[k].n_src > (long long)kind_maxnum(c->cfg, k) || c->kd[k].n_tgt == 0) return false;
  return true;
}
double* direct_w_stream(const tloam_ctx* c, int k, int parity) {
  return c->kd[k].c_buf.p + (size_t)(parity ? SS_W2 : SS_W) * c->kd[k].c_stride;
}
// the Solve of outer iteration `iter` reads weight stream iter & 1 (c->cv), its finish writes the other one
void direct_set_parity(tloam_ctx* c, int iter) {
  for (int k = 0; k < kKinds; ++k) c->cv.k[k].w = direct_w_stream(c, k, iter & 1);
}
// rows of the hand-over buffer of the finish, compact (riding) or direct
constexpr size_t kFinRowsDoubles = (size_t)(4 * 256 + 64) * 16;   // the blocks' rows + the rows of their groups of 64
// (a quarter of the) one-wave blocks of a direct finish: a fixed function of the capacity, so that the riding form and the launch of
// its own cut the rows alike.  (Twice as many blocks on ONE ticket measured 52 instead of 31 us for the frame's last finish: the
// arrivals on the ticket are what it waits for -- hence the two-level hand-over of finish_direct_block.)
int direct_wblocks(const tloam_ctx* c) {
  size_t cap = 0;
  for (int k = 0; k < kKinds; ++k) cap += c->kd[k].c_cap;
  return (int)std::min<size_t>(256, std::max<size_t>(64, cap / 2048));
}
int* direct_blk_cnt(const tloam_ctx* c, int iter) { return c->blk_cnt.p + (size_t)(iter & 1) * c->blk_cnt_n * kKinds; }
// built: 1 the set of outer iteration `iter` was built in it, 0 it is the previous one, -1 the device knows (GnState::run_build)
FinishLargeArgs direct_finish_args(tloam_ctx* c, const CorrView* cv_iter, const WeightParams* wp, const HostMirror& hm, OuterCtl ctl,
                                   int iter, int riding, int built) {
  ctl.direct = riding ? 2 : 1;
  FinishLargeArgs fin{cv_iter, wp, c->seg_n.p, c->sums16.p, hm, ctl, c->fin_rows.p, c->fin_tickets.p, direct_wblocks(c), {}, nullptr, 0, 0};
  for (int k = 0; k < kKinds; ++k) fin.w_next[k] = direct_w_stream(c, k, (iter + 1) & 1);
  fin.blk_cnt = direct_blk_cnt(c, iter);
  fin.nblk = (int)c->blk_cnt_n;
  fin.built = built;
  return fin;
}

// K3 launch; when the bench armed the timer, with a HIP event pair bound to the dispatch itself
// Every kK3SampleStride-th launch carries the pair (stride 3 is coprime to the 5 sweeps of a Solve and the 20 of
// a frame, so over a few frames every position is sampled equally): timing EVERY launch through
// hipExtLaunchKernelGGL cost ~8 % of the 1 M frame.
constexpr int kK3SampleStride = 3;
// the device-side period counter of the GN iterations, once tloam_gn_iter_timer has armed it (null otherwise: the kernels skip it)
unsigned long long* iter_span_of(const tloam_ctx* c) { return c->iter_timing ? c->iter_span.p : nullptr; }
int launch_k3_timed(tloam_ctx* c, bool force) {
  const bool sample = c->k3_timing && (c->k3_seq++ % kK3SampleStride) == 0;
  const int idx = c->batch_launches++;
  if (sample) {
    if (c->ev_used + 2 > c->ev_pool.size()) {
      const size_t old = c->ev_pool.size();
      c->ev_pool.resize(old + 256);
      for (size_t i = old; i < c->ev_pool.size(); ++i) HIPC(c, hipEventCreate(&c->ev_pool[i]));
    }
    launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, force, c->stream, c->ev_pool[c->ev_used],
              c->ev_pool[c->ev_used + 1]);
    c->ev_used += 2;
    c->ev_batch_idx.push_back(idx);
  } else {
    launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, force, c->stream);
  }
  return TLOAM_OK;
}
// the same for the one-launch GN iteration (k3_sweep_step): the pair then brackets sweep + fold + step; the streaming part
// alone is what the kernel's own span counter measures (K3Step::span, read by tloam_k3_timer_span)
int launch_k3_step_timed(tloam_ctx* c) {
  const bool sample = c->k3_timing && (c->k3_seq++ % kK3SampleStride) == 0;
  const int idx = c->batch_launches++;
  const MboxView* mb = (exchanging(c) && c->comm == COMM_MAILBOX) ? &c->mbox : nullptr;
  if (sample) {
    if (c->ev_used + 2 > c->ev_pool.size()) {
      const size_t old = c->ev_pool.size();
      c->ev_pool.resize(old + 256);
      for (size_t i = old; i < c->ev_pool.size(); ++i) HIPC(c, hipEventCreate(&c->ev_pool[i]));
    }
    launch_k3_step(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, c->k3_ticket.p, c->k3_span.p, mb, c->stream,
                   c->ev_pool[c->ev_used], c->ev_pool[c->ev_used + 1], iter_span_of(c));
    c->ev_used += 2;
    c->ev_batch_idx.push_back(idx);
  } else {
    launch_k3_step(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, c->k3_ticket.p, c->k3_span.p, mb, c->stream, nullptr,
                   nullptr, iter_span_of(c));
  }
  return TLOAM_OK;
}
// fold the recorded event pairs into the accumulated timers (stream must be idle).  The launches of the batch belong
// to `nsolve` Solves starting at batch positions start[i]; of each, the first working[i] launches did a sweep, the
// later ones were no-op launches after `done`.
int harvest_k3_events_multi(tloam_ctx* c, int nsolve, const int* start, const int* working) {
  c->batch_launches = 0;
  if (!c->k3_timing) { c->ev_used = 0; c->ev_batch_idx.clear(); return TLOAM_OK; }
  const size_t pairs = c->ev_used / 2;
  for (size_t i = 0; i < pairs; ++i) {
    float ms = 0.f;
    HIPC(c, hipEventElapsedTime(&ms, c->ev_pool[2 * i], c->ev_pool[2 * i + 1]));
    c->k3_all_us += (double)ms * 1e3;
    c->k3_all_launches += 1;
    const int b = c->ev_batch_idx[i];
    int sv = 0;
    while (sv + 1 < nsolve && start[sv + 1] <= b) ++sv;
    if (b - start[sv] < working[sv]) {
      c->k3_total_us += (double)ms * 1e3;
      c->k3_launches += 1;
    }
  }
  c->ev_used = 0;
  c->ev_batch_idx.clear();
  return TLOAM_OK;
}
int harvest_k3_events(tloam_ctx* c, int working) {
  const int zero = 0;
  return harvest_k3_events_multi(c, 1, &zero, &working);
}

// Result of an outer iteration on the host.  With the mirror the finish kernel has been handed
// {pinned state, sequence number}: poll the number (a word in host memory the device writes last); the stream
// is only queried now and then, to notice a failed launch instead of spinning forever.  Otherwise, or if the
// stream drained without the number arriving, copy the state and synchronise.
HostMirror next_mirror(tloam_ctx* c, int slot = 0) {
  HostMirror hm;
  hm.out = c->h_mirror_dev + slot;
  hm.seq = ++c->mirror_seq;
  return hm;
}
int wait_state(tloam_ctx* c, const HostMirror& hm, int slot = 0) {
  const auto t0 = std::chrono::steady_clock::now();
  struct Acc {
    tloam_ctx* c;
    std::chrono::steady_clock::time_point t0;
    ~Acc() { c->wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
  } acc{c, t0};
  if (hm.out) {
    // all three segments of the slot (MirrorSlot), each verified against the number, then the prefix out of them
    const MirrorSlot* ms = c->h_mirror + slot;
    int rc = TLOAM_OK;
    unsigned long long pay[3][7];
    for (int sgm = 0; sgm < 3 && rc == TLOAM_OK; ++sgm) rc = wait_segment(c, &ms->w[sgm * 8], hm.seq, pay[sgm]);
    if (rc < 0) return rc;
    if (rc == TLOAM_OK) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(c->h_state + slot);
      for (int w = 0; w < kMirrorWords; ++w) dst[w] = pay[w / 7][w % 7];
      c->h_state[slot].host_seq = hm.seq;
      return TLOAM_OK;
    }
  }
  HIPC(c, hipMemcpyAsync(c->h_state + slot, c->state.p, sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return TLOAM_OK;
}

// one ceres::Solve on the current correspondence set, device resident: 1 + 4 sweeps at most;
// sweeps after a tolerance exit are no-op launches (GnState.done).
constexpr int kSolveSweeps = 5;  // max_num_iterations 4 -> at most 1 + 4 evaluations per Solve
bool solve_small_path(const tloam_ctx* c) {
  return one_rank(c) && c->k3_single && !c->no_persistent_solve && solve_small_fits(c->k3_grid, c->device_cus);
}
// prep: the launch also prepares the factor set (only with solve_small_path and SlotView::flagb, see self_prepare_path)
// finish: ... and finishes the outer iteration, possibly running the following ones too (SolveFinish; needs prep).
// wp: the weight thresholds of the outer iteration this Solve belongs to (null: a Solve outside scanMatching) -- the
// one-launch Solve adds up the finish sums of its last evaluation for the finish kernel that follows.
int enqueue_solve(tloam_ctx* c, bool armed, int sweeps, const WeightParams* wp = nullptr, const SolvePrep* prep = nullptr,
                  const SolveFinish* finish = nullptr) {
  if (!armed) launch_solve_init(c->state.p, c->stream);  // scan_match re-arms the minimiser in its finish kernel
  if (sweeps > 0 && solve_small_path(c)) {
    // KITTI-size set: the whole Solve (up to `sweeps` evaluations) is one launch (k_solve_all)
    SolveFinish F;
    if (prep && finish) {
      F = *finish;
    } else {
      memset(&F, 0, sizeof(F));
      if (wp) { F.have_wp = 1; F.wp[0] = *wp; }
    }
    F.iter_span = iter_span_of(c);
    const int sabotage = c->dbg_fail_handover > 0 ? (c->dbg_fail_handover--, 1 << 16) : 0;   // test hook, see k_solve_all
    launch_solve_small(c->cv, c->state.p, c->partials.p, c->k3_ticket.p, c->k3_grid, sweeps | sabotage, prep, c->seg_n.p, &F, c->stream);
    c->batch_launches++;
    return TLOAM_OK;
  }
  // One GN iteration = ONE launch whatever the size of the set (round 4): the streaming sweep's last block folds the rows and
  // advances the minimiser (k3_sweep_step); with a mailbox it also posts, gathers and advances -- sweep + exchange + step.
  // RCCL / callback contexts keep sweep | collective | step: the collective is enqueued by the host between two launches.
  const bool one_launch = c->fused_large && (one_rank(c) ? !c->k3_single : c->comm == COMM_MAILBOX);
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    if (one_launch) {
      const int rc = launch_k3_step_timed(c);
      if (rc != TLOAM_OK) return rc;
      continue;
    }
    if (exchanging(c)) {
      // sharded GN iteration = 2 launches (+ the collective): the sweep, whose last block folds the rows into the
      // 48-double buffer (the 42 normal-equation scalars + cost) and -- with the mailbox -- stores it straight into
      // every rank's buffer over xGMI; then the step, which (mailbox) adds the ranks' rows in rank order itself
      K3Fuse fuse;
      memset(&fuse, 0, sizeof(fuse));
      fuse.ticket = c->k3_ticket.p;
      fuse.out48 = c->red48.p;
      if (c->comm == COMM_MAILBOX) fuse.mb = c->mbox;
      launch_k3_fused(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, false, fuse, c->stream);
      c->batch_launches++;
      if (c->comm == COMM_MAILBOX) {
        launch_gn_step_mbox(c->state.p, c->mbox, c->stream, iter_span_of(c));
      } else {
        const int rc = allreduce(c, c->red48.p, kReduceBuf);
        if (rc != TLOAM_OK) return rc;
        launch_gn_step(c->state.p, c->red48.p, c->stream, iter_span_of(c));
      }
    } else if (c->k3_single) {
      // KITTI-size set: one launch per GN iteration (k_sweep_step_small)
      launch_sweep_step_small(c->cv, c->state.p, c->partials.p, c->k3_ticket.p, c->k3_grid, c->stream, iter_span_of(c));
      c->batch_launches++;
    } else {
      const int rc = launch_k3_timed(c, false);
      if (rc != TLOAM_OK) return rc;
      launch_reduce_and_step(c->partials.p, c->k3_grid, c->state.p, c->stream, iter_span_of(c));
    }
  }
  return TLOAM_OK;
}

// the per-block rows of the sweeps (and, below 4096 words, the tagged rows / finish segments of the one-launch Solve): cleared
// when (re)allocated, so that nothing in them ever carries a valid check word that this context did not write
int reserve_partials(tloam_ctx* c) {
  const size_t before = c->partials.cap;
  HIPC(c, c->partials.reserve(std::max<size_t>((size_t)c->k3_grid * kAccStride, 4096)));
  if (c->partials.cap != before) HIPC(c, hipMemsetAsync(c->partials.p, 0, c->partials.cap * sizeof(double), c->stream));
  return TLOAM_OK;
}

double alg_bytes_of(const int n[kKinds]) {
  // plane 72 B, line 88 B, point 64 B per correspondence (fp64 SoA, cost write included)
  return 72.0 * ((double)n[TLOAM_KIND_PLANAR] + (double)n[TLOAM_KIND_GROUND]) + 88.0 * (double)n[TLOAM_KIND_EDGE] +
         64.0 * (double)n[TLOAM_KIND_SPHERE];
}

}  // namespace

extern "C" {

// ---- scanMatching, stepwise ---------------------------------------------------------------------
int tloam_sm_begin(tloam_ctx* c, const double predict[16], const double* omega3) {
  if (!c || !predict) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  for (int k = 0; k < kKinds; ++k)  // the reference asserts (registration.cpp:928-929)
    if (c->kd[k].n_src_full < 10 || c->kd[k].n_tgt < 10) return TLOAM_E_TOO_FEW_POINTS;
  for (int k = 0; k < kKinds; ++k)  // a hand-over that failed half way (staging, upload) left nothing registered
    if (!c->kd[k].src_set || !c->kd[k].tgt_set) { c->last_error = "a source / target hand-over failed: hand the frame over again"; return TLOAM_E_NOT_READY; }
  Pose P;
  if (!pose_from_matrix(predict, &P)) return TLOAM_E_BAD_POSE;  // SOPHUS_ENSURE in the reference
  double x[6];
  se3_log(P, x);  // :881
  if (sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]) < 1e-2) {  // :884-886
    double u[3] = {0.0, 0.0, 1.0};
    if (omega3) {
      const double nn = sqrt(omega3[0] * omega3[0] + omega3[1] * omega3[1] + omega3[2] * omega3[2]);
      if (nn > 0.0) { u[0] = omega3[0] / nn; u[1] = omega3[1] / nn; u[2] = omega3[2] / nn; }
    }
    x[3] = u[0] * 1e-4; x[4] = u[1] * 1e-4; x[5] = u[2] * 1e-4;
  }
  int rc = ensure_common(c);
  if (rc != TLOAM_OK) return rc;
  // ---- per-source-slot arrays (:931-949 weights = 1, residual slots = 0)
  size_t off = 0;
  for (int k = 0; k < kKinds; ++k) {
    c->sv.slot_off[k] = (int)off;
    c->sv.src_lo[k] = (int)c->kd[k].src_lo;
    off += c->kd[k].n_src;
  }
  c->sv.slot_off[kKinds] = (int)off;
  const size_t ns = std::max<size_t>(off, 1);
  HIPC(c, c->sx.reserve(ns)); HIPC(c, c->sy.reserve(ns)); HIPC(c, c->sz.reserve(ns)); HIPC(c, c->w_src.reserve(ns));
  HIPC(c, c->raw.reserve(ns * 8));
  HIPC(c, c->flags.reserve(ns + 1)); HIPC(c, c->scan.reserve(ns + 1));
  HIPC(c, c->scan_tmp.reserve(scan_tmp_elems(ns + 1)));
  c->sv.sx = c->sx.p; c->sv.sy = c->sy.p; c->sv.sz = c->sz.p; c->sv.w_src = c->w_src.p;
  c->sv.raw = c->raw.p;
  c->sv.flags = c->flags.p; c->sv.scan = c->scan.p;
  c->direct = direct_applies(c, off);
  c->set_stale = false;
  c->w_parity = 0;
  if (c->direct) {
    HIPC(c, c->fin_rows.reserve(kFinRowsDoubles));
    c->blk_cnt_n = (off + 63) / 64;   // one-wave blocks of the thread-per-query search (logical: the sorted queries, 64 each)
    HIPC(c, c->blk_cnt.reserve(2 * c->blk_cnt_n * kKinds + 8));
    HIPC(c, c->row_of_pos.reserve(off + 64));
    if (!c->fin_tickets.p) {
      HIPC(c, c->fin_tickets.reserve(128));
      HIPC(c, hipMemsetAsync(c->fin_tickets.p, 0, c->fin_tickets.cap * sizeof(int), c->stream));
    }
  }
  // ---- compact segments: at most min(n_src, maxnum) factors per kind (a direct set: one row per source point, which is the same)
  size_t total_cap = 0;
  for (int k = 0; k < kKinds; ++k) {
    const size_t cap = std::min<size_t>(c->kd[k].n_src, (size_t)std::max(kind_maxnum(c->cfg, k), 0));
    rc = reserve_seg(c, k, cap);
    if (rc != TLOAM_OK) return rc;
    total_cap += round_up(std::max<size_t>(cap, 1), kChunk);
  }
  c->prebuilt = false;
  {
    int caps[kKinds];
    for (int k = 0; k < kKinds; ++k) caps[k] = (int)c->kd[k].c_cap;
    k3_plan(caps, c->device_cus, &c->k3_grid, &c->k3_single, &c->k3_wide);
    (void)total_cap;
  }
  {
    // the one-launch Solve compacts the factor set itself when every kind's flag bytes fit a wave (SlotView::flagb)
    bool fits = solve_small_path(c) && prepare_small_fits(c->sv);
    for (int k = 0; k < kKinds; ++k) fits = fits && c->kd[k].n_src <= (size_t)kFlagbStride;
    c->sv.flagb = nullptr;
    if (fits) {
      HIPC(c, c->flagb.reserve((size_t)kKinds * kFlagbStride));
      c->sv.flagb = c->flagb.p;
    }
  }
  rc = reserve_partials(c);
  if (rc != TLOAM_OK) return rc;
  // ---- the start of the frame -- scan-frame sources AoS -> SoA slots, weights = 1 (:931-949), flag-scan terminator,
  //      minimiser state zeroed with `parameters` = x (passed by value) and armed for the first Solve -- rides on the
  //      first launch of the grid build
  FrameInitHook hook;
  memset(&hook, 0, sizeof(hook));
  for (int k = 0; k < kKinds; ++k) { hook.fi.src_aos[k] = c->kd[k].src_ptr; hook.fi.slot_off[k] = c->sv.slot_off[k]; }
  hook.fi.slot_off[kKinds] = c->sv.slot_off[kKinds];
  for (int i = 0; i < 6; ++i) hook.fi.x[i] = x[i];
  hook.fi.no_eval_reuse = c->dbg_no_eval_reuse ? 1 : 0;
  hook.fi.direct = c->direct ? 1 : 0;
  if (c->direct) direct_set_parity(c, 0);
  // 1 M-class single-rank frames: the first pass of the query sort rides on the last launch of the grid build (QueryBinRide)
  QueryBinRide qbin;
  memset(&qbin, 0, sizeof(qbin));
  c->qbin_rode = false;
  if (one_rank(c) && !c->no_qbin_ride && direct_set_size((int)std::min<size_t>(off, (size_t)INT32_MAX))) {
    qbin.sv = c->sv;
    for (int k = 0; k < kKinds; ++k) {
      qbin.bp.radius[k] = kind_radius(c->cfg, k);
      qbin.bp.maxnum[k] = kind_maxnum(c->cfg, k);
      qbin.bp.active[k] = kind_active(c->cfg, k);
    }
    qbin.bp.edge_dir_thres = c->cfg.edge_dir_thres;
    qbin.st = c->state.p;
    hook.qbin = &qbin;
  }
  hook.b = FrameInitBufs{c->sx.p, c->sy.p, c->sz.p, c->w_src.p, c->flags.p, c->state.p, c->seg_n.p};
  hook.n_slots = c->sv.slot_off[kKinds];
  // ---- :889-915 four search structures over the submap clouds: one launch per build phase for all kinds
  {
    double radius[kKinds];
    GridView views[kKinds];
    for (int k = 0; k < kKinds; ++k) radius[k] = kind_radius(c->cfg, k);
    // a sharded rank searches only the kinds it holds source points of (tloam_shard_ranges_frame): the other grids are not built
    if (c->nranks > 1)
      for (int k = 0; k < kKinds; ++k)
        if (c->kd[k].n_src == 0) radius[k] = 0.0;
    if (c->grids_ahead && c->grids_next_gen == c->tgt_gen && one_rank(c)) {
      // built when the targets were handed over (tloam_set_target_frame): they become the context's search structures now;
      // the frame's start is a launch of its own, below.  Used once: a second scanMatching over the same targets builds its own
      std::swap(c->grids, c->grids_next);
      for (int k = 0; k < kKinds; ++k) { c->kd[k].gv = c->gv_next[k]; c->kd[k].grid_valid = true; }
      c->grids_ahead = false;
    } else {
      rc = build_grids(c, c->grids, radius, views, &hook);
      if (rc != TLOAM_OK) return rc;
      c->qbin_rode = hook.qbin_done;
      // (a kind whose grid was skipped -- radius forced to 0 above: a sharded rank without source points of it -- has an EMPTY
      //  view: it is not a search structure getFitnessScore or anybody else may use)
      for (int k = 0; k < kKinds; ++k) { c->kd[k].gv = views[k]; c->kd[k].grid_valid = radius[k] > 0.0; }
    }
  }
  if (!hook.consumed) {  // (no grid launch: cannot happen with >= 10 targets per kind, kept for safety)
    GridView gviews[kKinds];
    for (int k = 0; k < kKinds; ++k) gviews[k] = c->kd[k].gv;
    const size_t ntiles = (size_t)build_tile_count(gviews, c->sv.slot_off);
    HIPC(c, c->tile_cnt.reserve(ntiles + 1 > c->tile_cnt.cap ? 2 * ntiles + 64 : ntiles + 1));   // (room to spare, as build_grids_over)
    hook.fi.tile_cnt = c->tile_cnt.p;
    hook.fi.n_tile_cnt = (int)ntiles + 1;
    launch_frame_init(hook.fi, hook.b, c->stream);
  }
  if (c->direct) {
    // a kind whose target cloud has no finite point has no search grid: its queries are not in the sorted order, and the rows of a
    // direct set ARE that order -- such a frame compacts (the frame's start, already enqueued, has set seg_n to the row counts)
    bool all = true;
    for (int k = 0; k < kKinds; ++k) all = all && c->kd[k].gv.n > 0;
    if (!all) {
      c->direct = false;
      HIPC(c, hipMemsetAsync(c->seg_n.p, 0, kKinds * sizeof(int), c->stream));
    }
  }
  c->wait_us = 0.0;
  c->mu = 1.0;  // :961
  c->noise_bound_sq = c->cfg.noise_bound * c->cfg.noise_bound;
  if (c->noise_bound_sq < 1e-16) c->noise_bound_sq = 1e-2;  // :963-964
  for (int k = 0; k < kKinds; ++k) { c->prev_cost[k] = INFINITY; c->cur_cost[k] = INFINITY; }  // :952-959
  c->iter = 0;
  c->active = true;
  c->have_build = false;
  memset(&c->stats, 0, sizeof(c->stats));
  memcpy(c->stats.se3, x, sizeof(x));
  c->ev_used = 0;
  c->ev_batch_idx.clear();
  c->batch_launches = 0;
  return TLOAM_OK;
}

// ---- pieces of one outer GNC iteration, shared by the stepwise API (the host decides between iterations) and by
//      tloam_scan_match's device-driven loop (every iteration enqueued at once, one host wait per frame) -----------
namespace {
constexpr int kMaxOuterFast = kMirrorSlots;   // outer iterations the device-driven loop plans for

void outer_params(const tloam_ctx* c, BuildParams* bp, GridView grids[kKinds]) {
  for (int k = 0; k < kKinds; ++k) {
    bp->radius[k] = kind_radius(c->cfg, k);
    bp->maxnum[k] = kind_maxnum(c->cfg, k);
    bp->active[k] = kind_active(c->cfg, k);
    grids[k] = c->kd[k].gv;
  }
  bp->edge_dir_thres = c->cfg.edge_dir_thres;
}
int outer_reserve(tloam_ctx* c, const GridView grids[kKinds]) { return tlh::reserve_query_sort(c, grids); }
}  // namespace
extern "C++" {
namespace tlh {
// (also called from the grid build when the sort's first pass rides on it: the SAME sizes, so that nothing is re-allocated -- and
//  lost -- between that pass and the rest of the sort)
int reserve_query_sort(tloam_ctx* c, const GridView grids[kKinds]) {
  const size_t n_slots = (size_t)c->sv.slot_off[kKinds];
  const size_t ntiles = (size_t)build_tile_count(grids, c->sv.slot_off);
  // (tile counts follow the bounding boxes like the cell tables: grow with room to spare)
  const size_t nt_res = (ntiles + 1 > c->tile_cnt.cap || ntiles + 1 > c->tile_scan.cap) ? 2 * ntiles + 64 : ntiles;
  {
    // The query-tile histogram is zeroed by the frame's start (k_frame_init / the extra blocks of the grid build's first launch),
    // which has been enqueued by now: a histogram that has to be re-allocated HERE -- the scan's table was shorter than the
    // histogram's, or the sizes asked for at the two places differ by the one element that crosses an allocation step -- is
    // zeroed again.  (Until round 5 it was not: the counting sort then ranked the queries on what the new block happened to
    // hold and scattered them out of bounds -- a GPU memory fault or a silently wrong query order on the first large frame of a
    // context, whenever the block was not fresh; found by tests/tools/fuzz_call_order.py.)
    const size_t before = c->tile_cnt.cap;
    HIPC(c, c->tile_cnt.reserve(nt_res + 1));
    if (c->tile_cnt.cap != before)
      HIPC(c, hipMemsetAsync(c->tile_cnt.p, 0, c->tile_cnt.cap * sizeof(unsigned long long), c->stream));
  }
  HIPC(c, c->tile_scan.reserve(nt_res + 1));
  HIPC(c, c->tile_fill.reserve(std::max<size_t>((size_t)nt_res, (size_t)n_slots + 1))  /* rank of every slot inside its tile */); HIPC(c, c->tile_of_slot.reserve(n_slots + 1));
  HIPC(c, c->qrec.reserve(n_slots + 1));
  HIPC(c, c->scan_tmp.reserve(scan_tmp_elems(std::max(nt_res + 1, n_slots + 1))));
  c->scan1p_q_use = !c->no_scan_1p && scan_1p_applies(ntiles + 1, c->device_cus);
  if (c->scan1p_q_use) {
    const size_t before = c->scan1p_q.cap;
    HIPC(c, c->scan1p_q.reserve(scan_1p_ctl_elems(nt_res + 1)));
    if (c->scan1p_q.cap != before) HIPC(c, hipMemsetAsync(c->scan1p_q.p, 0, c->scan1p_q.cap * sizeof(unsigned long long), c->stream));
  }
  return TLOAM_OK;
}
}  // namespace tlh
}  // extern "C++"
namespace {
// :976-1020 the four builders (K1 + K2), the flag scan, the index-order caps.  Small single-rank frames: the scan, the
// caps, the compaction AND the alternative (refresh) are one launch (k_prepare_small) -- `also_refresh` says whether this
// call stands for both alternatives of a device-gated iteration.
bool prepare_small_path(const tloam_ctx* c) { return one_rank(c) && prepare_small_fits(c->sv); }
// the Solve launch that follows prepares the set itself: no k_prepare_small
bool self_prepare_path(const tloam_ctx* c) { return c->sv.flagb != nullptr && prepare_small_path(c) && solve_small_path(c); }
// ride: the finish of the previous outer iteration rides on this search launch (large single-rank sets, device-driven loop:
// k_build_finish_large; the search then runs on GnState::spec_build instead of `gate`)
int enqueue_build(tloam_ctx* c, const BuildParams& bp, const GridView grids[kKinds], bool rebin, const int* gate,
                  const int* refresh_gate = nullptr, bool prepare_in_solve = false, const FinishLargeArgs* ride = nullptr, int iter = 0) {
  const size_t n_slots = (size_t)c->sv.slot_off[kKinds];
  // direct set: the search writes the rows itself -- no flag scan, no compaction, no refresh (the Solve reads the live weight stream)
  const DirectSet ds{c->direct ? 1 : 0, rebin ? 1 : 0, (ride && !rebin) ? 0 : 1, c->direct ? direct_blk_cnt(c, iter) : nullptr,
                     c->tile_of_slot.p, c->tile_scan.p, c->direct ? c->row_of_pos.p : nullptr};
  if (ride && !rebin) {
    const size_t ntiles = (size_t)build_tile_count(grids, c->sv.slot_off);
    launch_build_finish_large(c->sv, grids, bp, c->state.p, c->tile_scan.p + ntiles, c->qrec.p, *ride, c->stream, c->direct ? &ds : nullptr);
  } else {
    launch_build(c->sv, grids, bp, c->state.p, c->tile_of_slot.p, c->tile_cnt.p, c->tile_scan.p, c->tile_fill.p,
                 c->qrec.p, c->scan_tmp.p, rebin, c->stream, gate, c->scan1p_q_use ? c->scan1p_q.p : nullptr, c->h_fault_dev + kFaultScan1p,
                 c->direct ? &c->cv : nullptr, c->direct ? &ds : nullptr, rebin && c->qbin_rode);
    if (rebin) c->qbin_rode = false;   // (consumed: a second sort of this frame -- a re-run -- bins for itself)
  }
  if (c->direct) return TLOAM_OK;
  if (prepare_in_solve) return TLOAM_OK;
  if (prepare_small_path(c)) {
    launch_prepare_small(c->sv, c->cv, bp, c->seg_n.p, c->state.p, gate, refresh_gate, c->stream);
    return TLOAM_OK;
  }
  if (one_rank(c)) {
    // single rank: the tile-local scan only -- the compaction adds the tiles' offsets itself -- and ONE launch for both
    // alternatives of a device-gated iteration (compaction, or the refresh of the unchanged set): two launches less
    const int tiles = scan_tiles_only(c->flags.p, c->scan.p, n_slots + 1, c->scan_tmp.p, c->stream, gate);
    if (tiles > 0) {
      launch_compact(c->sv, c->cv, bp, c->seg_n.p, nullptr, 0, 1, c->state.p, c->stream, gate, refresh_gate, c->scan_tmp.p, tiles);
      return TLOAM_OK;
    }
  }
  launch_exclusive_scan_u64(c->flags.p, c->scan.p, n_slots + 1, c->scan_tmp.p, c->stream, gate);
  const double* rank_counts = nullptr;
  if (exchanging(c)) {
    launch_rank_counts(c->sv, c->rank_counts.p, c->rank, c->nranks, c->stream);
    const int rc = allreduce(c, c->rank_counts.p, c->nranks * kKinds);
    if (rc != TLOAM_OK) return rc;
    rank_counts = c->rank_counts.p;
  }
  // seg_n: every kind with slots and a positive cap is rewritten by the compaction, the others keep the 0 of
  // k_frame_init; only a sharded rank can find its cap already filled by the lower ranks and write nothing
  if (exchanging(c)) HIPC(c, hipMemsetAsync(c->seg_n.p, 0, kKinds * sizeof(int), c->stream));
  launch_compact(c->sv, c->cv, bp, c->seg_n.p, rank_counts, c->rank, c->nranks, c->state.p, c->stream, gate);
  if (refresh_gate) launch_refresh(c->sv, c->cv, c->stream, refresh_gate);
  return TLOAM_OK;
}
// budget of K3 sweeps of outer iteration `iter`: the most it needed in the last three frames
int planned_sweeps_for(tloam_ctx* c, int iter) {
  if ((int)c->planned_sweeps.size() < 3 * (iter + 1)) c->planned_sweeps.resize(3 * ((size_t)iter + 1), 0);  // 0 = no history yet
  const int* hist = &c->planned_sweeps[3 * (size_t)iter];
  int planned = hist[0] == 0 ? kSolveSweeps  // first frame of this context: the full budget
                             : std::min(std::max(std::max(hist[0], hist[1]), std::max(hist[2], 1)), kSolveSweeps);
  if (solve_small_path(c)) planned = kSolveSweeps;   // one launch runs the Solve to its end: nothing to predict
  if (c->dbg_planned_sweeps > 0) planned = std::min(c->dbg_planned_sweeps, kSolveSweeps);
  return planned;
}
WeightParams weight_params(const tloam_ctx* c, double mu, const BuildParams& bp) {
  WeightParams wp;
  wp.th1 = (mu + 1) / mu * c->noise_bound_sq;   // :1049
  wp.th2 = mu / (mu + 1) * c->noise_bound_sq;   // :1050
  wp.mu = mu;
  wp.noise_bound_sq = c->noise_bound_sq;
  for (int k = 0; k < kKinds; ++k) wp.active[k] = bp.active[k];
  return wp;
}
// :1049-1086 thresholds + weight update, :1091-1094 cost sums, publish (+ device-side loop control when ctl.fast)
size_t total_seg_cap(const tloam_ctx* c) {
  size_t cap = 0;
  for (int k = 0; k < kKinds; ++k) cap += c->kd[k].c_cap;
  return cap;
}
// one 1024-thread block does weights + sums + publish in a single launch
bool finish_small_path(const tloam_ctx* c) { return one_rank(c) && total_seg_cap(c) <= 16384; }
int enqueue_finish(tloam_ctx* c, const WeightParams& wp, const HostMirror& hm, const OuterCtl& ctl, int iter = 0, int built = 1) {
  if (c->direct) {   // (c->cv carries the weight stream of outer iteration `iter`: direct_set_parity)
    const FinishLargeArgs fin = direct_finish_args(c, &c->cv, &wp, hm, ctl, iter, /*riding=*/0, built);
    launch_finish_direct(fin, c->state.p, c->stream);
    return TLOAM_OK;
  }
  // fixed function of the capacity (so the summation tree, hence the bits, do not depend on timing)
  const size_t cap = total_seg_cap(c);
  const int wblocks = (int)std::min<size_t>(256, std::max<size_t>(64, cap / 2048));
  if (finish_small_path(c)) {
    launch_weights_finish_small(c->cv, c->sv, wp, c->seg_n.p, c->sums16.p, c->state.p, hm, ctl, c->stream);
    return TLOAM_OK;
  }
  launch_weights(c->cv, c->sv, wp, c->wpart.p, wblocks, c->state.p, c->stream);
  if (exchanging(c)) {
    launch_outer_finish(c->wpart.p, wblocks, c->seg_n.p, nullptr, c->state.p, c->sums16.p, hm, ctl, c->stream);
    const int rc = allreduce(c, c->sums16.p, 16);
    if (rc != TLOAM_OK) return rc;
    launch_outer_publish(c->sums16.p, c->state.p, hm, c->comm == COMM_MAILBOX ? c->mbox.ctr + 1 : nullptr, c->stream);
  } else {
    launch_outer_finish(c->wpart.p, wblocks, c->seg_n.p, c->state.p, c->state.p, c->sums16.p, hm, ctl, c->stream);  // + publish + re-arm
  }
  return TLOAM_OK;
}
// host bookkeeping of a finished outer iteration from the mirrored state S (:1089-1121); returns whether the loop ends.
// *weight_violation: the reference's assert (:871) would have fired in this iteration.
bool account_outer(tloam_ctx* c, int iter, const GnState& S, double mu, int sweeps_before, bool* weight_violation) {
  // (an iteration that ran inside the launch of an earlier one was never planned: planned_sweeps_for has not sized the history)
  if (c->planned_sweeps.size() < 3 * ((size_t)iter + 1)) c->planned_sweeps.resize(3 * ((size_t)iter + 1), 0);
  int* hist = &c->planned_sweeps[3 * (size_t)iter];
  {
    const int used = std::min(std::max(S.gn_sweeps - sweeps_before, 1), kSolveSweeps);
    if (hist[0] == 0) hist[1] = hist[2] = used;  // the first observation stands for the whole window
    else { hist[2] = hist[1]; hist[1] = hist[0]; }
    hist[0] = used;
  }
  c->mu = mu * exp((double)(iter + 1) * c->cfg.gnc_factor);  // :1089
  tloam_stats& st = c->stats;
  st.outer_iterations = iter + 1;
  st.gn_evaluations = S.gn_evaluations;
  st.gn_sweeps = S.gn_sweeps;
  st.host_wait_us = (int32_t)c->wait_us;
  st.gn_iterations = S.gn_iterations;
  st.accepted_steps = S.accepted_steps;
  *weight_violation = S.bad_weights > st.weight_range_violations;
  st.weight_range_violations = S.bad_weights;
  st.mu = c->mu;
  st.solver_cost = S.x_cost;
  memcpy(st.se3, S.x, sizeof(double) * 6);
  int nn[kKinds];
  for (int k = 0; k < kKinds; ++k) {
    c->cur_cost[k] = S.kind_cost[k];
    st.kind_cost[k] = S.kind_cost[k];
    st.n_corr[k] = S.n_corr[k];
    nn[k] = S.n_corr[k];
  }
  c->k3_alg_bytes = alg_bytes_of(nn);
  bool fin = false;
  if (fabs(c->cur_cost[TLOAM_KIND_PLANAR] - c->prev_cost[TLOAM_KIND_PLANAR]) < c->cfg.cost_threshold) {  // :1108
    st.converged_early = 1;
    fin = true;
  } else {
    for (int k = 0; k < kKinds; ++k) c->prev_cost[k] = c->cur_cost[k];  // :1113-1116 (slots re-zeroed by the next compaction)
    c->iter = iter + 1;
    if (c->iter >= c->cfg.max_iterations) fin = true;
  }
  if (fin) c->iter = c->cfg.max_iterations;
  return fin;
}
// :1027-1033.  When iteration 0 reaches this point no Evaluate() has run yet, so every residual slot the
// reference takes maxCoeff() over is still the 0 it was initialised with (:931-949); the literal formula
// then gives mu = 1/(0 - 1) = -1 -> 1e-10.
double initial_mu(const tloam_ctx* c) {
  const double max_residual = 0.0;
  double mu = 1 / (2 * max_residual / c->noise_bound_sq - 1.0);
  if (mu <= 0) mu = 1e-10;
  return mu;
}
}  // namespace

int tloam_sm_outer(tloam_ctx* c, int* done, tloam_stats* stats) {
  if (!c || !c->active) return TLOAM_E_NOT_READY;
  HIPC(c, hipSetDevice(c->device));
  const int iter = c->iter;
  if (iter >= c->cfg.max_iterations) {  // loop condition :966
    if (done) *done = 1;
    if (stats) *stats = c->stats;
    return TLOAM_OK;
  }
  int rc;
  BuildParams bp;
  GridView grids[kKinds];
  outer_params(c, &bp, grids);
  rc = outer_reserve(c, grids);
  if (rc != TLOAM_OK) return rc;
  // The correspondence search is a pure function of (pose, clouds).  In the reference's GNC dynamics the
  // outer iterations after the first usually reject every step, so the pose -- hence every
  // neighbour list, fit and gate -- is bit-identical to the previous outer iteration: then only the
  // captured weights and the zeroed side-channel slots of the compact set have to be refreshed.
  const bool same_pose = iter > 0 && c->have_build && memcmp(c->build_x, c->stats.se3, sizeof(c->build_x)) == 0 &&
                         !c->dbg_no_build_reuse;
  if (c->direct) direct_set_parity(c, iter);
  if (!same_pose) {
    rc = enqueue_build(c, bp, grids, /*rebin=*/iter == 0, nullptr, nullptr, false, nullptr, iter);
    if (rc != TLOAM_OK) return rc;
    memcpy(c->build_x, c->stats.se3, sizeof(c->build_x));
    c->have_build = true;
  } else if (!c->direct) {
    launch_refresh(c->sv, c->cv, c->stream);
  }
  if (iter == 0) c->mu = initial_mu(c);
  // ---- :1036-1047 ceres::Solve, device resident.  Only as many sweeps as this outer iteration needed in the
  //      last three frames are enqueued (typically 2 of 5 from the second iteration on: the retried rejected steps
  //      are served by the evaluation reuse); the weight update and the finish kernel are gated on the
  //      minimiser having terminated, and raise `incomplete` otherwise -- then the Solve is topped up.
  const int planned = planned_sweeps_for(c, iter);
  const double mu = c->mu;
  const WeightParams wp = weight_params(c, mu, bp);
  rc = enqueue_solve(c, /*armed=*/true, planned, &wp);  // armed by sm_begin / the previous iteration's finish kernel
  if (rc != TLOAM_OK) return rc;
  const OuterCtl host_decides{c->cfg.cost_threshold, 0, 0};
  const int sweeps_before = c->stats.gn_sweeps;
  for (int attempt = 0;; ++attempt) {
    const HostMirror hm = next_mirror(c);
    rc = enqueue_finish(c, wp, hm, host_decides, iter, same_pose ? 0 : 1);
    if (rc != TLOAM_OK) return rc;
    rc = wait_state(c, hm);
    if (rc != TLOAM_OK) return rc;
    if (!c->h_state->incomplete) break;
    if (c->h_state->incomplete == OS_COMM_ERROR) {
      c->last_error = exchanging(c) ? "mailbox exchange timed out: a peer rank did not post (dead process or diverged call sequence)"
                                    : "in-launch hand-over of the fused GN iteration timed out (a block of the grid never posted its row)";
      return exchanging(c) ? TLOAM_E_RCCL : TLOAM_E_HIP;
    }
    if (attempt > 0 || planned >= kSolveSweeps) {
      c->last_error = "the minimiser did not terminate within its evaluation budget";
      return TLOAM_E_INVALID;
    }
    rc = enqueue_solve(c, /*armed=*/true, kSolveSweeps - planned, &wp);  // top up, then weights + finish again
    if (rc != TLOAM_OK) return rc;
  }
  const GnState& S = *c->h_state;
  rc = harvest_k3_events(c, S.gn_sweeps - sweeps_before);
  if (rc != TLOAM_OK) return rc;
  bool weight_violation = false;
  const bool fin = account_outer(c, iter, S, mu, sweeps_before, &weight_violation);
  if (c->direct) c->w_parity = (iter + 1) & 1;   // (c->cv stays on the stream this Solve captured)
  if (done) *done = fin ? 1 : 0;
  if (stats) *stats = c->stats;
  return weight_violation ? TLOAM_E_WEIGHT_RANGE : TLOAM_OK;  // the iteration is complete either way (:871)
}

// ---- scanMatching with the outer GNC loop driven from the device ------------------------------------------------
// Every outer iteration of the frame is enqueued up front -- builders + scan + compaction gated on "the pose moved",
// the refresh gated on "it did not", the planned sweeps, the finish kernel, which makes the loop decisions of
// registration.cpp:1108-1121 itself (plateau break, max_iterations, the next iteration's gates) and mirrors the
// iteration's result into its own pinned slot -- and the host waits ONCE, for the last slot.  (The stepwise API keeps
// the host in the loop: one round trip of ~12 us plus ~4 us of launch catch-up per following kernel and per outer
// iteration on an otherwise idle GPU.)  A Solve that runs out of its planned budget stops the device loop; the host tops
// it up and re-enters the loop behind the top-up.
// KITTI-size frames (self_prepare_path + finish in the Solve launch, DeviceLoopPlan::in_launch_finish): a frame is grid
// build + (search + Solve launch) per RUN of outer iterations -- the Solve launch ends its iteration itself and goes on
// with the next one while the pose stands still.  Only the launches of the first kEnqueueAhead iterations are enqueued up
// front; the host waits for the result slots IN ORDER and adds a (search, Solve) pair when a slot carries OS_NEEDS_HOST.
namespace {
struct DeviceLoopPlan {
  int planned[kMaxOuterFast] = {}, solve_start[kMaxOuterFast] = {}, used[kMaxOuterFast] = {};
  double mus[kMaxOuterFast] = {};
  HostMirror hms[kMaxOuterFast];
  // finish-in-the-Solve mode (SolveFinish): the launches of iterations [0, enq_end) are in the stream; a later iteration is
  // enqueued when the device says that it is needed (OS_NEEDS_HOST) -- see scan_match_device_loop
  bool in_launch_finish = false;
  int enq_end = 0;
  SolveFinish F;
  SolvePrep prep;
};
// iterations enqueued ahead of the device's verdicts in finish-in-the-Solve mode: iteration 0 almost always moves the pose
// (so the search and the Solve of iteration 1 will run), the later ones almost never do (they run inside the launch of
// iteration 1): launches for them would be no-ops that the frame's successor has to queue behind.
constexpr int kEnqueueAhead = 2;   // (1 and 4 measured the same within noise and are bit-identical: round 3 / 4 A/Bs)
// finish-in-the-Solve mode: the launches of iterations [from, to): the search if the pose moved (always in the frame's
// first), then the Solve + finish -- a launch that returns at once when an earlier one has already run its iteration
int enqueue_iterations_in_launch_mode(tloam_ctx* c, int from, int to, const BuildParams& bp, const GridView grids[kKinds],
                                      DeviceLoopPlan& P) {
  GnState* st = c->state.p;
  for (int iter = from; iter < to; ++iter) {
    int rc = enqueue_build(c, bp, grids, /*rebin=*/iter == 0, iter == 0 ? nullptr : &st->run_build, nullptr, /*prepare_in_solve=*/true);
    if (rc != TLOAM_OK) return rc;
    P.prep.run_build = iter == 0 ? nullptr : &st->run_build;
    P.prep.run_refresh = iter == 0 ? nullptr : &st->run_refresh;
    P.F.first_iter = iter;
    P.planned[iter] = planned_sweeps_for(c, iter);
    P.solve_start[iter] = c->batch_launches;
    rc = enqueue_solve(c, /*armed=*/true, P.planned[iter], &P.F.wp[iter], &P.prep, &P.F);
    if (rc != TLOAM_OK) return rc;
  }
  if (to > P.enq_end) P.enq_end = to;
  return TLOAM_OK;
}
// enqueue outer iterations first .. M-1.  first == 0: the frame's first iteration (always builds).  first > 0: a restart
// behind a stand-alone finish of iteration first - 1 that has set the gates (run_build / run_refresh) on the device.
int enqueue_outer_iterations(tloam_ctx* c, int first, double mu, const BuildParams& bp, const GridView grids[kKinds],
                             DeviceLoopPlan& P) {
  const int M = c->cfg.max_iterations;
  GnState* st = c->state.p;
  const int* run_build = &st->run_build;
  const int* run_refresh = &st->run_refresh;
  // KITTI-size frames: the finish of iteration k-1 does not get a launch of its own, it rides on the correspondence
  // search of iteration k (k_build_finish_small: they are independent of each other); the last one stands alone
  const bool ride = prepare_small_path(c) && finish_small_path(c) && build_finish_small_fits(c->sv);
  // ... and 1 M-class frames the same way with k_weights + k_outer_finish (k_build_finish_large)
  const bool ride_large = !ride && one_rank(c) && !finish_small_path(c) && build_finish_large_fits(c->sv);
  const int wblocks_large = (int)std::min<size_t>(256, std::max<size_t>(64, total_seg_cap(c) / 2048));   // as enqueue_finish
  if (ride_large) HIPC(c, c->fin_rows.reserve((size_t)4 * 256 * 8));
  bool pending = false;   // the finish of the previous iteration has not been enqueued yet (it rides on this search)
  bool pending_large = false;
  WeightParams wp_prev;
  OuterCtl ctl_prev{0.0, 0, 0};
  int rc = TLOAM_OK;
  // the scan + caps + compaction (or the refresh) of an iteration: a launch of its own (k_prepare_small), or the prologue of
  // the one-launch Solve (SolvePrep)
  const bool in_solve = self_prepare_path(c);
  SolvePrep prep;
  memset(&prep, 0, sizeof(prep));
  prep.sv = c->sv;
  for (int k = 0; k < kKinds; ++k) prep.maxnum[k] = bp.maxnum[k];
  // ... and the finish of an iteration (weights, sums, loop decisions, result slot): a launch of its own / riding on the next
  // search, or the tail of the one-launch Solve, which then goes on with the next iteration itself while the pose stands still
  const bool fin_in_solve = in_solve && finish_small_path(c) && M <= kMaxOuterInLaunch;
  P.in_launch_finish = fin_in_solve;
  if (fin_in_solve) {
    SolveFinish& F = P.F;
    memset(&F, 0, sizeof(F));
    F.enabled = 1;
    F.have_wp = 1;
    F.n_iter = M;
    F.cost_threshold = c->cfg.cost_threshold;
    F.sums16 = c->sums16.p;
    F.iter_span = iter_span_of(c);
    double m = mu;
    for (int iter = first; iter < M; ++iter) {
      P.mus[iter] = m;
      P.hms[iter] = next_mirror(c, iter);
      F.wp[iter] = weight_params(c, m, bp);
      F.hm[iter] = P.hms[iter];
      m = m * exp((double)(iter + 1) * c->cfg.gnc_factor);  // :1089
    }
    P.prep = prep;
    P.enq_end = first;
    return enqueue_iterations_in_launch_mode(c, first, std::min(M, first + kEnqueueAhead), bp, grids, P);
  }
  for (int iter = first; iter < M; ++iter) {
    if (c->direct) direct_set_parity(c, iter);   // what this iteration's Solve (and its finish) read; the search does not care
    if (iter == 0) {
      rc = enqueue_build(c, bp, grids, /*rebin=*/true, nullptr, nullptr, in_solve);
      prep.run_build = nullptr;
      prep.run_refresh = nullptr;
    } else if (pending) {
      FinishSmallArgs fin{&c->cv, &wp_prev, c->seg_n.p, c->sums16.p, P.hms[iter - 1], ctl_prev, c->wpart.p, c->k3_ticket.p + 1};
      launch_build_finish_small(c->sv, grids, bp, st, fin, c->stream);
      if (!in_solve) launch_prepare_small(c->sv, c->cv, bp, c->seg_n.p, st, run_build, run_refresh, c->stream);
      prep.run_build = run_build;
      prep.run_refresh = run_refresh;
      pending = false;
    } else if (pending_large) {
      FinishLargeArgs fin{&c->cv, &wp_prev, c->seg_n.p, c->sums16.p, P.hms[iter - 1], ctl_prev, c->fin_rows.p, c->k3_ticket.p + 1,
                          wblocks_large, {}};
      CorrView cv_prev = c->cv;
      if (c->direct) {   // the finish of iteration iter - 1 reads ITS weight stream and writes this iteration's
        for (int k = 0; k < kKinds; ++k) cv_prev.k[k].w = direct_w_stream(c, k, (iter - 1) & 1);
        fin = direct_finish_args(c, &cv_prev, &wp_prev, P.hms[iter - 1], ctl_prev, iter - 1, /*riding=*/1, iter - 1 == 0 ? 1 : -1);
      }
      rc = enqueue_build(c, bp, grids, /*rebin=*/false, run_build, run_refresh, in_solve, &fin, iter);
      prep.run_build = run_build;
      prep.run_refresh = run_refresh;
      pending_large = false;
    } else {
      rc = enqueue_build(c, bp, grids, /*rebin=*/false, run_build, run_refresh, in_solve, nullptr, iter);   // both alternatives, device-gated
      prep.run_build = run_build;
      prep.run_refresh = run_refresh;
    }
    if (rc != TLOAM_OK) return rc;
    P.planned[iter] = planned_sweeps_for(c, iter);
    P.solve_start[iter] = c->batch_launches;
    const WeightParams wp_iter = weight_params(c, mu, bp);
    rc = enqueue_solve(c, /*armed=*/true, P.planned[iter], &wp_iter, in_solve ? &prep : nullptr);
    if (rc != TLOAM_OK) return rc;
    P.mus[iter] = mu;
    P.hms[iter] = next_mirror(c, iter);
    const OuterCtl ctl{c->cfg.cost_threshold, 1, iter == M - 1 ? 1 : 0};
    if (ride && iter < M - 1) {
      wp_prev = weight_params(c, mu, bp);
      ctl_prev = ctl;
      pending = true;
    } else if (ride_large && iter < M - 1) {
      wp_prev = weight_params(c, mu, bp);
      ctl_prev = ctl;
      pending_large = true;
    } else {
      rc = enqueue_finish(c, weight_params(c, mu, bp), P.hms[iter], ctl, iter, iter == 0 ? 1 : -1);
      if (rc != TLOAM_OK) return rc;
    }
    mu = mu * exp((double)(iter + 1) * c->cfg.gnc_factor);  // :1089
  }
  return TLOAM_OK;
}

int scan_match_device_loop(tloam_ctx* c, bool* weight_violation) {
  const int M = c->cfg.max_iterations;
  BuildParams bp;
  GridView grids[kKinds];
  outer_params(c, &bp, grids);
  int rc = outer_reserve(c, grids);
  if (rc != TLOAM_OK) return rc;
  GnState* st = c->state.p;
  DeviceLoopPlan P;
  rc = enqueue_outer_iterations(c, 0, initial_mu(c), bp, grids, P);
  if (rc != TLOAM_OK) return rc;
  if (!P.in_launch_finish) {
    rc = wait_state(c, P.hms[M - 1], M - 1);   // the last slot is written last (stream order), whatever the frame did
    if (rc != TLOAM_OK) return rc;
  }
  // ---- the frame's bookkeeping, iteration by iteration, from the mirrored slots
  int topups = 0;
  for (int iter = 0; iter < M; ++iter) {
    if (iter < M - 1 || P.in_launch_finish) {
      // (all launches enqueued: written before the last slot, already there -- this only unpacks it.  Finish-in-the-Solve
      //  mode: the slots are waited for in order -- the frame's result is there when its last iteration's is, and a launch
      //  may have to be added on the way)
      rc = wait_state(c, P.hms[iter], iter);
      if (rc != TLOAM_OK) return rc;
    }
    GnState* Sm = &c->h_state[iter];
    const bool needs_host = (Sm->incomplete & OS_NEEDS_HOST) != 0;
    if (Sm->incomplete & OS_SET_STALE) c->set_stale = true;   // (direct set: the loop ended beside a search that had already run)
    Sm->incomplete &= ~(int)(OS_NEEDS_HOST | OS_SET_STALE);
    const GnState* S = Sm;
    if (S->host_seq != P.hms[iter].seq) {
      c->last_error = "device-driven loop: the result slot of an outer iteration was not written";
      return TLOAM_E_HIP;
    }
    if (S->incomplete == OS_SKIPPED) break;   // the loop had ended before this iteration
    if (S->incomplete == OS_COMM_ERROR) {
      c->last_error = "in-launch hand-over of the fused GN iteration timed out (a block of the grid never posted its row)";
      c->hand_over_timed_out = true;
      return TLOAM_E_HIP;
    }
    if (S->incomplete == OS_INCOMPLETE) {
      // The Solve of this iteration ran out of its planned budget: the device stopped the loop there (the sweeps of
      // the later iterations, gated only on `done`, have meanwhile continued this same Solve; their builds, refreshes
      // and finish kernels were gated off).  Top the Solve up to its full budget, finish the iteration with the DEVICE
      // deciding as usual, and enqueue the rest of the frame behind it: one more wait instead of a host round trip per
      // remaining outer iteration.
      if (++topups > M) {
        c->last_error = "the minimiser did not terminate within its evaluation budget";
        return TLOAM_E_INVALID;
      }
      HIPC(c, hipMemsetAsync(&st->stop, 0, sizeof(int), c->stream));
      P.solve_start[iter] = c->batch_launches;
      P.planned[iter] = kSolveSweeps;
      const WeightParams wp_top = weight_params(c, P.mus[iter], bp);
      if (c->direct) direct_set_parity(c, iter);
      rc = enqueue_solve(c, /*armed=*/true, kSolveSweeps, &wp_top);
      if (rc != TLOAM_OK) return rc;
      P.hms[iter] = next_mirror(c, iter);
      // (the finish that found the Solve unfinished has cleared the device's gates: whether this iteration built its set is the
      //  host's to say -- the pose the previous iteration ended at against the pose of the last build, as the bookkeeping below)
      const int built_top = (iter == 0 || memcmp(c->build_x, c->stats.se3, sizeof(c->build_x)) != 0) ? 1 : 0;
      rc = enqueue_finish(c, wp_top, P.hms[iter], OuterCtl{c->cfg.cost_threshold, 1, iter == M - 1 ? 1 : 0}, iter, built_top);
      if (rc != TLOAM_OK) return rc;
      if (iter + 1 < M) {
        rc = enqueue_outer_iterations(c, iter + 1, P.mus[iter] * exp((double)(iter + 1) * c->cfg.gnc_factor), bp, grids, P);
        if (rc != TLOAM_OK) return rc;
      }
      if (!P.in_launch_finish) {
        rc = wait_state(c, P.hms[M - 1], M - 1);
        if (rc != TLOAM_OK) return rc;
      }
      if (iter < M - 1 || P.in_launch_finish) {
        rc = wait_state(c, P.hms[iter], iter);
        if (rc != TLOAM_OK) return rc;
      }
      S = &c->h_state[iter];
      if (S->incomplete == OS_INCOMPLETE) {
        c->last_error = "the minimiser did not terminate within its evaluation budget";
        return TLOAM_E_INVALID;
      }
      if (S->incomplete == OS_COMM_ERROR) {
        c->last_error = "in-launch hand-over of the fused GN iteration timed out (a block of the grid never posted its row)";
        c->hand_over_timed_out = true;
        return TLOAM_E_HIP;
      }
    }
    const int sweeps_before = c->stats.gn_sweeps;
    // the compact set of this iteration was (re)built iff the pose had moved since the last build
    if (iter == 0 || memcmp(c->build_x, c->stats.se3, sizeof(c->build_x)) != 0) memcpy(c->build_x, c->stats.se3, sizeof(c->build_x));
    c->have_build = true;
    c->mu = P.mus[iter];
    bool wv = false;
    const bool fin = account_outer(c, iter, *S, P.mus[iter], sweeps_before, &wv);
    P.used[iter] = std::min(S->gn_sweeps - sweeps_before, P.planned[iter]);
    if (wv) *weight_violation = true;
    if (fin) break;
    // the launch that ran this iteration has ended because the pose moved: the search and the Solve of the next one, unless
    // they are in the stream already
    if (P.in_launch_finish && needs_host && iter + 1 >= P.enq_end && iter + 1 < M) {
      const bool trace = getenv("TLOAM_DEBUG_RESUME") != nullptr;   // development aid / test census: how often the host adds a launch
      if (trace) fprintf(stderr, "[tloam resume] outer iteration %d enqueued by the host\n", iter + 1);
      rc = enqueue_iterations_in_launch_mode(c, iter + 1, iter + 2, bp, grids, P);
      if (rc != TLOAM_OK) return rc;
    }
  }
  rc = harvest_k3_events_multi(c, M, P.solve_start, P.used);
  if (rc != TLOAM_OK) return rc;
  if (c->direct) {   // what the getters and the timing helpers see: the weights the LAST Solve captured; the current ones: the other stream
    const int K = std::max(c->stats.outer_iterations, 1);
    direct_set_parity(c, K - 1);
    c->w_parity = K & 1;
  }
  return 0;
}
}  // namespace

int tloam_sm_end(tloam_ctx* c, double result[16], tloam_stats* stats) {
  if (!c || !c->active || !result) return TLOAM_E_NOT_READY;
  const Pose T = se3_exp(c->stats.se3);  // :1124 exp(se3_pose_).matrix()
  pose_to_matrix(T, result);
  if (stats) *stats = c->stats;
  c->active = false;
  return check_device_faults(c);   // (every iteration's result has been waited for: the frame's kernels are done)
}

int tloam_scan_match(tloam_ctx* c, const double predict[16], const double* omega3, double result[16],
                     double* scan_xyz, size_t n_scan, tloam_stats* stats) {
  if (n_scan > kMaxPoints) return TLOAM_E_INVALID;
  static const bool stamps = getenv("TLOAM_HOST_STAMPS") != nullptr;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double hs0 = stamps ? now_us() : 0.0;
  if (c) c->hs_entry = hs0;
  int rc = tloam_sm_begin(c, predict, omega3);
  if (rc != TLOAM_OK) return rc;
  const double hs1 = stamps ? now_us() : 0.0;
  // A frame that fails on the way -- an unwritten result slot, a minimiser that "did not terminate", a time-out inside a launch --
  // may have failed on the garbage a timed-out look-back scan left behind: the frame is closed, the fault words are looked at
  // (check_device_faults moves the context to the forms that wait for nothing and clears them: the NEXT frame is not blamed), and
  // if it was the scan the frame is run again, once per context, as tloam_sm_end's path does.
  auto frame_failed = [&](int rc_in) -> int {
    (void)hipStreamSynchronize(c->stream);
    c->active = false;
    const bool scan_fault_now = c->h_fault && __atomic_load_n(&c->h_fault[kFaultScan1p], __ATOMIC_ACQUIRE) != 0u;
    (void)check_device_faults(c);
    if (scan_fault_now && !c->scan1p_retried) {
      c->scan1p_retried = true;
      return tloam_scan_match(c, predict, omega3, result, scan_xyz, n_scan, stats);
    }
    return rc_in;
  };
  int done = 0;
  bool weight_violation = false;
  // device-driven outer loop where the stepwise machinery is not asked for: one rank, a pinned mirror, the planned
  // iterations fit the result slots, no development knob that needs the host between iterations
  if (one_rank(c) && c->cfg.max_iterations >= 1 && c->cfg.max_iterations <= kMaxOuterFast &&
      !c->dbg_no_build_reuse && !c->no_device_loop) {
    const bool persistent = solve_small_path(c);
    rc = scan_match_device_loop(c, &weight_violation);
    if (rc == TLOAM_E_HIP && persistent && c->hand_over_timed_out) {
      // A block of the one-launch Solve was never scheduled beside the others (a device shared with long-running kernels,
      // fewer usable CUs than the attribute says): the waits inside the launch are bounded, the frame is intact in HBM --
      // solve it again with one launch per GN iteration, and keep this context on that path.
      c->no_persistent_solve = true;
      c->persistent_solve_timed_out = true;
      c->fallback_events++;
      c->hand_over_timed_out = false;
      (void)hipStreamSynchronize(c->stream);
      c->active = false;
      rc = tloam_sm_begin(c, predict, omega3);
      if (rc != TLOAM_OK) return rc;
      weight_violation = false;
      rc = scan_match_device_loop(c, &weight_violation);
    }
    if (rc < 0) return frame_failed(rc);
    done = rc == 0 ? 1 : 0;
  }
  while (!done) {
    rc = tloam_sm_outer(c, &done, nullptr);
    if (rc == TLOAM_E_WEIGHT_RANGE) { weight_violation = true; continue; }  // reported after the solve
    if (rc != TLOAM_OK) return frame_failed(rc);
  }
  const double hs2 = stamps ? now_us() : 0.0;
  // (which bounded wait ran out, if any: tloam_sm_end's check_device_faults clears the words)
  const bool scan_fault = c->h_fault && __atomic_load_n(&c->h_fault[kFaultScan1p], __ATOMIC_ACQUIRE) != 0u;
  rc = tloam_sm_end(c, result, stats);
  if (stamps && c) {
    const double hs3 = now_us();
    if (c->hs_exit > 0.0) c->hs[0] += hs0 - c->hs_exit;   // the caller, between two calls
    c->hs[1] += hs1 - hs0;                                 // sm_begin: set-up + the grid build's launches
    c->hs[2] += hs2 - hs1 - c->wait_us;                    // the loop: enqueueing, bookkeeping (without the wait)
    c->hs[3] += c->wait_us;
    c->hs[4] += hs3 - hs2;                                 // sm_end
    c->hs_exit = hs3;
    if (++c->hs_n % 200 == 0) {
      fprintf(stderr, "[tloam host stamps] per frame over 200: caller %.2f  begin %.2f (first launch called at %.2f, issued at %.2f)  "
                      "loop-without-wait %.2f  wait %.2f  end %.2f us\n",
              c->hs[0] / 200, c->hs[1] / 200, c->hs[5] / 200, c->hs[6] / 200, c->hs[2] / 200, c->hs[3] / 200, c->hs[4] / 200);
      for (int i = 0; i < 8; ++i) c->hs[i] = 0.0;
    }
  }
  if (rc == TLOAM_E_HIP && scan_fault && !c->scan1p_retried) {
    // a look-back scan of this frame gave up (check_device_faults): the clouds are intact in HBM and the context has been
    // switched to the multi-launch scans -- run the frame again, once
    c->scan1p_retried = true;
    return tloam_scan_match(c, predict, omega3, result, scan_xyz, n_scan, stats);
  }
  if (rc != TLOAM_OK) return rc;
  if (scan_xyz && n_scan > 0) {  // :1126-1128 out_result_.scan_cloud->Transform(curr_frame_pose)
    HIPC(c, c->misc.reserve(3 * n_scan));
    HIPC(c, hipMemcpyAsync(c->misc.p, scan_xyz, sizeof(double) * 3 * n_scan, hipMemcpyHostToDevice, c->stream));
    launch_transform_cloud(c->misc.p, n_scan, result, c->stream);
    HIPC(c, hipMemcpyAsync(scan_xyz, c->misc.p, sizeof(double) * 3 * n_scan, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  return weight_violation ? TLOAM_E_WEIGHT_RANGE : TLOAM_OK;
}

// ---- getFitnessScore (registration.cpp:257-296) -------------------------------------------------
int tloam_fitness(tloam_ctx* c, double* fitness, double* rmse) {
  if (!c || !fitness || !rmse) return TLOAM_E_INVALID;
  *fitness = 0.0;
  *rmse = 0.0;
  if (!(c->cfg.fitness_thres > 0.0)) return TLOAM_OK;  // :258-261 (a NaN threshold finds nobody either)
  if (c->active) return TLOAM_E_NOT_READY;  // between sm_begin and sm_end the context belongs to the solve
  HIPC(c, hipSetDevice(c->device));
  const int blocks = 64;
  HIPC(c, c->misc.reserve(4096));
  const int order[kKinds] = {TLOAM_KIND_EDGE, TLOAM_KIND_SPHERE, TLOAM_KIND_PLANAR, TLOAM_KIND_GROUND};  // :287-290
  double fit_local[kKinds] = {0, 0, 0, 0}, err_local[kKinds] = {0, 0, 0, 0};
  for (int o = 0; o < kKinds; ++o) {
    const int k = order[o];
    KindData& K = c->kd[k];
    // the kd-trees are the ones built by the last scanMatching (:889-915); none yet -> no hits
    if (!K.grid_valid || K.n_src == 0 || !K.src_set || !K.src_ptr) continue;   // (a hand-over that failed registered nothing)
    // raw scan-frame source points (:271): this kind's AoS block as SoA, in scratch of its own (the slot arrays
    // sx/sy/sz belong to scan_match: SlotView holds their addresses)
    HIPC(c, c->fit_x.reserve(K.n_src)); HIPC(c, c->fit_y.reserve(K.n_src)); HIPC(c, c->fit_z.reserve(K.n_src));
    launch_aos_to_soa(K.src_ptr, K.n_src, c->fit_x.p, c->fit_y.p, c->fit_z.p, c->stream);
    launch_fitness(K.gv, c->fit_x.p, c->fit_y.p, c->fit_z.p, (int)K.n_src, c->cfg.fitness_thres, c->misc.p, blocks, c->stream);
    HIPC(c, hipMemcpyAsync(c->h_small, c->misc.p, sizeof(double) * blocks * 2, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < blocks; ++b) { err_local[k] += c->h_small[2 * b]; fit_local[k] += c->h_small[2 * b + 1]; }
  }
  if (c->nranks > 1) {  // sharded sources: hits and squared errors add up across ranks
    HIPC(c, c->misc.reserve(16));
    for (int k = 0; k < kKinds; ++k) { c->h_small[k] = fit_local[k]; c->h_small[4 + k] = err_local[k]; }
    HIPC(c, hipMemcpyAsync(c->misc.p, c->h_small, sizeof(double) * 8, hipMemcpyHostToDevice, c->stream));
    int rc = allreduce(c, c->misc.p, 8);
    if (rc != TLOAM_OK) return rc;
    HIPC(c, hipMemcpyAsync(c->h_small, c->misc.p, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < kKinds; ++k) { fit_local[k] = c->h_small[k]; err_local[k] = c->h_small[4 + k]; }
  }
  for (int o = 0; o < kKinds; ++o) {
    const int k = order[o];
    if (fit_local[k] > 0.0) {  // :278-284
      *fitness += fit_local[k] / (double)c->kd[k].n_src_full;
      *rmse += sqrt(err_local[k] / fit_local[k]);
    }
  }
  return TLOAM_OK;
}

// ---- introspection --------------------------------------------------------------------------------
static int download_soa3(tloam_ctx* c, const double* x, const double* y, const double* z, size_t n, double* aos) {
  std::vector<double> tmp(3 * n);
  HIPC(c, hipMemcpy(tmp.data(), x, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIPC(c, hipMemcpy(tmp.data() + n, y, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIPC(c, hipMemcpy(tmp.data() + 2 * n, z, sizeof(double) * n, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) { aos[3 * i] = tmp[i]; aos[3 * i + 1] = tmp[n + i]; aos[3 * i + 2] = tmp[2 * n + i]; }
  return TLOAM_OK;
}

// ---- a direct set through the getters: rebuilt first if it is stale, then filtered (idx >= 0) and put into source-index order ----
static int regen_direct_set(tloam_ctx* c) {
  // the rows hold the geometry of a search that ran on the last Solve's own verdict before the loop ended (OS_SET_STALE): search
  // again at the pose the SOLVED set was built at (x_build) -- same queries, same order, same arithmetic: the same rows
  BuildParams bp;
  GridView grids[kKinds];
  outer_params(c, &bp, grids);
  HIPC(c, c->state_scratch.reserve(1));
  HIPC(c, hipMemcpyAsync(c->state_scratch.p, c->state.p, sizeof(GnState), hipMemcpyDeviceToDevice, c->stream));
  launch_pose_from_x_build(c->state_scratch.p, c->stream);
  const DirectSet ds{1, 0, 0, nullptr, c->tile_of_slot.p, c->tile_scan.p, c->row_of_pos.p};
  launch_build(c->sv, grids, bp, c->state_scratch.p, c->tile_of_slot.p, c->tile_cnt.p, c->tile_scan.p, c->tile_fill.p, c->qrec.p,
               c->scan_tmp.p, /*rebin=*/false, c->stream, nullptr, nullptr, nullptr, &c->cv, &ds);
  HIPC(c, hipStreamSynchronize(c->stream));
  c->set_stale = false;
  return TLOAM_OK;
}
static int get_correspondences_direct(tloam_ctx* c, int kind, size_t capacity, size_t* n, int32_t* src_index, double* a, double* b,
                                      double* d, double* w, double* cost) {
  if (c->set_stale) { const int rc = regen_direct_set(c); if (rc != TLOAM_OK) return rc; }
  const size_t rows = (size_t)(c->sv.slot_off[kind + 1] - c->sv.slot_off[kind]);   // (of the frame the solve began with, not of the cloud registered now)
  const CorrSeg& s = c->cv.k[kind];
  std::vector<int> idx(rows);
  if (rows > 0) HIPC(c, hipMemcpy(idx.data(), s.idx, sizeof(int) * rows, hipMemcpyDeviceToHost));
  std::vector<std::pair<int, size_t>> order;   // (source index, row) of the factors
  order.reserve(rows);
  for (size_t r = 0; r < rows; ++r)
    if (idx[r] >= 0) order.emplace_back(idx[r], r);
  std::sort(order.begin(), order.end());
  const size_t m = order.size();
  *n = m;
  if (m > capacity) return TLOAM_E_INVALID;
  if (m == 0) return TLOAM_OK;
  std::vector<double> tmp(rows);
  auto stream = [&](const double* dev, double* out, size_t stride_out, size_t off) -> int {
    HIPC(c, hipMemcpy(tmp.data(), dev, sizeof(double) * rows, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; ++i) out[i * stride_out + off] = tmp[order[i].second];
    return TLOAM_OK;
  };
  int rc = TLOAM_OK;
  if (src_index) for (size_t i = 0; i < m; ++i) src_index[i] = order[i].first;
  if (a && ((rc = stream(s.ax, a, 3, 0)) || (rc = stream(s.ay, a, 3, 1)) || (rc = stream(s.az, a, 3, 2)))) return rc;
  if (b && kind == TLOAM_KIND_EDGE && ((rc = stream(s.bx, b, 3, 0)) || (rc = stream(s.by, b, 3, 1)) || (rc = stream(s.bz, b, 3, 2)))) return rc;
  if (d && kind <= TLOAM_KIND_GROUND && (rc = stream(s.d, d, 1, 0))) return rc;
  if (w && (rc = stream(s.w, w, 1, 0))) return rc;
  if (cost && (rc = stream(s.cost, cost, 1, 0))) return rc;
  return TLOAM_OK;
}

int tloam_get_correspondences(tloam_ctx* c, int kind, size_t capacity, size_t* n, int32_t* src_index, double* a,
                              double* b, double* d, double* w, double* cost) {
  if (!c || kind < 0 || kind >= kKinds || !n) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  if (c->direct && !c->prebuilt) return get_correspondences_direct(c, kind, capacity, n, src_index, a, b, d, w, cost);
  int segn[kKinds];
  HIPC(c, hipMemcpy(segn, c->seg_n.p, sizeof(segn), hipMemcpyDeviceToHost));
  const size_t m = (size_t)segn[kind];
  *n = m;
  if (m > capacity) return TLOAM_E_INVALID;
  if (m == 0) return TLOAM_OK;
  const CorrSeg& s = c->cv.k[kind];
  int rc;
  if (src_index) HIPC(c, hipMemcpy(src_index, s.idx, sizeof(int) * m, hipMemcpyDeviceToHost));
  if (a && (rc = download_soa3(c, s.ax, s.ay, s.az, m, a)) != TLOAM_OK) return rc;
  if (b && kind == TLOAM_KIND_EDGE && (rc = download_soa3(c, s.bx, s.by, s.bz, m, b)) != TLOAM_OK) return rc;
  if (d && kind <= TLOAM_KIND_GROUND) HIPC(c, hipMemcpy(d, s.d, sizeof(double) * m, hipMemcpyDeviceToHost));
  if (w) HIPC(c, hipMemcpy(w, s.w, sizeof(double) * m, hipMemcpyDeviceToHost));
  if (cost) HIPC(c, hipMemcpy(cost, s.cost, sizeof(double) * m, hipMemcpyDeviceToHost));
  return TLOAM_OK;
}

int tloam_get_weights(tloam_ctx* c, int kind, size_t capacity, size_t* n, double* w) {
  if (!c || kind < 0 || kind >= kKinds || !n) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  // the weights are those of the frame the last scanMatching BEGAN with (its slot table), whatever source cloud has been handed over
  // since (until round 6 the size came from the registered cloud: a larger cloud handed over after a solve made this a copy past
  // the end of the weights -- tests/tools/fuzz_call_order.py, TLOAM_E_HIP from a getter)
  const bool begun = c->w_src.p != nullptr && c->sv.slot_off[kKinds] > 0;
  const size_t m = begun ? (size_t)(c->sv.slot_off[kind + 1] - c->sv.slot_off[kind]) : c->kd[kind].n_src;
  *n = m;
  if (m > capacity || !begun) return TLOAM_E_INVALID;
  HIPC(c, hipStreamSynchronize(c->stream));
  if (c->direct && !c->prebuilt) {   // the current GNC weights live in the rows' weight stream `w_parity`: back to source-index order
    if (w && m > 0 && !c->have_build) {   // (no search has run in this frame yet: registration.cpp:931-949, every weight is 1)
      for (size_t i = 0; i < m; ++i) w[i] = 1.0;
    } else if (w && m > 0) {
      std::vector<int> idx(m);
      std::vector<double> wr(m);
      HIPC(c, hipMemcpy(idx.data(), c->cv.k[kind].idx, sizeof(int) * m, hipMemcpyDeviceToHost));
      HIPC(c, hipMemcpy(wr.data(), direct_w_stream(c, kind, c->w_parity), sizeof(double) * m, hipMemcpyDeviceToHost));
      for (size_t r = 0; r < m; ++r) w[(size_t)(idx[r] >= 0 ? idx[r] : ~idx[r]) - (size_t)c->sv.src_lo[kind]] = wr[r];
    }
    return TLOAM_OK;
  }
  if (w && m > 0) HIPC(c, hipMemcpy(w, c->w_src.p + c->sv.slot_off[kind], sizeof(double) * m, hipMemcpyDeviceToHost));
  return TLOAM_OK;
}

int tloam_knn(tloam_ctx* c, int kind, const double* q, size_t nq, double radius, int k, int32_t* out_idx,
              double* out_d2, int32_t* out_cnt) {
  if (!c || kind < 0 || kind >= kKinds || !q || k < 1 || k > kMaxK || !(radius > 0.0) || !out_idx || !out_d2 || !out_cnt ||
      nq > kMaxPoints / (size_t)k)
    return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  KindData& K = c->kd[kind];
  if (!K.tgt_set || K.n_tgt == 0) {
    for (size_t i = 0; i < nq; ++i) out_cnt[i] = 0;
    for (size_t i = 0; i < nq * (size_t)k; ++i) { out_idx[i] = -1; out_d2[i] = 0.0; }
    return TLOAM_OK;
  }
  int rc;
  GridBuffers tmp;  // a grid over the target currently set, sized for this radius; the scanMatching grids stay intact
  GridView views[kKinds];
  {
    double radii[kKinds] = {0, 0, 0, 0};
    radii[kind] = radius;
    rc = build_grids(c, tmp, radii, views);
    if (rc != TLOAM_OK) { tmp.release(); return rc; }
  }
  DBuf<double> qa, qx, qy, qz, d2;
  DBuf<int> idx, cnt;
  auto cleanup = [&]() { qa.release(); qx.release(); qy.release(); qz.release(); d2.release(); idx.release(); cnt.release(); tmp.release(); };
  hipError_t e = hipSuccess;
  if ((e = qa.reserve(3 * nq + 3)) != hipSuccess || (e = qx.reserve(nq + 1)) != hipSuccess ||
      (e = qy.reserve(nq + 1)) != hipSuccess || (e = qz.reserve(nq + 1)) != hipSuccess ||
      (e = d2.reserve(nq * k + 1)) != hipSuccess || (e = idx.reserve(nq * k + 1)) != hipSuccess ||
      (e = cnt.reserve(nq + 1)) != hipSuccess) {
    cleanup();
    c->last_error = hipGetErrorString(e);
    return TLOAM_E_HIP;
  }
  if (nq > 0) {
    (void)hipMemcpyAsync(qa.p, q, sizeof(double) * 3 * nq, hipMemcpyHostToDevice, c->stream);
    launch_aos_to_soa(qa.p, nq, qx.p, qy.p, qz.p, c->stream);
    launch_knn(views[kind], qx.p, qy.p, qz.p, (int)nq, radius, k, idx.p, d2.p, cnt.p, c->stream);
    (void)hipMemcpyAsync(out_idx, idx.p, sizeof(int) * nq * k, hipMemcpyDeviceToHost, c->stream);
    (void)hipMemcpyAsync(out_d2, d2.p, sizeof(double) * nq * k, hipMemcpyDeviceToHost, c->stream);
    (void)hipMemcpyAsync(out_cnt, cnt.p, sizeof(int) * nq, hipMemcpyDeviceToHost, c->stream);
  }
  e = hipStreamSynchronize(c->stream);
  cleanup();
  if (e != hipSuccess) { c->last_error = hipGetErrorString(e); return TLOAM_E_HIP; }
  return check_device_faults(c);
}

// ---- pre-built correspondence sets ------------------------------------------------------------------
int tloam_set_correspondences(tloam_ctx* c, int res_type, size_t n, const double* p, const double* a, const double* b,
                              const double* d, const double* w) {
  if (!c || res_type < 0 || res_type >= TLOAM_NUM_RES || n > kMaxPoints) return TLOAM_E_INVALID;
  if (n > 0 && (!p || !a || !w || (res_type == TLOAM_RES_LINE && !b) || (res_type == TLOAM_RES_PLANE && !d)))
    return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  int rc = ensure_common(c);
  if (rc != TLOAM_OK) return rc;
  const int kind = res_type == TLOAM_RES_PLANE ? TLOAM_KIND_PLANAR : (res_type == TLOAM_RES_LINE ? TLOAM_KIND_EDGE : TLOAM_KIND_SPHERE);
  if (!c->prebuilt) {
    c->direct = false;   // a pre-built set is compact
    HIPC(c, hipMemsetAsync(c->seg_n.p, 0, 8 * sizeof(int), c->stream));
    for (int k = 0; k < kKinds; ++k) {
      rc = reserve_seg(c, k, 1);
      if (rc != TLOAM_OK) return rc;
      c->kd[k].pre_n_full = 0;
    }
    c->prebuilt = true;
    c->active = false;
  }
  size_t lo = 0, hi = n;
  tloam_shard_range(n, c->rank, c->nranks, &lo, &hi);
  const size_t m = hi - lo;
  KindData& K = c->kd[kind];
  K.pre_lo = lo;
  K.pre_n_full = n;
  rc = reserve_seg(c, kind, m);
  if (rc != TLOAM_OK) return rc;
  HIPC(c, c->misc.reserve(3 * std::max<size_t>(m, 1)));
  const CorrSeg& s = c->cv.k[kind];
  if (m > 0) {
    HIPC(c, hipMemcpyAsync(c->misc.p, p + 3 * lo, sizeof(double) * 3 * m, hipMemcpyHostToDevice, c->stream));
    launch_aos_to_soa(c->misc.p, m, s.px, s.py, s.pz, c->stream);
    HIPC(c, hipMemcpyAsync(c->misc.p, a + 3 * lo, sizeof(double) * 3 * m, hipMemcpyHostToDevice, c->stream));
    launch_aos_to_soa(c->misc.p, m, s.ax, s.ay, s.az, c->stream);
    if (res_type == TLOAM_RES_LINE) {
      HIPC(c, hipMemcpyAsync(c->misc.p, b + 3 * lo, sizeof(double) * 3 * m, hipMemcpyHostToDevice, c->stream));
      launch_aos_to_soa(c->misc.p, m, s.bx, s.by, s.bz, c->stream);
    }
    if (res_type == TLOAM_RES_PLANE) HIPC(c, hipMemcpyAsync(s.d, d + lo, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(s.w, w + lo, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemsetAsync(s.cost, 0, sizeof(double) * m, c->stream));
    std::vector<int> ids(m);
    for (size_t i = 0; i < m; ++i) ids[i] = (int)(lo + i);
    HIPC(c, hipMemcpyAsync(s.idx, ids.data(), sizeof(int) * m, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  const int mi = (int)m;
  HIPC(c, hipMemcpyAsync(c->seg_n.p + kind, &mi, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  size_t total_cap = 0;
  int nn[kKinds];
  for (int k = 0; k < kKinds; ++k) {
    total_cap += c->kd[k].c_cap;
    nn[k] = (k == kind) ? mi : 0;
  }
  {  // algorithmic bytes of one sweep over the whole (job-wide) pre-built set
    int full[kKinds] = {(int)c->kd[0].pre_n_full, 0, (int)c->kd[2].pre_n_full, (int)c->kd[3].pre_n_full};
    c->k3_alg_bytes = alg_bytes_of(full);
    (void)nn;
  }
  {
    int caps[kKinds];
    for (int k = 0; k < kKinds; ++k) caps[k] = (int)c->kd[k].c_cap;
    k3_plan(caps, c->device_cus, &c->k3_grid, &c->k3_single, &c->k3_wide);
    (void)total_cap;
  }
  return reserve_partials(c);
}

int tloam_accumulate(tloam_ctx* c, const double se3[6], double H[36], double g[6], double* cost) {
  if (!c || !se3) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->partials.p) return TLOAM_E_NOT_READY;
  memcpy(c->h_small, se3, sizeof(double) * 6);
  HIPC(c, hipMemcpyAsync(c->se3_dev.p, c->h_small, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
  launch_set_eval(c->state.p, c->se3_dev.p, c->stream);
  int rc = launch_k3_timed(c, true);
  if (rc != TLOAM_OK) return rc;
  launch_reduce(c->partials.p, c->k3_grid, c->state.p, c->red48.p, c->stream);
  rc = allreduce(c, c->red48.p, kReduceBuf);
  if (rc != TLOAM_OK) return rc;
  HIPC(c, hipMemcpyAsync(c->h_small + 8, c->red48.p, sizeof(double) * kReduceBuf, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  rc = harvest_k3_events(c, 1);
  if (rc != TLOAM_OK) return rc;
  const double* t = c->h_small + 8;
  if (H) {
    int u = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) { H[i * 6 + j] = t[u]; H[j * 6 + i] = t[u]; ++u; }
  }
  if (g) for (int i = 0; i < 6; ++i) g[i] = t[21 + i];
  if (cost) *cost = t[27];
  return TLOAM_OK;
}

int tloam_get_normal_equations(tloam_ctx* c, double H[36], double g[6], double* cost) {
  if (!c) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  GnState* S = (GnState*)malloc(sizeof(GnState));
  if (!S) return TLOAM_E_INVALID;
  const hipError_t e = hipMemcpy(S, c->state.p, sizeof(GnState), hipMemcpyDeviceToHost);
  if (e == hipSuccess) {
    if (H) memcpy(H, S->H, sizeof(double) * 36);
    if (g) memcpy(g, S->g, sizeof(double) * 6);
    if (cost) *cost = S->x_cost;
  }
  free(S);
  HIPC(c, e);
  return TLOAM_OK;
}

int tloam_get_costs(tloam_ctx* c, int res_type, size_t capacity, size_t* n, double* cost) {
  if (!c || res_type < 0 || res_type >= TLOAM_NUM_RES || !n) return TLOAM_E_INVALID;
  const int kind = res_type == TLOAM_RES_PLANE ? TLOAM_KIND_PLANAR : (res_type == TLOAM_RES_LINE ? TLOAM_KIND_EDGE : TLOAM_KIND_SPHERE);
  return tloam_get_correspondences(c, kind, capacity, n, nullptr, nullptr, nullptr, nullptr, nullptr, cost);
}

int tloam_solve(tloam_ctx* c, double se3[6], tloam_stats* stats) {
  if (!c || !se3) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->partials.p) return TLOAM_E_NOT_READY;
  HIPC(c, hipMemsetAsync(c->state.p, 0, sizeof(GnState), c->stream));
  memcpy(c->h_small, se3, sizeof(double) * 6);
  HIPC(c, hipMemcpyAsync(c->state.p, c->h_small, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
  if (c->dbg_no_eval_reuse) {
    static const int one = 1;
    HIPC(c, hipMemcpyAsync(&c->state.p->no_eval_reuse, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  int rc = enqueue_solve(c, /*armed=*/false, c->dbg_max_sweeps > 0 ? c->dbg_max_sweeps : kSolveSweeps);
  if (rc != TLOAM_OK) return rc;
  HIPC(c, hipMemcpyAsync(c->h_state, c->state.p, sizeof(GnState), hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  const GnState& S = *c->h_state;
  rc = harvest_k3_events(c, S.gn_sweeps);
  if (rc != TLOAM_OK) return rc;
  memcpy(se3, S.x, sizeof(double) * 6);
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->gn_evaluations = S.gn_evaluations;
    stats->gn_sweeps = S.gn_sweeps;
    stats->gn_iterations = S.gn_iterations;
    stats->accepted_steps = S.accepted_steps;
    stats->solver_cost = S.x_cost;
    memcpy(stats->se3, S.x, sizeof(double) * 6);
  }
  return TLOAM_OK;
}

int tloam_time_accumulate(tloam_ctx* c, const double se3[6], int launches, double* mean_us) {
  if (!c || !se3 || launches < 1 || !mean_us) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->partials.p) return TLOAM_E_NOT_READY;
  memcpy(c->h_small, se3, sizeof(double) * 6);
  HIPC(c, hipMemcpyAsync(c->se3_dev.p, c->h_small, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
  launch_set_eval(c->state.p, c->se3_dev.p, c->stream);
  hipEvent_t e0, e1;
  HIPC(c, hipEventCreate(&e0));
  HIPC(c, hipEventCreate(&e1));
  HIPC(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < launches; ++i) launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, true, c->stream);
  HIPC(c, hipEventRecord(e1, c->stream));
  HIPC(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_us = (double)ms * 1e3 / launches;
  return TLOAM_OK;
}

// Sharded contexts (collective call: every rank, same arguments): `launches` sweeps of this rank's block of the
// current set at se3, each followed (with_exchange != 0) by the exchange of the 48 doubles exactly as a GN iteration
// does it -- mailbox: posted by the sweep's last block, gathered by a one-wave kernel; RCCL / callback: all-reduce of
// the folded buffer -- bracketed by one HIP event pair.  with_exchange == 0: the sweeps alone (the last block
// still folds the rows).  The difference of the two is the latency the exchange adds to a GN iteration.
int tloam_time_sharded_sweep(tloam_ctx* c, const double se3[6], int launches, int with_exchange, double* mean_us) {
  if (!c || !se3 || launches < 1 || !mean_us) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->partials.p) return TLOAM_E_NOT_READY;
  memcpy(c->h_small, se3, sizeof(double) * 6);
  HIPC(c, hipMemcpyAsync(c->se3_dev.p, c->h_small, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
  launch_set_eval(c->state.p, c->se3_dev.p, c->stream);
  K3Fuse fuse;
  memset(&fuse, 0, sizeof(fuse));
  fuse.ticket = c->k3_ticket.p;
  fuse.out48 = c->red48.p;
  const bool mbox = with_exchange && c->comm == COMM_MAILBOX && exchanging(c);
  if (mbox) fuse.mb = c->mbox;
  hipEvent_t e0, e1;
  HIPC(c, hipEventCreate(&e0));
  HIPC(c, hipEventCreate(&e1));
  HIPC(c, hipEventRecord(e0, c->stream));
  int rc = TLOAM_OK;
  for (int i = 0; i < launches && rc == TLOAM_OK; ++i) {
    launch_k3_fused(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_wide, true, fuse, c->stream);
    if (mbox) launch_mbox_gather_only(c->red48.p, c->mbox, c->stream);
    else if (with_exchange) rc = allreduce(c, c->red48.p, kReduceBuf);
  }
  HIPC(c, hipEventRecord(e1, c->stream));
  HIPC(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_us = (double)ms * 1e3 / launches;
  return rc;
}

// Timing helper for the bench (roofline_k1): `launches` back-to-back runs of the correspondence-search kernel
// (K1 + K2: SearchHybrid + the four builders) over the source slots of the last scan_match -- same pose, same grids,
// same query order; the kernel only rewrites the raw records and flags it wrote before -- bracketed by one HIP event
// pair.  *queries = source points searched per launch.
int tloam_time_build(tloam_ctx* c, int launches, double* mean_us, int64_t* queries) {
  if (!c || launches < 1 || !mean_us) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (c->active || !c->have_build || !c->qrec.p) return TLOAM_E_NOT_READY;
  BuildParams bp;
  GridView grids[kKinds];
  outer_params(c, &bp, grids);
  hipEvent_t e0, e1;
  HIPC(c, hipEventCreate(&e0));
  HIPC(c, hipEventCreate(&e1));
  HIPC(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < launches; ++i)
    launch_build(c->sv, grids, bp, c->state.p, c->tile_of_slot.p, c->tile_cnt.p, c->tile_scan.p, c->tile_fill.p, c->qrec.p,
                 c->scan_tmp.p, /*rebin=*/false, c->stream, nullptr);
  HIPC(c, hipEventRecord(e1, c->stream));
  HIPC(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mean_us = (double)ms * 1e3 / launches;
  if (queries) *queries = (int64_t)c->sv.slot_off[kKinds];
  return TLOAM_OK;
}

int tloam_k3_timer(tloam_ctx* c, int reset, double* total_us, int64_t* launches, double* algorithmic_bytes) {
  if (!c) return TLOAM_E_INVALID;
  if (total_us) *total_us = c->k3_total_us;
  if (launches) *launches = c->k3_launches;
  if (algorithmic_bytes) *algorithmic_bytes = c->k3_alg_bytes;
  if (reset) {
    c->k3_total_us = c->k3_all_us = 0.0;
    c->k3_launches = c->k3_all_launches = 0;
  }
  c->k3_timing = true;  // first call arms the per-launch event pairs
  return TLOAM_OK;
}

// test aid: the device SE(3) arithmetic of the minimiser step (k_debug_se3), n items of (x, delta) -> 26 doubles each
int tloam_debug_se3(tloam_ctx* c, int n, const double* x, const double* delta, double* out26) {
  if (!c || n < 1 || !x || !delta || !out26) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, c->misc.reserve((size_t)n * 38 + 8));
  double* dx = c->misc.p; double* dd = dx + 6 * (size_t)n; double* dout = dd + 6 * (size_t)n;
  HIPC(c, hipMemcpyAsync(dx, x, sizeof(double) * 6 * n, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dd, delta, sizeof(double) * 6 * n, hipMemcpyHostToDevice, c->stream));
  launch_debug_se3(dx, dd, n, dout, c->stream);
  HIPC(c, hipMemcpyAsync(out26, dout, sizeof(double) * 26 * n, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return TLOAM_OK;
}

// debugging aid: raw copy of the device-resident minimiser state (layout: tl_common.hpp GnState)
int tloam_debug_state(tloam_ctx* c, double* out, int n_doubles) {
  if (!c || !out) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t bytes = std::min(sizeof(GnState), sizeof(double) * (size_t)n_doubles);
  HIPC(c, hipMemcpy(out, c->state.p, bytes, hipMemcpyDeviceToHost));
  return (int)(sizeof(GnState) / sizeof(double));
}

int tloam_debug_partials(tloam_ctx* c, double* out, int n_doubles) {
  if (!c || !out) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t n = std::min(c->partials.cap, (size_t)std::max(n_doubles, 0));   // (rows, then whatever a profiling build put behind them)
  HIPC(c, hipMemcpy(out, c->partials.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return c->k3_grid;
}

// every K3 launch since the last reset, no-op launches (after a tolerance exit) included: the population
// `rocprofv3 --kernel-trace --stats` averages over
int tloam_k3_timer_all(tloam_ctx* c, double* total_us, int64_t* launches) {
  if (!c) return TLOAM_E_INVALID;
  if (total_us) *total_us = c->k3_all_us;
  if (launches) *launches = c->k3_all_launches;
  return TLOAM_OK;
}

// The period of a GN iteration as the DEVICE clocks it (iter_span_note, tl_gn.hip): between the ends of two consecutive
// minimiser steps of one Solve -- sweep, launch boundaries, fold, exchange, step.  The first call arms it.
int tloam_gn_iter_timer(tloam_ctx* c, int reset, double* total_us, int64_t* iterations) {
  if (!c) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  unsigned long long h[4] = {0, 0, 0, 0};
  if (c->iter_span.p) {
    HIPC(c, hipMemcpyAsync(h, c->iter_span.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (reset) HIPC(c, hipMemsetAsync(c->iter_span.p, 0, sizeof(h), c->stream));
  }
  if (total_us) *total_us = (double)h[1] * 0.01;   // 100 MHz wall clock
  if (iterations) *iterations = (int64_t)h[2];
  c->iter_timing = true;
  return TLOAM_OK;
}

int tloam_k3_span(tloam_ctx* c, int reset, double* total_us, int64_t* launches) {
  if (!c) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  unsigned long long h[4] = {0, 0, 0, 0};
  if (c->k3_span.p) {
    HIPC(c, hipMemcpyAsync(h, c->k3_span.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    if (reset) HIPC(c, hipMemsetAsync(c->k3_span.p, 0, sizeof(h), c->stream));
  }
  if (total_us) *total_us = (double)h[1] * 0.01;   // 100 MHz wall clock
  if (launches) *launches = (int64_t)h[2];
  return TLOAM_OK;
}

}  // extern "C"
