// tl_ctx.hpp -- host-side context of the C ABI (include/tloam_hip.h), shared by the API translation units (tl_api.hip:
// lifetime, tl_api_frames.hip: HBM residency + search grids, tl_api_match.hip: the scanMatching driver, tl_api_comm.hip: multi-GPU
// exchange, tl_api_submap.hip: device-resident submap, tl_api_feature.hip: PCA features).
#pragma once

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "tl_common.hpp"


namespace tlh {
using namespace tl;


// ------------------------------------------------------------------------------------------------
//  grow-only device buffer
// ------------------------------------------------------------------------------------------------
template <class T>
struct DBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    size_t want = std::max(n, cap + cap / 2);
    want = (want + 63) & ~size_t(63);
    T* q = nullptr;
    hipError_t e = hipMalloc((void**)&q, want * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    static const bool trace = getenv("TLOAM_DEBUG_ALLOC") != nullptr;   // development aid: (re)allocations inside the timed path show up here
    if (trace) fprintf(stderr, "[tloam alloc] %zu -> %zu elements of %zu B\n", cap, want, sizeof(T));
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct KindData {
  // source (this rank's block)
  size_t n_src_full = 0, src_lo = 0, n_src = 0;
  DBuf<double> src_aos;          // storage when the cloud came through tloam_set_source (one kind at a time)
  const double* src_ptr = nullptr;   // this rank's block of the cloud, AoS: src_aos.p, or inside the frame's src_pack (tloam_set_source_frame)
  bool src_set = false;
  // target as given by set_target
  size_t n_tgt = 0;
  DBuf<double> tgt_aos, tx, ty, tz;
  bool tgt_set = false;
  // view into the shared search-grid buffers built at sm_begin (the deep copy KDTreeFlann::SetGeometry
  // makes, :898-913)
  GridView gv{};
  bool grid_valid = false;
  // compact correspondence segment
  // ONE allocation per kind: kSegStreams arrays of `c_stride` doubles each, in the order of tl::SegStream -- K3 gets
  // the planar segment as (base, stride) in its first kernel arguments, which the command processor preloads into
  // SGPRs, so a wave can request its first chunk before any scalar load has returned
  DBuf<int> c_idx;
  DBuf<double> c_buf;
  size_t c_cap = 0, c_stride = 0;
  size_t pre_lo = 0, pre_n_full = 0;  // pre-built sets: this rank's block
};

// the clouds of one registered frame (what setInputSource / setInputTarget hand over), movable as a unit
struct FrameClouds {
  size_t n_src_full[tl::kKinds] = {}, src_lo[tl::kKinds] = {}, n_src[tl::kKinds] = {}, n_tgt[tl::kKinds] = {};
  DBuf<double> src_aos[tl::kKinds], tgt_aos[tl::kKinds], tx[tl::kKinds], ty[tl::kKinds], tz[tl::kKinds];
  DBuf<double> src_pack;                       // the four source clouds of a Frame handed over in one piece (tloam_set_source_frame)
  const double* src_ptr[tl::kKinds] = {};
  bool src_set[tl::kKinds] = {}, tgt_set[tl::kKinds] = {};
  double tgt_box[tl::kKinds][6] = {};
  bool tgt_box_valid[tl::kKinds] = {};
  void release() {
    for (int k = 0; k < tl::kKinds; ++k) { src_aos[k].release(); tgt_aos[k].release(); tx[k].release(); ty[k].release(); tz[k].release(); src_ptr[k] = nullptr; }
    src_pack.release();
  }
};

struct GridBuffers {
  DBuf<double4> gp;
  DBuf<int> cell_start, cell_of_pt, rank_of_pt;
  DBuf<unsigned long long> cell_cnt, cell_scan, scan_tmp;
  DBuf<unsigned long long> scan1p;   // control words of the single-pass scan (large tables), zero when allocated
  DBuf<double> bbox;
  void release() {
    gp.release(); cell_start.release(); cell_of_pt.release(); rank_of_pt.release();
    cell_cnt.release(); cell_scan.release(); scan_tmp.release(); scan1p.release(); bbox.release();
  }
};


enum CommMode { COMM_NONE = 0, COMM_CALLBACK = 1, COMM_RCCL = 2, COMM_MAILBOX = 3 };

// ---- device-resident submap (front_end.cpp:201-275): frame buffers + scratch of the crop/voxel pipeline ----
struct RingFrame {
  DBuf<double> aos;   // the frame's cloud, sensor frame, as handed over
  size_t n = 0;
  double pose[16];
};
struct SubmapState {
  bool inited = false;
  unsigned long long pending_seq = 0ull;  // sequence number the last kernel of the update in flight stores into the host slot
  tloam_submap_config cfg;
  std::vector<RingFrame*> planar_ring, sphere_ring;  // oldest first (std::deque in the reference)
  DBuf<double> in_aos, wx, wy, wz, min_partial, vmin;
  DBuf<unsigned long long> keys, cnt, off, leader, leader_scan, scan_tmp, counts;
  DBuf<int> slot_of_pt, urank, members, sorted, overflow;
  void release() {
    for (auto* f : planar_ring) { f->aos.release(); delete f; }
    for (auto* f : sphere_ring) { f->aos.release(); delete f; }
    planar_ring.clear(); sphere_ring.clear();
    in_aos.release(); wx.release(); wy.release(); wz.release(); min_partial.release(); vmin.release();
    keys.release(); cnt.release(); off.release(); leader.release(); leader_scan.release(); scan_tmp.release();
    counts.release(); slot_of_pt.release(); urank.release(); members.release(); sorted.release(); overflow.release();
    inited = false;
  }
};

// scratch of the PCA feature path (grow-only, kept across calls)
struct FeatBuffers {
  DBuf<double> aos, x, y, z, flatness, cvr, sphericity, normal;
  DBuf<double> f2, gf, out;        // candidate flatness of both lists ([0, n) planar, [n, 2n) sphere), grouped by bucket, ranked + packed
  DBuf<int> num_sum, neigh, idx2, gi, bkt, pos;
  DBuf<unsigned long long> flags, scan, scan_tmp;
  DBuf<tl::FeatRankCtl> rank_ctl;
  GridBuffers grid;
  void release() {
    aos.release(); x.release(); y.release(); z.release(); flatness.release(); cvr.release(); sphericity.release();
    normal.release(); f2.release(); gf.release(); out.release(); num_sum.release(); neigh.release();
    idx2.release(); gi.release(); bkt.release(); pos.release(); flags.release(); scan.release();
    scan_tmp.release(); rank_ctl.release(); grid.release();
  }
};

// up to four SoA clouds (x, y, z, n) a search grid is built over -- the registered targets, or any other cloud
struct CloudRef { const double *x, *y, *z; size_t n; };
}  // namespace tlh

using namespace tlh;

struct tloam_ctx {
  tloam_tls_config cfg;
  SubmapState submap;
  FeatBuffers feat;
  int device = 0;
  hipStream_t stream = nullptr;
  KindData kd[kKinds];
  // concatenated per-source-slot arrays of the current scan_match
  DBuf<double> sx, sy, sz, w_src, raw;
  DBuf<double> fit_x, fit_y, fit_z;  // getFitnessScore scratch
  DBuf<unsigned long long> flags, scan, scan_tmp, tile_cnt, tile_scan;
  DBuf<unsigned long long> scan1p_q;   // control words of the single-pass scan of the query-sort histogram, zero when allocated
  bool scan1p_q_use = false;           // ... and whether this frame's query sort takes it (outer_reserve: scan_1p_applies on this device)
  bool grids_ahead = false;    // the search grids in `grids_next` were built over the registered targets at hand-over (tloam_set_target_frame) and are still theirs
  // ... which is also CHECKED: every path that changes a registered target cloud advances tgt_gen, the grids built ahead remember
  // the generation they were built over, and tloam_sm_begin swaps them in only if that is still the current one
  unsigned long long tgt_gen = 0, grids_next_gen = ~0ull;
  bool no_grid_ahead = false;  // TLOAM_NO_GRID_AHEAD: the grids are always built inside scanMatching (A/B, tests)
  bool no_scan_1p = false;     // set after a single-pass look-back scan timed out (blocks not co-resident): the multi-launch scans from then on
  DBuf<unsigned char> flagb;   // SlotView::flagb
  DBuf<double> fin_rows;       // hand-over rows of the finish riding on a thread-per-query search (k_build_finish_large)
  bool fused_large = false;    // TLOAM_FUSED_LARGE: a GN iteration of a large set as ONE launch (k3_sweep_step; sharded + mailbox: sweep, exchange
                               // and step).  Measured slower than sweep + step as two launches: off by default
  DBuf<int> tile_of_slot, tile_fill;
  DBuf<double4> qrec;  // tile-sorted query records (x, y, z, slot)
  GridBuffers grids;  // the four search grids of the last scanMatching (shared buffers)
  GridBuffers grids_next;      // ... and the set built AHEAD, over targets just handed over (tloam_set_target_frame); swapped in by the next sm_begin
  tl::GridView gv_next[tl::kKinds];
  SlotView sv{};
  CorrView cv{};
  DBuf<int> seg_n;
  DBuf<double> partials, red48, sums16, wpart, rank_counts, se3_dev, bbox_dev, misc;
  DBuf<GnState> state;
  GnState* h_state = nullptr;  // pinned mirror
  double* h_small = nullptr;   // pinned scratch (>= 64*6*4 doubles)
  int k3_grid = 1;
  bool k3_single = false;
  bool k3_wide = false;        // the streaming sweep goes out as blocks of eight waves (k3_plan); k3_grid = blocks launched = rows
  int dbg_max_sweeps = 0;          // development knobs, read from the environment once at create
  bool dbg_no_build_reuse = false;
  bool dbg_no_eval_reuse = false;
  bool no_device_loop = false;     // TLOAM_NO_DEVICE_LOOP: tloam_scan_match keeps the host in the outer loop (A/B, tests)
  bool no_persistent_solve = false;  // TLOAM_NO_PERSISTENT_SOLVE, or set by tloam_scan_match after an in-launch hand-over timed out:
                                     // KITTI-size Solves run one launch per GN iteration instead of k_solve_all
  bool hand_over_timed_out = false;  // the last TLOAM_E_HIP of the device loop was OS_COMM_ERROR on one rank
  int dbg_fail_handover = 0;       // TLOAM_DEBUG_FAIL_HANDOVER=n: the next n one-launch Solves time out in their first hand-over (test hook)
  int device_cus = 0;              // multiProcessorCount of the device (k_solve_all and the single-pass scans need all their blocks resident at once)
  double* h_bbox = nullptr;        // pinned, device-visible: [4][64][6] bounding-box rows
  double* h_bbox_dev = nullptr;
  // bounds of the registered target clouds, taken at hand-over (set_target*: the call synchronises anyway), so that
  // scanMatching can size its search grids without a host round trip of its own
  double tgt_box[tl::kKinds][6];
  bool tgt_box_valid[tl::kKinds] = {false, false, false, false};
  double wait_us = 0.0;            // time the host spent waiting for the device in the current scan_match
  double hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // development aid (TLOAM_HOST_STAMPS): where the calling thread's time goes, per frame
  long hs_n = 0;
  double hs_exit = 0.0;
  double hs_entry = 0.0;          // > 0: stamping, entry time of the scan_match in progress (us, steady clock)
  tl::MirrorSlot* h_mirror = nullptr;       // pinned, device-visible result slots (HostMirror targets), 64-byte aligned
  tl::MirrorSlot* h_mirror_dev = nullptr;   // ... as the device addresses them
  unsigned long long mirror_seq = 0;
  // fault words a kernel raises when a bounded in-launch wait runs out (pinned, device-visible; see check_device_faults):
  // [0] single-pass look-back scan (tl_nn.hip), [1] k_vox_emit's look-back (tl_submap.hip)
  unsigned* h_fault = nullptr;
  unsigned* h_fault_dev = nullptr;
  bool scan1p_retried = false;     // tloam_scan_match has re-run a frame after a look-back scan timed out (once per context)
  bool vox_ticket = false;         // set after k_vox_emit's look-back timed out: its blocks take start tickets from then on
  DBuf<double> src_pack;                        // the registered Frame's four source clouds in one piece (tloam_set_source_frame)
  // pinned staging of tloam_set_source_frame: the borrowed host clouds are copied here (two halves, used alternately; an
  // event per half says when the device has read it) and go to HBM with ONE asynchronous copy -- the call returns without
  // waiting for the device, the frame's first kernel is ordered behind the copy by the stream
  double* h_stage[2] = {nullptr, nullptr};
  double* h_stage_dev[2] = {nullptr, nullptr};  // the halves as the device addresses them (kernels that read the staging in place)
  size_t h_stage_cap[2] = {0, 0};               // doubles
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};
  int stage_next = 0;
  std::vector<tlh::FrameClouds*> frame_store;   // tloam_frame_stash / tloam_frame_select
  int frame_selected = -1;                      // slot whose clouds are the registered ones (-1: the context's own)
  int dbg_planned_sweeps = 0;      // TLOAM_PLANNED_SWEEPS: force the sweep budget per Solve (exercises the top-up)
  std::vector<int> planned_sweeps; // per outer iteration x 3: sweeps the Solve needed in the last three frames
  bool prebuilt = false;
  // comm
  int rank = 0, nranks = 1;
  CommMode comm = COMM_NONE;
  // A mailbox or RCCL set-up with nranks == 1: the context exchanges with ITSELF -- every launch of the sharded forms runs (fused
  // sweep + post, gather + step, the side exchanges), the exchange is a loop-back, the results are those of the single-rank forms
  // bit for bit (one row folded onto +0.0).  What times the sharded forms at shard size on ONE GPU (bench: shard_size_iterations)
  // and makes a one-rank RCCL communicator actually carry the all-reduce (tests/test_gpu_multirank.py)
  bool loopback = false;
  tloam_allreduce_fn cb = nullptr;
  void* cb_user = nullptr;
  void* nccl_comm = nullptr;
  // one-shot peer exchange (tl_common.hpp MboxView): the local buffer (fine-grained device memory, exported through
  // HIP IPC), the peers' buffers as mapped here, the device-resident exchange counter, the ticket of the fused sweep
  double* mbox_local = nullptr;
  void* mbox_opened[tl::kMaxRanks] = {};   // hipIpcOpenMemHandle results (closed at destroy)
  tl::MboxView mbox{};
  DBuf<unsigned long long> mbox_ctr;
  DBuf<int> k3_ticket;
  DBuf<unsigned long long> k3_span;    // K3Step::span: streaming span of the one-launch GN iterations (100 MHz ticks, launches)
  DBuf<unsigned long long> iter_span;  // iter_span_note (tl_gn.hip): [0] last stamp, [1] ticks, [2] periods; handed to the kernels once
  bool iter_timing = false;            // tloam_gn_iter_timer has armed it
  // bounded in-launch waits that ran out on this context (look-back scan, voxel look-back, one-launch Solve hand-over): every
  // one moved the context to a form that waits for nothing -- tloam_get_info reports which, and how often
  int fallback_events = 0;
  bool persistent_solve_timed_out = false;   // no_persistent_solve was set by a time-out, not by TLOAM_NO_PERSISTENT_SOLVE
  // the DIRECT factor set of large frames (tl_common.hpp DirectSet): chosen per frame by tloam_sm_begin
  bool no_qbin_ride = false;   // TLOAM_NO_QBIN_RIDE: the query sort's first pass as a launch of its own (A/B)
  bool qbin_rode = false;      // this frame's query binning rode on the grid build (launch_build skips its own)
  bool no_direct_set = false;  // TLOAM_NO_DIRECT_SET: large frames compact as the others do (A/B, tests)
  bool direct = false;         // this frame's set is direct: rows = tile-sorted queries, holes, two weight streams
  bool set_stale = false;      // ... and its rows hold the geometry of a search whose set was never solved (OS_SET_STALE): rebuilt at
                               // x_build before a getter reads them
  int w_parity = 0;            // weight stream the CURRENT GNC weights are in (the captured ones of the last Solve: the other)
  DBuf<GnState> state_scratch; // a copy of the state with T_cur = exp(x_build), for that rebuild
  DBuf<int> row_of_pos;        // DirectSet::row_of_pos: the row of every sorted query position
  DBuf<int> fin_tickets;       // FinishDirect::ticket: the top ticket + one per group of 64 finish blocks (zero between launches)
  DBuf<int> blk_cnt;           // DirectSet::blk_cnt: [2 parities][search blocks][4]
  size_t blk_cnt_n = 0;        // search blocks of this frame
  // scanMatching host state
  bool active = false;
  bool have_build = false;   // the compact set matches build_x
  double build_x[6] = {0, 0, 0, 0, 0, 0};
  int iter = 0;
  double mu = 1.0, noise_bound_sq = 1e-4;
  double prev_cost[kKinds], cur_cost[kKinds];
  tloam_stats stats;
  // K3 timing (bench roofline)
  bool k3_timing = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<int> ev_batch_idx;   // position of each sampled launch inside its batch (solve)
  int batch_launches = 0;          // K3 launches enqueued since the last harvest
  long long k3_seq = 0;            // all K3 launches of this context
  double k3_total_us = 0.0, k3_all_us = 0.0;  // working sweeps only / every K3 launch incl. no-ops
  int64_t k3_launches = 0, k3_all_launches = 0;
  double k3_alg_bytes = 0.0;  // algorithmic bytes of ONE sweep over the current set
  std::string last_error;
};

#define HIPC(ctx, expr)                                                                 \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);            \
      return TLOAM_E_HIP;                                                               \
    }                                                                                   \
  } while (0)

// pinned result slots: one per outer iteration of a device-driven frame (slot 0: stepwise API); the last one also
// carries the sizes of a submap update back (tl_api_submap.hip)
constexpr int kMirrorSlots = 8;
constexpr int kFaultWords = 16;
constexpr int kFaultScan1p = 0, kFaultVoxEmit = 1;   // tloam_ctx::h_fault
// Points a single cloud / correspondence set / query batch may hold: slots, cells and ranks are 32-bit integers throughout, and the
// four kinds of a frame share one slot space -- 2^28 points (6.4 GB as doubles) per cloud keeps every sum over the four kinds
// (<= 2^30) and every 3 n inside it (2^29 did not: four clouds at the bound sum to INT_MAX + 1).  Sums that grow behind the
// entry points -- a submap's accumulated clouds -- are checked where they are formed (submap_update_body).
// More is TLOAM_E_INVALID at the entry point, not an overflow behind it.
constexpr size_t kMaxPoints = (size_t)1 << 28;

namespace tlh {
// ---- small helpers shared by the API units
inline double kind_radius(const tloam_tls_config& c, int k) {
  switch (k) {
    case TLOAM_KIND_PLANAR: return c.planar_dist_thres;
    case TLOAM_KIND_GROUND: return c.ground_dist_thres;
    case TLOAM_KIND_EDGE: return c.edge_dist_thres;
    default: return c.sphere_dist_thres;
  }
}
inline int kind_maxnum(const tloam_tls_config& c, int k) {
  switch (k) {
    case TLOAM_KIND_PLANAR: return c.planar_maxnum;
    case TLOAM_KIND_GROUND: return c.ground_maxnum;
    case TLOAM_KIND_EDGE: return c.edge_maxnum;
    default: return c.sphere_maxnum;
  }
}
// registration.cpp:979-1016: factor_num 4 -> all four builders, 3 -> planar+ground+edge, 2 -> planar+ground
inline int kind_active(const tloam_tls_config& c, int k) {
  if (c.factor_num == 4) return 1;
  if (c.factor_num == 3) return k != TLOAM_KIND_SPHERE;
  if (c.factor_num == 2) return k == TLOAM_KIND_PLANAR || k == TLOAM_KIND_GROUND;
  return 0;
}
inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
inline bool one_rank(const tloam_ctx* c) { return c->nranks == 1 && !c->loopback; }    // the single-rank launch forms apply
inline bool exchanging(const tloam_ctx* c) { return c->nranks > 1 || c->loopback; }    // the sharded launch forms run
// tl_api_comm.hip
int allreduce(tloam_ctx* c, double* dev, int count);   // sum all-reduce of a small device buffer of doubles across the context's ranks
void comm_release(tloam_ctx* c);
void comm_rccl_info(const tloam_ctx* c, int32_t* count, int32_t* user_rank);   // ncclCommCount / ncclCommUserRank of the context's communicator, -1 without one
// tl_api_match.hip
int reserve_seg(tloam_ctx* c, int k, size_t n);        // compact correspondence segment of kind k for n factors
int ensure_common(tloam_ctx* c);                       // the context's small fixed device buffers
int reserve_query_sort(tloam_ctx* c, const tl::GridView grids[tl::kKinds]);   // the buffers of the frame's query sort, sized by the grids
// tl_api_frames.hip
void exchange_clouds(tloam_ctx* c, FrameClouds& F);    // the registered clouds <-> a FrameClouds (pointers and counts only)
int check_device_faults(tloam_ctx* c);
int wait_word(tloam_ctx* c, const unsigned long long* p, unsigned long long seq);
int wait_segment(tloam_ctx* c, const unsigned long long* seg, unsigned long long seq, unsigned long long payload[7]);
int stage_and_upload(tloam_ctx* c, const double* const parts[], const size_t counts[], int nparts, double* dev_dst, size_t offs[]);
int stage_in_place(tloam_ctx* c, const double* const parts[], const size_t counts[], int nparts, size_t offs[], const double** dev_view,
                   int* half);
int stage_release(tloam_ctx* c, int half, bool completed);
size_t staged_size(const size_t counts[], int nparts);
int build_grids_over(tloam_ctx* c, GridBuffers& G, const double radius[tl::kKinds], const CloudRef clouds[tl::kKinds],
                     tl::GridView out[tl::kKinds], const double (*known_boxes)[6] = nullptr,
                     tl::FrameInitHook* frame = nullptr);
int build_grids(tloam_ctx* c, GridBuffers& G, const double radius[tl::kKinds], tl::GridView out[tl::kKinds],
                tl::FrameInitHook* frame = nullptr);
void reduce_box_rows(const double* box_rows, double boxes[tl::kKinds][6]);   // rows of k_bbox_all / k_ingest_targets -> (lo, hi) per kind
int enqueue_target_bounds(tloam_ctx* c);
void finish_target_bounds(tloam_ctx* c);
}  // namespace tlh
