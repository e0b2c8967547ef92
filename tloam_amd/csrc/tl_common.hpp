// tl_common.hpp -- shared device-side data structures of the MI355X T-LOAM pose-optimisation path.
//
// HBM layout (all fp64 SoA, one array per component, 256-B aligned, sized for 288 GB parts):
//   source clouds   sx/sy/sz[k][n_src_local], GNC weights w_src[k][n]      (one slot per source point)
//   target clouds   tx/ty/tz[k][n_tgt] (as given) + cell-sorted packed copy gp[k] (x,y,z,idx) + cell table
//   raw records     per source slot: a(3) b(3) d + flags        (written by the builder, K1+K2)
//   compact set     one SoA segment per kind: idx, px py pz, ax ay az, [bx by bz], [d], w, cost
//                   (streamed by K3; cost written back)
//   partials        [gridDim][32] per-block sums of H(21) g(6) cost(1)
//   GnState         the Ceres-minimiser / dogleg state machine + pose, device resident
#pragma once
#include <cstddef>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tloam_hip.h"
#include "tl_se3.hpp"

namespace tl {

constexpr int kKinds = 4;
constexpr int kAccN = 28;        // 21 upper-triangular H + 6 g + 1 cost
constexpr int kAccStride = 32;   // padded row of the partials buffer
constexpr int kReduceBuf = 48;   // the all-reduced buffer: 28 used + per-kind tail
constexpr int kChunk = 128;      // correspondences per wave-chunk in K3 (64 lanes x 2)
constexpr int kMaxK = 8;         // max neighbours of tloam_knn
constexpr int kMaxRanks = 16;    // ranks one frame can be sharded over (rank_counts rows, mailbox slots); a node has 8 GPUs

// residual type of a kind (registration.cpp:981-992 builder -> cost functor)
__host__ __device__ inline int res_type_of_kind(int kind) {
  return kind == TLOAM_KIND_EDGE ? TLOAM_RES_LINE : (kind == TLOAM_KIND_SPHERE ? TLOAM_RES_POINT : TLOAM_RES_PLANE);
}

// ---- uniform grid over a target cloud (the device stand-in for KDTreeFlann) -----------------
struct GridView {
  const double4* gp;  // cell-sorted targets, packed (x, y, z, original index as integer bits): ONE 32-byte
                      // record per candidate, so a cell's points share cache lines
  const int* cell_start;  // ncell + 1
  double org[3];
  double inv_cell;    // 1 / cell
  double cell;
  int dim[3];
  int n;
};

// ---- compact correspondence set: one SoA segment per kind ------------------------------------
// stream order inside a kind's buffer (stream s starts at base + s * stride doubles)
enum SegStream : int { SS_PX = 0, SS_PY, SS_PZ, SS_AX, SS_AY, SS_AZ, SS_W, SS_D, SS_BX, SS_BY, SS_BZ, SS_COST, SS_W2, kSegStreams };
// (SS_W2: the second weight stream of a DIRECT set -- see DirectSet below; unused by compact sets)
struct CorrSeg {
  int* idx;                 // global source index of each correspondence
  double *px, *py, *pz;     // source point (sensor frame)
  double *ax, *ay, *az;     // plane normal | line point a | target point
  double *bx, *by, *bz;     // line point b (edge only, else null)
  double* d;                // plane offset (planar/ground only, else null)
  double* w;                // TLS weight captured at build time
  double* cost;             // side channel (`mutable double* cost`, registration.hpp:51,76,96)
  int cap;                  // capacity (multiple of kChunk)
  int stride;               // doubles between consecutive streams of the kind's buffer (SegStream order)
};
struct CorrView {
  CorrSeg k[kKinds];
  const int* seg_n;         // DEVICE: number of valid entries per kind [4]
};

// ---- the DIRECT factor set of large frames (round 6) ------------------------------------------------------------------------
// Frames searched one thread per query (> 131072 source points), on one rank, with every kind active and caps that cannot bind
// (a kind's cap >= its source points: the 1 M-class frames) do not compact at all.  The search processes its queries in the
// TILE-SORTED order of the frame's query sort; ROW r of kind k's segment IS the query at sorted position slot_off[k] + r, for the
// whole frame, and the search writes the factor's geometry straight into that row (coalesced: consecutive lanes are consecutive
// rows).  A query without a factor leaves a HOLE: a row that adds exact zeros to every sum and whose side-channel cost is exactly
// 0 -- plane n = 0, d = 0 (r = 0, J = 0 whatever the weight); line a = b = 0 (the sweep takes 1 / |a - b| as 0); point q.x = NaN
// (the sweep takes weight and q.x as 0) -- so the sweep and the weight update run over the rows as they are, ~10 % of them holes.
//   p, idx     written by the frame's first search (the rows never move); idx[r] = the source index, ~index (negative) on a hole:
//              what the finish counts the factors by and the getters filter on
//   a, b, d    written by every search
//   w          TWO streams: the Solve of outer iteration k reads stream k & 1 -- the weights "captured by value at construction"
//              (registration.hpp:51,76,96) --, its finish writes every row of stream (k + 1) & 1 (updateWeight, registration.cpp:
//              858-876; a row it leaves alone is copied).  No capture pass, no per-slot weight array
//   cost       written by every sweep (every row: a hole's is 0), read by the finish
// What is gone: the flag scan, the compaction (~45 us per outer iteration of the 1 M frame) and the raw records.  What it costs:
// the holes in the sweep (+10 % rows).  The sums run over the rows in tile order instead of index order: same factors, another
// rounding of the last bits (the parity bar on the pose is 1e-9).  Frames whose caps CAN bind, sharded frames and everything
// smaller keep the compact set: "first N valid in index order" (registration.cpp:448, :538, :592, :735) needs the index order.
struct DirectSet {
  int on;          // 0: raw records + flags (compacted afterwards)
  int first;       // the frame's first search: also writes p, idx, and weight stream 0 = 1 (registration.cpp:931-949)
  int set_x_build; // the launch records the pose the set is built at (GnState::x_build); 0 when a finish in the SAME launch does
  // factors per kind found by every one-wave block of THIS search, [blocks][4]: what n_corr is added up from -- by the finish of
  // the outer iteration the set is solved in, not from the rows (the NEXT search may be rewriting them beside that finish).
  // Two buffers, by the parity of the outer iteration the search belongs to.
  int* blk_cnt;
  // The ROW of a sorted position.  The frame's query sort places the queries of one bin (a cell, or a tile) in the ARRIVAL order of
  // an atomic: two runs over the same frame permute them, and a factor set whose rows followed the sorted positions would add its
  // sums up in another order every time (same factors, other last bits).  The rows therefore take the bins' queries in SLOT order:
  // the frame's first search works out, per query, how many queries of its bin have a lower slot (bins hold one or two queries on
  // average; beyond kDirectBinMax the arrival order stands) and keeps the row of every sorted position for the later searches.
  // Run to run the same rows, the same sums, the same bits -- as with a compact set.
  const int* tile_of_slot;
  const unsigned long long* tile_scan;
  int* row_of_pos;
};
constexpr int kDirectBinMax = 1024;

// ---- per-source-slot arrays of one scan_match (concatenated over kinds) ----------------------
constexpr int kFlagbStride = 4096;   // 64 lanes x 64 slots
struct SlotView {
  const double *sx, *sy, *sz;  // source points (sensor frame)
  double* w_src;               // GNC weights
  // raw builder output: one 64-byte record per slot (a[3], b[3], d, pad) -- K1 writes it with four 16-byte
  // stores into ONE line (the seven separate SoA arrays cost a 32-byte sector per 8-byte scattered store:
  // PMC WRITE_SIZE 269 MB per 1 M queries for 64 MB of records)
  double* raw;
  unsigned long long* flags;   // low 32: counted, high 32: valid (then scanned in place -> exclusive prefix)
  unsigned long long* scan;    // exclusive scan of flags
  // the same two bits per slot as ONE BYTE (bit 0 valid, bit 1 counted) at [kind * kFlagbStride + slot-in-kind], or null:
  // what the one-launch Solve reads when it compacts the factor set itself (k_solve_all, SolvePrep) -- a lane takes in
  // 64 slots with four 16-byte loads.  Only for frames whose kinds have at most kFlagbStride source points each.
  unsigned char* flagb;
  int slot_off[kKinds + 1];    // concatenated slot ranges per kind
  int src_lo[kKinds];          // global index of local slot 0 (sharded contexts)
};

// ---- Ceres TrustRegionMinimizer + DoglegStrategy state -----------------
enum GnPhase : int { PH_ITER0 = 0, PH_CAND = 1 };

struct GnState {
  // ---- host-visible prefix: what the host reads after an outer iteration / a Solve.  The finish kernels
  //      mirror exactly these kMirrorWords 8-byte words (two 64-byte PCIe writes) and then host_seq.
  double x[6];        // accepted iterate == `parameters` (registration.hpp:328)
  double x_cost;
  double kind_cost[kKinds];
  int n_corr[kKinds];
  // counters (whole scan_match)
  int gn_evaluations, gn_iterations, accepted_steps;
  int gn_sweeps;      // sweeps actually consumed (<= gn_evaluations, see gn_consume_wave)
  int bad_weights;
  int incomplete;     // set by the gated weight/finish kernels when the Solve had not terminated yet
  unsigned long long host_seq;  // sequence number of the host mirror (see HostMirror); right after the prefix
  // ---- device-only from here
  double x_cand[6];   // candidate evaluated by the sweep in flight
  Pose T_eval;        // exp(point being swept)
  Rt Rt_eval;         // the same as rotation matrix + translation (what K3 streams against)
  Pose T_cur;         // exp(x), used by the builders
  // minimiser
  double x_norm, gmax, model_cost_change;
  double g[6], H[36]; // robustified normal equations at x
  double S[6];        // Jacobi scaling (fixed at iteration 0 of each Solve)
  // dogleg (DoglegStrategy).  Kept in the form the step's hot path needs (tl_step.hpp): D = the clamped diagonal
  // itself (min_diagonal_^2 .. max_diagonal_^2, i.e. Ceres' diagonal_ SQUARED), gn = the Gauss-Newton step in the
  // Jacobi-scaled space (-y of (Hs + mu D) y = gs, i.e. gauss_newton_step_ / diagonal_), gn_norm = | gauss_newton_step_ |
  double radius, mu, alpha, step_norm, gn_norm;
  double D[6], grad[6], gn[6], U[12], sg[2], sB[4];
  int reuse, subspace_1d;
  int cand_gn;        // the candidate in the state (x_cand, T_eval, model_cost_change) is the Gauss-Newton step of the dogleg data above
  int phase, iteration, invalid, step_successful, done;
  int no_eval_reuse;  // development knob (TLOAM_NO_EVAL_REUSE): every evaluation runs its own sweep
  int comm_error;     // a mailbox exchange timed out (sharded contexts): published to the host as incomplete = 3
  // ---- outer GNC loop driven from the device (tloam_scan_match enqueues ALL outer iterations at once and waits once)
  double x_build[6];  // `parameters` the compact set was built at (written by the compaction)
  double prev_planar; // planar side-channel cost sum of the previous outer iteration (registration.cpp:1108), +inf at start
  int stop;           // 0 running | 1 the loop has ended (plateau break :1108 or max_iterations) | 2 Solve out of budget
  int run_build;      // gates of the NEXT outer iteration's launches: builders + scan + compaction run iff the pose moved,
  int run_refresh;    // the refresh iff it did not (see k_refresh); both 0 once the loop has ended
  int spec_build;     // written by the minimiser step: 1 iff the Solve has terminated at a pose other than x_build -- the gate of
                      // a correspondence search that runs CONCURRENTLY with the finish of that Solve (k_build_finish_small)
  int next_outer;     // device-driven loop: the outer iteration to run next (advanced by every finish).  A one-launch Solve that
                      // also finishes its iteration may run the FOLLOWING ones too (SolveFinish); the launches enqueued for
                      // those find their iteration taken and return
  int fin_valid;      // fin_sum / fin_bad hold the finish sums of the Solve that has just ended (k_solve_all); consumed by the finish
  double fin_sum[kKinds], fin_bad;
  double dbg[12];     // LAST: phase time stamps of the step kernel (TLOAM_STEP_PROFILE builds only)
};
constexpr int kMirrorWords = 16;  // x[6], x_cost, kind_cost[4], n_corr[4] (2 words), 6 ints (3 words)
static_assert(offsetof(GnState, host_seq) == kMirrorWords * 8, "host-visible prefix of GnState");
// End of an outer iteration: the finish kernel copies the host-visible prefix of the state into pinned host
// memory and then stores the sequence number, so the host reads the result by polling one word instead of paying a
// copy kernel plus a stream synchronisation per outer iteration (registration.cpp:1108 is a host decision).
// out == nullptr: off.
// The slot is written WITHOUT any fence: it is three 64-byte segments, each carrying seven words of the prefix and,
// last, check_mix(sequence number) XOR seg_word of each of those seven words; one wave stores all 24 words with one instruction.  In
// practice every segment leaves the GPU as one aligned 64-byte write (a full cache line for the host); the host does
// not rely on that: it accepts a segment only when the check word agrees with the payload it reads
// (tlh::wait_segment), so a torn segment reads as "not there yet".  (A system-scope release would first write back
// every dirty L2 line of the kernel, ~2 us; writing the number behind the words with only a wave-level wait is NOT
// safe: the two cache lines travel through different L2 channels.)
struct MirrorSlot {
  unsigned long long w[24];
};
// The check word of every 64-byte hand-over segment (result slots, block rows, pose broadcasts, finish sums, submap sizes) is
// check_mix(number) XOR the seven payload words.  Consecutive numbers differ in a few low bits; multiplied by an odd 64-bit
// constant they differ in about half of all bits, so a segment whose check word is still the old one and of which ONE payload
// word has arrived can only pass if that word changed by exactly that 64-bit pattern -- not by a counter step or a
// last-mantissa-bit change.  Zeroed memory checks for number 0 only; numbers start at 1.
__host__ __device__ inline unsigned long long check_mix(unsigned long long number) { return number * 0x9E3779B97F4A7C15ull; }
// The segments the HOST reads (result slots, submap sizes) enter their check word through seg_word: position-dependent and
// non-linear in the word.  With a plain XOR of the payload a segment whose check word had arrived and whose payload had not
// passed whenever the old and the new payload XORed alike -- (58, 58) after (0, 0), (459, 16) after (458, 17): the sizes of a
// submap update read back as those of the update before in 2-4 % of contexts (round 5, tests/tools/stress_rows.py).
__host__ __device__ inline unsigned long long seg_word(unsigned long long w, int pos) {
  unsigned long long v = (w + 0x9E3779B97F4A7C15ull * (unsigned long long)(pos + 1)) * 0xBF58476D1CE4E5B9ull;
  return v ^ (v >> 29);
}
struct HostMirror {
  MirrorSlot* out;
  unsigned long long seq;
};
// status of an outer iteration as the host reads it in GnState.incomplete
// OS_NEEDS_HOST (a flag on top of the status, result slot only): the launch that finished this iteration inside the Solve
// has ended because the pose moved -- the loop goes on with a correspondence search, which the host enqueues unless it
// already has
// OS_SET_STALE (a flag like OS_NEEDS_HOST, direct sets only): the loop ended in the launch whose search had already run on the
// Solve's own verdict -- the rows hold the geometry of a set that was never solved; the host rebuilds them before anybody reads
enum OuterStatus : int { OS_OK = 0, OS_INCOMPLETE = 1, OS_COMM_ERROR = 3, OS_PLATEAU = 4, OS_SKIPPED = 8, OS_NEEDS_HOST = 16, OS_SET_STALE = 32 };
// device-driven loop control handed to the finish kernels (fast == 0: the host decides, as in the stepwise API)
struct OuterCtl {
  double cost_threshold;  // registration.cpp:1108
  int fast;               // 1: evaluate the plateau test / pose comparison on the device and set the gates
  int last;               // this is outer iteration max_iterations - 1
  int direct;             // direct set: a finish that sends the loop into a rebuild records the pose of that build (x_build) itself
                          // -- the search that does the rebuild runs beside it in the same launch -- and flags a stale set
};

// ---- one-shot peer exchange of a sharded context ("mailbox") -------------
// Every rank owns a small buffer in fine-grained (uncached) device memory that its peers map through HIP IPC:
//   [2 parities][kMaxRanks writers][kMboxSlot doubles]   (<= 64 values ... , word kMboxSlot-1 = the exchange id)
// Exchange number id (1, 2, ...; counted on the device, identically on every rank): rank r stores its values
// into slot [id & 1][r] of EVERY rank's buffer over xGMI, fences at system scope, stores the id; a reader waits for
// the id in all nranks slots of its OWN buffer and adds the rows in rank order (deterministic, identical on all
// ranks).  Two parities suffice: a rank can only post id + 2 after it has gathered id + 1, which every rank posts
// only after it has finished gathering id.
constexpr int kMboxSlot = 80;   // up to 64 values + padding; the last word carries the exchange id
struct MboxView {
  double* peer[kMaxRanks];        // device-visible address of every rank's buffer (peer[rank] = the local one)
  unsigned long long* ctr;        // [0] exchanges completed so far (device-resident, survives frames); [1] error flag
  int rank, nranks;
};
constexpr size_t kMboxDoubles = (size_t)2 * kMaxRanks * kMboxSlot;

// ---- host-callable launchers (defined in tl_nn.hip / tl_gn.hip) ------------------------------
struct ScanTemp;  // opaque

// AoS double[3] -> SoA, for a sub-range
void launch_aos_to_soa(const double* aos, size_t n, double* x, double* y, double* z, hipStream_t s);

// exclusive scan of u64 values; tmp must hold scan_tmp_elems(n) u64
size_t scan_tmp_elems(size_t n);
// gate != nullptr: a device flag; the launches do nothing where it reads 0
void launch_exclusive_scan_u64(const unsigned long long* in, unsigned long long* out, size_t n,
                               unsigned long long* tmp, hipStream_t s, const int* gate = nullptr);

// grid build for all four kinds at once (one launch per phase, blockIdx.y = kind): bbox (finished on the
// host), histogram over the CONCATENATED cell tables, one exclusive scan, finalize, scatter
struct GridSet {
  const double* tx[kKinds];
  const double* ty[kKinds];
  const double* tz[kKinds];
  int n[kKinds];
  int tgt_off[kKinds];          // offset of the kind's points in the concatenated point arrays
  long long cell_base[kKinds];  // offset of the kind's cells in the concatenated cell arrays
  long long ncell[kKinds];
  double org[kKinds][3];
  double inv_cell[kKinds];
  int dim[kKinds][3];
};
struct IngestArgs {   // k_ingest_targets: the four target clouds as uploaded (AoS) and where their SoA copies go
  const double* aos[kKinds];
  double *x[kKinds], *y[kKinds], *z[kKinds];
  int n[kKinds];
};
void launch_ingest_targets(const IngestArgs& A, double* bbox_rows /*[kind][64][6]*/, hipStream_t s);
struct FrameInit {
  const double* src_aos[kKinds];
  int direct;         // direct set: seg_n[k] = the kind's rows (= its source points) for the whole frame instead of 0
  int slot_off[kKinds + 1];
  double x[6];
  int no_eval_reuse;  // development knob, copied into the state
  unsigned long long* tile_cnt;  // query-tile histogram of the first builder pass, zeroed here (n_tile_cnt entries)
  int n_tile_cnt;
};
struct FrameInitBufs {
  double *sx, *sy, *sz, *w_src;
  unsigned long long* flags;
  GnState* st;
  int* seg_n;
};
// start of a scan_match folded into the first launch of the grid build (k_grid_count_all): `on` = 0 leaves the launch
// as it was; `consumed` tells the caller whether a launch carried it (no targets at all = no grid launch)
struct BuildParams {
  double radius[kKinds];
  int maxnum[kKinds];
  int active[kKinds];
  double edge_dir_thres;
};
// The first pass of the frame's query sort (k_query_bin: the bin of every source point under the predicted pose, one returning
// atomic each) needs the grids' DIMENSIONS, not the grids: on the 1 M-class frames it rides on the last launch of the grid build
// (the scatter of the targets into their cells -- writes against atomics) as one more row of blocks instead of following it.
struct QueryBinRide {
  SlotView sv;
  BuildParams bp;
  const GnState* st;
  int* tile_of_slot;
  int* rank_in_tile;
  unsigned long long* tile_cnt;
};
struct FrameInitHook {
  FrameInit fi;
  FrameInitBufs b;
  int n_slots;
  bool consumed;
  // in: non-null = the caller wants the query binning to ride (it has filled sv / bp / st); out: qbin_done
  QueryBinRide* qbin;
  bool qbin_done;
};
#if defined(__HIPCC__)
// start-of-Solve values of the minimiser (Ceres defaults: initial_trust_region_radius 1e4, min_mu 1e-8);
// the sweep point is the current pose
__device__ __forceinline__ void arm_solver(GnState& s) {
  s.radius = 1e4;
  s.mu = 1e-8;
  s.reuse = 0;
  s.cand_gn = 0;
  s.subspace_1d = 0;
  s.phase = PH_ITER0;
  s.iteration = 0;
  s.invalid = 0;
  s.step_successful = 1;
  s.done = 0;
  s.gmax = 1e300;
  s.T_eval = s.T_cur;
  s.Rt_eval = to_rt(s.T_cur);
}
// Start of a scan_match: de-interleave the four source clouds into the concatenated SoA slot arrays, set the GNC
// weights to 1 and the flag-scan terminator to 0 (registration.cpp:931-949); block 0 also zeroes the minimiser state,
// installs `parameters` = log(predict) (:881) and arms the first Solve.  Called by `nblocks` blocks of 256 threads.
__device__ __forceinline__ void frame_init_body(const FrameInit& fi, const FrameInitBufs& b, int block, int nblocks) {
  const int n = fi.slot_off[kKinds];
  for (int slot = block * 256 + (int)threadIdx.x; slot < n; slot += nblocks * 256) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < kKinds; ++q) k += (slot >= fi.slot_off[q]) ? 1 : 0;
    const double* a = fi.src_aos[k] + 3 * (size_t)(slot - fi.slot_off[k]);
    b.sx[slot] = a[0]; b.sy[slot] = a[1]; b.sz[slot] = a[2];
    b.w_src[slot] = 1.0;
  }
  for (int i = block * 256 + (int)threadIdx.x; i < fi.n_tile_cnt; i += nblocks * 256) fi.tile_cnt[i] = 0ull;
  if (block == 0) {
    constexpr int kWords = (int)(sizeof(GnState) / sizeof(double));
    for (int i = threadIdx.x; i < kWords; i += 256) reinterpret_cast<double*>(b.st)[i] = 0.0;
    if (threadIdx.x < 8)
      b.seg_n[threadIdx.x] = (fi.direct && threadIdx.x < kKinds) ? fi.slot_off[threadIdx.x + 1] - fi.slot_off[threadIdx.x] : 0;
    if (threadIdx.x == 0) b.flags[n] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {
      GnState* st = b.st;
#pragma unroll
      for (int i = 0; i < 6; ++i) st->x[i] = fi.x[i];
      st->T_cur = se3_exp(st->x);
      st->no_eval_reuse = fi.no_eval_reuse;
      st->prev_planar = __builtin_inf();   // registration.cpp:952-959
      st->run_build = 1;                   // the first outer iteration always builds
      arm_solver(*st);
    }
  }
}
#endif
void launch_frame_init(const FrameInit& fi, const FrameInitBufs& b, hipStream_t s);
void launch_bbox_all(const GridSet& gs, double* out /*[4][64][6], device or pinned host*/, hipStream_t s);
void launch_grid_count_all(const GridSet& gs, unsigned long long* cell_cnt, int* cell_of_pt, int* rank_of_pt,
                           hipStream_t s, const FrameInitHook* frame = nullptr);
void launch_grid_finalize_all(const GridSet& gs, const unsigned long long* cell_scan, int* cell_start,
                              unsigned long long* cell_cnt /* re-zeroed */, hipStream_t s);
// medium-size builds: tile-local scan only (returns the tile count, 0 = not applicable), then finalize + scatter in ONE
// launch that adds the tiles' offsets itself -- three launches per build instead of five
int scan_tiles_only(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* totals, hipStream_t s,
                    const int* gate = nullptr);
void launch_grid_finalize_scatter_all(const GridSet& gs, const unsigned long long* cell_scan, const unsigned long long* totals,
                                      int tiles, int* cell_start, unsigned long long* cell_cnt, const int* cell_of_pt,
                                      const int* rank_of_pt, double4* gp, hipStream_t s);
void launch_grid_scatter_all(const GridSet& gs, const int* cell_of_pt, const unsigned long long* cell_scan,
                             const int* rank_of_pt, double4* gp, hipStream_t s);
// large tables (more than 1024 scan tiles: the 1 M-class frames): ONE single-pass launch scans the histogram, writes cell_start
// and re-zeroes the histogram (no scan array), then the scatter -- count | scan + finalize | scatter.  ctl: scan_1p_ctl_elems(n)
// words, zero when allocated (k_scan_1p, tl_nn.hip).  Only where every block of the launch is resident at once on THIS device
// (scan_1p_applies: tiles <= device_cus); fault: pinned host word the kernel raises if its bounded look-back times out
size_t scan_1p_ctl_elems(size_t n);
bool scan_1p_applies(size_t n, int device_cus);
void launch_scan_counts_1p(const unsigned long long* in, unsigned long long* out, size_t n, unsigned long long* ctl, unsigned* fault,
                           hipStream_t s);
void launch_grid_scan_finalize_scatter_1p(const GridSet& gs, unsigned long long* cell_cnt, size_t ncells_plus_1, int* cell_start,
                                          unsigned long long* ctl, unsigned* fault, const int* cell_of_pt, const int* rank_of_pt, double4* gp,
                                          hipStream_t s, const QueryBinRide* qbin = nullptr, const GridView* views = nullptr);

// start-of-frame initialisation, one launch
// K1+K2: per source slot kNN + fit + gates -> raw records + flags
// scan1p_ctl: control words of the single-pass scan of the query-sort histogram (large frames; the caller has checked
// scan_1p_applies), or null: the multi-launch scan; scan1p_fault: see launch_scan_counts_1p
// direct_cv != null: the search writes the DIRECT set (DirectSet) -- st is then written (x_build) where ds.set_x_build says so
void launch_build(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                  int* tile_of_slot, unsigned long long* tile_cnt, unsigned long long* tile_scan, int* tile_fill,
                  double4* qrec, unsigned long long* scan_tmp, bool rebin, hipStream_t s, const int* gate = nullptr,
                  unsigned long long* scan1p_ctl = nullptr, unsigned* scan1p_fault = nullptr, const CorrView* direct_cv = nullptr,
                  const DirectSet* ds = nullptr, bool binned = false);   // binned: the sort's first pass rode on the grid build
bool direct_set_size(int n_slots);   // frames the thread-per-query search takes (the size class of the direct set)
int build_tile_count(const GridView grids[kKinds], const int slot_off[kKinds + 1]);  // bins of the query counting sort (per kind: its tiles, or the cells of its tiles)
// cap + compaction (after the flag scan)
// refresh_gate != null: the launch also stands for the refresh alternative (see CompactArgs); tiles > 0: `sv.scan` holds
// tile-local scans and `totals` the tiles' totals (scan_tiles_only)
void launch_compact(const SlotView& sv, const CorrView& cv, const BuildParams& bp, int* seg_n,
                    const double* rank_counts, int rank, int nranks, GnState* st, hipStream_t s, const int* gate = nullptr,
                    const int* refresh_gate = nullptr, const unsigned long long* totals = nullptr, int tiles = 0);
void launch_rank_counts(const SlotView& sv, double* rank_counts, int rank, int nranks, hipStream_t s);
// same correspondences, new outer iteration: re-capture the weights, zero the side-channel slots
void launch_refresh(const SlotView& sv, const CorrView& cv, hipStream_t s, const int* gate = nullptr);
// small single-rank frames: flag scan + caps + compaction (run_build, or always if null) or refresh (run_refresh) in ONE launch
bool prepare_small_fits(const SlotView& sv);
void launch_prepare_small(const SlotView& sv, const CorrView& cv, const BuildParams& bp, int* seg_n, GnState* st,
                          const int* run_build, const int* run_refresh, hipStream_t s);
// generic hybrid search (tloam_knn / fitness)
void launch_knn(const GridView& g, const double* qx, const double* qy, const double* qz, int nq,
                double radius, int k, int* out_idx, double* out_d2, int* out_cnt, hipStream_t s);
void launch_fitness(const GridView& g, const double* qx, const double* qy, const double* qz, int nq,
                    double radius, double* partial /*[blocks*2]*/, int blocks, hipStream_t s);

// ---- device-resident submap maintenance (tl_submap.hip; front_end.cpp:201-275) ----------------
// One Crop + VoxelDownSample launch sequence over up to TWO clouds stored back to back ("segments": the edge and
// the ground submap of an update -- same pipeline, different box and voxel size).  Points [0, n0) are segment 0,
// [n0, n) segment 1; the segment is part of the voxel key, so the segments never mix, and because the voxels are
// emitted in order of first occurrence segment 0's come out first.  One cloud: n0 == n.
struct VoxelJob {
  const double *x, *y, *z;   // input clouds, SoA, concatenated
  size_t n, n0;
  double lo[2][3], hi[2][3]; // crop boxes, inclusive (+-inf: no crop)
  double voxel[2];           // voxel sizes
  unsigned long long mask;   // hash-table capacity - 1 (power of two >= 2 n)
};
struct VoxelWork {           // scratch, sized by the caller (see voxel_table_size)
  double* min_partial;       // [256][6]
  double* vmin;              // [2][3] voxel_min_bound per segment
  unsigned long long *keys, *cnt, *off;   // [cap + 1]
  int *slot_of_pt, *urank, *members, *sorted;  // [n]
  unsigned long long *leader, *leader_scan;    // [n + 1]
  unsigned long long* scan_tmp;
  int* overflow;             // set when a voxel index leaves [0, 2^21)
  unsigned long long* n_out; // [2] receives the sizes of the down-sampled clouds
  // != nullptr: the sizes, the overflow flag and host_seq are also stored into one 64-byte segment of pinned host memory
  // ([0] n_out[0], [1] n_out[1], [2] overflow, [7] host_seq) by ONE instruction of the last kernel -- the host polls the
  // number instead of paying two device-to-host copies and a stream synchronisation
  unsigned long long* host_seg;
  unsigned long long host_seq;
  double* out[2][3];         // (x, y, z) of the down-sampled cloud per segment
  // k_vox_emit's look-back scan: blocks take their places from blockIdx while the whole grid is resident on this device
  // (vox_emit_resident_blocks), from a start ticket otherwise; fault: pinned host word raised if the bounded look-back times out
  int use_ticket;
  unsigned* fault;
};
size_t voxel_table_size(size_t n);
int vox_emit_resident_blocks(int device_cus);   // blocks of k_vox_emit the device holds at once, with a margin (tl_submap.hip)
void launch_transform_to_soa(const double* aos, size_t n, const double M[16], double* ox, double* oy, double* oz,
                             hipStream_t s);
void launch_transform_to_soa2(const double* aos, size_t n, const double M[16], double* ax, double* ay, double* az,
                              double* bx, double* by, double* bz, hipStream_t s);
struct AssembleArgs {          // per segment: old submap cloud (SoA) followed by the new scan cloud (AoS, to transform)
  const double *ox[2], *oy[2], *oz[2];
  const double* aos[2];
  size_t n_old[2], n_new[2], base[2];
  double M[16];                // column-major pose
};
void launch_assemble(const AssembleArgs& A, double* wx, double* wy, double* wz, hipStream_t s);
int transform_ring_max();  // frames one launch_transform_ring call takes
void launch_transform_ring(int count, const double* const aos[], const size_t n[], const double* const poses[],
                           double* ax, double* ay, double* az, double* bx, double* by, double* bz, hipStream_t s);
void launch_copy3(const double* ax, const double* ay, const double* az, size_t n, double* ox, double* oy, double* oz,
                  hipStream_t s);
void launch_soa_to_aos(const double* x, const double* y, const double* z, size_t n, double* aos, hipStream_t s);
void launch_blit_doubles(const double* src_host_view, double* dst, size_t n, hipStream_t s);   // pinned host -> device by a kernel
void launch_crop_voxel(const VoxelJob& J, const VoxelWork& W, hipStream_t s, bool front_done = false);
// the front of a submap update in one launch (k_submap_front): planar ring transform + assembly of the edge / ground clouds +
// the crop + voxel job's table clear and min bound; launch_crop_voxel(front_done = true) follows
size_t submap_front_rows(size_t n_ring_max, size_t n_seg_max);
void launch_submap_front(int count, const double* const aos[], const size_t n[], const double* const poses[], const AssembleArgs& A,
                         const VoxelJob& J, const VoxelWork& W, double* px, double* py, double* pz, double* qx, double* qy, double* qz,
                         double* wx, double* wy, double* wz, hipStream_t s, int copy_frame = -1, double* copy_dst = nullptr);
// (copy_dst: frame `copy_frame`'s cloud is read from pinned host memory and copied to this device buffer on the way; every
//  AoS cloud the launch reads must start on a 16-byte boundary and may be read up to one double past its end)

// ---- PCA feature extraction (tl_feature.hip; feature_extract.cpp:47-197) -----------------------
struct FeatArgs {
  GridView g;                 // grid over the cloud itself (cell >= radius)
  const double *x, *y, *z;    // the cloud, SoA, original order
  int n;
  double radius;
  int K, min_neigh;
  double *flatness, *cvr, *sphericity, *normal;  // [n], [n], [n], [3n] AoS
  int *num_sum, *neigh;                          // [n], [n * K] (-1 padded)
};
struct FeatSelect { double cvr_submap, planar_submap_thres, planar_vertic_thres; };
void launch_pca_info(const FeatArgs& A, hipStream_t s);
struct FeatRankCtl {   // control block of the ranking (tl_feature.hip): two lists, 0 planar, 1 sphere
  unsigned long long kmin[2], kmax[2];   // order-preserving keys of the smallest / largest flatness of a list
  int hist[2][4096 + 1];                 // bucket sizes, then bucket starts (+ the list's length)
};
void launch_feat_select(const FeatArgs& A, const FeatSelect& S, unsigned long long* flags, unsigned long long* scan,
                        unsigned long long* scan_tmp, double* f2, int* idx2, FeatRankCtl* ctl, int* bkt, int* pos, double* gf, int* gi,
                        double* out, hipStream_t s);

// K3 and the minimiser
int k3_grid_for(int total_cap, int device_cus);
// one wave per chunk (small sets) vs the streaming variant (wide: blocks of eight waves, see tl_gn.hip); *grid = blocks launched = rows
void k3_plan(const int cap[kKinds], int device_cus, int* grid, bool* single, bool* wide);
void launch_k3(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, bool force, hipStream_t s,
               hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// sharded contexts: the sweep whose LAST block (ticket counter) also folds the block rows into out48 and, with a
// mailbox, posts them to every rank -- the sweep of a sharded GN iteration is then 2 launches (+ the collective)
struct K3Fuse {
  int* ticket;       // zero between launches
  double* out48;     // this rank's totals (all-reduced by RCCL / the callback when there is no mailbox)
  MboxView mb;       // mb.nranks == 0: no mailbox
};
void launch_k3_fused(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, bool force,
                     const K3Fuse& fuse, hipStream_t s);
// one GN iteration in ONE launch, whatever the size of the set: the sweep whose last block folds the rows and advances the
// minimiser (mailbox contexts: posts, gathers and advances) -- k3_sweep_step, tl_gn.hip.  span: see K3Step (may be null)
// iter_span (every launcher of a kernel that ends a GN iteration): the device-side period counter of tloam_gn_iter_timer, or null
void launch_k3_step(const CorrView& cv, GnState* st, double* partials, int grid, bool single, bool wide, int* ticket, unsigned long long* span,
                    const MboxView* mb_or_null, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                    unsigned long long* iter_span = nullptr);
void launch_gn_step_mbox(GnState* st, const MboxView& mb, hipStream_t s, unsigned long long* iter_span = nullptr);   // gather (rank order) + consume
void launch_mbox_allreduce(double* buf, int count, const MboxView& mb, hipStream_t s);  // buf <- sum over ranks
void launch_mbox_gather_only(double* out48, const MboxView& mb, hipStream_t s);
void launch_reduce(const double* partials, int grid, GnState* st, double* out48, hipStream_t s);
void launch_solve_init(GnState* st, hipStream_t s);                   // begin one ceres::Solve at st->x
void launch_set_eval(GnState* st, const double* se3_dev, hipStream_t s);
void launch_pose_from_x_build(GnState* st, hipStream_t s);   // st->T_cur = exp(st->x_build)
void launch_gn_step(GnState* st, const double* in48, hipStream_t s, unsigned long long* iter_span = nullptr);  // consume a reduced sweep
void launch_reduce_and_step(const double* partials, int grid, GnState* st, hipStream_t s, unsigned long long* iter_span = nullptr);
// small sets on one rank: sweep + step in ONE launch (ticket: zero between launches)
void launch_sweep_step_small(const CorrView& cv, GnState* st, double* partials, int* ticket, int grid, hipStream_t s,
                             unsigned long long* iter_span = nullptr);
// sets whose sweep is a grid of a dozen blocks: a whole ceres::Solve (up to max_sweeps evaluations) in ONE launch
bool solve_small_fits(int grid, int device_cus);
// The factor set of an outer iteration prepared by the Solve launch itself (no k_prepare_small launch in front of it):
// flagb == null means "already prepared" (stepwise API, topped-up Solves).
struct SolvePrep {
  SlotView sv;
  int maxnum[kKinds];
  const int* run_build;     // device gates of the outer loop; null: always build (the frame's first iteration)
  const int* run_refresh;
};
// K4
struct WeightParams {
  double th1, th2, mu, noise_bound_sq;
  int active[kKinds];
};
// The finish of the outer iterations a one-launch Solve may run itself (device-driven loop, KITTI-size sets): every wave
// updates the weights of the factors it holds and adds their side-channel costs up, the consumer wave makes the loop
// decisions and writes the result slot; if the loop goes on with an unchanged pose the launch refreshes its correspondences
// in registers and runs the next Solve as well.  enabled == 0: the launch ends with the Solve and leaves the sums in the
// state for the finish kernel (GnState::fin_valid).
constexpr int kMaxOuterInLaunch = 8;
struct SolveFinish {
  int enabled;
  int have_wp;               // wp[first_iter ..] are set: the launch adds up the finish sums of its Solve(s)
  int first_iter, n_iter;    // outer iteration this launch was enqueued for; max_iterations
  double cost_threshold;     // registration.cpp:1108
  double* sums16;
  unsigned long long* iter_span;   // the lead's stepper notes the period of its GN iterations here (tloam_gn_iter_timer), or null
  WeightParams wp[kMaxOuterInLaunch];   // per outer iteration
  HostMirror hm[kMaxOuterInLaunch];
};
void launch_solve_small(const CorrView& cv, GnState* st, double* partials, int* ticket, int grid, int max_sweeps,
                        const SolvePrep* prep_or_null, int* seg_n, const SolveFinish* finish_or_null, hipStream_t s);
// (gated on st->done: see k_weights)
void launch_weights(const CorrView& cv, const SlotView& sv, const WeightParams& wp, double* partial /*[blocks*8]*/,
                    int blocks, const GnState* st, hipStream_t s);
void launch_outer_finish(const double* partial, int blocks, const int* seg_n, GnState* st_or_null, GnState* gate,
                         double* sums8, HostMirror hm, OuterCtl ctl, hipStream_t s);
void launch_outer_publish(const double* sums8, GnState* st, HostMirror hm, const unsigned long long* comm_err_or_null,
                          hipStream_t s);
// weights + finish in one launch for small single-rank sets
// KITTI-size frames, device-driven loop: the finish of outer iteration k-1 (the first sixteen blocks) and the
// correspondence search of iteration k (the other blocks, gated on GnState::spec_build) in ONE launch -- they are
// independent of each other
bool build_finish_small_fits(const SlotView& sv);
struct FinishSmallArgs {
  const CorrView* cv;
  const WeightParams* wp;
  const int* seg_n;
  double* sums16;
  HostMirror hm;
  OuterCtl ctl;
  double* rows;   // [16][8] hand-over rows of the sixteen finish blocks
  int* ticket;    // zero between launches
};
void launch_build_finish_small(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                               const FinishSmallArgs& fin, hipStream_t s);
// large single-rank sets (thread-per-query search), device-driven loop: the same for k_weights + k_outer_finish -- the
// first 4 * wblocks one-wave blocks of the search launch are the finish (rows: [4 * wblocks][8] doubles)
struct FinishLargeArgs {
  const CorrView* cv;
  const WeightParams* wp;
  const int* seg_n;
  double* sums16;
  HostMirror hm;
  OuterCtl ctl;
  double* rows;
  int* ticket;
  int wblocks;
  // direct set: the weight stream the finish WRITES (every row), per kind; cv->k[].w is the one it reads.  null: compact set
  double* w_next[kKinds];
  // ... and where n_corr comes from: the per-block counts of the search that built the set solved in this iteration (blk_cnt,
  // nblk blocks) if built != 0 -- built < 0: GnState::run_build says whether it was (device-driven loop) --, else unchanged
  const int* blk_cnt;
  int nblk, built;
};
// the finish of a direct set as a launch of its own (stepwise API, the frame's last outer iteration, a topped-up Solve): the same
// one-wave blocks as the riding form, bit for bit
void launch_finish_direct(const FinishLargeArgs& fin, GnState* st, hipStream_t s);
bool build_finish_large_fits(const SlotView& sv);
void launch_build_finish_large(const SlotView& sv, const GridView grids[kKinds], const BuildParams& bp, GnState* st,
                               const unsigned long long* n_sorted, const double4* qrec, const FinishLargeArgs& fin, hipStream_t s,
                               const DirectSet* ds = nullptr);
void launch_weights_finish_small(const CorrView& cv, const SlotView& sv, const WeightParams& wp, const int* seg_n,
                                 double* sums16, GnState* st, HostMirror hm, OuterCtl ctl, hipStream_t s);
void launch_transform_cloud(double* aos, size_t n, const double M[16], hipStream_t s);
void launch_debug_se3(const double* x, const double* delta, int n, double* out /* 26 per item */, hipStream_t s);

}  // namespace tl
