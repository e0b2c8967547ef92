// tl_api_frames.hip -- HBM residency of the C ABI (include/tloam_hip.h): setInputSource / setInputTarget
// (registration.cpp:232-248) through pinned staging, the search grids over the registered targets (the role of
// KDTreeFlann::SetGeometry, :889-915), frames staged ahead of their solve, and the host's waits on pinned result words.
#include <chrono>
#include "tl_ctx.hpp"

using namespace tl;

namespace tlh {
// Poll a word in pinned host memory that a kernel stores last (HostMirror).  The stream is only queried
// now and then, to notice a failed launch instead of spinning forever.  TLOAM_OK: the word arrived; 1: the stream
// drained without it (the caller reads the result the slow way).
int wait_word(tloam_ctx* c, const unsigned long long* p, unsigned long long seq) {
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) return TLOAM_OK;
    if ((spins & 0x7ffu) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) return __atomic_load_n(p, __ATOMIC_ACQUIRE) == seq ? TLOAM_OK : 1;
      if (e != hipErrorNotReady) HIPC(c, e);
    }
    __builtin_ia32_pause();
  }
}
// One 64-byte segment of a result slot (MirrorSlot: seven payload words, then check_mix(sequence number) XOR seg_word of every
// payload word): wait until the check word agrees with the payload read -- a segment that has only partly arrived does not
// check, whichever part it is -- and copy the payload out.  Same return convention as wait_word.
int wait_segment(tloam_ctx* c, const unsigned long long* seg, unsigned long long number, unsigned long long payload[7]) {
  const unsigned long long seq = check_mix(number);   // (what the check word carries, tl_common.hpp)
  for (unsigned spins = 1;; ++spins) {
    unsigned long long w[8], x = 0ull;
    for (int i = 0; i < 8; ++i) { w[i] = __atomic_load_n(seg + i, __ATOMIC_ACQUIRE); x ^= i < 7 ? tl::seg_word(w[i], i) : w[i]; }
    if (x == seq) {
      for (int i = 0; i < 7; ++i) payload[i] = w[i];
      return TLOAM_OK;
    }
    if ((spins & 0x7ffu) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {
        x = 0ull;
        for (int i = 0; i < 8; ++i) { w[i] = __atomic_load_n(seg + i, __ATOMIC_ACQUIRE); x ^= i < 7 ? tl::seg_word(w[i], i) : w[i]; }
        if (x != seq) return 1;
        for (int i = 0; i < 7; ++i) payload[i] = w[i];
        return TLOAM_OK;
      }
      if (e != hipErrorNotReady) HIPC(c, e);
    }
    __builtin_ia32_pause();
  }
}
// After a host synchronisation: did a kernel of this context give up one of its bounded in-launch waits?  The single-pass scans
// (tl_nn.hip scan1p_tile) and k_vox_emit's look-back (tl_submap.hip) spin on blocks of their own launch, which is only safe
// while all of them are resident at once; the host only picks those forms where the device's CU count says they are, and
// should that ever be wrong (a device shared with long-running kernels) the wait runs out after ~1 s, the kernel raises a word
// in pinned memory and finishes with garbage.  Here the context is switched to the forms that wait for nothing (multi-launch
// scans, start tickets) for good, everything derived from the garbage is invalidated, and the caller gets TLOAM_E_HIP (or, in
// tloam_scan_match, runs the frame again).
int check_device_faults(tloam_ctx* c) {
  if (!c->h_fault) return TLOAM_OK;
  int rc = TLOAM_OK;
  if (__atomic_load_n(&c->h_fault[kFaultScan1p], __ATOMIC_ACQUIRE) != 0u) {
    __atomic_store_n(&c->h_fault[kFaultScan1p], 0u, __ATOMIC_RELEASE);
    c->no_scan_1p = true;
    c->fallback_events++;
    c->grids_ahead = false; c->tgt_gen++;
    for (int k = 0; k < kKinds; ++k) c->kd[k].grid_valid = false;
    c->have_build = false;
    c->last_error = "a single-pass scan timed out in its look-back (its blocks were not resident together): the context now uses the multi-launch scans";
    rc = TLOAM_E_HIP;
  }
  if (__atomic_load_n(&c->h_fault[kFaultVoxEmit], __ATOMIC_ACQUIRE) != 0u) {
    __atomic_store_n(&c->h_fault[kFaultVoxEmit], 0u, __ATOMIC_RELEASE);
    c->vox_ticket = true;
    c->fallback_events++;
    c->grids_ahead = false; c->tgt_gen++;
    c->last_error = "the voxel down-sampling timed out in its look-back (its blocks were not resident together): the context now uses start tickets; "
                    "the submap of this update is undefined -- initialise it again";
    rc = TLOAM_E_HIP;
  }
  return rc;
}
// Borrowed host arrays -> device, without waiting for the device: the pieces are copied into a pinned staging half (two halves
// used alternately; an event per half says when the device has read it -- long ago in the reference's call pattern, waited for
// otherwise), every piece on a 16-byte boundary (offs[i], in doubles; `total` out), and either
//   stage_and_upload: go to `dev_dst` with ONE asynchronous copy on the context's stream (same layout there), or
//   stage_in_place:   stay where they are for the caller's kernels to read across PCIe (*dev_view = the half as the device sees
//                     it); the caller reports the end of that use with stage_release.
// counts in doubles.
static int stage_fill(tloam_ctx* c, const double* const parts[], const size_t counts[], int nparts, size_t offs[], size_t* total_out,
                      int* half_out) {
  size_t total = 0;
  for (int i = 0; i < nparts; ++i) {
    offs[i] = total;
    total += counts[i] + (counts[i] & 1u);   // the next piece starts on an even double
  }
  *total_out = total;
  *half_out = -1;
  if (total == 0) return TLOAM_OK;
  const int h = c->stage_next;
  c->stage_next ^= 1;
  if (c->stage_busy[h]) {
    HIPC(c, hipEventSynchronize(c->stage_ev[h]));
    c->stage_busy[h] = false;
  }
  if (total + 2 > c->h_stage_cap[h]) {   // (+ 2: a kernel reading a piece in 16-byte steps may touch one double past its end)
    if (c->h_stage[h]) (void)hipHostFree(c->h_stage[h]);
    c->h_stage[h] = nullptr;
    c->h_stage_dev[h] = nullptr;
    c->h_stage_cap[h] = 0;
    const size_t want = total + total / 2 + 2;
    HIPC(c, hipHostMalloc((void**)&c->h_stage[h], want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    c->h_stage_cap[h] = want;
    c->h_stage[h][want - 1] = c->h_stage[h][want - 2] = 0.0;
    if (hipHostGetDevicePointer((void**)&c->h_stage_dev[h], c->h_stage[h], 0) != hipSuccess) c->h_stage_dev[h] = nullptr;
  }
  if (!c->stage_ev[h]) HIPC(c, hipEventCreateWithFlags(&c->stage_ev[h], hipEventDisableTiming));
  for (int i = 0; i < nparts; ++i) {
    if (counts[i] > 0) memcpy(c->h_stage[h] + offs[i], parts[i], sizeof(double) * counts[i]);
    if (counts[i] & 1u) c->h_stage[h][offs[i] + counts[i]] = 0.0;   // the padding double is defined
  }
  *half_out = h;
  return TLOAM_OK;
}
int stage_and_upload(tloam_ctx* c, const double* const parts[], const size_t counts[], int nparts, double* dev_dst, size_t offs[]) {
  size_t total = 0;
  int h = -1;
  const int rc = stage_fill(c, parts, counts, nparts, offs, &total, &h);
  if (rc != TLOAM_OK || h < 0) return rc;
  // up to a few MB a kernel that reads the pinned block in place does the copy (for 226 KB the copy command costs the calling
  // thread and the copy engine more than a launch: 0.197 / 0.201 against 0.206 / 0.206 ms set_source + scan_match, round 4)
  if (c->h_stage_dev[h] && total <= (size_t)1 << 19)
    launch_blit_doubles(c->h_stage_dev[h], dev_dst, total, c->stream);
  else
    HIPC(c, hipMemcpyAsync(dev_dst, c->h_stage[h], sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipEventRecord(c->stage_ev[h], c->stream));
  c->stage_busy[h] = true;
  return TLOAM_OK;
}
int stage_in_place(tloam_ctx* c, const double* const parts[], const size_t counts[], int nparts, size_t offs[], const double** dev_view,
                   int* half) {
  size_t total = 0;
  *dev_view = nullptr;
  const int rc = stage_fill(c, parts, counts, nparts, offs, &total, half);
  if (rc != TLOAM_OK || *half < 0) return rc;
  if (!c->h_stage_dev[*half]) return TLOAM_E_NOT_READY;   // (the caller looked at stage_in_place_ok first)
  *dev_view = c->h_stage_dev[*half];
  return TLOAM_OK;
}
// completed: the caller has waited for the kernels that read the half; otherwise an event behind them is recorded
int stage_release(tloam_ctx* c, int half, bool completed) {
  if (half < 0) return TLOAM_OK;
  if (completed) { c->stage_busy[half] = false; return TLOAM_OK; }
  HIPC(c, hipEventRecord(c->stage_ev[half], c->stream));
  c->stage_busy[half] = true;
  return TLOAM_OK;
}
size_t staged_size(const size_t counts[], int nparts) {   // doubles the pieces take up, padding included
  size_t total = 0;
  for (int i = 0; i < nparts; ++i) total += counts[i] + (counts[i] & 1u);
  return total;
}
// The four search grids share one set of buffers (points and cell tables concatenated), so that every
// phase of the build is ONE launch for all kinds: bbox -> (host: dims) -> histogram -> scan -> finalize ->
// scatter.  `GridBuffers` owns the storage; ctx->grids is the set built by scanMatching, tloam_knn uses a
// temporary one.
// radius[k] <= 0: kind not rebuilt (its view is left empty).  One host synchronisation (bounding boxes).
// rows of launch_bbox_all ([kind][64][6]) -> (lo[3], hi[3]) per kind
void reduce_box_rows(const double* box_rows, double boxes[kKinds][6]) {
  for (int k = 0; k < kKinds; ++k) {
    double* b = boxes[k];
    b[0] = b[1] = b[2] = 1e300;
    b[3] = b[4] = b[5] = -1e300;
    for (int r = 0; r < 64; ++r) {
      const double* row = box_rows + ((size_t)k * 64 + r) * 6;
      for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], row[a]); b[3 + a] = std::max(b[3 + a], row[3 + a]); }
    }
  }
}
// known_boxes: the clouds' bounds are already on the host (targets: taken at set_target) -- no launch, no wait
int build_grids_over(tloam_ctx* c, GridBuffers& G, const double radius[kKinds], const CloudRef clouds[kKinds],
                     GridView out[kKinds], const double (*known_boxes)[6], FrameInitHook* frame) {
  GridSet gs;
  memset(&gs, 0, sizeof(gs));
  size_t tgt_total = 0;
  for (int k = 0; k < kKinds; ++k) {
    const bool use = radius[k] > 0.0 && clouds[k].n > 0;
    gs.tx[k] = clouds[k].x; gs.ty[k] = clouds[k].y; gs.tz[k] = clouds[k].z;
    gs.n[k] = use ? (int)clouds[k].n : 0;
    gs.tgt_off[k] = (int)tgt_total;
    tgt_total += (size_t)gs.n[k];
  }
  double boxes[kKinds][6];
  if (known_boxes) {
    memcpy(boxes, known_boxes, sizeof(boxes));
  } else {
    // rows straight into pinned host memory: no copy kernel.  (Publishing a completion word from the last of the
    // 256 blocks -- system-scope fence per block -- was measured: it costs more than this synchronisation.)
    launch_bbox_all(gs, c->h_bbox_dev, c->stream);
    HIPC(c, hipStreamSynchronize(c->stream));
    reduce_box_rows(c->h_bbox, boxes);
  }
  long long cell_total = 0;
  for (int k = 0; k < kKinds; ++k) {
    const double* lo = boxes[k];
    const double* hi = boxes[k] + 3;
    GridView& g = out[k];
    memset(&g, 0, sizeof(g));
    gs.cell_base[k] = cell_total;
    // (a cloud without a single finite point has no bounds -- the rows come back as they were initialised, lo > hi -- and
    //  nothing to search: no grid, like an empty cloud.  Sizing a cell table from such a box was a GPU memory fault until
    //  round 5: tests/tools/fuzz_call_order.py, a one-point cloud whose point is NaN)
    if (gs.n[k] > 0 && !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) gs.n[k] = 0;
    if (gs.n[k] == 0) { gs.ncell[k] = 0; gs.dim[k][0] = gs.dim[k][1] = gs.dim[k][2] = 1; gs.inv_cell[k] = 1.0; continue; }
    double cell = radius[k] * (1.0 + 1e-6);  // every target within `radius` of a query lies in its 27 cells
    double dims[3];
    for (;;) {
      double cells = 1.0;
      for (int a = 0; a < 3; ++a) {
        dims[a] = floor((hi[a] - lo[a]) / cell) + 1.0;
        cells *= dims[a];
      }
      if (cells <= 4.0e6) break;  // dense cell table bound (u64 histogram + scan per frame)
      cell *= 1.25;
    }
    g.cell = cell;
    g.inv_cell = 1.0 / cell;
    long long ncell = 1;
    for (int a = 0; a < 3; ++a) {
      g.org[a] = lo[a];
      g.dim[a] = (int)dims[a];
      ncell *= g.dim[a];
      gs.org[k][a] = lo[a];
      gs.dim[k][a] = g.dim[a];
    }
    g.n = gs.n[k];
    gs.inv_cell[k] = g.inv_cell;
    gs.ncell[k] = ncell;
    cell_total += ncell;
  }
  const size_t nc = (size_t)std::max<long long>(cell_total, 1);
  HIPC(c, G.gp.reserve(std::max<size_t>(tgt_total, 1))); HIPC(c, G.cell_of_pt.reserve(std::max<size_t>(tgt_total, 1)));
  HIPC(c, G.rank_of_pt.reserve(tgt_total + 1));
  // The cell count follows the bounding boxes, which change from frame to frame: a table that has to grow does so with
  // room to spare (a re-allocation inside scanMatching costs ~0.7 ms -- three times the frame)
  const size_t nc_res = (nc + 1 > G.cell_cnt.cap || nc + kKinds + 1 > G.cell_start.cap) ? 2 * nc + 64 : nc;
  HIPC(c, G.cell_start.reserve(nc_res + kKinds + 1));
  {
    // the cell histogram is all-zero between builds (k_grid_finalize_all re-zeroes what a build used): only a
    // (re)allocation has to be cleared
    const size_t before = G.cell_cnt.cap;
    HIPC(c, G.cell_cnt.reserve(nc_res + 1));
    if (G.cell_cnt.cap != before)
      HIPC(c, hipMemsetAsync(G.cell_cnt.p, 0, G.cell_cnt.cap * sizeof(unsigned long long), c->stream));
  }
  HIPC(c, G.cell_scan.reserve(nc_res + 1));
  HIPC(c, G.scan_tmp.reserve(scan_tmp_elems(nc_res + 1)));
  for (int k = 0; k < kKinds; ++k) {
    out[k].gp = G.gp.p + gs.tgt_off[k];
    out[k].cell_start = G.cell_start.p + gs.cell_base[k] + k;
  }
  if (cell_total == 0) return TLOAM_OK;
  if (frame) {  // the start of the scan_match rides on the first launch (the query-tile histogram is sized by the grids)
    const size_t ntiles = (size_t)build_tile_count(out, frame->fi.slot_off);
    HIPC(c, c->tile_cnt.reserve(ntiles + 1 > c->tile_cnt.cap ? 2 * ntiles + 64 : ntiles + 1));
    frame->fi.tile_cnt = c->tile_cnt.p;
    frame->fi.n_tile_cnt = (int)ntiles + 1;
    frame->consumed = true;
  }
  const bool stamp = frame && c->hs_entry > 0.0;   // (TLOAM_HOST_STAMPS: when the frame's FIRST launch went out / had been issued)
  if (stamp) c->hs[5] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->hs_entry;
  launch_grid_count_all(gs, G.cell_cnt.p, G.cell_of_pt.p, G.rank_of_pt.p, c->stream, frame);
  if (stamp) c->hs[6] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->hs_entry;
  if (!c->no_scan_1p && scan_1p_applies(nc + 1, c->device_cus)) {
    // 1 M-class tables: count | scan + finalize in ONE single-pass launch | scatter (three launches and one pass over the table
    // less than tile scan + scan of the totals + add + finalize)
    const size_t before = G.scan1p.cap;
    HIPC(c, G.scan1p.reserve(scan_1p_ctl_elems(nc_res + 1)));
    if (G.scan1p.cap != before) HIPC(c, hipMemsetAsync(G.scan1p.p, 0, G.scan1p.cap * sizeof(unsigned long long), c->stream));
    const QueryBinRide* qbin = nullptr;
    if (frame && frame->qbin) {
      // the query sort's buffers, sized exactly as the sort itself will ask for them (nothing may be re-allocated in between)
      const int rc = reserve_query_sort(c, out);
      if (rc != TLOAM_OK) return rc;
      frame->qbin->tile_of_slot = c->tile_of_slot.p;
      frame->qbin->rank_in_tile = c->tile_fill.p;
      frame->qbin->tile_cnt = c->tile_cnt.p;
      // (the frame's start, riding on the first launch above, zeroes the histogram it was handed: the same array only if the
      //  reservation just made did not move it)
      if (frame->fi.tile_cnt == c->tile_cnt.p) { qbin = frame->qbin; frame->qbin_done = true; }
    }
    launch_grid_scan_finalize_scatter_1p(gs, G.cell_cnt.p, nc + 1, G.cell_start.p, G.scan1p.p, c->h_fault_dev + kFaultScan1p, G.cell_of_pt.p,
                                         G.rank_of_pt.p, G.gp.p, c->stream, qbin, out);
    return TLOAM_OK;
  }
  const int tiles = scan_tiles_only(G.cell_cnt.p, G.cell_scan.p, nc + 1, G.scan_tmp.p, c->stream);
  if (tiles > 0) {
    launch_grid_finalize_scatter_all(gs, G.cell_scan.p, G.scan_tmp.p, tiles, G.cell_start.p, G.cell_cnt.p, G.cell_of_pt.p,
                                     G.rank_of_pt.p, G.gp.p, c->stream);
  } else {
    launch_exclusive_scan_u64(G.cell_cnt.p, G.cell_scan.p, nc + 1, G.scan_tmp.p, c->stream);
    launch_grid_finalize_all(gs, G.cell_scan.p, G.cell_start.p, G.cell_cnt.p, c->stream);
    launch_grid_scatter_all(gs, G.cell_of_pt.p, G.cell_scan.p, G.rank_of_pt.p, G.gp.p, c->stream);
  }
  return TLOAM_OK;
}
int build_grids(tloam_ctx* c, GridBuffers& G, const double radius[kKinds], GridView out[kKinds], FrameInitHook* frame) {
  CloudRef clouds[kKinds];
  bool known = true;
  for (int k = 0; k < kKinds; ++k) {
    const KindData& K = c->kd[k];
    clouds[k] = CloudRef{K.tx.p, K.ty.p, K.tz.p, K.tgt_set ? K.n_tgt : 0};
    if (radius[k] > 0.0 && clouds[k].n > 0 && !c->tgt_box_valid[k]) known = false;
  }
  return build_grids_over(c, G, radius, clouds, out, known ? c->tgt_box : nullptr, frame);
}
// bounds of the target clouds registered so far, taken while the hand-over call is synchronising anyway
int enqueue_target_bounds(tloam_ctx* c) {
  GridSet gs;
  memset(&gs, 0, sizeof(gs));
  for (int k = 0; k < kKinds; ++k) {
    const KindData& K = c->kd[k];
    gs.tx[k] = K.tx.p; gs.ty[k] = K.ty.p; gs.tz[k] = K.tz.p;
    gs.n[k] = K.tgt_set ? (int)K.n_tgt : 0;
  }
  launch_bbox_all(gs, c->h_bbox_dev, c->stream);
  return TLOAM_OK;
}
void finish_target_bounds(tloam_ctx* c) {  // after the stream has been synchronised
  reduce_box_rows(c->h_bbox, c->tgt_box);
  for (int k = 0; k < kKinds; ++k) c->tgt_box_valid[k] = c->kd[k].tgt_set && c->kd[k].n_tgt > 0;
}
// exchange the registered clouds of the context with a FrameClouds (pointers and counts only)
void exchange_clouds(tloam_ctx* c, FrameClouds& F) {
  for (int k = 0; k < kKinds; ++k) {
    KindData& K = c->kd[k];
    std::swap(K.n_src_full, F.n_src_full[k]); std::swap(K.src_lo, F.src_lo[k]); std::swap(K.n_src, F.n_src[k]);
    std::swap(K.n_tgt, F.n_tgt[k]);
    std::swap(K.src_aos, F.src_aos[k]); std::swap(K.tgt_aos, F.tgt_aos[k]);
    std::swap(K.src_ptr, F.src_ptr[k]);
    std::swap(K.tx, F.tx[k]); std::swap(K.ty, F.ty[k]); std::swap(K.tz, F.tz[k]);
    std::swap(K.src_set, F.src_set[k]); std::swap(K.tgt_set, F.tgt_set[k]);
    for (int a = 0; a < 6; ++a) std::swap(c->tgt_box[k][a], F.tgt_box[k][a]);
    std::swap(c->tgt_box_valid[k], F.tgt_box_valid[k]);
    K.grid_valid = false;   // the search grids belong to the frame they were built over
  }
  c->grids_ahead = false; c->tgt_gen++;
  std::swap(c->src_pack, F.src_pack);
  c->have_build = false;
}
}  // namespace tlh

extern "C" {

// ---- setInputSource / setInputTarget (registration.cpp:232-248) --------------------------------
namespace {
int set_source_async(tloam_ctx* c, int kind, const double* xyz, size_t n) {
  if (kind < 0 || kind >= kKinds || (n > 0 && !xyz) || n > kMaxPoints) return TLOAM_E_INVALID;
  KindData& K = c->kd[kind];
  size_t lo = 0, hi = n;
  tloam_shard_range(n, c->rank, c->nranks, &lo, &hi);
  K.n_src_full = n;
  K.src_lo = lo;
  K.n_src = hi - lo;
  HIPC(c, K.src_aos.reserve(3 * std::max<size_t>(K.n_src, 1)));
  if (K.n_src > 0)
    HIPC(c, hipMemcpyAsync(K.src_aos.p, xyz + 3 * lo, sizeof(double) * 3 * K.n_src, hipMemcpyHostToDevice, c->stream));
  K.src_ptr = K.src_aos.p;
  K.src_set = true;
  return TLOAM_OK;
}
// setInputSource(const Frame&): the four clouds through pinned staging and ONE asynchronous copy, no host synchronisation
// (front_end.cpp:314 is followed at once by scanMatching, :321: the wait moves to that call's first wait for the device)
int set_source_frame_packed(tloam_ctx* c, const double* const xyz[4], const size_t n[4]) {
  size_t off[kKinds] = {0, 0, 0, 0};
  size_t lo4[kKinds], hi4[kKinds], cnt4[kKinds];
  for (int k = 0; k < kKinds; ++k)   // (refused as a whole, before anything of the registered frame has been touched)
    if ((n[k] > 0 && !xyz[k]) || n[k] > kMaxPoints) return TLOAM_E_INVALID;
  tloam_shard_ranges_frame(n, c->rank, c->nranks, lo4, hi4);
  for (int k = 0; k < kKinds; ++k) {
    KindData& K = c->kd[k];
    K.n_src_full = n[k];
    K.src_lo = lo4[k];
    K.n_src = hi4[k] - lo4[k];
    cnt4[k] = 3 * K.n_src;
  }
  const size_t total = std::max<size_t>(tlh::staged_size(cnt4, kKinds), 3);
  if (total > c->src_pack.cap) {
    // (a kernel of an earlier frame may still read the old block: nothing of this context is in flight in the reference's
    //  call pattern, but a growing buffer is rare enough to afford the certainty)
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, c->src_pack.reserve(total));
  }
  const double* parts[kKinds];
  for (int k = 0; k < kKinds; ++k) parts[k] = c->kd[k].n_src > 0 ? xyz[k] + 3 * c->kd[k].src_lo : nullptr;
  const int rc = tlh::stage_and_upload(c, parts, cnt4, kKinds, c->src_pack.p, off);
  for (int k = 0; k < kKinds; ++k) {
    // (a failed staging / upload leaves the block undefined: the sources are NOT registered, the next solve says so)
    c->kd[k].src_ptr = rc == TLOAM_OK ? c->src_pack.p + off[k] : nullptr;
    c->kd[k].src_set = rc == TLOAM_OK;
  }
  return rc;
}
int set_target_async(tloam_ctx* c, int kind, const double* xyz, size_t n, bool convert = true) {
  if (kind < 0 || kind >= kKinds || (n > 0 && !xyz) || n > kMaxPoints) return TLOAM_E_INVALID;
  KindData& K = c->kd[kind];
  K.n_tgt = n;
  c->tgt_box_valid[kind] = false;
  c->grids_ahead = false; c->tgt_gen++;
  const size_t m = std::max<size_t>(n, 1);
  HIPC(c, K.tgt_aos.reserve(3 * m));
  HIPC(c, K.tx.reserve(m)); HIPC(c, K.ty.reserve(m)); HIPC(c, K.tz.reserve(m));
  if (n > 0) {
    HIPC(c, hipMemcpyAsync(K.tgt_aos.p, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    if (convert) launch_aos_to_soa(K.tgt_aos.p, n, K.tx.p, K.ty.p, K.tz.p, c->stream);  // AoS -> SoA on the device
  }
  K.tgt_set = true;
  return TLOAM_OK;
}
}  // namespace

int tloam_set_source(tloam_ctx* c, int kind, const double* xyz, size_t n) {
  if (!c) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  const int rc = set_source_async(c, kind, xyz, n);
  if (rc != TLOAM_OK) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));  // the host buffer is only borrowed for the call
  return TLOAM_OK;
}

int tloam_set_target(tloam_ctx* c, int kind, const double* xyz, size_t n) {
  if (!c) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  int rc = set_target_async(c, kind, xyz, n);
  if (rc == TLOAM_OK) rc = enqueue_target_bounds(c);
  if (rc != TLOAM_OK) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));
  finish_target_bounds(c);
  return TLOAM_OK;
}

int tloam_set_source_frame(tloam_ctx* c, const double* const xyz[4], const size_t n[4]) {
  if (!c || !xyz || !n) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  return set_source_frame_packed(c, xyz, n);   // (the host buffers have been copied out when this returns)
}

int tloam_set_target_frame(tloam_ctx* c, const double* const xyz[4], const size_t n[4]) {
  if (!c || !xyz || !n) return TLOAM_E_INVALID;
  for (int k = 0; k < kKinds; ++k)   // (refused as a whole, before any of the registered targets has been replaced)
    if ((n[k] > 0 && !xyz[k]) || n[k] > kMaxPoints) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  int rc = TLOAM_OK;
  // four copies, then ONE launch that converts all four clouds and takes their bounds (rows into pinned memory)
  for (int k = 0; k < kKinds && rc == TLOAM_OK; ++k) rc = set_target_async(c, k, xyz[k], n[k], /*convert=*/false);
  if (rc == TLOAM_OK) {
    IngestArgs A;
    for (int k = 0; k < kKinds; ++k) {
      KindData& K = c->kd[k];
      A.aos[k] = K.tgt_aos.p; A.x[k] = K.tx.p; A.y[k] = K.ty.p; A.z[k] = K.tz.p;
      A.n[k] = (int)K.n_tgt;
    }
    launch_ingest_targets(A, c->h_bbox_dev, c->stream);
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  if (rc == TLOAM_OK) finish_target_bounds(c);
  // The four search grids (registration.cpp:889-915 builds its kd-trees at the top of scanMatching) are enqueued HERE, behind the
  // hand-over's own synchronisation and not waited for: the targets are final once setInputTarget returns, the next scan is a
  // sensor period away, and the ~24 us of launches leave the bracket around scanMatching (front_end.cpp:320-322).  The grids
  // stay valid until a target changes; a frame brought in by tloam_frame_select is built over inside scanMatching as before.
  if (rc == TLOAM_OK && one_rank(c) && !c->no_grid_ahead) {
    double radius[kKinds];
    GridView views[kKinds];
    bool all = true;
    for (int k = 0; k < kKinds; ++k) { radius[k] = kind_radius(c->cfg, k); all = all && c->tgt_box_valid[k]; }
    if (all) {
      // (into a second set of buffers: until the next scanMatching the context's search structures are those of the LAST one,
      //  as the reference's kd-trees are -- getFitnessScore in between sees them, :257-296)
      rc = build_grids(c, c->grids_next, radius, views, nullptr);
      if (rc == TLOAM_OK) {
        for (int k = 0; k < kKinds; ++k) c->gv_next[k] = views[k];
        c->grids_ahead = true;
        c->grids_next_gen = c->tgt_gen;
      }
    }
  }
  return rc;
}

// ---- frames staged ahead of their solve ------------------------------------------------------------
int tloam_frame_stash(tloam_ctx* c, int slot) {
  if (!c || slot < 0 || slot > (1 << 20)) return TLOAM_E_INVALID;
  if (c->active) return TLOAM_E_NOT_READY;
  HIPC(c, hipSetDevice(c->device));
  if ((size_t)slot >= c->frame_store.size()) c->frame_store.resize((size_t)slot + 1, nullptr);
  if (c->frame_selected == slot) {
    // the slot's frame is the registered one (possibly just updated through tloam_set_*): kd[] holds it, the slot holds the
    // context's own clouds.  Exchange them back -- the frame goes into the slot, the context's own clouds become the
    // registered ones again ("select -1") -- so that a later select(slot) finds the frame, not the context's clouds
    exchange_clouds(c, *c->frame_store[slot]);
    c->frame_selected = -1;
    return TLOAM_OK;
  }
  if (c->frame_selected >= 0) return TLOAM_E_NOT_READY;   // another slot's frame is registered: select -1 first
  if (!c->frame_store[slot]) {
    c->frame_store[slot] = new (std::nothrow) FrameClouds();
    if (!c->frame_store[slot]) return TLOAM_E_INVALID;
  } else {
    HIPC(c, hipStreamSynchronize(c->stream));   // nothing in flight may still read the buffers being replaced
    c->frame_store[slot]->release();
    *c->frame_store[slot] = FrameClouds();
  }
  exchange_clouds(c, *c->frame_store[slot]);
  return TLOAM_OK;
}

int tloam_frame_select(tloam_ctx* c, int slot) {
  if (!c || slot < -1) return TLOAM_E_INVALID;
  if (c->active) return TLOAM_E_NOT_READY;
  if (slot >= 0 && ((size_t)slot >= c->frame_store.size() || !c->frame_store[slot])) return TLOAM_E_INVALID;
  if (slot == c->frame_selected) return TLOAM_OK;
  if (c->frame_selected >= 0) exchange_clouds(c, *c->frame_store[c->frame_selected]);   // the context's own clouds back
  if (slot >= 0) exchange_clouds(c, *c->frame_store[slot]);
  c->frame_selected = slot;
  return TLOAM_OK;
}

}  // extern "C"
