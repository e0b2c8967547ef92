// tl_api.hip -- the C ABI of include/tloam_hip.h: context, HBM residency, the host driver of
// LocalRegistration::scanMatching (registration.cpp:879-1133) and the multi-GPU exchange.
//
// Host side mirrors the reference's control flow (outer GNC loop, mu schedule, plateau test);
// every per-point / per-correspondence computation is a HIP kernel (tl_nn.hip, tl_gn.hip).
// There is no CPU fallback: without a usable device every computing entry point returns
// TLOAM_E_HIP.
#include <chrono>

#include "tl_ctx.hpp"

using namespace tl;

struct Uid128 { char bytes[128]; };  // == ncclUniqueId (rccl.h: char internal[128]), passed BY VALUE

namespace {
// ---- RCCL, loaded at run time so the library also loads where librccl is absent ----------------
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Uid128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
bool load_rccl(std::string* err) {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // prefer the copy already in the process
    if (h) break;
  }
  if (!h)
    for (const char* nm : names) {
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) { if (err) *err = std::string("dlopen librccl: ") + dlerror(); return false; }
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Uid128, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    if (err) *err = "librccl: missing symbols";
    return false;
  }
  g_rccl.handle = h;
  return true;
}
constexpr int kNcclFloat64 = 8;  // ncclDataType_t ncclFloat64 (rccl.h)
constexpr int kNcclSum = 0;      // ncclRedOp_t ncclSum
}  // namespace

namespace {

double kind_radius(const tloam_tls_config& c, int k) {
  switch (k) {
    case TLOAM_KIND_PLANAR: return c.planar_dist_thres;
    case TLOAM_KIND_GROUND: return c.ground_dist_thres;
    case TLOAM_KIND_EDGE: return c.edge_dist_thres;
    default: return c.sphere_dist_thres;
  }
}
int kind_maxnum(const tloam_tls_config& c, int k) {
  switch (k) {
    case TLOAM_KIND_PLANAR: return c.planar_maxnum;
    case TLOAM_KIND_GROUND: return c.ground_maxnum;
    case TLOAM_KIND_EDGE: return c.edge_maxnum;
    default: return c.sphere_maxnum;
  }
}
// registration.cpp:979-1016: factor_num 4 -> all four builders, 3 -> planar+ground+edge, 2 -> planar+ground
int kind_active(const tloam_tls_config& c, int k) {
  if (c.factor_num == 4) return 1;
  if (c.factor_num == 3) return k != TLOAM_KIND_SPHERE;
  if (c.factor_num == 2) return k == TLOAM_KIND_PLANAR || k == TLOAM_KIND_GROUND;
  return 0;
}
size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

int sync_stream(tloam_ctx* c) {
  HIPC(c, hipStreamSynchronize(c->stream));
  return TLOAM_OK;
}

// sum all-reduce of a small device buffer of doubles across the ranks of this context
int allreduce(tloam_ctx* c, double* dev, int count) {
  if (c->nranks <= 1 || c->comm == COMM_NONE) return TLOAM_OK;
  if (c->comm == COMM_MAILBOX) {
    if (count > 64) { c->last_error = "mailbox exchange: more than 64 values"; return TLOAM_E_INVALID; }
    launch_mbox_allreduce(dev, count, c->mbox, c->stream);
    return TLOAM_OK;
  }
  if (c->comm == COMM_CALLBACK) {
    const int rc = c->cb(c->cb_user, dev, count, (void*)c->stream);
    if (rc != 0) { c->last_error = "allreduce callback failed"; return TLOAM_E_RCCL; }
    return TLOAM_OK;
  }
  const int rc = g_rccl.AllReduce(dev, dev, (size_t)count, kNcclFloat64, kNcclSum, c->nccl_comm, c->stream);
  if (rc != 0) {
    c->last_error = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return TLOAM_E_RCCL;
  }
  return TLOAM_OK;
}

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

}  // namespace

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
// One 64-byte segment of a result slot (MirrorSlot: seven payload words, then the sequence number XORed with them): wait
// until the XOR of the eight words equals `seq` -- a segment that has only partly arrived does not check -- and copy the
// payload out.  Same return convention as wait_word.
int wait_segment(tloam_ctx* c, const unsigned long long* seg, unsigned long long number, unsigned long long payload[7]) {
  const unsigned long long seq = check_mix(number);   // (what the check word carries, tl_common.hpp)
  for (unsigned spins = 1;; ++spins) {
    unsigned long long w[8], x = 0ull;
    for (int i = 0; i < 8; ++i) { w[i] = __atomic_load_n(seg + i, __ATOMIC_ACQUIRE); x ^= w[i]; }
    if (x == seq) {
      for (int i = 0; i < 7; ++i) payload[i] = w[i];
      return TLOAM_OK;
    }
    if ((spins & 0x7ffu) == 0) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) {
        x = 0ull;
        for (int i = 0; i < 8; ++i) { w[i] = __atomic_load_n(seg + i, __ATOMIC_ACQUIRE); x ^= w[i]; }
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
    c->grids_ahead = false;
    for (int k = 0; k < kKinds; ++k) c->kd[k].grid_valid = false;
    c->have_build = false;
    c->last_error = "a single-pass scan timed out in its look-back (its blocks were not resident together): the context now uses the multi-launch scans";
    rc = TLOAM_E_HIP;
  }
  if (__atomic_load_n(&c->h_fault[kFaultVoxEmit], __ATOMIC_ACQUIRE) != 0u) {
    __atomic_store_n(&c->h_fault[kFaultVoxEmit], 0u, __ATOMIC_RELEASE);
    c->vox_ticket = true;
    c->grids_ahead = false;
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
    const size_t ntiles = (size_t)build_tile_count(out, frame->n_slots);
    HIPC(c, c->tile_cnt.reserve(ntiles + 1 > c->tile_cnt.cap ? 2 * ntiles + 64 : ntiles + 1));
    frame->fi.tile_cnt = c->tile_cnt.p;
    frame->fi.n_tile_cnt = (int)ntiles + 1;
    frame->consumed = true;
  }
  launch_grid_count_all(gs, G.cell_cnt.p, G.cell_of_pt.p, G.rank_of_pt.p, c->stream, frame);
  if (!c->no_scan_1p && scan_1p_applies(nc + 1, c->device_cus)) {
    // 1 M-class tables: count | scan + finalize in ONE single-pass launch | scatter (three launches and one pass over the table
    // less than tile scan + scan of the totals + add + finalize)
    const size_t before = G.scan1p.cap;
    HIPC(c, G.scan1p.reserve(scan_1p_ctl_elems(nc_res + 1)));
    if (G.scan1p.cap != before) HIPC(c, hipMemsetAsync(G.scan1p.p, 0, G.scan1p.cap * sizeof(unsigned long long), c->stream));
    launch_grid_scan_finalize_scatter_1p(gs, G.cell_cnt.p, nc + 1, G.cell_start.p, G.scan1p.p, c->h_fault_dev + kFaultScan1p, G.cell_of_pt.p,
                                         G.rank_of_pt.p, G.gp.p, c->stream);
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
}  // namespace tlh

namespace {

// K3 launch; when the bench armed the timer, with a HIP event pair bound to the dispatch itself
// Every kK3SampleStride-th launch carries the pair (stride 3 is coprime to the 5 sweeps of a Solve and the 20 of
// a frame, so over a few frames every position is sampled equally): timing EVERY launch through
// hipExtLaunchKernelGGL cost ~8 % of the 1 M frame.
constexpr int kK3SampleStride = 3;
int launch_k3_timed(tloam_ctx* c, bool force) {
  const bool sample = c->k3_timing && (c->k3_seq++ % kK3SampleStride) == 0;
  const int idx = c->batch_launches++;
  if (sample) {
    if (c->ev_used + 2 > c->ev_pool.size()) {
      const size_t old = c->ev_pool.size();
      c->ev_pool.resize(old + 256);
      for (size_t i = old; i < c->ev_pool.size(); ++i) HIPC(c, hipEventCreate(&c->ev_pool[i]));
    }
    launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, force, c->stream, c->ev_pool[c->ev_used],
              c->ev_pool[c->ev_used + 1]);
    c->ev_used += 2;
    c->ev_batch_idx.push_back(idx);
  } else {
    launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, force, c->stream);
  }
  return TLOAM_OK;
}
// the same for the one-launch GN iteration (k3_sweep_step): the pair then brackets sweep + fold + step; the streaming part
// alone is what the kernel's own span counter measures (K3Step::span, read by tloam_k3_timer_span)
int launch_k3_step_timed(tloam_ctx* c) {
  const bool sample = c->k3_timing && (c->k3_seq++ % kK3SampleStride) == 0;
  const int idx = c->batch_launches++;
  const MboxView* mb = (c->nranks > 1 && c->comm == COMM_MAILBOX) ? &c->mbox : nullptr;
  if (sample) {
    if (c->ev_used + 2 > c->ev_pool.size()) {
      const size_t old = c->ev_pool.size();
      c->ev_pool.resize(old + 256);
      for (size_t i = old; i < c->ev_pool.size(); ++i) HIPC(c, hipEventCreate(&c->ev_pool[i]));
    }
    launch_k3_step(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_ticket.p, c->k3_span.p, mb, c->stream,
                   c->ev_pool[c->ev_used], c->ev_pool[c->ev_used + 1]);
    c->ev_used += 2;
    c->ev_batch_idx.push_back(idx);
  } else {
    launch_k3_step(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, c->k3_ticket.p, c->k3_span.p, mb, c->stream);
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
  return c->nranks == 1 && c->k3_single && !c->no_fused_small && !c->no_persistent_solve && solve_small_fits(c->k3_grid, c->device_cus);
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
    const int sabotage = c->dbg_fail_handover > 0 ? (c->dbg_fail_handover--, 1 << 16) : 0;   // test hook, see k_solve_all
    launch_solve_small(c->cv, c->state.p, c->partials.p, c->k3_ticket.p, c->k3_grid, sweeps | sabotage, prep, c->seg_n.p, &F, c->stream);
    c->batch_launches++;
    return TLOAM_OK;
  }
  // One GN iteration = ONE launch whatever the size of the set (round 4): the streaming sweep's last block folds the rows and
  // advances the minimiser (k3_sweep_step); with a mailbox it also posts, gathers and advances -- sweep + exchange + step.
  // RCCL / callback contexts keep sweep | collective | step: the collective is enqueued by the host between two launches.
  const bool one_launch = c->fused_large && (c->nranks == 1 ? !(c->k3_single && !c->no_fused_small) : c->comm == COMM_MAILBOX);
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    if (one_launch) {
      const int rc = launch_k3_step_timed(c);
      if (rc != TLOAM_OK) return rc;
      continue;
    }
    if (c->nranks > 1) {
      // sharded GN iteration = 2 launches (+ the collective): the sweep, whose last block folds the rows into the
      // 48-double buffer (the 42 normal-equation scalars + cost) and -- with the mailbox -- stores it straight into
      // every rank's buffer over xGMI; then the step, which (mailbox) adds the ranks' rows in rank order itself
      K3Fuse fuse;
      memset(&fuse, 0, sizeof(fuse));
      fuse.ticket = c->k3_ticket.p;
      fuse.out48 = c->red48.p;
      if (c->comm == COMM_MAILBOX) fuse.mb = c->mbox;
      launch_k3_fused(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, false, fuse, c->stream);
      c->batch_launches++;
      if (c->comm == COMM_MAILBOX) {
        launch_gn_step_mbox(c->state.p, c->mbox, c->stream);
      } else {
        const int rc = allreduce(c, c->red48.p, kReduceBuf);
        if (rc != TLOAM_OK) return rc;
        launch_gn_step(c->state.p, c->red48.p, c->stream);
      }
    } else if (c->k3_single && !c->no_fused_small) {
      // KITTI-size set: one launch per GN iteration (k_sweep_step_small)
      launch_sweep_step_small(c->cv, c->state.p, c->partials.p, c->k3_ticket.p, c->k3_grid, c->stream);
      c->batch_launches++;
    } else {
      const int rc = launch_k3_timed(c, false);
      if (rc != TLOAM_OK) return rc;
      launch_reduce_and_step(c->partials.p, c->k3_grid, c->state.p, c->stream);
    }
  }
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
    HIPC(c, c->k3_span.reserve(4));     // K3Step::span
    HIPC(c, hipMemsetAsync(c->k3_span.p, 0, 4 * sizeof(unsigned long long), c->stream));
  }
  c->cv.seg_n = c->seg_n.p;
  return TLOAM_OK;
}

double alg_bytes_of(const int n[kKinds]) {
  // plane 72 B, line 88 B, point 64 B per correspondence (fp64 SoA, cost write included)
  return 72.0 * ((double)n[TLOAM_KIND_PLANAR] + (double)n[TLOAM_KIND_GROUND]) + 88.0 * (double)n[TLOAM_KIND_EDGE] +
         64.0 * (double)n[TLOAM_KIND_SPHERE];
}

}  // namespace

namespace {
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
  c->grids_ahead = false;
  std::swap(c->src_pack, F.src_pack);
  c->have_build = false;
}
}  // namespace

// ================================================================================================
//  C ABI
// ================================================================================================
extern "C" {

int tloam_abi_version(void) { return TLOAM_ABI_VERSION; }

const char* tloam_status_string(int s) {
  switch (s) {
    case TLOAM_OK: return "ok";
    case TLOAM_E_INVALID: return "invalid argument";
    case TLOAM_E_TOO_FEW_POINTS: return "fewer than 10 points in a feature cloud";
    case TLOAM_E_BAD_POSE: return "predicted pose is not a rigid transform";
    case TLOAM_E_HIP: return "HIP error / no gfx950 device";
    case TLOAM_E_RCCL: return "RCCL error";
    case TLOAM_E_NOT_READY: return "call sequence violated";
    case TLOAM_E_WEIGHT_RANGE: return "GNC weight outside [0,1]";
    default: return "unknown status";
  }
}

void tloam_default_config(tloam_tls_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));  // config/mapping/lidar_odometry.yaml:23-39
  c->k_corr = 10;
  c->factor_num = 4;
  c->edge_dist_thres = 1.0;
  c->edge_dir_thres = 0.85;
  c->edge_maxnum = 1200;
  c->sphere_dist_thres = 0.5;
  c->sphere_maxnum = 200;
  c->planar_dist_thres = 0.5;
  c->planar_maxnum = 2500;
  c->ground_dist_thres = 0.5;
  c->ground_maxnum = 2000;
  c->max_iterations = 4;
  c->cost_threshold = 5e-9;
  c->gnc_factor = 11.8;
  c->noise_bound = 0.01;
  c->fitness_thres = 0.02;
}

int tloam_create(const tloam_tls_config* cfg, int device_id, tloam_ctx** out) {
  if (!cfg || !out) return TLOAM_E_INVALID;
  *out = nullptr;
  // kernel arguments in device memory (a frame is a chain of short dependent launches); only effective if this is
  // the process's first HIP call, never overrides the host's own setting
  (void)setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return TLOAM_E_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return TLOAM_E_HIP;
  tloam_ctx* c = new (std::nothrow) tloam_ctx();
  if (!c) return TLOAM_E_INVALID;
  c->cfg = *cfg;
  c->device = device_id;
  if (const char* e = getenv("TLOAM_DEBUG_MAX_SWEEPS")) c->dbg_max_sweeps = atoi(e);
  c->dbg_no_build_reuse = getenv("TLOAM_NO_BUILD_REUSE") != nullptr;
  c->dbg_no_eval_reuse = getenv("TLOAM_NO_EVAL_REUSE") != nullptr;
  c->no_device_loop = getenv("TLOAM_NO_DEVICE_LOOP") != nullptr;
  c->no_fused_small = getenv("TLOAM_NO_FUSED_SMALL") != nullptr;
  c->no_persistent_solve = getenv("TLOAM_NO_PERSISTENT_SOLVE") != nullptr;
  c->no_grid_ahead = getenv("TLOAM_NO_GRID_AHEAD") != nullptr;
  if (const char* e = getenv("TLOAM_DEBUG_FAIL_HANDOVER")) c->dbg_fail_handover = atoi(e);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) cus = 0;
    c->device_cus = cus;
  }
  c->no_ride_large = getenv("TLOAM_NO_RIDE_LARGE") != nullptr;
  c->fused_large = getenv("TLOAM_FUSED_LARGE") != nullptr;
  c->no_self_prepare = getenv("TLOAM_NO_SELF_PREPARE") != nullptr;
  c->no_finish_in_solve = getenv("TLOAM_NO_FINISH_IN_SOLVE") != nullptr;
  c->enqueue_ahead = getenv("TLOAM_ENQUEUE_AHEAD") ? std::max(1, atoi(getenv("TLOAM_ENQUEUE_AHEAD"))) : 0;
  if (const char* e = getenv("TLOAM_PLANNED_SWEEPS")) c->dbg_planned_sweeps = atoi(e);
  memset(&c->stats, 0, sizeof(c->stats));
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&c->h_state, sizeof(GnState) * kMirrorSlots, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostMalloc((void**)&c->h_small, sizeof(double) * 4096, hipHostMallocDefault) != hipSuccess) {
    delete c;
    return TLOAM_E_HIP;
  }
  memset(c->h_state, 0, sizeof(GnState) * kMirrorSlots);
  {
    // pinned, device-visible host memory the kernels write their results into (result slots, bounding-box rows, fault words):
    // required -- there is no copy + synchronise variant of the paths that use them
    constexpr size_t kBoxBytes = sizeof(double) * ((size_t)kKinds * 64 * 6 + 8);
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
    if (hipHostMalloc((void**)&c->h_mirror, sizeof(MirrorSlot) * kMirrorSlots, flags) != hipSuccess ||
        ((uintptr_t)c->h_mirror & 63u) != 0 || hipHostGetDevicePointer((void**)&c->h_mirror_dev, c->h_mirror, 0) != hipSuccess ||
        hipHostMalloc((void**)&c->h_bbox, kBoxBytes, flags) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->h_bbox_dev, c->h_bbox, 0) != hipSuccess ||
        hipHostMalloc((void**)&c->h_fault, sizeof(unsigned) * kFaultWords, flags) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->h_fault_dev, c->h_fault, 0) != hipSuccess) {
      (void)hipGetLastError();
      tloam_destroy(c);
      return TLOAM_E_HIP;
    }
    memset(c->h_mirror, 0, sizeof(MirrorSlot) * kMirrorSlots);
    memset(c->h_bbox, 0, kBoxBytes);
    memset(c->h_fault, 0, sizeof(unsigned) * kFaultWords);
  }
  if (ensure_common(c) != TLOAM_OK) { tloam_destroy(c); return TLOAM_E_HIP; }
  (void)hipMemsetAsync(c->state.p, 0, sizeof(GnState), c->stream);
  (void)hipMemsetAsync(c->seg_n.p, 0, 8 * sizeof(int), c->stream);
  (void)hipStreamSynchronize(c->stream);
  *out = c;
  return TLOAM_OK;
}

void tloam_destroy(tloam_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl_comm);
  for (int r = 0; r < kMaxRanks; ++r)
    if (c->mbox_opened[r]) (void)hipIpcCloseMemHandle(c->mbox_opened[r]);
  if (c->mbox_local) (void)hipFree(c->mbox_local);
  c->scan1p_q.release(); c->mbox_ctr.release(); c->k3_ticket.release(); c->k3_span.release(); c->fin_rows.release(); c->flagb.release();
  for (auto& e : c->ev_pool) (void)hipEventDestroy(e);
  // (a slot still selected: its clouds are in kd[], the context's own in the slot -- put them back first, so that every
  //  buffer is released exactly once below)
  if (c->frame_selected >= 0) { exchange_clouds(c, *c->frame_store[c->frame_selected]); c->frame_selected = -1; }
  for (int k = 0; k < kKinds; ++k) {
    KindData& K = c->kd[k];
    K.src_aos.release(); K.tgt_aos.release(); K.tx.release(); K.ty.release(); K.tz.release();
    K.c_idx.release(); K.c_buf.release();
  }
  c->src_pack.release();
  for (int h = 0; h < 2; ++h) {
    if (c->h_stage[h]) (void)hipHostFree(c->h_stage[h]);
    if (c->stage_ev[h]) (void)hipEventDestroy(c->stage_ev[h]);
  }
  c->sx.release(); c->sy.release(); c->sz.release(); c->w_src.release();
  c->fit_x.release(); c->fit_y.release(); c->fit_z.release();
  c->raw.release(); c->flags.release(); c->scan.release(); c->scan_tmp.release(); c->seg_n.release();
  c->tile_cnt.release(); c->tile_scan.release(); c->tile_of_slot.release(); c->tile_fill.release(); c->qrec.release();
  c->partials.release(); c->red48.release(); c->sums16.release(); c->wpart.release(); c->rank_counts.release();
  c->se3_dev.release(); c->bbox_dev.release(); c->misc.release(); c->state.release(); c->grids.release(); c->grids_next.release();
  for (auto* f : c->frame_store)
    if (f) { f->release(); delete f; }
  c->frame_store.clear();
  c->submap.release();
  c->feat.release();
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->h_mirror) (void)hipHostFree(c->h_mirror);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->h_bbox) (void)hipHostFree(c->h_bbox);
  if (c->h_fault) (void)hipHostFree(c->h_fault);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* tloam_last_error(const tloam_ctx* c) { return c ? c->last_error.c_str() : ""; }

void tloam_shard_range(size_t n, int rank, int nranks, size_t* lo, size_t* hi) {
  if (nranks < 1) nranks = 1;
  if (rank < 0) rank = 0;
  if (rank >= nranks) rank = nranks - 1;
  // contiguous index blocks keep the reference's "first N valid in index order" cap semantics
  const size_t a = (size_t)(((unsigned __int128)n * (unsigned)rank) / (unsigned)nranks);
  const size_t b = (size_t)(((unsigned __int128)n * (unsigned)(rank + 1)) / (unsigned)nranks);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

// A whole Frame over the ranks: the four source clouds laid end to end (planar | ground | edge | sphere), the line cut into
// nranks equal pieces, every rank the intersection of its piece with each cloud.  Per kind that is still contiguous index
// blocks in rank order (so the cap prefix over the lower ranks' counts, registration.cpp:448/:538/:592/:735, holds as it
// is), every rank gets the same number of queries whatever the mix, and a rank touches one or two kinds instead of four:
// it builds only THOSE kinds' search grids (tloam_sm_begin) -- the target-grid build is what a sharded frame replicates.
void tloam_shard_ranges_frame(const size_t n[4], int rank, int nranks, size_t lo[4], size_t hi[4]) {
  if (nranks < 1) nranks = 1;
  if (rank < 0) rank = 0;
  if (rank >= nranks) rank = nranks - 1;
  unsigned __int128 total = 0;
  for (int k = 0; k < kKinds; ++k) total += n[k];
  const size_t a = (size_t)((total * (unsigned)rank) / (unsigned)nranks), b = (size_t)((total * (unsigned)(rank + 1)) / (unsigned)nranks);
  size_t base = 0;
  for (int k = 0; k < kKinds; ++k) {
    const size_t l = std::min(std::max(a, base), base + n[k]), h = std::min(std::max(b, base), base + n[k]);
    lo[k] = l - base;
    hi[k] = h - base;
    base += n[k];
  }
}

// ---- setInputSource / setInputTarget (registration.cpp:232-248) --------------------------------
namespace {
int set_source_async(tloam_ctx* c, int kind, const double* xyz, size_t n) {
  if (kind < 0 || kind >= kKinds || (n > 0 && !xyz)) return TLOAM_E_INVALID;
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
  tloam_shard_ranges_frame(n, c->rank, c->nranks, lo4, hi4);
  for (int k = 0; k < kKinds; ++k) {
    if (n[k] > 0 && !xyz[k]) return TLOAM_E_INVALID;
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
  if (kind < 0 || kind >= kKinds || (n > 0 && !xyz)) return TLOAM_E_INVALID;
  KindData& K = c->kd[kind];
  K.n_tgt = n;
  c->tgt_box_valid[kind] = false;
  c->grids_ahead = false;
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
  if (rc == TLOAM_OK && c->nranks == 1 && !c->no_grid_ahead) {
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
  // ---- compact segments: at most min(n_src, maxnum) factors per kind
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
    k3_plan(caps, &c->k3_grid, &c->k3_single);
    (void)total_cap;
  }
  {
    // the one-launch Solve compacts the factor set itself when every kind's flag bytes fit a wave (SlotView::flagb)
    bool fits = solve_small_path(c) && prepare_small_fits(c->sv) && !c->no_self_prepare;
    for (int k = 0; k < kKinds; ++k) fits = fits && c->kd[k].n_src <= (size_t)kFlagbStride;
    c->sv.flagb = nullptr;
    if (fits) {
      HIPC(c, c->flagb.reserve((size_t)kKinds * kFlagbStride));
      c->sv.flagb = c->flagb.p;
    }
  }
  HIPC(c, c->partials.reserve(std::max<size_t>((size_t)c->k3_grid * kAccStride, 4096)));
  // ---- the start of the frame -- scan-frame sources AoS -> SoA slots, weights = 1 (:931-949), flag-scan terminator,
  //      minimiser state zeroed with `parameters` = x (passed by value) and armed for the first Solve -- rides on the
  //      first launch of the grid build
  FrameInitHook hook;
  memset(&hook, 0, sizeof(hook));
  for (int k = 0; k < kKinds; ++k) { hook.fi.src_aos[k] = c->kd[k].src_ptr; hook.fi.slot_off[k] = c->sv.slot_off[k]; }
  hook.fi.slot_off[kKinds] = c->sv.slot_off[kKinds];
  for (int i = 0; i < 6; ++i) hook.fi.x[i] = x[i];
  hook.fi.no_eval_reuse = c->dbg_no_eval_reuse ? 1 : 0;
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
    if (c->grids_ahead && c->nranks == 1) {
      // built when the targets were handed over (tloam_set_target_frame): they become the context's search structures now;
      // the frame's start is a launch of its own, below.  Used once: a second scanMatching over the same targets builds its own
      std::swap(c->grids, c->grids_next);
      for (int k = 0; k < kKinds; ++k) { c->kd[k].gv = c->gv_next[k]; c->kd[k].grid_valid = true; }
      c->grids_ahead = false;
    } else {
      rc = build_grids(c, c->grids, radius, views, &hook);
      if (rc != TLOAM_OK) return rc;
      for (int k = 0; k < kKinds; ++k) { c->kd[k].gv = views[k]; c->kd[k].grid_valid = true; }
    }
  }
  if (!hook.consumed) {  // (no grid launch: cannot happen with >= 10 targets per kind, kept for safety)
    GridView gviews[kKinds];
    for (int k = 0; k < kKinds; ++k) gviews[k] = c->kd[k].gv;
    const size_t ntiles = (size_t)build_tile_count(gviews, c->sv.slot_off[kKinds]);
    HIPC(c, c->tile_cnt.reserve(ntiles + 1 > c->tile_cnt.cap ? 2 * ntiles + 64 : ntiles + 1));   // (room to spare, as build_grids_over)
    hook.fi.tile_cnt = c->tile_cnt.p;
    hook.fi.n_tile_cnt = (int)ntiles + 1;
    launch_frame_init(hook.fi, hook.b, c->stream);
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
int outer_reserve(tloam_ctx* c, const GridView grids[kKinds]) {
  const size_t n_slots = (size_t)c->sv.slot_off[kKinds];
  const size_t ntiles = (size_t)build_tile_count(grids, c->sv.slot_off[kKinds]);
  // (tile counts follow the bounding boxes like the cell tables: grow with room to spare)
  const size_t nt_res = (ntiles + 1 > c->tile_cnt.cap || ntiles + 1 > c->tile_scan.cap) ? 2 * ntiles + 64 : ntiles;
  HIPC(c, c->tile_cnt.reserve(nt_res + 1)); HIPC(c, c->tile_scan.reserve(nt_res + 1));
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
// :976-1020 the four builders (K1 + K2), the flag scan, the index-order caps.  Small single-rank frames: the scan, the
// caps, the compaction AND the alternative (refresh) are one launch (k_prepare_small) -- `also_refresh` says whether this
// call stands for both alternatives of a device-gated iteration.
bool prepare_small_path(const tloam_ctx* c) { return c->nranks == 1 && prepare_small_fits(c->sv) && !c->no_fused_small; }
// the Solve launch that follows prepares the set itself: no k_prepare_small
bool self_prepare_path(const tloam_ctx* c) { return c->sv.flagb != nullptr && prepare_small_path(c) && solve_small_path(c); }
// ride: the finish of the previous outer iteration rides on this search launch (large single-rank sets, device-driven loop:
// k_build_finish_large; the search then runs on GnState::spec_build instead of `gate`)
int enqueue_build(tloam_ctx* c, const BuildParams& bp, const GridView grids[kKinds], bool rebin, const int* gate,
                  const int* refresh_gate = nullptr, bool prepare_in_solve = false, const FinishLargeArgs* ride = nullptr) {
  const size_t n_slots = (size_t)c->sv.slot_off[kKinds];
  if (ride && !rebin) {
    const size_t ntiles = (size_t)build_tile_count(grids, c->sv.slot_off[kKinds]);
    launch_build_finish_large(c->sv, grids, bp, c->state.p, c->tile_scan.p + ntiles, c->qrec.p, *ride, c->stream);
  } else {
    launch_build(c->sv, grids, bp, c->state.p, c->tile_of_slot.p, c->tile_cnt.p, c->tile_scan.p, c->tile_fill.p,
                 c->qrec.p, c->scan_tmp.p, rebin, c->stream, gate, c->scan1p_q_use ? c->scan1p_q.p : nullptr, c->h_fault_dev + kFaultScan1p);
  }
  if (prepare_in_solve) return TLOAM_OK;
  if (prepare_small_path(c)) {
    launch_prepare_small(c->sv, c->cv, bp, c->seg_n.p, c->state.p, gate, refresh_gate, c->stream);
    return TLOAM_OK;
  }
  if (c->nranks == 1 && !c->no_fused_small) {
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
  if (c->nranks > 1) {
    launch_rank_counts(c->sv, c->rank_counts.p, c->rank, c->nranks, c->stream);
    const int rc = allreduce(c, c->rank_counts.p, c->nranks * kKinds);
    if (rc != TLOAM_OK) return rc;
    rank_counts = c->rank_counts.p;
  }
  // seg_n: every kind with slots and a positive cap is rewritten by the compaction, the others keep the 0 of
  // k_frame_init; only a sharded rank can find its cap already filled by the lower ranks and write nothing
  if (c->nranks > 1) HIPC(c, hipMemsetAsync(c->seg_n.p, 0, kKinds * sizeof(int), c->stream));
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
bool finish_small_path(const tloam_ctx* c) { return c->nranks == 1 && total_seg_cap(c) <= 16384; }
int enqueue_finish(tloam_ctx* c, const WeightParams& wp, const HostMirror& hm, const OuterCtl& ctl) {
  // fixed function of the capacity (so the summation tree, hence the bits, do not depend on timing)
  const size_t cap = total_seg_cap(c);
  const int wblocks = (int)std::min<size_t>(256, std::max<size_t>(64, cap / 2048));
  if (finish_small_path(c)) {
    launch_weights_finish_small(c->cv, c->sv, wp, c->seg_n.p, c->sums16.p, c->state.p, hm, ctl, c->stream);
    return TLOAM_OK;
  }
  launch_weights(c->cv, c->sv, wp, c->wpart.p, wblocks, c->state.p, c->stream);
  if (c->nranks > 1) {
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
  if (!same_pose) {
    rc = enqueue_build(c, bp, grids, /*rebin=*/iter == 0, nullptr);
    if (rc != TLOAM_OK) return rc;
    memcpy(c->build_x, c->stats.se3, sizeof(c->build_x));
    c->have_build = true;
  } else {
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
    rc = enqueue_finish(c, wp, hm, host_decides);
    if (rc != TLOAM_OK) return rc;
    rc = wait_state(c, hm);
    if (rc != TLOAM_OK) return rc;
    if (!c->h_state->incomplete) break;
    if (c->h_state->incomplete == OS_COMM_ERROR) {
      c->last_error = c->nranks > 1 ? "mailbox exchange timed out: a peer rank did not post (dead process or diverged call sequence)"
                                    : "in-launch hand-over of the fused GN iteration timed out (a block of the grid never posted its row)";
      return c->nranks > 1 ? TLOAM_E_RCCL : TLOAM_E_HIP;
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
constexpr int kEnqueueAhead = 2;   // (TLOAM_ENQUEUE_AHEAD, read when the context is created: tloam_ctx::enqueue_ahead)
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
  const bool ride_large = !ride && c->nranks == 1 && !finish_small_path(c) && build_finish_large_fits(c->sv) && !c->no_ride_large;
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
  const bool fin_in_solve = in_solve && finish_small_path(c) && !c->no_finish_in_solve && M <= kMaxOuterInLaunch;
  P.in_launch_finish = fin_in_solve;
  if (fin_in_solve) {
    SolveFinish& F = P.F;
    memset(&F, 0, sizeof(F));
    F.enabled = 1;
    F.have_wp = 1;
    F.n_iter = M;
    F.cost_threshold = c->cfg.cost_threshold;
    F.sums16 = c->sums16.p;
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
    return enqueue_iterations_in_launch_mode(c, first, std::min(M, first + (c->enqueue_ahead > 0 ? c->enqueue_ahead : kEnqueueAhead)), bp, grids, P);
  }
  for (int iter = first; iter < M; ++iter) {
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
      const FinishLargeArgs fin{&c->cv, &wp_prev, c->seg_n.p, c->sums16.p, P.hms[iter - 1], ctl_prev, c->fin_rows.p, c->k3_ticket.p + 1,
                                wblocks_large};
      rc = enqueue_build(c, bp, grids, /*rebin=*/false, run_build, run_refresh, in_solve, &fin);
      prep.run_build = run_build;
      prep.run_refresh = run_refresh;
      pending_large = false;
    } else {
      rc = enqueue_build(c, bp, grids, /*rebin=*/false, run_build, run_refresh, in_solve);   // both alternatives, device-gated
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
      rc = enqueue_finish(c, weight_params(c, mu, bp), P.hms[iter], ctl);
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
    Sm->incomplete &= ~(int)OS_NEEDS_HOST;
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
      rc = enqueue_solve(c, /*armed=*/true, kSolveSweeps, &wp_top);
      if (rc != TLOAM_OK) return rc;
      P.hms[iter] = next_mirror(c, iter);
      rc = enqueue_finish(c, wp_top, P.hms[iter], OuterCtl{c->cfg.cost_threshold, 1, iter == M - 1 ? 1 : 0});
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

// development aid (TLOAM_HOST_PROFILE=1, single frame stream only): where the calling thread's time goes per
// tloam_scan_match, averaged, printed to stderr every 200 calls
struct HostProf { double begin_us = 0, enqueue_us = 0, wait_us = 0, tail_us = 0, gap_us = 0; long n = 0; std::chrono::steady_clock::time_point last_ret; bool have_last = false; };
static HostProf g_hp;
static const bool g_hp_on = getenv("TLOAM_HOST_PROFILE") != nullptr;
int tloam_scan_match(tloam_ctx* c, const double predict[16], const double* omega3, double result[16],
                     double* scan_xyz, size_t n_scan, tloam_stats* stats) {
  const auto hp_t0 = std::chrono::steady_clock::now();
  int rc = tloam_sm_begin(c, predict, omega3);
  if (rc != TLOAM_OK) return rc;
  const auto hp_t1 = std::chrono::steady_clock::now();
  const double hp_wait0 = c->wait_us;
  int done = 0;
  bool weight_violation = false;
  // device-driven outer loop where the stepwise machinery is not asked for: one rank, a pinned mirror, the planned
  // iterations fit the result slots, no development knob that needs the host between iterations
  if (c->nranks == 1 && c->cfg.max_iterations >= 1 && c->cfg.max_iterations <= kMaxOuterFast &&
      !c->dbg_no_build_reuse && !c->no_device_loop) {
    const bool persistent = solve_small_path(c);
    rc = scan_match_device_loop(c, &weight_violation);
    if (rc == TLOAM_E_HIP && persistent && c->hand_over_timed_out) {
      // A block of the one-launch Solve was never scheduled beside the others (a device shared with long-running kernels,
      // fewer usable CUs than the attribute says): the waits inside the launch are bounded, the frame is intact in HBM --
      // solve it again with one launch per GN iteration, and keep this context on that path.
      c->no_persistent_solve = true;
      c->hand_over_timed_out = false;
      (void)hipStreamSynchronize(c->stream);
      c->active = false;
      rc = tloam_sm_begin(c, predict, omega3);
      if (rc != TLOAM_OK) return rc;
      weight_violation = false;
      rc = scan_match_device_loop(c, &weight_violation);
    }
    if (rc < 0) return rc;
    done = rc == 0 ? 1 : 0;
  }
  while (!done) {
    rc = tloam_sm_outer(c, &done, nullptr);
    if (rc == TLOAM_E_WEIGHT_RANGE) { weight_violation = true; continue; }  // reported after the solve
    if (rc != TLOAM_OK) return rc;
  }
  rc = tloam_sm_end(c, result, stats);
  if (rc == TLOAM_E_HIP && c->no_scan_1p && !c->scan1p_retried) {
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
  if (g_hp_on) {
    const auto t2 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    g_hp.begin_us += us(hp_t0, hp_t1);
    g_hp.wait_us += c->wait_us - hp_wait0;
    g_hp.enqueue_us += us(hp_t1, t2) - (c->wait_us - hp_wait0);   // enqueue + bookkeeping + end, without the waits
    if (g_hp.have_last) g_hp.gap_us += us(g_hp.last_ret, hp_t0);
    g_hp.last_ret = t2; g_hp.have_last = true;
    if (++g_hp.n % 200 == 0)
      fprintf(stderr, "[tloam host] per call: sm_begin %.1f us, enqueue + bookkeeping %.1f, waiting %.1f, outside the call %.1f\n",
              g_hp.begin_us / g_hp.n, g_hp.enqueue_us / g_hp.n, g_hp.wait_us / g_hp.n, g_hp.gap_us / g_hp.n);
  }
  return weight_violation ? TLOAM_E_WEIGHT_RANGE : TLOAM_OK;
}

// ---- getFitnessScore (registration.cpp:257-296) -------------------------------------------------
int tloam_fitness(tloam_ctx* c, double* fitness, double* rmse) {
  if (!c || !fitness || !rmse) return TLOAM_E_INVALID;
  *fitness = 0.0;
  *rmse = 0.0;
  if (c->cfg.fitness_thres <= 0.0) return TLOAM_OK;  // :258-261
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
    if (!K.grid_valid || K.n_src == 0) continue;
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

int tloam_get_correspondences(tloam_ctx* c, int kind, size_t capacity, size_t* n, int32_t* src_index, double* a,
                              double* b, double* d, double* w, double* cost) {
  if (!c || kind < 0 || kind >= kKinds || !n) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
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
  const size_t m = c->kd[kind].n_src;
  *n = m;
  if (m > capacity || !c->w_src.p) return TLOAM_E_INVALID;
  HIPC(c, hipStreamSynchronize(c->stream));
  if (w && m > 0) HIPC(c, hipMemcpy(w, c->w_src.p + c->sv.slot_off[kind], sizeof(double) * m, hipMemcpyDeviceToHost));
  return TLOAM_OK;
}

int tloam_knn(tloam_ctx* c, int kind, const double* q, size_t nq, double radius, int k, int32_t* out_idx,
              double* out_d2, int32_t* out_cnt) {
  if (!c || kind < 0 || kind >= kKinds || !q || k < 1 || k > kMaxK || !(radius > 0.0) || !out_idx || !out_d2 || !out_cnt)
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
  if (!c || res_type < 0 || res_type >= TLOAM_NUM_RES) return TLOAM_E_INVALID;
  if (n > 0 && (!p || !a || !w || (res_type == TLOAM_RES_LINE && !b) || (res_type == TLOAM_RES_PLANE && !d)))
    return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  int rc = ensure_common(c);
  if (rc != TLOAM_OK) return rc;
  const int kind = res_type == TLOAM_RES_PLANE ? TLOAM_KIND_PLANAR : (res_type == TLOAM_RES_LINE ? TLOAM_KIND_EDGE : TLOAM_KIND_SPHERE);
  if (!c->prebuilt) {
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
    k3_plan(caps, &c->k3_grid, &c->k3_single);
    (void)total_cap;
  }
  HIPC(c, c->partials.reserve(std::max<size_t>((size_t)c->k3_grid * kAccStride, 4096)));
  return TLOAM_OK;
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
  for (int i = 0; i < launches; ++i) launch_k3(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, true, c->stream);
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
  const bool mbox = with_exchange && c->comm == COMM_MAILBOX && c->nranks > 1;
  if (mbox) fuse.mb = c->mbox;
  hipEvent_t e0, e1;
  HIPC(c, hipEventCreate(&e0));
  HIPC(c, hipEventCreate(&e1));
  HIPC(c, hipEventRecord(e0, c->stream));
  int rc = TLOAM_OK;
  for (int i = 0; i < launches && rc == TLOAM_OK; ++i) {
    launch_k3_fused(c->cv, c->state.p, c->partials.p, c->k3_grid, c->k3_single, true, fuse, c->stream);
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

int tloam_debug_raise_fault(tloam_ctx* c, int which) {
  if (!c || which < 0 || which >= kFaultWords || !c->h_fault) return TLOAM_E_INVALID;
  __atomic_store_n(&c->h_fault[which], 1u, __ATOMIC_RELEASE);
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

// ---- multi-GPU -------------------------------------------------------------------------------------
int tloam_rccl_unique_id(void* out128) {
  if (!out128) return TLOAM_E_INVALID;
  std::string err;
  if (!load_rccl(&err)) return TLOAM_E_RCCL;
  Uid128 id;
  memset(&id, 0, sizeof(id));
  if (g_rccl.GetUniqueId(&id) != 0) return TLOAM_E_RCCL;
  memcpy(out128, &id, sizeof(id));
  return TLOAM_OK;
}

int tloam_comm_init_rccl(tloam_ctx* c, int rank, int nranks, const void* unique_id128) {
  if (!c || !unique_id128 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!load_rccl(&c->last_error)) return TLOAM_E_RCCL;
  Uid128 id;
  memcpy(&id, unique_id128, sizeof(id));
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (rc != 0) {
    c->last_error = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return TLOAM_E_RCCL;
  }
  c->nccl_comm = comm;
  c->rank = rank;
  c->nranks = nranks;
  c->comm = COMM_RCCL;
  return TLOAM_OK;
}

int tloam_comm_init_callback(tloam_ctx* c, int rank, int nranks, tloam_allreduce_fn fn, void* user) {
  if (!c || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return TLOAM_E_INVALID;
  c->rank = rank;
  c->nranks = nranks;
  c->cb = fn;
  c->cb_user = user;
  c->comm = COMM_CALLBACK;
  return TLOAM_OK;
}

// (c) one-shot peer exchange over xGMI, no collective library on the data path: every rank exports a small
//     fine-grained buffer through HIP IPC, maps its peers', and from then on a sharded GN iteration is the sweep
//     (its last block stores the 48 doubles into every rank's buffer) and the step (adds them in rank order).
static_assert(sizeof(hipIpcMemHandle_t) == 64, "tloam_comm_mailbox_export hands out 64 bytes");
int tloam_comm_mailbox_export(tloam_ctx* c, void* handle64) {
  if (!c || !handle64) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  if (!c->mbox_local) {
    const size_t bytes = sizeof(double) * kMboxDoubles;
    void* p = nullptr;
    // uncached fine-grained device memory: peers' stores land in memory, local polls read memory
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    HIPC(c, e);
    HIPC(c, hipMemset(p, 0, bytes));
    c->mbox_local = (double*)p;
  }
  hipIpcMemHandle_t h;
  HIPC(c, hipIpcGetMemHandle(&h, c->mbox_local));
  memcpy(handle64, &h, sizeof(h));
  return TLOAM_OK;
}

int tloam_comm_init_mailbox(tloam_ctx* c, int rank, int nranks, const void* handles64) {
  if (!c || !handles64 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return TLOAM_E_INVALID;
  if (!c->mbox_local) return TLOAM_E_NOT_READY;  // export first
  HIPC(c, hipSetDevice(c->device));
  memset(&c->mbox, 0, sizeof(c->mbox));
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { c->mbox.peer[r] = c->mbox_local; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles64 + 64 * (size_t)r, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      c->last_error = std::string("hipIpcOpenMemHandle(rank ") + std::to_string(r) + "): " + hipGetErrorString(e);
      (void)hipGetLastError();
      return TLOAM_E_RCCL;
    }
    c->mbox_opened[r] = p;
    c->mbox.peer[r] = (double*)p;
  }
  HIPC(c, c->mbox_ctr.reserve(4));
  HIPC(c, hipMemset(c->mbox_ctr.p, 0, 4 * sizeof(unsigned long long)));
  c->mbox.ctr = c->mbox_ctr.p;
  c->mbox.rank = rank;
  c->mbox.nranks = nranks;
  c->rank = rank;
  c->nranks = nranks;
  c->comm = COMM_MAILBOX;
  return TLOAM_OK;
}

// ---- SE(3) helpers (host) ------------------------------------------------------------------------------
int tloam_se3_exp(const double se3[6], double T[16]) {
  if (!se3 || !T) return TLOAM_E_INVALID;
  pose_to_matrix(se3_exp(se3), T);
  return TLOAM_OK;
}
int tloam_se3_log(const double T[16], double se3[6]) {
  if (!se3 || !T) return TLOAM_E_INVALID;
  Pose P;
  if (!pose_from_matrix(T, &P)) return TLOAM_E_BAD_POSE;
  se3_log(P, se3);
  return TLOAM_OK;
}
int tloam_se3_plus(const double x[6], const double delta[6], double out[6]) {
  if (!x || !delta || !out) return TLOAM_E_INVALID;
  se3_plus(x, delta, out);
  return TLOAM_OK;
}

}  // extern "C"
