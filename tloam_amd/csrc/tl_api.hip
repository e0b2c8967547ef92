// tl_api.hip -- lifetime of the C ABI's context (include/tloam_hip.h): create / destroy, configuration, status strings,
// the sharding rules, the host SE(3) helpers.  The other units: tl_api_frames.hip (HBM residency, search grids),
// tl_api_match.hip (the scanMatching driver), tl_api_comm.hip (multi-GPU exchange), tl_api_submap.hip, tl_api_feature.hip.
//
// There is no CPU fallback: without a usable device every computing entry point returns TLOAM_E_HIP.
#include "tl_ctx.hpp"

using namespace tl;

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
  c->no_persistent_solve = getenv("TLOAM_NO_PERSISTENT_SOLVE") != nullptr;
  c->no_grid_ahead = getenv("TLOAM_NO_GRID_AHEAD") != nullptr;
  c->no_direct_set = getenv("TLOAM_NO_DIRECT_SET") != nullptr;
  c->no_qbin_ride = getenv("TLOAM_NO_QBIN_RIDE") != nullptr;
  if (const char* e = getenv("TLOAM_DEBUG_FAIL_HANDOVER")) c->dbg_fail_handover = atoi(e);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) cus = 0;
    c->device_cus = cus;
  }
  c->fused_large = getenv("TLOAM_FUSED_LARGE") != nullptr;
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
  tlh::comm_release(c);
  c->scan1p_q.release(); c->k3_ticket.release(); c->k3_span.release(); c->iter_span.release(); c->state_scratch.release(); c->blk_cnt.release(); c->row_of_pos.release(); c->fin_tickets.release(); c->fin_rows.release(); c->flagb.release();
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

int tloam_get_info(tloam_ctx* c, tloam_ctx_info* out) {
  if (!c || !out) return TLOAM_E_INVALID;
  memset(out, 0, sizeof(*out));
  out->abi_version = TLOAM_ABI_VERSION;
  out->device = c->device;
  out->device_cus = c->device_cus;
  out->comm_mode = (int32_t)c->comm;
  out->rank = c->rank;
  out->nranks = c->nranks;
  out->loopback = c->loopback ? 1 : 0;
  out->direct_set = (c->direct && !c->prebuilt) ? 1 : 0;
  out->set_stale = c->set_stale ? 1 : 0;
  tlh::comm_rccl_info(c, &out->rccl_comm_count, &out->rccl_comm_rank);
  out->fallbacks_taken = (c->no_scan_1p ? TLOAM_FALLBACK_SCAN : 0) | (c->vox_ticket ? TLOAM_FALLBACK_VOXEL : 0) |
                         (c->persistent_solve_timed_out ? TLOAM_FALLBACK_SOLVE : 0);
  out->fallback_events = c->fallback_events;
  out->k3_grid = c->k3_grid;
  out->k3_single = c->k3_single ? 1 : 0;
  out->k3_wide = c->k3_wide ? 1 : 0;
  out->one_launch_solve = (one_rank(c) && c->k3_single && !c->no_persistent_solve && solve_small_fits(c->k3_grid, c->device_cus)) ? 1 : 0;
  return TLOAM_OK;
}

int tloam_debug_raise_fault(tloam_ctx* c, int which) {
  if (!c || which < 0 || which >= kFaultWords || !c->h_fault) return TLOAM_E_INVALID;
  __atomic_store_n(&c->h_fault[which], 1u, __ATOMIC_RELEASE);
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
