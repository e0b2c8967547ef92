// tl_api_submap.hip -- C ABI of the device-resident submap (include/tloam_hip.h: tloam_submap_*, tloam_get_target):
// FrontEnd::updateSubmap (front_end.cpp:201-275) and the first-frame branch (:283-304) driven on the device
// (kernels in tl_submap.hip).  The result is written straight into the SoA target arrays of the context.
#include "tl_ctx.hpp"

using namespace tl;

extern "C" {

// ---- submap maintenance on the device (front_end.cpp:201-275, :283-304) ---------------------------
void tloam_submap_default_config(tloam_submap_config* cfg) {
  if (!cfg) return;
  cfg->planar_frame_size = 3;
  cfg->sphere_frame_size = 3;
  cfg->edge_crop_box_length = 100.0;
  cfg->ground_crop_box_length = 100.0;
  cfg->edge_down_sample_submap = 0.3;
  cfg->ground_down_sample_submap = 0.45;
  cfg->ground_down_sample = 0.3;
}

namespace {
int submap_reserve_work(tloam_ctx* c, size_t n) {
  SubmapState& S = c->submap;
  const size_t m = std::max<size_t>(n, 1), cap = voxel_table_size(m);
  HIPC(c, S.min_partial.reserve(256 * 6)); HIPC(c, S.vmin.reserve(8)); HIPC(c, S.counts.reserve(8));
  if (!S.overflow.p) {   // [0] overflow flag, [1] the ticket of k_vox_min2: zero between launches, so zero before the first
    HIPC(c, S.overflow.reserve(8));
    HIPC(c, hipMemsetAsync(S.overflow.p, 0, 8 * sizeof(int), c->stream));
  }
  HIPC(c, S.keys.reserve(cap + 1)); HIPC(c, S.cnt.reserve(cap + 1)); HIPC(c, S.off.reserve(cap + 1));
  HIPC(c, S.slot_of_pt.reserve(m)); HIPC(c, S.urank.reserve(m)); HIPC(c, S.members.reserve(m)); HIPC(c, S.sorted.reserve(m));
  HIPC(c, S.leader.reserve(m + 1)); HIPC(c, S.leader_scan.reserve(m + 1));
  HIPC(c, S.scan_tmp.reserve(scan_tmp_elems(std::max(cap + 1, m + 1))));
  return TLOAM_OK;
}
// One launch sequence for one or two clouds stored back to back in (wx, wy, wz): segment s = points
// [s ? n0 : 0, s ? n : n0) -> target[kind[s]] = VoxelDownSample(Crop(segment, box[s]), voxel[s]); sizes to counts[s].
// A single cloud: n0 == n, kind[1] ignored.
struct CropVoxelSeg { int kind; size_t n; const double* lo; const double* hi; double voxel; };
int submap_job(tloam_ctx* c, const CropVoxelSeg seg[2], int nseg, VoxelJob* Jout, VoxelWork* Wout) {
  SubmapState& S = c->submap;
  const size_t n0 = seg[0].n, n = n0 + (nseg > 1 ? seg[1].n : 0);
  int rc = submap_reserve_work(c, n);
  if (rc != TLOAM_OK) return rc;
  VoxelJob J;
  J.x = S.wx.p; J.y = S.wy.p; J.z = S.wz.p;
  J.n = n;
  J.n0 = n0;
  VoxelWork W;
  for (int s = 0; s < 2; ++s) {
    const CropVoxelSeg& G = seg[s < nseg ? s : 0];
    for (int a = 0; a < 3; ++a) { J.lo[s][a] = G.lo[a]; J.hi[s][a] = G.hi[a]; }
    J.voxel[s] = G.voxel;
    KindData& K = c->kd[G.kind];
    if (s < nseg) {
      const size_t m = std::max<size_t>(G.n, 1);
      HIPC(c, K.tx.reserve(m)); HIPC(c, K.ty.reserve(m)); HIPC(c, K.tz.reserve(m));
    }
    W.out[s][0] = K.tx.p; W.out[s][1] = K.ty.p; W.out[s][2] = K.tz.p;
  }
  J.mask = voxel_table_size(std::max<size_t>(n, 1)) - 1;
  W.min_partial = S.min_partial.p; W.vmin = S.vmin.p;
  W.keys = S.keys.p; W.cnt = S.cnt.p; W.off = S.off.p;
  W.slot_of_pt = S.slot_of_pt.p; W.urank = S.urank.p; W.members = S.members.p; W.sorted = S.sorted.p;
  W.leader = S.leader.p; W.leader_scan = S.leader_scan.p; W.scan_tmp = S.scan_tmp.p; W.overflow = S.overflow.p;
  W.n_out = S.counts.p;
  // the last slot of the result mirror is free outside scanMatching: the sizes come back through it
  W.host_seg = &c->h_mirror_dev[kMirrorSlots - 1].w[0];
  W.host_seq = ++c->mirror_seq;
  W.use_ticket = (c->vox_ticket || (long long)(n + 256) / 256 > (long long)vox_emit_resident_blocks(c->device_cus)) ? 1 : 0;
  W.fault = c->h_fault_dev + kFaultVoxEmit;
  S.pending_seq = W.host_seq;
  *Jout = J;
  *Wout = W;
  return TLOAM_OK;
}
int submap_crop_voxel(tloam_ctx* c, const CropVoxelSeg seg[2], int nseg) {
  VoxelJob J;
  VoxelWork W;
  const int rc = submap_job(c, seg, nseg, &J, &W);
  if (rc != TLOAM_OK) return rc;
  launch_crop_voxel(J, W, c->stream);
  return TLOAM_OK;
}
int submap_upload(tloam_ctx* c, const double* xyz, size_t n) {  // host AoS -> in_aos (device)
  SubmapState& S = c->submap;
  HIPC(c, S.in_aos.reserve(3 * std::max<size_t>(n, 1)));
  if (n > 0) HIPC(c, hipMemcpyAsync(S.in_aos.p, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
  return TLOAM_OK;
}
int submap_finish(tloam_ctx* c, size_t* n_edge, size_t* n_ground) {  // the ONE host sync of an update
  SubmapState& S = c->submap;
  unsigned long long h[2] = {0, 0};
  int ov = 0;
  bool have = false;
  if (S.pending_seq) {  // the sizes arrive in pinned memory with the last kernel (everything before it has completed)
    const unsigned long long* seg = &c->h_mirror[kMirrorSlots - 1].w[0];
    unsigned long long pay[7];
    const int rc = wait_segment(c, seg, S.pending_seq, pay);
    if (rc < 0) return rc;
    if (rc == TLOAM_OK) { h[0] = pay[0]; h[1] = pay[1]; ov = (int)pay[2]; have = true; }
    S.pending_seq = 0ull;
  }
  if (!have) {
    HIPC(c, hipMemcpyAsync(h, S.counts.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipMemcpyAsync(&ov, S.overflow.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  if (tlh::check_device_faults(c) != TLOAM_OK) return TLOAM_E_HIP;   // (k_vox_emit's bounded look-back ran out: see there)
  if (ov) {
    c->last_error = "[VoxelDownSample] voxel_size is too small.";  // PointCloud2.cpp:370-372
    return TLOAM_E_INVALID;
  }
  *n_edge = (size_t)h[0];
  *n_ground = (size_t)h[1];
  return TLOAM_OK;
}
const double kNoLo[3] = {-INFINITY, -INFINITY, -INFINITY}, kNoHi[3] = {INFINITY, INFINITY, INFINITY};
}  // namespace

int tloam_submap_init(tloam_ctx* c, const tloam_submap_config* cfg, const double* planar, size_t n_planar,
                      const double* sphere, size_t n_sphere, const double* edge, size_t n_edge, const double* ground,
                      size_t n_ground) {
  if (!c || (n_planar && !planar) || (n_sphere && !sphere) || (n_edge && !edge) || (n_ground && !ground) ||
      n_planar > kMaxPoints || n_sphere > kMaxPoints || n_edge > kMaxPoints || n_ground > kMaxPoints)
    return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  SubmapState& S = c->submap;
  tloam_submap_config want;
  if (cfg) want = *cfg;
  else tloam_submap_default_config(&want);
  if (want.planar_frame_size < 1 || want.sphere_frame_size < 1 || !(want.edge_down_sample_submap > 0.0) ||
      !(want.ground_down_sample_submap > 0.0) || !(want.ground_down_sample > 0.0))
    return TLOAM_E_INVALID;  // "[VoxelDownSample] voxel_size <= 0." (PointCloud2.cpp:361-363); the submap in place stays
  S.release();
  S.cfg = want;
  // :286 / :290-291 submap += cloud on empty submaps: the clouds as given
  int rc = tloam_set_target(c, TLOAM_KIND_EDGE, edge, n_edge);
  if (rc == TLOAM_OK) rc = tloam_set_target(c, TLOAM_KIND_PLANAR, planar, n_planar);
  if (rc == TLOAM_OK) rc = tloam_set_target(c, TLOAM_KIND_SPHERE, sphere, n_sphere);
  if (rc != TLOAM_OK) return rc;
  // :287 ground += ground->VoxelDownSample(ground_down_sample)
  rc = submap_upload(c, ground, n_ground);
  if (rc != TLOAM_OK) return rc;
  const size_t m = std::max<size_t>(n_ground, 1);
  HIPC(c, S.wx.reserve(m)); HIPC(c, S.wy.reserve(m)); HIPC(c, S.wz.reserve(m));
  launch_aos_to_soa(S.in_aos.p, n_ground, S.wx.p, S.wy.p, S.wz.p, c->stream);
  rc = submap_reserve_work(c, n_ground);
  if (rc != TLOAM_OK) return rc;
  {  // one cloud; its size lands in counts[0]
    const CropVoxelSeg seg[2] = {{TLOAM_KIND_GROUND, n_ground, kNoLo, kNoHi, S.cfg.ground_down_sample},
                                 {TLOAM_KIND_GROUND, 0, kNoLo, kNoHi, S.cfg.ground_down_sample}};
    rc = submap_crop_voxel(c, seg, 1);
  }
  if (rc != TLOAM_OK) return rc;
  size_t ne = 0, ng = 0;
  rc = submap_finish(c, &ne, &ng);
  if (rc != TLOAM_OK) return rc;
  (void)ng;
  c->kd[TLOAM_KIND_GROUND].n_tgt = ne;  // (single-cloud job: counts[0])
  c->kd[TLOAM_KIND_GROUND].tgt_set = true;
  c->tgt_box_valid[TLOAM_KIND_GROUND] = false;
  c->grids_ahead = false; c->tgt_gen++;
  S.inited = true;
  return TLOAM_OK;
}

static int submap_update_body(tloam_ctx* c, const double pose[16], const double* planar, size_t n_planar,
                              const double* sphere, size_t n_sphere, const double* edge, size_t n_edge,
                              const double* ground, size_t n_ground);

int tloam_submap_update(tloam_ctx* c, const double pose[16], const double* planar, size_t n_planar,
                        const double* sphere, size_t n_sphere, const double* edge, size_t n_edge,
                        const double* ground, size_t n_ground) {
  if (!c || !pose || (n_planar && !planar) || (n_sphere && !sphere) || (n_edge && !edge) || (n_ground && !ground) ||
      n_planar > kMaxPoints || n_sphere > kMaxPoints || n_edge > kMaxPoints || n_ground > kMaxPoints)
    return TLOAM_E_INVALID;
  if (!c->submap.inited) return TLOAM_E_NOT_READY;
  // Open3D's Transform takes any 4x4 (front_end.cpp:246-247 hands it an Isometry3d's matrix: no orthogonality test anywhere on
  // this path), but a non-finite entry makes the crop box of :250-262 and every transformed point non-finite: a bad pose here
  // (the reference would go on with NaN clouds)
  for (int i = 0; i < 16; ++i)
    if (!(pose[i] - pose[i] == 0.0)) return TLOAM_E_BAD_POSE;
  HIPC(c, hipSetDevice(c->device));
  const int rc = submap_update_body(c, pose, planar, n_planar, sphere, n_sphere, edge, n_edge, ground, n_ground);
  // The host clouds are borrowed for the call only and are copied asynchronously; the success path ends in the one
  // synchronisation of submap_finish -- every error path must drain the stream before the buffers go back.
  if (rc != TLOAM_OK) (void)hipStreamSynchronize(c->stream);
  return rc;
}

static int submap_update_body(tloam_ctx* c, const double pose[16], const double* planar, size_t n_planar,
                              const double* sphere, size_t n_sphere, const double* edge, size_t n_edge,
                              const double* ground, size_t n_ground) {
  SubmapState& S = c->submap;
  (void)sphere;
  bool fused_front = false;
  for (int k = 0; k < kKinds; ++k) c->tgt_box_valid[k] = false;  // the targets are about to be rebuilt on the device
  c->grids_ahead = false; c->tgt_gen++;
  // :202-218 push the frame into both buffers, keep the newest *_frame_size
  // The three clouds the device needs (planar, edge, ground) are copied end to end into pinned staging (every cloud on a 16-byte
  // boundary) and the update's front launch reads them THERE, across PCIe, through LDS: the planar cloud is copied to the newest
  // ring frame's device buffer on the way (it is read again by the next planar_frame_size - 1 updates), the edge and ground clouds
  // are needed by this launch only.  No copy command: a hipMemcpyAsync costs the calling thread ~10 us and the copy engine about
  // as much before the first kernel can start, more than the update's kernels take.  (Three pageable copies: ~25 us each.)
  // (Measured against the staged upload, round 4: 0.093-0.097 against 0.102-0.106 ms per update.)  More ring frames than one front
  // launch takes: ONE asynchronous copy into the ring frame's buffer instead.
  const bool in_place = S.cfg.planar_frame_size <= transform_ring_max();
  const double* stage_view = nullptr;   // the staging half as the device sees it (in_place)
  size_t stage_off[3] = {0, 0, 0};      // first double of planar | edge | ground in the staged block
  int stage_half = -1;
  auto push = [&](std::vector<RingFrame*>& ring, const double* xyz, size_t n, int keep, bool upload) -> int {
    RingFrame* f = nullptr;
    if ((int)ring.size() >= keep) {  // recycle the frame that falls out
      f = ring.front();
      ring.erase(ring.begin());
      while ((int)ring.size() >= keep) { ring.front()->aos.release(); delete ring.front(); ring.erase(ring.begin()); }
    } else {
      f = new RingFrame();
    }
    ring.push_back(f);
    f->n = n;
    memcpy(f->pose, pose, sizeof(double) * 16);
    if (!upload) return TLOAM_OK;
    const double* parts[3] = {xyz, edge, ground};
    const size_t counts[3] = {3 * n, 3 * n_edge, 3 * n_ground};
    const size_t all = std::max<size_t>(tlh::staged_size(counts, 3), 2) + 2;   // (+ 2: read / written in 16-byte steps)
    if (f->aos.cap < all) HIPC(c, hipStreamSynchronize(c->stream));   // regrowth: nothing may be in flight
    HIPC(c, f->aos.reserve(all));
    if (in_place) {
      const int rc = tlh::stage_in_place(c, parts, counts, 3, stage_off, &stage_view, &stage_half);
      if (rc != TLOAM_E_NOT_READY) return rc;
      stage_view = nullptr;   // no device view of the pinned block on this system: the pieces are staged, copy them
      HIPC(c, hipMemcpyAsync(f->aos.p, c->h_stage[stage_half], sizeof(double) * tlh::staged_size(counts, 3), hipMemcpyHostToDevice, c->stream));
      return tlh::stage_release(c, stage_half, /*completed=*/false);
    }
    return tlh::stage_and_upload(c, parts, counts, 3, f->aos.p, stage_off);
  };
  // The sphere buffer is kept for its bookkeeping only (sizes, poses, frame count): nothing ever reads its points --
  // the sphere submap is rebuilt from the PLANAR buffer (front_end.cpp:221) -- so they are not uploaded.
  int rc = push(S.sphere_ring, nullptr, n_sphere, S.cfg.sphere_frame_size, /*upload=*/false);
  if (rc == TLOAM_OK) rc = push(S.planar_ring, planar, n_planar, S.cfg.planar_frame_size, /*upload=*/true);
  if (rc != TLOAM_OK) return rc;
  // :220-243 both submaps are rebuilt from submap_planar_buffer (the sphere loop iterates the PLANAR buffer)
  size_t total = 0;
  for (auto* f : S.planar_ring) total += f->n;
  {
    KindData& P = c->kd[TLOAM_KIND_PLANAR];
    KindData& Q = c->kd[TLOAM_KIND_SPHERE];
    const size_t m = std::max<size_t>(total, 1);
    if (P.tx.cap < m || Q.tx.cap < m) HIPC(c, hipStreamSynchronize(c->stream));  // regrowth: nothing may be in flight
    HIPC(c, P.tx.reserve(m)); HIPC(c, P.ty.reserve(m)); HIPC(c, P.tz.reserve(m));
    HIPC(c, Q.tx.reserve(m)); HIPC(c, Q.ty.reserve(m)); HIPC(c, Q.tz.reserve(m));
    if ((int)S.planar_ring.size() <= transform_ring_max()) {
      // all buffered frames, both submaps: part of the update's front launch below (k_submap_front)
      fused_front = true;
    } else {
      size_t off = 0;
      for (auto* f : S.planar_ring) {  // one launch per buffered frame writes both submaps
        launch_transform_to_soa2(f->aos.p, f->n, f->pose, P.tx.p + off, P.ty.p + off, P.tz.p + off, Q.tx.p + off,
                                 Q.ty.p + off, Q.tz.p + off, c->stream);
        off += f->n;
      }
    }
    P.n_tgt = Q.n_tgt = total;
    P.tgt_set = Q.tgt_set = true;
  }
  // :246-264 edge / ground: submap += scan->Transform(pose); Crop(pose.translation() +- L)->VoxelDownSample
  // Both clouds go through ONE launch sequence (two segments of one job, tl_common.hpp VoxelJob): half the launches
  // of two separate jobs -- the update is bound by the host's launch rate, not by the device.
  struct Acc { int kind; const double* xyz; size_t n; double L, voxel; };
  const Acc accs[2] = {{TLOAM_KIND_EDGE, edge, n_edge, S.cfg.edge_crop_box_length, S.cfg.edge_down_sample_submap},
                       {TLOAM_KIND_GROUND, ground, n_ground, S.cfg.ground_crop_box_length, S.cfg.ground_down_sample_submap}};
  size_t n_old[2], n_in[2], n_all = 0;
  for (int s = 0; s < 2; ++s) {
    const KindData& K = c->kd[accs[s].kind];
    n_old[s] = K.tgt_set ? K.n_tgt : 0;
    n_in[s] = n_old[s] + accs[s].n;
    n_all += n_in[s];
  }
  // (an accumulated cloud is a cloud: the bound of the entry points holds for what they add up to as well)
  if (n_in[0] > kMaxPoints || n_in[1] > kMaxPoints) {
    c->last_error = "submap update: an accumulated cloud would exceed the 2^28 points a cloud may hold";
    return TLOAM_E_INVALID;
  }
  {  // a buffer about to be regrown (hipFree) must not be in use by the kernels still in flight: synchronise
     // only then -- in steady state the capacities suffice and the update runs without a host wait
    const size_t m = std::max<size_t>(n_all, 1);
    bool grow = S.wx.cap < m || S.slot_of_pt.cap < m || S.keys.cap < voxel_table_size(m) + 1 || S.leader.cap < m + 1;
    for (int s = 0; s < 2; ++s) grow = grow || c->kd[accs[s].kind].tx.cap < std::max<size_t>(n_in[s], 1);
    if (grow) HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, S.wx.reserve(m)); HIPC(c, S.wy.reserve(m)); HIPC(c, S.wz.reserve(m));
    // The edge / ground submaps are INPUT (the old points, read by the assembly) and OUTPUT (the crop + voxel job writes the
    // new submap into the same arrays, sized for all n_in points): an array that has to grow keeps its old points.  (DBuf::reserve
    // does not; until round 5 the one-launch front read the old points through the pointer of a block that the job's
    // reserve had just freed -- right only as long as nobody else was handed that block in between.)
    for (int s = 0; s < 2; ++s) {
      KindData& K = c->kd[accs[s].kind];
      const size_t need = std::max<size_t>(n_in[s], 1);
      DBuf<double>* arr[3] = {&K.tx, &K.ty, &K.tz};
      for (DBuf<double>* b : arr) {
        if (need <= b->cap) continue;
        DBuf<double> nb;
        HIPC(c, nb.reserve(std::max(need, b->cap + b->cap / 2)));
        if (n_old[s] > 0 && b->p) {
          const hipError_t e = hipMemcpy(nb.p, b->p, sizeof(double) * n_old[s], hipMemcpyDeviceToDevice);
          if (e != hipSuccess) { nb.release(); c->last_error = hipGetErrorString(e); return TLOAM_E_HIP; }
        }
        b->release();
        *b = nb;
      }
    }
  }
  double lo[2][3], hi[2][3];
  {
    AssembleArgs A;
    size_t base = 0;
    for (int s = 0; s < 2; ++s) {
      const Acc& a = accs[s];
      KindData& K = c->kd[a.kind];
      // (staged with the planar cloud, behind it: in the pinned block itself, or uploaded to the newest ring frame's buffer)
      const double* stage = (stage_view ? stage_view : S.planar_ring.back()->aos.p) + stage_off[1 + s];
      A.ox[s] = K.tx.p; A.oy[s] = K.ty.p; A.oz[s] = K.tz.p;
      A.aos[s] = stage;
      A.n_old[s] = n_old[s]; A.n_new[s] = a.n; A.base[s] = base;
      for (int x = 0; x < 3; ++x) { lo[s][x] = pose[12 + x] - a.L; hi[s][x] = pose[12 + x] + a.L; }  // :250-254, :259-262
      base += n_in[s];
    }
    for (int i = 0; i < 16; ++i) A.M[i] = pose[i];
    const CropVoxelSeg seg[2] = {{accs[0].kind, n_in[0], lo[0], hi[0], accs[0].voxel},
                                 {accs[1].kind, n_in[1], lo[1], hi[1], accs[1].voxel}};
    if (fused_front) {
      // ONE launch for the planar ring, the assembly of both clouds and the head of the crop + voxel job (table emptied, min
      // bound): the update is front | insert | emit
      size_t ring_max = 0;
      const double* aos[16]; size_t nn[16]; const double* poses[16];
      int cnt = 0;
      for (auto* f : S.planar_ring) { aos[cnt] = f->aos.p; nn[cnt] = f->n; poses[cnt] = f->pose; ring_max = std::max(ring_max, f->n); ++cnt; }
      if (stage_view) aos[cnt - 1] = stage_view + stage_off[0];   // the newest frame: read in the staging, copied to f->aos on the way
      const size_t rows = submap_front_rows(ring_max, std::max(n_in[0], n_in[1]));
      if (S.min_partial.cap < rows * 6) HIPC(c, hipStreamSynchronize(c->stream));   // regrowth: nothing may be in flight
      HIPC(c, S.min_partial.reserve(rows * 6));
      VoxelJob J;
      VoxelWork W;
      rc = submap_job(c, seg, 2, &J, &W);
      if (rc != TLOAM_OK) return rc;
      KindData& P = c->kd[TLOAM_KIND_PLANAR];
      KindData& Q = c->kd[TLOAM_KIND_SPHERE];
      launch_submap_front(cnt, aos, nn, poses, A, J, W, P.tx.p, P.ty.p, P.tz.p, Q.tx.p, Q.ty.p, Q.tz.p, S.wx.p, S.wy.p, S.wz.p, c->stream,
                          cnt - 1, stage_view ? S.planar_ring.back()->aos.p : nullptr);
      launch_crop_voxel(J, W, c->stream, /*front_done=*/true);
    } else {
      launch_assemble(A, S.wx.p, S.wy.p, S.wz.p, c->stream);  // [old | Transform(new)] of both clouds, one launch
      rc = submap_crop_voxel(c, seg, 2);
      if (rc != TLOAM_OK) return rc;
    }
  }
  size_t ne = 0, ng = 0;
  rc = submap_finish(c, &ne, &ng);
  // (the update's last kernel has completed, or -- on an error -- the caller drains the stream: the staging half is free)
  if (stage_view) (void)tlh::stage_release(c, stage_half, /*completed=*/true);
  if (rc != TLOAM_OK) return rc;
  c->kd[TLOAM_KIND_EDGE].n_tgt = ne;
  c->kd[TLOAM_KIND_GROUND].n_tgt = ng;
  c->kd[TLOAM_KIND_EDGE].tgt_set = c->kd[TLOAM_KIND_GROUND].tgt_set = true;
  return TLOAM_OK;
}

int tloam_get_target(tloam_ctx* c, int kind, size_t capacity, size_t* n, double* xyz) {
  if (!c || kind < 0 || kind >= kKinds || !n) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  const KindData& K = c->kd[kind];
  *n = K.tgt_set ? K.n_tgt : 0;
  if (*n == 0) return TLOAM_OK;
  if (capacity < *n || !xyz) return TLOAM_E_INVALID;
  HIPC(c, c->misc.reserve(3 * *n));
  launch_soa_to_aos(K.tx.p, K.ty.p, K.tz.p, *n, c->misc.p, c->stream);
  HIPC(c, hipMemcpyAsync(xyz, c->misc.p, sizeof(double) * 3 * *n, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return TLOAM_OK;
}

}  // extern "C"
