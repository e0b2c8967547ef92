// tl_api_feature.hip -- C ABI of the PCA feature extraction (include/tloam_hip.h: tloam_pca_info,
// tloam_extract_planar_sphere): featureExtract::calculatePCAInfo / extractPlanarSphere
// (feature_extract.cpp:47-122, :133-197) driven on the device (kernels in tl_feature.hip).
#include "tl_ctx.hpp"

using namespace tl;

extern "C" {

// ---- PCA feature extraction (feature_extract.cpp:47-122, :133-197) --------------------------------
void tloam_feature_default_config(tloam_feature_config* cfg) {
  if (!cfg) return;
  cfg->radius = 0.2; cfg->K = 20; cfg->min_neigh = 10; cfg->planar_num = 500; cfg->sphere_num = 300;
  cfg->cvr_scan = 0.25; cfg->cvr_submap = 0.15; cfg->planar_scan_thres = 0.75; cfg->planar_submap_thres = 0.65;
  cfg->planar_vertic_thres = 0.25;
}

namespace {
// calculatePCAInfo on the device; the per-point arrays stay in F
int feature_pca(tloam_ctx* c, const tloam_feature_config& cfg, const double* xyz, size_t n, FeatBuffers& F, FeatArgs* out) {
  if (cfg.K < 3 || cfg.K > 20 || !(cfg.radius >= 0.0)) return TLOAM_E_INVALID;  // assert(r_ >= 0.0 && K_ >= 3) :55
  const size_t m = std::max<size_t>(n, 1);
  HIPC(c, F.aos.reserve(3 * m)); HIPC(c, F.x.reserve(m)); HIPC(c, F.y.reserve(m)); HIPC(c, F.z.reserve(m));
  HIPC(c, F.flatness.reserve(m)); HIPC(c, F.cvr.reserve(m)); HIPC(c, F.sphericity.reserve(m)); HIPC(c, F.normal.reserve(3 * m));
  HIPC(c, F.num_sum.reserve(m)); HIPC(c, F.neigh.reserve(m * (size_t)cfg.K));
  if (n > 0) {
    // (the borrowed cloud straight through the copy command: for 2.4 MB that beats pinned staging + copy kernel, 0.425 against
    //  0.478 ms per call, three interleaved rounds -- the staging pays below ~1 MB, where the command's fixed cost dominates)
    HIPC(c, hipMemcpyAsync(F.aos.p, xyz, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
  }
  GridView views[kKinds];
  // (the reference's assert admits r_ == 0: SearchHybrid then finds nobody -- not even the point itself, its squared distance 0
  //  is not below 0 -- and every point comes out with zero neighbours.  The grid still needs cells of SOME size: the walk's own
  //  radius test does the rest)
  double radii[kKinds] = {cfg.radius > 0.0 ? cfg.radius : 1.0, 0, 0, 0};
  CloudRef clouds[kKinds] = {{F.x.p, F.y.p, F.z.p, n}, {nullptr, nullptr, nullptr, 0}, {nullptr, nullptr, nullptr, 0},
                             {nullptr, nullptr, nullptr, 0}};
  double boxes[kKinds][6];
  const double (*known)[6] = nullptr;
  if (n > 0) {
    // AoS -> SoA and the cloud's bounds in ONE launch, the rows straight into pinned memory (k_ingest_targets, as
    // tloam_set_target_frame): the grid is sized without a bounds launch of its own
    IngestArgs I;
    memset(&I, 0, sizeof(I));
    I.aos[0] = F.aos.p; I.x[0] = F.x.p; I.y[0] = F.y.p; I.z[0] = F.z.p; I.n[0] = (int)n;
    launch_ingest_targets(I, c->h_bbox_dev, c->stream);
    HIPC(c, hipStreamSynchronize(c->stream));
    tlh::reduce_box_rows(c->h_bbox, boxes);
    known = boxes;
  }
  int rc = build_grids_over(c, F.grid, radii, clouds, views, known);  // KDTreeFlann::SetGeometry(cloud) :57
  if (rc != TLOAM_OK) return rc;
  FeatArgs A;
  A.g = views[0];
  A.x = F.x.p; A.y = F.y.p; A.z = F.z.p;
  A.n = (int)n;
  A.radius = cfg.radius; A.K = cfg.K; A.min_neigh = cfg.min_neigh;
  A.flatness = F.flatness.p; A.cvr = F.cvr.p; A.sphericity = F.sphericity.p; A.normal = F.normal.p;
  A.num_sum = F.num_sum.p; A.neigh = F.neigh.p;
  if (n > 0 && A.g.n <= 0) {
    // no search structure: the cloud holds no finite point (build_grids_over).  Every point then has no neighbour -- nanoflann's
    // result set never admits a NaN / infinite distance -- and keeps what calculatePCAInfo initialises: zeros, indices -1.
    // (The PCA pass takes its points in the GRID's order: without a grid it would write nothing at all.)
    HIPC(c, hipMemsetAsync(F.flatness.p, 0, sizeof(double) * n, c->stream)); HIPC(c, hipMemsetAsync(F.cvr.p, 0, sizeof(double) * n, c->stream));
    HIPC(c, hipMemsetAsync(F.sphericity.p, 0, sizeof(double) * n, c->stream)); HIPC(c, hipMemsetAsync(F.normal.p, 0, sizeof(double) * 3 * n, c->stream));
    HIPC(c, hipMemsetAsync(F.num_sum.p, 0, sizeof(int) * n, c->stream));
    HIPC(c, hipMemsetAsync(F.neigh.p, 0xff, sizeof(int) * n * (size_t)cfg.K, c->stream));
  } else {
    launch_pca_info(A, c->stream);
  }
  *out = A;
  return TLOAM_OK;
}
}  // namespace

int tloam_pca_info(tloam_ctx* c, const tloam_feature_config* cfg, const double* xyz, size_t n, double* flatness,
                   double* cvr, double* sphericity, double* normal, int32_t* num_sum, int32_t* neigh) {
  if (!c || !cfg || (n > 0 && !xyz) || n > kMaxPoints) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  FeatBuffers& F = c->feat;
  FeatArgs A;
  int rc = feature_pca(c, *cfg, xyz, n, F, &A);
  if (rc == TLOAM_OK && n > 0) {
    hipError_t e = hipSuccess;
    auto get = [&](void* dst, const void* src, size_t bytes) {
      if (dst && e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    };
    get(flatness, F.flatness.p, sizeof(double) * n); get(cvr, F.cvr.p, sizeof(double) * n);
    get(sphericity, F.sphericity.p, sizeof(double) * n); get(normal, F.normal.p, sizeof(double) * 3 * n);
    get(num_sum, F.num_sum.p, sizeof(int) * n); get(neigh, F.neigh.p, sizeof(int) * n * (size_t)cfg->K);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->last_error = hipGetErrorString(e); rc = TLOAM_E_HIP; }
  }
  (void)hipStreamSynchronize(c->stream);
  if (rc == TLOAM_OK) rc = tlh::check_device_faults(c);   // (the grid build's single-pass scan: bounded look-back)
  return rc;
}

int tloam_extract_planar_sphere(tloam_ctx* c, const tloam_feature_config* cfg, const double* xyz, size_t n,
                                int32_t* planar_scan, size_t* n_ps, int32_t* planar_submap, size_t* n_pm,
                                int32_t* sphere_scan, size_t* n_ss, int32_t* sphere_submap, size_t* n_sm) {
  if (!c || !cfg || (n > 0 && !xyz) || n > kMaxPoints || !n_ps || !n_pm || !n_ss || !n_sm) return TLOAM_E_INVALID;
  *n_ps = *n_pm = *n_ss = *n_sm = 0;
  if (n == 0) return TLOAM_OK;  // "cloud_in_ does not contain points" (:50-53): the lists stay empty
  if (!planar_scan || !planar_submap || !sphere_scan || !sphere_submap) return TLOAM_E_INVALID;
  HIPC(c, hipSetDevice(c->device));
  FeatBuffers& F = c->feat;
  FeatArgs A;
  int rc = feature_pca(c, *cfg, xyz, n, F, &A);
  std::vector<double> blob;   // the ranked lists as k_rank_final packs them: flatness planar | sphere, then the indices
  size_t np = 0, ns = 0;
  unsigned long long total = 0;
  if (rc == TLOAM_OK) {
    hipError_t e = hipSuccess;
    if ((e = F.flags.reserve(n + 1)) == hipSuccess && (e = F.scan.reserve(n + 1)) == hipSuccess &&
        (e = F.scan_tmp.reserve(scan_tmp_elems(n + 1))) == hipSuccess && (e = F.f2.reserve(2 * n)) == hipSuccess &&
        (e = F.gf.reserve(2 * n)) == hipSuccess && (e = F.out.reserve(3 * n + 2)) == hipSuccess &&
        (e = F.idx2.reserve(2 * n)) == hipSuccess && (e = F.gi.reserve(2 * n)) == hipSuccess &&
        (e = F.bkt.reserve(2 * n)) == hipSuccess && (e = F.pos.reserve(2 * n)) == hipSuccess &&
        (e = F.rank_ctl.reserve(1)) == hipSuccess) {
      const FeatSelect S{cfg->cvr_submap, cfg->planar_submap_thres, cfg->planar_vertic_thres};
      launch_feat_select(A, S, F.flags.p, F.scan.p, F.scan_tmp.p, F.f2.p, F.idx2.p, F.rank_ctl.p, F.bkt.p, F.pos.p, F.gf.p, F.gi.p,
                         F.out.p, c->stream);
      e = hipMemcpyAsync(&total, F.scan.p + n, sizeof(total), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      np = (size_t)(total >> 32); ns = (size_t)(total & 0xffffffffull);
      // ONE copy for both ranked lists: (np + ns) doubles, then (np + ns) ints
      blob.resize(np + ns + (np + ns + 1) / 2);
      if (e == hipSuccess && np + ns) e = hipMemcpy(blob.data(), F.out.p, sizeof(double) * (np + ns) + sizeof(int) * (np + ns), hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) { c->last_error = hipGetErrorString(e); rc = TLOAM_E_HIP; }
  }
  (void)hipStreamSynchronize(c->stream);
  if (rc == TLOAM_OK) rc = tlh::check_device_faults(c);
  if (rc != TLOAM_OK) return rc;
  // :178-190 on the ranked lists
  const double* pf = blob.data();
  const double* sf = pf + np;
  const int* pidx = reinterpret_cast<const int*>(blob.data() + np + ns);
  for (size_t id = 0; id < np; ++id) {
    if (id < (size_t)std::max(cfg->planar_num, 0) || pf[id] > cfg->planar_scan_thres) planar_scan[(*n_ps)++] = pidx[id];
    planar_submap[(*n_pm)++] = pidx[id];
  }
  for (size_t id = 0; id < ns; ++id) {  // the RANK is stored, not the point index (:186, :188)
    if (id < (size_t)std::max(cfg->sphere_num, 0) || sf[id] > cfg->cvr_scan) sphere_scan[(*n_ss)++] = (int32_t)id;
    sphere_submap[(*n_sm)++] = (int32_t)id;
  }
  return TLOAM_OK;
}

}  // extern "C"
