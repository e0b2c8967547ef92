/* submap_oracle.c -- CPU restatement of the reference's submap maintenance.
 *
 * TEST INFRASTRUCTURE ONLY (see tloam_oracle.h): the checker for tloam_submap_init / tloam_submap_update.
 * "parity unpinned": the reference holds no test or golden vector for this path, and Open3D 0.12 (whose
 * TransformPoints / AxisAlignedBoundingBox the reference's PointCloud2 calls) is not under /root/reference;
 * the arithmetic below is restated from the reference's own PointCloud2.cpp and the cited upstream behaviour.
 *
 * Follows
 *   FrontEnd::updateSubmap                       src/front_end/front_end.cpp:201-275
 *   first-frame branch of updateLidarOdometry    src/front_end/front_end.cpp:283-304
 *   PointCloud2::Transform                       src/open3d/PointCloud2.cpp:71-75  (Open3D TransformPoints:
 *                                                new = T * (x,y,z,1); p = new.head<3>() / new(3))
 *   PointCloud2::operator+=                      src/open3d/PointCloud2.cpp:96-132
 *   PointCloud2::Crop(AxisAlignedBoundingBox)    src/open3d/PointCloud2.cpp:551-559 (inclusive bounds, index order)
 *   PointCloud2::VoxelDownSample                 src/open3d/PointCloud2.cpp:358-403, AccumulatedPoint :246-291
 *
 * VoxelDownSample's output order is std::unordered_map iteration order (unspecified); this restatement emits
 * voxels in order of first occurrence.  Quirk kept: the sphere submap is rebuilt from the PLANAR frame buffer
 * (front_end.cpp:221 `for (auto& sphere_frame : submap_planar_buffer)`).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tloam_hip.h"

/* ---- PointCloud2::Transform -------------------------------------------------------------------- */
void orc_pc_transform(const double M[16], double* xyz, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    double r[4];
    for (int a = 0; a < 4; ++a) r[a] = ((M[a] * x + M[4 + a] * y) + M[8 + a] * z) + M[12 + a] * 1.0;
    xyz[3 * i] = r[0] / r[3];
    xyz[3 * i + 1] = r[1] / r[3];
    xyz[3 * i + 2] = r[2] / r[3];
  }
}

/* ---- PointCloud2::Crop: indices with min <= p <= max, order kept -------------------------------- */
size_t orc_pc_crop(const double* in, size_t n, const double lo[3], const double hi[3], double* out) {
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    const double* p = in + 3 * i;
    if (p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2]) {
      out[3 * m] = p[0]; out[3 * m + 1] = p[1]; out[3 * m + 2] = p[2];
      ++m;
    }
  }
  return m;
}

/* ---- PointCloud2::VoxelDownSample --------------------------------------------------------------- */
typedef struct { int32_t k[3]; int32_t used; int32_t out; int32_t num; double s[3]; } vox_cell;

static uint64_t vox_hash(const int32_t k[3]) {
  uint64_t x = (uint64_t)(uint32_t)k[0] * 0x9E3779B97F4A7C15ull;
  x ^= (uint64_t)(uint32_t)k[1] * 0xC2B2AE3D27D4EB4Full + (x << 6) + (x >> 2);
  x ^= (uint64_t)(uint32_t)k[2] * 0x165667B19E3779F9ull + (x << 6) + (x >> 2);
  x ^= x >> 29;
  return x;
}

/* returns the number of output points, -1 for "voxel_size <= 0" / "voxel_size is too small" (:361-372) */
long orc_pc_voxel_down_sample(const double* in, size_t n, double voxel, double* out) {
  if (!(voxel > 0.0)) return -1;
  /* GetMinBound / GetMaxBound: elementwise; an empty cloud has (0,0,0) */
  double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const double v = in[3 * i + a];
      if (i == 0 || v < mn[a]) mn[a] = v;
      if (i == 0 || v > mx[a]) mx[a] = v;
    }
  double vmin[3], vmax[3], ext = 0.0;
  for (int a = 0; a < 3; ++a) {
    vmin[a] = mn[a] - voxel * 0.5;  /* :366 */
    vmax[a] = mx[a] + voxel * 0.5;  /* :367 */
    if (vmax[a] - vmin[a] > ext) ext = vmax[a] - vmin[a];
  }
  if (voxel * 2147483647.0 < ext) return -1; /* :368-372 */
  size_t cap = 1024;
  while (cap < 2 * n + 2) cap <<= 1;
  vox_cell* tab = (vox_cell*)calloc(cap, sizeof(vox_cell));
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (n + 1)); /* table slot of the j-th voxel seen */
  long nout = 0;
  for (size_t i = 0; i < n; ++i) {
    int32_t k[3];
    for (int a = 0; a < 3; ++a) k[a] = (int32_t)floor((in[3 * i + a] - vmin[a]) / voxel); /* :380-383 */
    size_t h = (size_t)(vox_hash(k) & (cap - 1));
    while (tab[h].used && (tab[h].k[0] != k[0] || tab[h].k[1] != k[1] || tab[h].k[2] != k[2])) h = (h + 1) & (cap - 1);
    if (!tab[h].used) {
      tab[h].used = 1;
      tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2];
      tab[h].out = (int32_t)nout;
      order[nout++] = (int32_t)h;
    }
    /* AccumulatedPoint::AddPoint (:253-269): point_ += p, in index order */
    tab[h].s[0] += in[3 * i]; tab[h].s[1] += in[3 * i + 1]; tab[h].s[2] += in[3 * i + 2];
    tab[h].num += 1;
  }
  for (long j = 0; j < nout; ++j) { /* GetAveragePoint (:271-273): point_ / double(num) */
    const vox_cell* c = &tab[order[j]];
    const double dn = (double)c->num;
    out[3 * j] = c->s[0] / dn; out[3 * j + 1] = c->s[1] / dn; out[3 * j + 2] = c->s[2] / dn;
  }
  free(order);
  free(tab);
  return nout;
}

/* ---- the submap object of FrontEnd ---------------------------------------------------------------- */
#define ORC_MAX_RING 64
typedef struct { double* xyz; size_t n; double pose[16]; } orc_frame;
typedef struct orc_submap {
  tloam_submap_config cfg;
  orc_frame planar_ring[ORC_MAX_RING], sphere_ring[ORC_MAX_RING];
  int n_planar_ring, n_sphere_ring;
  double* cloud[4]; /* submap.{planar,ground,edge,sphere}_feature, kinds as TLOAM_KIND_* */
  size_t n[4];
  int inited;
} orc_submap;

static void set_copy(double** dst, size_t* dn, const double* src, size_t n) {
  free(*dst);
  *dst = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  if (n) memcpy(*dst, src, sizeof(double) * 3 * n);
  *dn = n;
}
static void ring_push(orc_frame* ring, int* count, int keep, const double pose[16], const double* xyz, size_t n) {
  /* emplace_back, then pop_front while size > keep (front_end.cpp:205-218) */
  orc_frame f;
  f.xyz = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  if (n) memcpy(f.xyz, xyz, sizeof(double) * 3 * n);
  f.n = n;
  memcpy(f.pose, pose, sizeof(double) * 16);
  ring[(*count)++] = f;
  while (*count > keep) {
    free(ring[0].xyz);
    memmove(ring, ring + 1, sizeof(orc_frame) * (size_t)(*count - 1));
    --(*count);
  }
}

orc_submap* orc_submap_create(const tloam_submap_config* cfg) {
  orc_submap* s = (orc_submap*)calloc(1, sizeof(orc_submap));
  if (cfg) s->cfg = *cfg;
  else {
    s->cfg.planar_frame_size = 3; s->cfg.sphere_frame_size = 3;
    s->cfg.edge_crop_box_length = 100.0; s->cfg.ground_crop_box_length = 100.0;
    s->cfg.edge_down_sample_submap = 0.3; s->cfg.ground_down_sample_submap = 0.45; s->cfg.ground_down_sample = 0.3;
  }
  return s;
}
void orc_submap_destroy(orc_submap* s) {
  if (!s) return;
  for (int i = 0; i < s->n_planar_ring; ++i) free(s->planar_ring[i].xyz);
  for (int i = 0; i < s->n_sphere_ring; ++i) free(s->sphere_ring[i].xyz);
  for (int k = 0; k < 4; ++k) free(s->cloud[k]);
  free(s);
}

/* front_end.cpp:283-304 */
int orc_submap_init(orc_submap* s, const double* planar, size_t n_planar, const double* sphere, size_t n_sphere,
                    const double* edge, size_t n_edge, const double* ground, size_t n_ground) {
  if (!s || s->cfg.planar_frame_size < 1 || s->cfg.sphere_frame_size < 1 || s->cfg.planar_frame_size >= ORC_MAX_RING ||
      s->cfg.sphere_frame_size >= ORC_MAX_RING)
    return TLOAM_E_INVALID;
  set_copy(&s->cloud[TLOAM_KIND_EDGE], &s->n[TLOAM_KIND_EDGE], edge, n_edge);          /* :286 */
  double* g = (double*)malloc(sizeof(double) * 3 * (n_ground ? n_ground : 1));
  const long ng = orc_pc_voxel_down_sample(ground, n_ground, s->cfg.ground_down_sample, g); /* :287 */
  if (ng < 0) { free(g); return TLOAM_E_INVALID; }
  set_copy(&s->cloud[TLOAM_KIND_GROUND], &s->n[TLOAM_KIND_GROUND], g, (size_t)ng);
  free(g);
  set_copy(&s->cloud[TLOAM_KIND_PLANAR], &s->n[TLOAM_KIND_PLANAR], planar, n_planar);  /* :290 */
  set_copy(&s->cloud[TLOAM_KIND_SPHERE], &s->n[TLOAM_KIND_SPHERE], sphere, n_sphere);  /* :291 */
  s->inited = 1;
  return TLOAM_OK;
}

static int accumulate_crop_voxel(orc_submap* s, int kind, const double pose[16], const double* scan, size_t n_scan,
                                 double L, double voxel) {
  /* *submap += scan->Transform(pose)  (front_end.cpp:246-247) */
  const size_t n_old = s->n[kind], n_in = n_old + n_scan;
  double* all = (double*)malloc(sizeof(double) * 3 * (n_in ? n_in : 1));
  if (n_old) memcpy(all, s->cloud[kind], sizeof(double) * 3 * n_old);
  if (n_scan) memcpy(all + 3 * n_old, scan, sizeof(double) * 3 * n_scan);
  orc_pc_transform(pose, all + 3 * n_old, n_scan);
  /* crop box = lidar_odom_pose.translation() +- L  (:250-254 / :259-262) */
  double lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = pose[12 + a] - L; hi[a] = pose[12 + a] + L; }
  double* cropped = (double*)malloc(sizeof(double) * 3 * (n_in ? n_in : 1));
  const size_t nc = orc_pc_crop(all, n_in, lo, hi, cropped);
  double* out = (double*)malloc(sizeof(double) * 3 * (nc ? nc : 1));
  const long nv = orc_pc_voxel_down_sample(cropped, nc, voxel, out); /* :257 / :264 */
  free(all);
  free(cropped);
  if (nv < 0) { free(out); return TLOAM_E_INVALID; }
  free(s->cloud[kind]);
  s->cloud[kind] = out;
  s->n[kind] = (size_t)nv;
  return TLOAM_OK;
}

/* front_end.cpp:201-275 */
int orc_submap_update(orc_submap* s, const double pose[16], const double* planar, size_t n_planar,
                      const double* sphere, size_t n_sphere, const double* edge, size_t n_edge, const double* ground,
                      size_t n_ground) {
  if (!s || !s->inited) return TLOAM_E_NOT_READY;
  ring_push(s->sphere_ring, &s->n_sphere_ring, s->cfg.sphere_frame_size, pose, sphere, n_sphere);
  ring_push(s->planar_ring, &s->n_planar_ring, s->cfg.planar_frame_size, pose, planar, n_planar);
  /* :220-243 -- BOTH loops run over submap_planar_buffer */
  size_t total = 0;
  for (int i = 0; i < s->n_planar_ring; ++i) total += s->planar_ring[i].n;
  const int kinds[2] = {TLOAM_KIND_SPHERE, TLOAM_KIND_PLANAR};
  for (int q = 0; q < 2; ++q) {
    double* all = (double*)malloc(sizeof(double) * 3 * (total ? total : 1));
    size_t off = 0;
    for (int i = 0; i < s->n_planar_ring; ++i) {
      const orc_frame* f = &s->planar_ring[i];
      if (f->n) memcpy(all + 3 * off, f->xyz, sizeof(double) * 3 * f->n);
      orc_pc_transform(f->pose, all + 3 * off, f->n);
      off += f->n;
    }
    free(s->cloud[kinds[q]]);
    s->cloud[kinds[q]] = all;
    s->n[kinds[q]] = total;
  }
  int rc = accumulate_crop_voxel(s, TLOAM_KIND_EDGE, pose, edge, n_edge, s->cfg.edge_crop_box_length,
                                 s->cfg.edge_down_sample_submap);
  if (rc != TLOAM_OK) return rc;
  return accumulate_crop_voxel(s, TLOAM_KIND_GROUND, pose, ground, n_ground, s->cfg.ground_crop_box_length,
                               s->cfg.ground_down_sample_submap);
}

int orc_submap_get(const orc_submap* s, int kind, size_t capacity, size_t* n, double* xyz) {
  if (!s || kind < 0 || kind >= 4 || !n) return TLOAM_E_INVALID;
  *n = s->n[kind];
  if (*n == 0) return TLOAM_OK;
  if (capacity < *n || !xyz) return TLOAM_E_INVALID;
  memcpy(xyz, s->cloud[kind], sizeof(double) * 3 * *n);
  return TLOAM_OK;
}
