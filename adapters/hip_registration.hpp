// hip_registration.hpp -- drop-in tloam::RegistrationInterface implementation backed by the MI355X
// library (include/tloam_hip.h).  Header-only; link with -ltloam_hip.
//
// Replaces tloam::LocalRegistration (reference src/models/registration/registration.cpp) behind the same
// plugin boundary (include/tloam/models/registration/registration_interface.hpp:40-48):
//
//     bool setInputSource(Frame&)                      registration.cpp:232-239
//     bool setInputTarget(Frame&)                      registration.cpp:241-248
//     bool scanMatching(Frame&, Isometry3d&, Isometry3d&)   registration.cpp:879-1133
//     std::pair<double,double> getFitnessScore()       registration.cpp:257-296
//
// Selection mirrors FrontEnd::initRegistraton (front_end.cpp:155-167): add
//     else if (method == "TLS_HIP") local_registration_ptr_ = std::make_shared<HipRegistration>(config_node["TLS"]);
// (INTEGRATION.md has the full patch).
//
// The marshalling core is a template over the frame / pose types so that it can be compiled and tested
// without Eigen, Open3D or ROS: it only needs
//     frame.<kind>_feature->points_   : contiguous array of 3 doubles per point  (PointCloud2.hpp:396)
//     pose.matrix().data()            : 16 doubles, column-major                 (Eigen::Isometry3d)
// When the reference's registration_interface.hpp has been included first, the concrete
// tloam::HipRegistration : RegistrationInterface is defined at the bottom of this file.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>

#include "../include/tloam_hip.h"

namespace tloam_hip {

// error convention of the reference: every method returns true; failures are asserts / ROS_WARN
// (SURVEY 8(b)).  Here a failing status is logged and mapped to false / (0,0).
inline bool ok(int status, const char* what, const tloam_ctx* ctx) {
  if (status == TLOAM_OK) return true;
  std::fprintf(stderr, "[tloam_hip] %s: %s %s\n", what, tloam_status_string(status), ctx ? tloam_last_error(ctx) : "");
  return false;
}

// PointsAccessor<Cloud>::data(cloud) -> const double* (AoS xyz), ::size(cloud) -> number of points.
// The default fits open3d::geometry::PointCloud2 (std::vector<Eigen::Vector3d> points_).
template <class Cloud>
struct PointsAccessor {
  // the reinterpret_casts below read `points_` as packed xyz doubles: true of std::vector<Eigen::Vector3d> (24 bytes per
  // point, no padding; PointCloud2.hpp:396) -- checked at compile time for whatever point type the cloud carries
  using Point = typename std::remove_cv<typename std::remove_reference<decltype(std::declval<const Cloud&>().points_[0])>::type>::type;
  static_assert(sizeof(Point) == 3 * sizeof(double), "points_ must be 24 bytes per point: packed double[3] (Eigen::Vector3d)");
  static const double* data(const Cloud& c) { return c.points_.empty() ? nullptr : reinterpret_cast<const double*>(c.points_.data()); }
  static double* mutable_data(Cloud& c) { return c.points_.empty() ? nullptr : reinterpret_cast<double*>(c.points_.data()); }
  static std::size_t size(const Cloud& c) { return c.points_.size(); }
};

template <class FrameT, class PoseT>
class HipRegistrationCore {
 public:
  explicit HipRegistrationCore(const tloam_tls_config& cfg, int device_id = 0) : cfg_(cfg) {
    status_ = tloam_create(&cfg, device_id, &ctx_);
    ok(status_, "tloam_create", nullptr);
  }
  ~HipRegistrationCore() { tloam_destroy(ctx_); }
  HipRegistrationCore(const HipRegistrationCore&) = delete;
  HipRegistrationCore& operator=(const HipRegistrationCore&) = delete;

  bool valid() const { return ctx_ != nullptr; }
  tloam_ctx* context() { return ctx_; }
  const tloam_stats& lastStats() const { return stats_; }

  // The unit vector the reference draws with Eigen::Vector3d::Random() when the predicted rotation is below 1e-2 rad
  // (registration.cpp:884-886).  Default: none -> the library uses (0,0,1), deterministic.  A host that wants the
  // reference's behaviour passes Eigen::Vector3d::Random().normalized().data() before each scanMatching.
  void setOmegaPerturbation(const double* unit3_or_null) {
    have_omega_ = unit3_or_null != nullptr;
    if (have_omega_) for (int i = 0; i < 3; ++i) omega_[i] = unit3_or_null[i];
  }

  // The reference's builders end with a diagnostic when they added few factors (registration.cpp:500-502 edge, :554-556 sphere,
  // :630-632 planar, :773-775 ground; ROS_WARN, the planar one std::cout).  The adapter repeats them after scanMatching from what
  // the solve reports -- same thresholds, same texts, on stderr (no ROS in this header) -- for the LAST outer iteration's factor
  // set (the reference prints them in every outer iteration: up to four times per frame).  On by default, like the reference.
  void setFewFactorWarnings(bool on) { warn_few_ = on; }

  bool setInputSource(FrameT& f) { return upload(f, /*source=*/true); }
  bool setInputTarget(FrameT& f) { return upload(f, /*source=*/false); }

  bool scanMatching(FrameT& out_result, PoseT& predict_pose, PoseT& result_pose) {
    if (!ctx_) return false;
    auto& scan = *out_result.scan_cloud;
    using Acc = PointsAccessor<typename std::remove_reference<decltype(scan)>::type>;
    double result[16];
    const int rc = tloam_scan_match(ctx_, predict_pose.matrix().data(), have_omega_ ? omega_ : nullptr, result,
                                    Acc::mutable_data(scan), Acc::size(scan), &stats_);
    if (rc == TLOAM_E_WEIGHT_RANGE) {
      // the reference's assert at registration.cpp:871 (its build sets no NDEBUG) would have aborted the node; the
      // solve itself ran to the end with the weights as computed and the result is written: warn and carry on
      std::fprintf(stderr, "[tloam_hip] scanMatching: %d GNC weight(s) outside [0,1] (registration.cpp:871)\n",
                   (int)stats_.weight_range_violations);
    } else if (!ok(rc, "tloam_scan_match", ctx_)) {
      return false;
    }
    for (int i = 0; i < 16; ++i) result_pose.matrix().data()[i] = result[i];  // registration.cpp:1124
    if (warn_few_) warnFewFactors();
    return true;
  }

  // registration.cpp:500-502, :554-556, :630-632, :773-775 -- the builders' own thresholds and texts.  The sphere builder tests
  // `sphere_sum`, which counts the SOURCE POINTS it looked at, not the factors it added (:551), and returns past the
  // warning once the cap is reached (:538): it warns exactly when the sphere cloud has <= 10 points.  The edge builder runs for
  // factor_num >= 3, the sphere builder for factor_num == 4 (:979-1016).
  void warnFewFactors() const {
    if (stats_.n_corr[TLOAM_KIND_EDGE] <= 20 && cfg_.factor_num >= 3) std::fprintf(stderr, "[ WARN] not enough edge points !!!\n");
    if (n_src_[TLOAM_KIND_SPHERE] <= 10 && cfg_.factor_num >= 4) std::fprintf(stderr, "[ WARN] not enough sphere point..\n");
    if (stats_.n_corr[TLOAM_KIND_PLANAR] < 20 && cfg_.factor_num >= 2) std::fprintf(stdout, "not enough ground points\n");   // (sic, :631: std::cout)
    if (stats_.n_corr[TLOAM_KIND_GROUND] <= 20 && cfg_.factor_num >= 2) std::fprintf(stderr, "[ WARN] not enough ground point..\n");
  }

  // ---- device-resident submap (optional; replaces the body of FrontEnd::updateSubmap, front_end.cpp:201-275,
  //      and of the first-frame branch :283-304).  The clouds are the ones the reference builds on the CPU
  //      anyway -- SelectByIndex(planar/sphere_submap_index), current_scan.edge/ground_feature -- and the
  //      resulting submap becomes the registration target without coming back to the host.
  template <class CloudPtr>
  bool submapInit(const tloam_submap_config& cfg, const CloudPtr& planar_submap, const CloudPtr& sphere_submap,
                  const CloudPtr& edge, const CloudPtr& ground) {
    using Acc = PointsAccessor<typename std::remove_reference<decltype(*planar_submap)>::type>;
    return ctx_ && ok(tloam_submap_init(ctx_, &cfg, Acc::data(*planar_submap), Acc::size(*planar_submap),
                                        Acc::data(*sphere_submap), Acc::size(*sphere_submap), Acc::data(*edge),
                                        Acc::size(*edge), Acc::data(*ground), Acc::size(*ground)),
                      "tloam_submap_init", ctx_);
  }
  template <class CloudPtr>
  bool submapUpdate(PoseT& lidar_odom_pose, const CloudPtr& planar_submap, const CloudPtr& sphere_submap,
                    const CloudPtr& edge_scan, const CloudPtr& ground_scan) {
    using Acc = PointsAccessor<typename std::remove_reference<decltype(*planar_submap)>::type>;
    return ctx_ && ok(tloam_submap_update(ctx_, lidar_odom_pose.matrix().data(), Acc::data(*planar_submap),
                                          Acc::size(*planar_submap), Acc::data(*sphere_submap), Acc::size(*sphere_submap),
                                          Acc::data(*edge_scan), Acc::size(*edge_scan), Acc::data(*ground_scan),
                                          Acc::size(*ground_scan)),
                      "tloam_submap_update", ctx_);
  }

  // ---- PCA feature extraction (optional; featureExtract::extractPlanarSphere, feature_extract.cpp:133-197): the four
  //      index vectors of the reference's signature, filled from the device lists.
  template <class Cloud>
  bool extractPlanarSphere(const tloam_feature_config& cfg, const Cloud& cloud, std::vector<std::size_t>& planar_scan_index,
                           std::vector<std::size_t>& planar_submap_index, std::vector<std::size_t>& sphere_scan_index,
                           std::vector<std::size_t>& sphere_submap_index) {
    if (!ctx_) return false;
    using Acc = PointsAccessor<Cloud>;
    const std::size_t n = Acc::size(cloud);
    std::vector<std::int32_t> buf(4 * (n ? n : 1));
    std::size_t cnt[4] = {0, 0, 0, 0};
    const int rc = tloam_extract_planar_sphere(ctx_, &cfg, Acc::data(cloud), n, buf.data(), &cnt[0], buf.data() + n, &cnt[1],
                                               buf.data() + 2 * n, &cnt[2], buf.data() + 3 * n, &cnt[3]);
    if (!ok(rc, "tloam_extract_planar_sphere", ctx_)) return false;
    std::vector<std::size_t>* out[4] = {&planar_scan_index, &planar_submap_index, &sphere_scan_index, &sphere_submap_index};
    for (int l = 0; l < 4; ++l) {
      out[l]->clear();  // the reference appends to vectors it has just swapped empty (front_end.cpp:299-302)
      out[l]->reserve(cnt[l]);
      for (std::size_t i = 0; i < cnt[l]; ++i) out[l]->push_back(static_cast<std::size_t>(buf[l * n + i]));
    }
    return true;
  }

  std::pair<double, double> getFitnessScore() {
    double fitness = 0.0, rmse = 0.0;
    if (!ctx_ || !ok(tloam_fitness(ctx_, &fitness, &rmse), "tloam_fitness", ctx_)) return {0.0, 0.0};
    return {fitness, rmse};
  }

 private:
  bool upload(FrameT& f, bool source) {
    if (!ctx_) return false;
    const double* ptr[4];
    size_t cnt[4];
    auto put = [&](int kind, const auto& cloud_ptr) {
      using Cloud = typename std::remove_reference<decltype(*cloud_ptr)>::type;
      ptr[kind] = PointsAccessor<Cloud>::data(*cloud_ptr);
      cnt[kind] = PointsAccessor<Cloud>::size(*cloud_ptr);
    };
    put(TLOAM_KIND_PLANAR, f.planar_feature);   // registration.cpp:233-236 / :242-245
    put(TLOAM_KIND_GROUND, f.ground_feature);
    put(TLOAM_KIND_EDGE, f.edge_feature);
    put(TLOAM_KIND_SPHERE, f.sphere_feature);
    if (source) for (int k = 0; k < 4; ++k) n_src_[k] = cnt[k];
    // the four clouds of the Frame in one call: one host synchronisation per frame
    return source ? ok(tloam_set_source_frame(ctx_, ptr, cnt), "tloam_set_source_frame", ctx_)
                  : ok(tloam_set_target_frame(ctx_, ptr, cnt), "tloam_set_target_frame", ctx_);
  }

  tloam_ctx* ctx_ = nullptr;
  tloam_tls_config cfg_;
  int status_ = TLOAM_OK;
  tloam_stats stats_{};
  std::size_t n_src_[4] = {0, 0, 0, 0};
  bool warn_few_ = true;
  bool have_omega_ = false;
  double omega_[3] = {0.0, 0.0, 1.0};
};

}  // namespace tloam_hip

// ---- concrete plugin, only where the reference's interface (Eigen + Open3D + yaml-cpp) is available ----
#ifdef TLOAM_REGISTRATION_INTERFACE_HPP
#include <yaml-cpp/yaml.h>
namespace tloam {
class HipRegistration : public RegistrationInterface {
 public:
  // same constructor argument as LocalRegistration (registration.cpp:182-206, initConfig :212-230)
  // (reference_random_omega: see setReferenceRandomOmega)
  explicit HipRegistration(const YAML::Node& node, int device_id = 0, bool reference_random_omega = false)
      : core_(fromYaml(node), device_id), reference_random_omega_(reference_random_omega) {}
  bool setInputSource(Frame& f) override { return core_.setInputSource(f); }
  bool setInputTarget(Frame& f) override { return core_.setInputTarget(f); }
  bool scanMatching(Frame& out, Eigen::Isometry3d& predict, Eigen::Isometry3d& result) override {
    if (reference_random_omega_) {
      // registration.cpp:884-886: when the predicted rotation is below 1e-2 rad the reference replaces it by
      // Eigen::Vector3d::Random().normalized() * 1e-4 -- a fresh draw per call, from Eigen's (std::rand) generator.  The library
      // uses the direction only when that branch is taken.
      const Eigen::Vector3d u = Eigen::Vector3d::Random().normalized();
      core_.setOmegaPerturbation(u.data());
    }
    return core_.scanMatching(out, predict, result);
  }
  // opt-in: draw the omega perturbation as the reference does (non-deterministic, like the reference).  Off: the library's
  // fixed (0, 0, 1) -- or whatever core().setOmegaPerturbation was given -- and two runs over the same input agree bit for bit.
  void setReferenceRandomOmega(bool on) {
    reference_random_omega_ = on;
    if (!on) core_.setOmegaPerturbation(nullptr);
  }
  std::pair<double, double> getFitnessScore() override { return core_.getFitnessScore(); }
  // beyond the interface: the device-resident submap entry points
  tloam_hip::HipRegistrationCore<Frame, Eigen::Isometry3d>& core() { return core_; }

  // the 16 keys of the `TLS:` block exactly as LocalRegistration::initConfig reads them (registration.cpp:212-230)
  static tloam_tls_config fromYaml(const YAML::Node& n) {
    tloam_tls_config c;
    tloam_default_config(&c);
    c.k_corr = n["k_corr"].as<int>();
    c.factor_num = n["factor_num"].as<int>();
    c.edge_dist_thres = n["edge_dist_thres"].as<double>();
    c.sphere_dist_thres = n["sphere_dist_thres"].as<double>();
    c.planar_dist_thres = n["planar_dist_thres"].as<double>();
    c.ground_dist_thres = n["ground_dist_thres"].as<double>();
    c.edge_dir_thres = n["edge_dir_thres"].as<double>();
    c.edge_maxnum = n["edge_maxnum"].as<int>();
    c.sphere_maxnum = n["sphere_maxnum"].as<int>();
    c.planar_maxnum = n["planar_maxnum"].as<int>();
    c.ground_maxnum = n["ground_maxnum"].as<int>();
    c.max_iterations = n["max_iterations"].as<int>();
    c.cost_threshold = n["cost_threshold"].as<double>();
    c.gnc_factor = n["gnc_factor"].as<double>();
    c.noise_bound = n["noise_bound"].as<double>();
    c.fitness_thres = n["fitness_thres"].as<double>();
    return c;
  }

 private:
  tloam_hip::HipRegistrationCore<Frame, Eigen::Isometry3d> core_;
  bool reference_random_omega_ = false;
};
}  // namespace tloam
#endif
