"""CPU: the CONCRETE plugin class `tloam::HipRegistration` (adapters/hip_registration.hpp, the part behind
`#ifdef TLOAM_REGISTRATION_INTERFACE_HPP`) compiled against the reference's REAL, UNMODIFIED interface header
(/root/reference/include/tloam/models/registration/registration_interface.hpp:40-48) and driven the way
`FrontEnd::initRegistraton` would drive it (front_end.cpp:155-167 with the branch INTEGRATION.md section 1 adds):
make_shared through the base pointer, the four virtuals, the 16 `TLS:` keys of the reference's own
config/mapping/lidar_odometry.yaml:23-39 read by `fromYaml` like `LocalRegistration::initConfig` reads them
(registration.cpp:212-230).

Eigen / Open3D / yaml-cpp are not in the image: tests/shim_include/ declares just the types the two headers name, with
the layouts the adapter relies on (README.md there; a compile check, not a build of the reference).  `-Wall -Wextra
-Werror=overloaded-virtual -Werror=suggest-override`: a signature that drifted from the interface would not override
and fails the build.  Skips where /root/reference is absent (it never travels to the GPU box)."""
import os
import re
import subprocess
import tempfile

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
IFACE = os.path.join(REF, "include", "tloam", "models", "registration", "registration_interface.hpp")
YAML_PATH = os.path.join(REF, "config", "mapping", "lidar_odometry.yaml")

pytestmark = pytest.mark.skipif(not os.path.exists(IFACE), reason="the reference tree is not on this machine")

SRC = r'''
#include <cstdio>
#include <memory>
#include <string>
#include <yaml-cpp/yaml.h>
#include "tloam/models/registration/registration_interface.hpp"      // the reference's own header, unmodified
#include "hip_registration.hpp"                                      // this repo: adapters/hip_registration.hpp

#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
using namespace tloam;

static_assert(sizeof(Eigen::Vector3d) == 24, "PointCloud2::points_ is read as packed xyz doubles");
static_assert(std::is_abstract<RegistrationInterface>::value, "the reference's interface");
static_assert(std::is_base_of<RegistrationInterface, HipRegistration>::value, "plugin derives from the reference's interface");
static_assert(!std::is_abstract<HipRegistration>::value, "every pure virtual of registration_interface.hpp:44-47 is overridden");
static_assert(std::is_constructible<HipRegistration, const YAML::Node&>::value, "same constructor argument as LocalRegistration");

// FrontEnd::initRegistraton (front_end.cpp:155-167) with the branch of INTEGRATION.md section 1; the "TLS" branch itself
// needs LocalRegistration (Ceres + Open3D) and is what cannot be compiled here
bool initRegistraton(std::shared_ptr<RegistrationInterface> &registration_ptr_, const YAML::Node &config_node) {
  std::string registration_method = config_node["local_registration_method"].as<std::string>();
  if (registration_method == "TLS_HIP") {
    registration_ptr_ = std::make_shared<HipRegistration>(config_node["TLS"]);
  } else {
    ROS_ERROR("Other methods are not yet supported\n");
    return false;
  }
  return true;
}

int main() {
  YAML::Node config_node;
  config_node["local_registration_method"] = std::string("TLS_HIP");
@ASSIGN@
  const tloam_tls_config c = HipRegistration::fromYaml(static_cast<const YAML::Node&>(config_node)["TLS"]);
  std::printf("cfg k_corr=%d factor_num=%d edge_dist_thres=%.17g edge_dir_thres=%.17g edge_maxnum=%d sphere_dist_thres=%.17g "
              "sphere_maxnum=%d planar_dist_thres=%.17g planar_maxnum=%d ground_dist_thres=%.17g ground_maxnum=%d "
              "max_iterations=%d cost_threshold=%.17g gnc_factor=%.17g noise_bound=%.17g fitness_thres=%.17g\n",
              c.k_corr, c.factor_num, c.edge_dist_thres, c.edge_dir_thres, c.edge_maxnum, c.sphere_dist_thres, c.sphere_maxnum,
              c.planar_dist_thres, c.planar_maxnum, c.ground_dist_thres, c.ground_maxnum, c.max_iterations, c.cost_threshold,
              c.gnc_factor, c.noise_bound, c.fitness_thres);

  std::shared_ptr<RegistrationInterface> local_registration_ptr_;
  if (!initRegistraton(local_registration_ptr_, config_node)) return 2;
  YAML::Node other;
  other["local_registration_method"] = std::string("NDT");
  std::shared_ptr<RegistrationInterface> none;
  if (initRegistraton(none, other) || none) return 3;                // any other string: "not yet supported", false

  Frame current_scan, local_map, result_frame;                       // the reference's Frame (registration_interface.hpp:19-38)
  for (auto* cl : {current_scan.edge_feature.get(), current_scan.sphere_feature.get(), current_scan.planar_feature.get(),
                   current_scan.ground_feature.get(), local_map.edge_feature.get(), local_map.sphere_feature.get(),
                   local_map.planar_feature.get(), local_map.ground_feature.get()})
    for (int i = 0; i < 32; ++i) cl->points_.push_back(Eigen::Vector3d(0.1 * i, 0.05 * (i % 7), 0.02 * (i % 3)));
  Eigen::Isometry3d predict_pose = Eigen::Isometry3d::Identity(), result_pose = Eigen::Isometry3d::Identity();
  // the four virtual calls through the BASE pointer, in the front end's order (front_end.cpp:267/:294, :314, :320-322)
  const bool t = local_registration_ptr_->setInputTarget(local_map);
  const bool s = local_registration_ptr_->setInputSource(current_scan);
  const bool m = local_registration_ptr_->scanMatching(result_frame, predict_pose, result_pose);
  const std::pair<double, double> fit = local_registration_ptr_->getFitnessScore();
  auto* hip = dynamic_cast<HipRegistration*>(local_registration_ptr_.get());
  const bool have_gpu = hip && hip->core().valid();
  // the reference's default behaviour at registration.cpp:884-886, opt-in: a fresh Eigen::Vector3d::Random() per call (the
  // identity prediction above takes that branch); and the builders' few-factor diagnostics (:500-502, :554-556, :630-632,
  // :773-775), which the 32-point clouds of this frame trigger
  hip->setReferenceRandomOmega(true);
  const bool m2 = local_registration_ptr_->scanMatching(result_frame, predict_pose, result_pose);
  hip->setReferenceRandomOmega(false);
  hip->core().setFewFactorWarnings(false);
  const bool m3 = local_registration_ptr_->scanMatching(result_frame, predict_pose, result_pose);
  if (have_gpu && !(m2 && m3)) return 4;
  HipRegistration drawn(static_cast<const YAML::Node&>(config_node)["TLS"], 0, /*reference_random_omega=*/true);
  std::printf("gpu=%d target=%d source=%d match=%d fitness=%g,%g\n", (int)have_gpu, (int)t, (int)s, (int)m, fit.first, fit.second);
  local_registration_ptr_.reset();                                   // virtual destructor through the base
  // no device: every call reports failure (there is no CPU fallback); with one they all succeed
  return (have_gpu ? (t && s && m) : (!t && !s && !m && fit.first == 0.0)) ? 0 : 1;
}
'''


def test_concrete_plugin_compiles_against_the_reference_interface_and_runs_through_the_base_pointer():
    tls = yaml.safe_load(open(YAML_PATH))["TLS"]          # the reference's own TLS block
    assert len(tls) == 16
    assign = "\n".join(f'  config_node["TLS"]["{k}"] = {("%d" % v) if isinstance(v, int) else repr(float(v))};' for k, v in tls.items())
    lib_dir = os.path.join(ROOT, "tloam_amd")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC.replace("@ASSIGN@", assign))
        inc = ["-I", os.path.join(ROOT, "tests", "shim_include"),       # in FRONT of the reference: its sensor_data.hpp needs Open3D + ROS
               "-I", os.path.join(REF, "include"), "-I", os.path.join(ROOT, "adapters")]
        cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror=overloaded-virtual", "-Werror=suggest-override", *inc, src, "-o", exe,
               "-L", lib_dir, "-l:libtloam_hip.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
        cc = subprocess.run(cmd, capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr
        # the interface header that was compiled is the reference's file, not a copy: the preprocessor says where it came from
        pp = subprocess.run(["g++", "-std=c++17", "-E", *inc, src], capture_output=True, text=True)
        assert f'"{IFACE}"' in pp.stdout
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = dict(re.findall(r"(\w+)=([-+.\w]+)", out.stdout.splitlines()[0]))
    for k, v in tls.items():                                # fromYaml read every key of the reference's block
        assert float(got[k]) == float(v), (k, got[k], v)
    assert "target=" in out.stdout
