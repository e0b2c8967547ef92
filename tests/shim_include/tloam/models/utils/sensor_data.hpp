// TEST-ONLY compile shim (see ../../../README.md).  The real header drags in open3d_to_ros.hpp (Open3D + ROS); what
// registration_interface.hpp needs from it is the cloud type, whose first data member is the point vector
// (reference include/tloam/open3d/PointCloud2.hpp:396: `std::vector<Eigen::Vector3d> points_;`).
#pragma once
#include <memory>
#include <utility>
#include <vector>
#include <Eigen/Dense>
namespace open3d {
namespace geometry {
class PointCloud2 {
 public:
  std::vector<Eigen::Vector3d> points_;
};
}  // namespace geometry
}  // namespace open3d
