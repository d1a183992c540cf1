// STAND-IN, NOT Open3D (see ../Eigen/Dense): the members of open3d::geometry::PointCloud the path touches
// (reference include/ESKF_LIO/Types.hpp:11, src/Registration.cpp:11-13, src/LocalMap.cpp:15).
#pragma once
#include <Eigen/Dense>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
namespace open3d {
namespace geometry {
class PointCloud {
 public:
  std::vector<Eigen::Vector3d> points_;
  std::vector<Eigen::Vector3d> normals_;
  std::vector<Eigen::Vector3d> colors_;
  std::vector<Eigen::Matrix3d> covariances_;
  PointCloud& Transform(const Eigen::Matrix4d&) { return *this; }
  bool HasCovariances() const { return !points_.empty() && covariances_.size() == points_.size(); }
};
}  // namespace geometry
namespace camera {
class PinholeCameraParameters {};
}  // namespace camera
namespace utility {
template <typename T>
struct hash_eigen {
  std::size_t operator()(const T& m) const {
    std::size_t seed = 0;
    for (int i = 0; i < 3; ++i) seed ^= std::hash<int>()(m(i)) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};
}  // namespace utility
}  // namespace open3d
