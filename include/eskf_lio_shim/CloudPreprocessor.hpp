// CloudPreprocessor.hpp — drop-in for the scan-preparation half of the reference's
// include/ESKF_LIO/CloudPreprocessor.hpp + src/CloudPreprocessor.cpp:
//   void voxelDownsampleAndEstimateCovariances(PointCloud & cloud) const
// (reference include/ESKF_LIO/CloudPreprocessor.hpp:35-36, src/CloudPreprocessor.cpp:76-127) with the
// same effect on the cloud: points_ becomes one point per occupied voxel (the voxel's first point in
// scan order), covariances_ their regularised 30-neighbour covariances.  The KD-tree build, the k-NN
// searches, Open3D's ComputeCovariance and the JacobiSVD all run on the MI355X behind
// vgicp_preprocess() (include/vgicp_hip.h).
// Behaviour kept: cloud.points_ / cloud.covariances_ are replaced in place; the neighbour count is
// KDTreeSearchParamKNN's default (30); fewer than 3 neighbours give the regularised identity.
// Behaviour changed on purpose: the output is in ascending scan order — the reference's order is
// the iteration order of an unordered_map (src/CloudPreprocessor.cpp:96-101), which no caller may
// rely on.  deskew() (src/CloudPreprocessor.cpp:25-74, row N4 of SURVEY.md 8(f)) stays on the host
// and is not part of this header.
#ifndef ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_
#define ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_

#include <cstdint>
#include <vector>

#include "LocalMap.hpp"

namespace ESKF_LIO
{

// The key CloudPreprocessor's YAML constructor reads for this half (config/hilti_config.yaml:
// cloud_preprocessor.voxel_size).
struct CloudPreprocessorConfig
{
  double voxelSize = 0.3;
  int knn = 30;  // open3d::geometry::KDTreeSearchParamKNN's default
};

class CloudPreprocessor
{
public:
  explicit CloudPreprocessor(const CloudPreprocessorConfig & config, vgicp_ctx * ctx = nullptr)
  : voxelSize_(config.voxelSize), knn_(config.knn), ctx_(ctx ? ctx : shim::defaultContext())
  {
  }

#if defined(ESKF_LIO_SHIM_HAVE_YAML)
  CloudPreprocessor(const YAML::Node & config)
  : voxelSize_(config["cloud_preprocessor"]["voxel_size"].as<double>()), knn_(30),
    ctx_(shim::defaultContext())
  {
  }
#endif

  void voxelDownsampleAndEstimateCovariances(PointCloud & cloud) const
  {
    auto & points = cloud.points_;
    auto & covariances = cloud.covariances_;
    const size_t n = points.size();
    std::vector<Vector3d> pointsDown(n);
    std::vector<Matrix3d> covsDown(n);
    size_t kept = 0;
    static_assert(sizeof(Vector3d) == 3 * sizeof(double), "points must be packed xyz triples");
    static_assert(sizeof(Matrix3d) == 9 * sizeof(double), "covariances must be packed 3x3 blocks");
    shim::check(
      ctx_,
      vgicp_preprocess(
        ctx_, n, n ? reinterpret_cast<const double *>(points.data()) : nullptr, voxelSize_, knn_, n,
        reinterpret_cast<double *>(pointsDown.data()), reinterpret_cast<double *>(covsDown.data()),
        nullptr, &kept),
      "vgicp_preprocess");
    pointsDown.resize(kept);
    covsDown.resize(kept);
    std::swap(points, pointsDown);
    std::swap(covariances, covsDown);
  }

  double voxelSize() const {return voxelSize_;}

private:
  CloudPreprocessor() = delete;

  double voxelSize_;
  int knn_;
  vgicp_ctx * ctx_;
};
}  // namespace ESKF_LIO

#endif  // ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_
