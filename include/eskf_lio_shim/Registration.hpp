// Registration.hpp — drop-in for the reference's include/ESKF_LIO/Registration.hpp +
// src/Registration.cpp: class ESKF_LIO::ICP with the same constructor keys and the same
//   Eigen::Isometry3d align(const PointCloud & cloud, const LocalMap & localMap,
//                           const Eigen::Isometry3d & guess)
// (reference include/ESKF_LIO/Registration.hpp:23-32), so src/ErrorStateKF.cpp:9,130 compiles
// against it unchanged.  align() forwards the raw buffers of the cloud to vgicp_align(): the whole
// loop of src/Registration.cpp:7-35 then runs on the MI355X (see include/vgicp_hip.h).
// Behaviour kept: inputs are not modified; non-convergence only prints "ICP not converged!"
// (src/Registration.cpp:30-32) and still returns the pose.  Behaviour changed on purpose: the
// reference's converged_ member is sticky across calls (Registration.hpp:50, set at
// Registration.cpp:23 and never reset), so after the first converged frame its message can never
// print again; here convergence is per call and readable through lastStats().  A HIP / argument
// error throws std::runtime_error (the reference has no error path at all).
#ifndef ESKF_LIO_SHIM_REGISTRATION_HPP_
#define ESKF_LIO_SHIM_REGISTRATION_HPP_

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "LocalMap.hpp"

namespace ESKF_LIO
{

// The keys ICP's YAML constructor reads (config/hilti_config.yaml:50-53).
struct RegistrationConfig
{
  int maxIteration = 100;
  double translationSquaredThreshold = 1.0e-6;
  double cosineThreshold = 0.9999;
  int chunkIterations = 0;  // vgicp_params.chunk_iterations; 0 = library default
};

class ICP
{
public:
  using PointVector = typename std::vector<Vector3d>;
  using CovarianceVector = typename std::vector<Matrix3d>;
  using Correspondence = typename std::tuple<PointVector, CovarianceVector, PointVector,
      CovarianceVector>;

  struct Stats
  {
    int iterations = 0;
    bool converged = false;
    double seconds = 0.0;
    double deviceSeconds = 0.0;
    std::vector<uint64_t> correspondenceCounts;
  };

  explicit ICP(const RegistrationConfig & config)
  : maxIteration_(config.maxIteration)
    , translationSquaredThreshold_(config.translationSquaredThreshold)
    , cosineThreshold_(config.cosineThreshold)
    , chunkIterations_(config.chunkIterations)
  {
  }

#if defined(ESKF_LIO_SHIM_HAVE_YAML)
  ICP(const YAML::Node & config)
  : maxIteration_(config["registration"]["max_iteration"].as<int>())
    , translationSquaredThreshold_(config["registration"]["translation_sq_threshold"].as<double>())
    , cosineThreshold_(config["registration"]["cosine_threshold"].as<double>())
  {
  }
#endif

  Isometry3d align(const PointCloud & cloud, const LocalMap & localMap, const Isometry3d & guess)
  {
    vgicp_ctx * ctx = localMap.context();
    vgicp_params params{};
    params.max_iteration = maxIteration_;
    params.chunk_iterations = chunkIterations_;
    params.translation_sq_threshold = translationSquaredThreshold_;
    params.cosine_threshold = cosineThreshold_;
    std::vector<uint64_t> counts(static_cast<size_t>(maxIteration_ > 0 ? maxIteration_ : 1), 0);
    vgicp_stats stats{};
    stats.corr_count = counts.data();
    double pose[16];
    // The cloud CloudPreprocessor::process just prepared is resident on the device already (src/Odometry.cpp:74 ->
    // src/ErrorStateKF.cpp:130 hand it over untouched): no second upload of its 96 bytes per point, and the frame's
    // one synchronisation is this call's.  Any other cloud — or that one after somebody changed it — goes up as it is.
    int rc = VGICP_OK;
    shim::ResidentStamp * resident = nullptr;
    bool done = false;
    {
      shim::TraceScope ts(shim::Trace::AlignVerify);
      resident = shim::residentIdentityOf(ctx, cloud);
    }
    shim::HashCrew & crew = shim::HashCrew::instance();
    if (resident && !resident->sampled && crew.helpers() > 0 &&
      cloud.points_.size() * (sizeof(Vector3d) + sizeof(Matrix3d)) >= (256u << 10))
    {
      // The full-hash check of a large cloud and the registration at the same time: helper threads read the host cloud
      // while this thread waits for the device, which registers its own copy.  The result counts only if the hash then
      // matches the stamp; a cloud edited since is registered again below from the host data, as the reference would
      // (vgicp_align_resident changes neither the resident scan nor the map).
      shim::FullHashJob job;
      shim::planFullHash(cloud, job, crew.helpers());
      crew.begin(job.chunks, job.count);
      {
        shim::TraceScope tsCall(shim::Trace::AlignCall);
        rc = vgicp_align_resident(ctx, shim::poseData(guess), &params, pose, &stats);
      }
      {
        shim::TraceScope ts(shim::Trace::AlignVerify);
        crew.finish();
        done = shim::foldFullHash(job) == resident->hash;
      }
      if (!done) {resident = nullptr;}
    } else if (resident) {
      shim::TraceScope ts(shim::Trace::AlignVerify);
      if (resident->hash != shim::sampleHash(cloud, resident->sampled)) {resident = nullptr;}
    }
    shim::TraceScope tsCall(shim::Trace::AlignCall);
    if (done) {
      lastUsedResidentScan_ = true;
    } else if (resident) {
      rc = vgicp_align_resident(ctx, shim::poseData(guess), &params, pose, &stats);
      lastUsedResidentScan_ = true;
    } else {
      const size_t n = cloud.points_.size();
      if (cloud.covariances_.size() != n) {
        throw std::runtime_error(
                "ICP::align: the cloud has " + std::to_string(n) + " points but " +
                std::to_string(cloud.covariances_.size()) + " covariances (a cloud prepared with a deferred host "
                "copy and changed since? call shim::materialize first)");
      }
      const double * pts = n ? cloud.points_.data()->data() : nullptr;
      const double * covs = n ? cloud.covariances_.data()->data() : nullptr;
      rc = vgicp_align(ctx, n, pts, covs, shim::poseData(guess), &params, pose, &stats);
      lastUsedResidentScan_ = false;
    }
    if (rc != VGICP_OK && rc != VGICP_ERR_DEGENERATE) {shim::check(ctx, rc, "vgicp_align");}

    lastStats_.iterations = stats.iterations;
    lastStats_.converged = stats.converged != 0;
    lastStats_.seconds = stats.seconds;
    lastStats_.deviceSeconds = stats.device_seconds;
    counts.resize(static_cast<size_t>(stats.iterations));
    lastStats_.correspondenceCounts = counts;
    if (!lastStats_.converged) {
      std::cout << "ICP not converged!\n";
    }
    return shim::poseFromData(pose);
  }

  const Stats & lastStats() const {return lastStats_;}
  // whether the last align() found its cloud resident on the device (prepared there by CloudPreprocessor::process)
  bool lastUsedResidentScan() const {return lastUsedResidentScan_;}

private:
  ICP() = delete;

  int maxIteration_;
  double translationSquaredThreshold_;
  double cosineThreshold_;
  int chunkIterations_ = 0;
  Stats lastStats_;
  bool lastUsedResidentScan_ = false;
};

}  // namespace ESKF_LIO

#endif  // ESKF_LIO_SHIM_REGISTRATION_HPP_
