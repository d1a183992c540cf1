// Compile-only check (CPU, this container): the shim's NATIVE-TYPES branch — the one a maintainer of the reference
// gets, with Eigen / Open3D / yaml-cpp present — against stand-in headers (tests/compile_native/stubs) that declare
// only what that branch and the reference's call sites use.  The calls below are the reference's, verbatim:
//   src/ErrorStateKF.cpp:126-130   guess from the filter state, icp_->align(*lidar.cloud, localMap, guess)
//   src/Odometry.cpp:61            localMap_->updateLocalMap(std::move(lidarMeasCopy->cloud), Eigen::Isometry3d::Identity())
//   src/Odometry.cpp:74,79,86      process(states, lidarMeas_); update(...); updateLocalMap(std::move(cloud), transform)
// It pins nothing numerically (the stand-ins compute nothing); it retires "this branch has never met a compiler".
#include <deque>
#include <memory>

#include "eskf_lio_shim/CloudPreprocessor.hpp"
#include "eskf_lio_shim/LocalMap.hpp"
#include "eskf_lio_shim/Registration.hpp"

#ifndef ESKF_LIO_SHIM_NATIVE_TYPES
#error "the native-types branch was not selected: the stand-in headers are not on the include path"
#endif

// State, LidarMeasurement, LidarMeasurementPtr: the reference's own include/ESKF_LIO/Types.hpp (on the include path
// of this check only; it is read where it lies, never copied)

Eigen::Isometry3d filter_update(  // the part of ErrorStateKF::update that touches the path (src/ErrorStateKF.cpp:126-130)
  const std::shared_ptr<ESKF_LIO::ICP> & icp_, const ESKF_LIO::LidarMeasurement & lidar,
  const ESKF_LIO::LocalMap & localMap, const Eigen::Quaterniond & attitude, const Eigen::Vector3d & position)
{
  Eigen::Isometry3d guess;
  guess.linear() = attitude.toRotationMatrix();
  guess.translation() = position;
  auto observation = icp_->align(*lidar.cloud, localMap, guess);
  return observation;
}

void odometry_frame(  // src/Odometry.cpp:61,86
  const std::shared_ptr<ESKF_LIO::LocalMap> & localMap_, ESKF_LIO::LidarMeasurementPtr lidarMeas_,
  const Eigen::Isometry3d & transform, bool first)
{
  auto lidarMeasCopy = lidarMeas_;
  lidarMeas_ = nullptr;
  if (first) {
    localMap_->updateLocalMap(std::move(lidarMeasCopy->cloud), Eigen::Isometry3d::Identity());
  } else {
    localMap_->updateLocalMap(std::move(lidarMeasCopy->cloud), transform);
  }
}

void preprocess_frame(  // src/Odometry.cpp:58-60,74: cloudPreprocessor_->process(states, lidarMeas_)
  const std::shared_ptr<ESKF_LIO::CloudPreprocessor> & cloudPreprocessor_, const std::deque<ESKF_LIO::State> & states,
  ESKF_LIO::LidarMeasurementPtr lidarMeas_)
{
  cloudPreprocessor_->process({}, lidarMeas_);
  cloudPreprocessor_->process(states, lidarMeas_);
}

std::shared_ptr<ESKF_LIO::CloudPreprocessor> make_preprocessor(const YAML::Node & config)  // src/Odometry.cpp:11
{
  return std::make_shared<ESKF_LIO::CloudPreprocessor>(config);
}

std::shared_ptr<ESKF_LIO::ICP> make_icp(const YAML::Node & config)  // src/ErrorStateKF.cpp:9
{
  return std::make_shared<ESKF_LIO::ICP>(config);
}

std::shared_ptr<ESKF_LIO::LocalMap> make_map(const YAML::Node & config)  // src/Odometry.cpp:12-16
{
  open3d::camera::PinholeCameraParameters visualizerConfig;
  return std::make_shared<ESKF_LIO::LocalMap>(config, visualizerConfig, false);
}
