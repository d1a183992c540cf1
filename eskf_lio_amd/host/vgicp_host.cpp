// vgicp_host.cpp — builds the C++ host mirror (include/eskf_lio_shim/: ESKF_LIO::LocalMap,
// ESKF_LIO::ICP and ESKF_LIO::CloudPreprocessor with the reference's method signatures) into libvgicp_host.so and gives it a small
// C surface so the Python parity tests can drive the SAME C++ objects a patched Odometry.cpp would
// (reference call sites: src/Odometry.cpp:61,86, src/ErrorStateKF.cpp:130).
// Compiled with plain g++ against the dependency-free stand-in types (no Eigen / Open3D here); all
// compute goes through libvgicp_hip.so.
#define ESKF_LIO_SHIM_FORCE_POD 1
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../include/eskf_lio_shim/CloudPreprocessor.hpp"
#include "../../include/eskf_lio_shim/Registration.hpp"
#include "../csrc/vgicp_math.h"

using ESKF_LIO::CloudPreprocessor;
using ESKF_LIO::ICP;
using ESKF_LIO::Isometry3d;
using ESKF_LIO::LocalMap;
using ESKF_LIO::PointCloud;

namespace
{
thread_local std::string g_error;

PointCloud makeCloud(size_t n, const double * points, const double * covs)
{
  PointCloud cloud;
  cloud.points_.resize(n);
  cloud.covariances_.resize(n);
  if (n) {
    std::memcpy(cloud.points_.data(), points, n * 24);
    std::memcpy(cloud.covariances_.data(), covs, n * 72);
  }
  return cloud;
}

template<typename F>
int guarded(F && f)
{
  try {
    f();
    return 0;
  } catch (const std::exception & e) {
    g_error = e.what();
    return 1;
  }
}
}  // namespace

extern "C" {

const char * host_last_error(void) {return g_error.c_str();}

// LocalMap(config) — the YAML constructor's keys as plain arguments.
LocalMap * host_localmap_create_config(
  double voxel_size, size_t max_points_per_voxel, double translation_sq_threshold,
  double cosine_threshold, int remove_distant_points, double distance_threshold,
  double remove_period, int device_resident, int keep_raw_points)
{
  LocalMap * out = nullptr;
  guarded(
    [&] {
      ESKF_LIO::LocalMapConfig c;
      c.voxelSize = voxel_size;
      c.maxNumPointsPerVoxel = max_points_per_voxel;
      c.translationSquaredThreshold = translation_sq_threshold;
      c.cosineThreshold = cosine_threshold;
      c.removeDistantPoints = remove_distant_points != 0;
      c.distanceThreshold = distance_threshold;
      c.removePeriod = remove_period;
      c.deviceResident = device_resident != 0;
      c.keepRawPoints = keep_raw_points != 0;
      out = new LocalMap(c);
    });
  return out;
}

// LocalMap(double voxelSize, size_t maxNumPointsPerVoxel, bool visualize = false)
LocalMap * host_localmap_create(double voxel_size, size_t max_points_per_voxel)
{
  LocalMap * out = nullptr;
  guarded([&] {out = new LocalMap(voxel_size, max_points_per_voxel);});
  return out;
}

void host_localmap_destroy(LocalMap * map) {delete map;}
// helper threads of the drop-in's full-hash check (CloudPreprocessorConfig::residentCheckThreads; 0 = the caller alone)
void host_hash_helpers(int n) {ESKF_LIO::shim::HashCrew::instance().setHelpers(n);}
size_t host_localmap_size(const LocalMap * map) {return map->size();}
// waits until the shadow grid's worker has filed every cloud handed to it (what LocalMap::grid() / save() do first);
// returns the host grid's voxel count
size_t host_localmap_drain(const LocalMap * map) {return map->grid().size();}

// updateLocalMap(cloud, transform, initialize); the transformed cloud is written back to
// points/covs, as the reference mutates the shared cloud in place (src/LocalMap.cpp:15).
int host_localmap_update(
  LocalMap * map, size_t n, double * points, double * covs, const double transform[16],
  int initialize)
{
  return guarded(
    [&] {
      auto cloud = std::make_shared<PointCloud>(makeCloud(n, points, covs));
      map->updateLocalMap(cloud, ESKF_LIO::shim::poseFromData(transform), initialize != 0);
      if (n) {
        std::memcpy(points, cloud->points_.data(), n * 24);
        std::memcpy(covs, cloud->covariances_.data(), n * 72);
      }
    });
}

// correspondenceMatching(points, covariances) -> the four arrays, M returned through *matched.
int host_localmap_match(
  const LocalMap * map, size_t n, const double * points, const double * covs, double * src_points,
  double * src_covs, double * map_points, double * map_covs, size_t * matched)
{
  return guarded(
    [&] {
      PointCloud cloud = makeCloud(n, points, covs);
      auto [sp, sc, mp, mc] = map->correspondenceMatching(cloud.points_, cloud.covariances_);
      *matched = sp.size();
      if (!sp.empty()) {
        std::memcpy(src_points, sp.data(), sp.size() * 24);
        std::memcpy(src_covs, sc.data(), sc.size() * 72);
        std::memcpy(map_points, mp.data(), mp.size() * 24);
        std::memcpy(map_covs, mc.data(), mc.size() * 72);
      }
    });
}

// Host-authoritative voxel statistics (order = container iteration order).
size_t host_localmap_export(
  const LocalMap * map, size_t capacity, int32_t * keys, double * means, double * covs,
  uint64_t * counts)
{
  size_t w = 0;
  if (map->deviceResident()) {
    if (vgicp_map_export(map->context(), capacity, keys, means, covs, counts, &w) != VGICP_OK) {return 0;}
    return w;
  }
  for (const auto & kv : map->grid()) {
    if (w == capacity) {break;}
    keys[3 * w] = kv.first.i;
    keys[3 * w + 1] = kv.first.j;
    keys[3 * w + 2] = kv.first.k;
    std::memcpy(means + 3 * w, kv.second.mean.data(), 24);
    std::memcpy(covs + 9 * w, kv.second.covariance.data(), 72);
    counts[w] = kv.second.numPoints;
    ++w;
  }
  return w;
}

int host_localmap_save(const LocalMap * map, const char * cloud_path, const char * trajectory_path)
{
  return guarded([&] {map->save(cloud_path, trajectory_path);});
}

// ICP(config) — registration.max_iteration / translation_sq_threshold / cosine_threshold
ICP * host_icp_create(
  int max_iteration, double translation_sq_threshold, double cosine_threshold, int chunk_iterations)
{
  ESKF_LIO::RegistrationConfig c;
  c.maxIteration = max_iteration;
  c.translationSquaredThreshold = translation_sq_threshold;
  c.cosineThreshold = cosine_threshold;
  c.chunkIterations = chunk_iterations;
  return new ICP(c);
}
void host_icp_destroy(ICP * icp) {delete icp;}

// icp->align(cloud, localMap, guess)
int host_icp_align(
  ICP * icp, size_t n, const double * points, const double * covs, const LocalMap * map,
  const double guess[16], double out_pose[16], int32_t * iterations, int32_t * converged,
  uint64_t * corr_count, size_t corr_capacity)
{
  return guarded(
    [&] {
      const PointCloud cloud = makeCloud(n, points, covs);
      const Isometry3d T = icp->align(cloud, *map, ESKF_LIO::shim::poseFromData(guess));
      std::memcpy(out_pose, ESKF_LIO::shim::poseData(T), 16 * sizeof(double));
      const auto & st = icp->lastStats();
      if (iterations) {*iterations = st.iterations;}
      if (converged) {*converged = st.converged ? 1 : 0;}
      for (size_t k = 0; k < st.correspondenceCounts.size() && k < corr_capacity; ++k) {
        corr_count[k] = st.correspondenceCounts[k];
      }
    });
}

// Developer aid (tools/probe_eager.py): switch the classes' host-time trace on / off; out (optional) receives
// Trace::Slots seconds followed by Trace::Slots call counts (as doubles) and is then reset.
void host_trace(int enable, double * out)
{
  auto & t = ESKF_LIO::shim::trace();
  if (out) {
    for (int k = 0; k < ESKF_LIO::shim::Trace::Slots; ++k) {
      out[k] = t.seconds[k];
      out[ESKF_LIO::shim::Trace::Slots + k] = static_cast<double>(t.calls[k]);
      t.seconds[k] = 0.0;
      t.calls[k] = 0;
    }
  }
  t.on = enable != 0;
}

// CloudPreprocessor(config) — cloud_preprocessor.voxel_size, sensors.lidar.extrinsics as a 4x4;
// host_copy: 0 = eager (the host cloud holds the prepared scan after process()), 1 = deferred, -1 = the default;
// sampled_check != 0: CloudPreprocessorConfig::residentCheck = Sampled (the default hashes every byte)
CloudPreprocessor * host_preprocessor_create(double voxel_size, const double T_il[16], int host_copy, int sampled_check)
{
  CloudPreprocessor * out = nullptr;
  guarded(
    [&] {
      ESKF_LIO::CloudPreprocessorConfig c;
      c.voxelSize = voxel_size;
      if (T_il) {std::memcpy(c.T_il, T_il, sizeof c.T_il);}
      if (host_copy == 0) {c.hostCopy = ESKF_LIO::CloudPreprocessorConfig::HostCopy::Eager;}
      if (host_copy == 1) {c.hostCopy = ESKF_LIO::CloudPreprocessorConfig::HostCopy::Deferred;}
      if (sampled_check) {c.residentCheck = ESKF_LIO::shim::ResidentCheck::Sampled;}
      out = new CloudPreprocessor(c);
    });
  return out;
}

// ---- one LiDAR frame through the drop-in classes exactly as src/Odometry.cpp:73-87 writes it ----
//   cloudPreprocessor_->process(states, lidarMeas_);                         (:74)
//   transform = kalmanFilter_->update(...) -> icp_->align(*lidar.cloud, localMap, guess)   (:79, ErrorStateKF.cpp:130)
//   localMap_->updateLocalMap(std::move(lidarMeasCopy->cloud), transform);   (:86)
// In three steps so that a caller can time host_frame_run alone (the measurement object is built before).
struct HostFrame
{
  ESKF_LIO::LidarMeasurementPtr meas;
  std::deque<ESKF_LIO::State> states;
  Isometry3d pose;
  int iterations = 0;
  bool usedResident = false;
  size_t corr0 = 0;
};

HostFrame * host_frame_begin(
  size_t n, const double * points, const double * point_time, size_t num_states, const double * states)
{
  HostFrame * f = new HostFrame;
  f->meas = std::make_shared<ESKF_LIO::LidarMeasurement>();
  f->meas->cloud = std::make_shared<PointCloud>();
  f->meas->cloud->points_.resize(n);
  if (n) {std::memcpy(f->meas->cloud->points_.data(), points, n * 24);}
  f->meas->pointTime.assign(point_time, point_time + n);
  f->states.resize(num_states);
  for (size_t s = 0; s < num_states; ++s) {
    f->states[s].timestamp = states[8 * s];
    for (int a = 0; a < 3; ++a) {f->states[s].position(a) = states[8 * s + 1 + a];}
    for (int a = 0; a < 4; ++a) {f->states[s].attitude.c[a] = states[8 * s + 4 + a];}
  }
  return f;
}

// CloudPreprocessor::stage on the frame's measurement (what a lidar callback would do when the sweep arrives): 1 staged
int host_frame_stage(HostFrame * f, const CloudPreprocessor * p)
{
  int staged = 0;
  const int rc = guarded([&] {staged = p->stage(f->meas) ? 1 : 0;});
  return rc == 0 ? staged : -1;
}

// mutate: 0 = the frame as the reference runs it; 1 = the caller edits the prepared cloud between process() and
// align() (its FIRST point moves by 1 mm: an element even the sampled check looks at); 2 = the caller resizes it;
// 3 = the caller edits an element the SAMPLED check never looks at (point 1 of thousands: the default check, which
// hashes every byte, must notice; ResidentCheck::Sampled does not — the documented price of that mode).
// first_frame: process({}, meas) + updateLocalMap(cloud, guess) without align (src/Odometry.cpp:60-61).
// stage_next: a later frame whose sweep "arrives" while this one is being processed: staged (CloudPreprocessor::stage)
// right after this frame's process(), where a lidar callback's thread would be copying beside the device's work.
int host_frame_run(
  HostFrame * f, const CloudPreprocessor * p, ICP * icp, LocalMap * map, const double guess[16], int first_frame,
  int mutate, HostFrame * stage_next, int move_cloud)
{
  return guarded(
    [&] {
      const Isometry3d g = ESKF_LIO::shim::poseFromData(guess);
      if (first_frame) {
        p->process({}, f->meas);
        f->pose = g;
        map->updateLocalMap(f->meas->cloud, g);
        return;
      }
      p->process(f->states, f->meas);
      if (stage_next) {p->stage(stage_next->meas);}
      if (mutate) {ESKF_LIO::shim::materialize(map->context(), *f->meas->cloud);}   // an editor needs the data first
      if (mutate == 1 && !f->meas->cloud->points_.empty()) {f->meas->cloud->points_[0](0) += 1e-3;}
      if (mutate == 3 && f->meas->cloud->points_.size() > 200) {f->meas->cloud->points_[1](0) += 1e-3;}
      if (mutate == 2 && f->meas->cloud->points_.size() > 1) {
        f->meas->cloud->points_.pop_back();
        f->meas->cloud->covariances_.pop_back();
      }
      f->pose = icp->align(*f->meas->cloud, *map, g);
      f->usedResident = icp->lastUsedResidentScan();
      f->iterations = icp->lastStats().iterations;
      f->corr0 = icp->lastStats().correspondenceCounts.empty() ? 0 : icp->lastStats().correspondenceCounts[0];
      // move_cloud: as src/Odometry.cpp:86 hands it over (std::move: the map becomes the cloud's only owner);
      // else a copy of the pointer stays with the frame so that host_frame_end can return the cloud
      if (move_cloud) {map->updateLocalMap(std::move(f->meas->cloud), f->pose);} else {map->updateLocalMap(f->meas->cloud, f->pose);}
    });
}

// What the frame left: pose, rounds, first round's correspondence count, whether align found the scan resident,
// and the host cloud as it is now (size through *host_points; copied when the arrays are given).
int host_frame_end(
  HostFrame * f, double out_pose[16], int32_t * iterations, int32_t * used_resident, uint64_t * corr0,
  size_t * host_points, size_t capacity, double * points, double * covs)
{
  const int rc = guarded(
    [&] {
      std::memcpy(out_pose, ESKF_LIO::shim::poseData(f->pose), 16 * sizeof(double));
      if (iterations) {*iterations = f->iterations;}
      if (used_resident) {*used_resident = f->usedResident ? 1 : 0;}
      if (corr0) {*corr0 = f->corr0;}
      if (!f->meas->cloud) {                         // moved into the map (host_frame_run with move_cloud)
        if (host_points) {*host_points = 0;}
        return;
      }
      const auto & cloud = *f->meas->cloud;
      if (host_points) {*host_points = cloud.points_.size();}
      if (points && covs && cloud.covariances_.size() == cloud.points_.size() && cloud.points_.size() <= capacity) {
        if (!cloud.points_.empty()) {
          std::memcpy(points, cloud.points_.data(), cloud.points_.size() * 24);
          std::memcpy(covs, cloud.covariances_.data(), cloud.covariances_.size() * 72);
        }
      }
    });
  delete f;
  return rc;
}
void host_preprocessor_destroy(CloudPreprocessor * p) {delete p;}

// preprocessor->voxelDownsampleAndEstimateCovariances(cloud): the cloud is n points (covariances are
// produced); out arrays hold n entries, *kept receives the new size of the cloud.
int host_preprocessor_downsample(
  const CloudPreprocessor * p, size_t n, const double * points, double * out_points, double * out_covs,
  size_t * kept)
{
  return guarded(
    [&] {
      PointCloud cloud;
      cloud.points_.resize(n);
      if (n) {std::memcpy(cloud.points_.data(), points, n * 24);}
      p->voxelDownsampleAndEstimateCovariances(cloud);
      *kept = cloud.points_.size();
      if (*kept) {
        std::memcpy(out_points, cloud.points_.data(), *kept * 24);
        std::memcpy(out_covs, cloud.covariances_.data(), *kept * 72);
      }
    });
}

// preprocessor->process(states, lidarMeas): extrinsic, deskew, down-sampling + covariances.
// states: num_states x 8 (timestamp, position, quaternion x y z w).
int host_preprocessor_process(
  const CloudPreprocessor * p, size_t n, const double * points, const double * point_time,
  size_t num_states, const double * states, double * out_points, double * out_covs, size_t * kept)
{
  return guarded(
    [&] {
      auto meas = std::make_shared<ESKF_LIO::LidarMeasurement>();
      meas->cloud = std::make_shared<PointCloud>();
      meas->cloud->points_.resize(n);
      if (n) {std::memcpy(meas->cloud->points_.data(), points, n * 24);}
      meas->pointTime.assign(point_time, point_time + n);
      std::deque<ESKF_LIO::State> queue(num_states);
      for (size_t s = 0; s < num_states; ++s) {
        queue[s].timestamp = states[8 * s];
        for (int a = 0; a < 3; ++a) {queue[s].position(a) = states[8 * s + 1 + a];}
        for (int a = 0; a < 4; ++a) {queue[s].attitude.c[a] = states[8 * s + 4 + a];}
      }
      p->process(queue, meas);
      ESKF_LIO::shim::materialize(ESKF_LIO::shim::defaultContext(), *meas->cloud);   // a deferred host copy is filled now
      *kept = meas->cloud->points_.size();
      if (*kept) {
        std::memcpy(out_points, meas->cloud->points_.data(), *kept * 24);
        std::memcpy(out_covs, meas->cloud->covariances_.data(), *kept * 72);
      }
      if (!meas->pointTime.empty()) {throw std::runtime_error("process() must clear pointTime");}
    });
}

// The module's own 6x6 solve (csrc/vgicp_math.h, the code the kernels run as their pivoted fallback),
// compiled for the host: lets a CPU-only test pin its operation order against the oracle's LDLT.
// lower21: packed lower triangle row by row; b: right-hand side; x: solution.
void host_math_ldlt6_solve(const double * lower21, const double * b, double * x)
{
  double work[vgicp::kLdltWork];
  vgicp::ldlt6_solve(lower21, b, x, work);
}

// Utils::se3ToSE3 as the module's host math states it -> column-major 4x4.
void host_math_se3_exp(const double * xi, double * out16)
{
  vgicp::Pose T;
  vgicp::se3_exp(xi, T);
  vgicp::pose_to_mat4(T, out16);
}

}  // extern "C"
