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
// rely on.
// process() and deskew() (include/ESKF_LIO/CloudPreprocessor.hpp:34,41-43, src/CloudPreprocessor.cpp:8-74)
// are here too: process() applies the LiDAR->IMU extrinsic, deskews with the IMU states and prepares the scan, in
// the reference's order, as ONE enqueue on the device (vgicp_scan_prepare_async) whose result stays resident for
// ICP::align and LocalMap::updateLocalMap (CloudPreprocessorConfig::hostCopy says what the host cloud then holds).  Where the reference's deskew would step
// off its state queue (no state at or before the end of the sweep, or none after it: undefined behaviour
// there) this one throws std::runtime_error.
#ifndef ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_
#define ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_

#include <cstdint>
#include <cstdlib>
#include <deque>
#include <memory>
#include <string>
#include <mutex>
#include <vector>

#include "LocalMap.hpp"
#if defined(ESKF_LIO_SHIM_NATIVE_TYPES)
#include "ESKF_LIO/Types.hpp"  // the reference's State / LidarMeasurement
#endif

namespace ESKF_LIO
{

// The key CloudPreprocessor's YAML constructor reads for this half (config/hilti_config.yaml:
// cloud_preprocessor.voxel_size).
struct CloudPreprocessorConfig
{
  double voxelSize = 0.3;
  int knn = 30;  // open3d::geometry::KDTreeSearchParamKNN's default
  // process() leaves the prepared scan RESIDENT on the device (vgicp_scan_prepare*), so that ICP::align and — with
  // LocalMapConfig::deviceResident — LocalMap::updateLocalMap work on it without another upload.  What the HOST
  // cloud holds when process() returns:
  //   Eager     the prepared scan, as the reference leaves it (one synchronisation + one download per frame);
  //   Deferred  the raw sweep it held before; the prepared scan is on the device only until somebody needs it on the
  //             host (shim::materialize(ctx, cloud); LocalMap's host-authoritative mode does it by itself).  This is
  //             the frame chain without host round trips: src/Odometry.cpp:73-87 never reads the cloud between
  //             process(), update() and updateLocalMap().
  // Default: Eager; VGICP_HOST_COPY=deferred in the environment (or cloud_preprocessor.host_copy: deferred in the
  // YAML file) selects Deferred without touching the caller.
  enum class HostCopy {Eager, Deferred};
  HostCopy hostCopy = defaultHostCopy();
  // how ICP::align / LocalMap::updateLocalMap later recognise "this host cloud is still the scan process() left on the
  // device" (LocalMap.hpp, shim::ResidentCheck): FullHash (default) sees an in-place edit of ANY element and falls back
  // to uploading the cloud, as the reference reads the host cloud every time; Sampled (~0.1 ms per frame cheaper) sees
  // resizes, reallocations and edits of 64 sampled elements only
  shim::ResidentCheck residentCheck = shim::ResidentCheck::FullHash;
  // helper threads of the FullHash check (shim::HashCrew, shared by every object of the process: the last configuration
  // made wins): they hash beside the caller, and in ICP::align while the caller waits for the device.  0: the caller
  // alone (one core, ~70 us per check and 35 000 points); at most 3.  YAML key cloud_preprocessor.resident_check_threads
  int residentCheckThreads = 2;
  // true: the prepared cloud's points come in the sequence the REFERENCE emits them (the iteration order of its
  // unordered_map, src/CloudPreprocessor.cpp:85-99) instead of ascending input index: a frame chain then reproduces the
  // reference's chain, not only its per-call results (VGICP_OPTION_REFERENCE_ORDER; ~1.5 ms per frame on the host, and
  // process() waits for the device).  YAML key cloud_preprocessor.reference_order: true
  bool referenceOrder = false;
  static HostCopy defaultHostCopy()
  {
    const char * env = std::getenv("VGICP_HOST_COPY");
    return (env && (env[0] == 'd' || env[0] == 'D')) ? HostCopy::Deferred : HostCopy::Eager;
  }
  // sensors.lidar.extrinsics (quaternion + translation) as the 4x4 the reference builds from them
  // (include/ESKF_LIO/CloudPreprocessor.hpp:20-28), column-major; identity by default
  double T_il[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

namespace shim
{
// Sweeps handed to the device module when they ARRIVED (CloudPreprocessor::stage, called from the lidar callback's
// thread): measurement object -> ticket of vgicp_sweep_stage.  process() looks its measurement up here.
struct StagedSweep
{
  vgicp_ctx * ctx;
  std::weak_ptr<const void> meas;   // the measurement object, watched: an entry whose measurement has died is dropped
  const void * pointData;           // (and its slot in the module released) instead of meeting a new one at its address
  size_t n;
  uint64_t ticket;
};
// under stagedSweepsMutex(): forget the sweeps whose measurement no longer exists (dropped unprocessed by the caller)
inline void dropDeadSweeps(std::vector<StagedSweep> & staged)
{
  for (size_t i = 0; i < staged.size(); ) {
    if (staged[i].meas.expired()) {
      (void)vgicp_sweep_unstage(staged[i].ctx, staged[i].ticket);
      staged.erase(staged.begin() + static_cast<std::ptrdiff_t>(i));
    } else {
      ++i;
    }
  }
}
inline std::mutex & stagedSweepsMutex()
{
  static std::mutex m;
  return m;
}
inline std::vector<StagedSweep> & stagedSweeps()
{
  static std::vector<StagedSweep> v;
  return v;
}
}  // namespace shim

class CloudPreprocessor
{
public:
  explicit CloudPreprocessor(const CloudPreprocessorConfig & config, vgicp_ctx * ctx = nullptr)
  : voxelSize_(config.voxelSize), knn_(config.knn), T_il_(shim::poseFromData(config.T_il)),
    hostCopy_(config.hostCopy), residentCheck_(config.residentCheck), ctx_(ctx ? ctx : shim::defaultContext())
  {
    if (config.referenceOrder) {
      shim::check(ctx_, vgicp_set_option(ctx_, VGICP_OPTION_REFERENCE_ORDER, 1), "vgicp_set_option");
    }
    shim::HashCrew::instance().setHelpers(config.residentCheckThreads);
  }

#if defined(ESKF_LIO_SHIM_HAVE_YAML)
  CloudPreprocessor(const YAML::Node & config)
  : voxelSize_(config["cloud_preprocessor"]["voxel_size"].as<double>()), knn_(30),
    ctx_(shim::defaultContext())
  {
    auto lidar = config["sensors"]["lidar"];
    auto lidar_quat = lidar["extrinsics"]["quaternion"].as<std::vector<double>>();
    auto lidar_trans = lidar["extrinsics"]["translation"].as<std::vector<double>>();
    Eigen::Quaterniond quat = Eigen::Map<Eigen::Quaterniond>(lidar_quat.data());
    T_il_.linear() = quat.toRotationMatrix();
    T_il_.translation() = Eigen::Map<Eigen::Vector3d>(lidar_trans.data());
    if (config["cloud_preprocessor"]["host_copy"].IsDefined()) {   // optional key, not in the reference's file
      const auto mode = config["cloud_preprocessor"]["host_copy"].as<std::string>();
      hostCopy_ = (mode == "deferred") ? CloudPreprocessorConfig::HostCopy::Deferred :
        CloudPreprocessorConfig::HostCopy::Eager;
    }
    if (config["cloud_preprocessor"]["reference_order"].IsDefined() &&
      config["cloud_preprocessor"]["reference_order"].as<bool>())
    {
      shim::check(ctx_, vgicp_set_option(ctx_, VGICP_OPTION_REFERENCE_ORDER, 1), "vgicp_set_option");
    }
    if (config["cloud_preprocessor"]["resident_check_threads"].IsDefined()) {   // optional key, not in the reference's file
      shim::HashCrew::instance().setHelpers(config["cloud_preprocessor"]["resident_check_threads"].as<int>());
    }
    if (config["cloud_preprocessor"]["resident_check"].IsDefined()) {   // optional key, not in the reference's file
      residentCheck_ = config["cloud_preprocessor"]["resident_check"].as<std::string>() == "sampled" ?
        shim::ResidentCheck::Sampled : shim::ResidentCheck::FullHash;
    }
  }
#endif

  // OPTIONAL, not in the reference: call it where the sweep arrives (the lidar callback, include/ESKF_LIO/Subscriber.hpp:
  // 80-103, once cloud and pointTime are filled).  The raw sweep is copied into page-locked memory of the device module
  // right away (CPU copies only; safe to call from the callback's thread while Odometry::run is inside process / align
  // / updateLocalMap), and the process() that later gets this measurement starts from there instead of copying first:
  // src/Odometry.cpp:43-48 pops a measurement long after it arrived.  The measurement must not be edited in between.
  // Returns false (and does nothing) when three sweeps are staged already.
  bool stage(const LidarMeasurementPtr & lidarMeas) const
  {
    const PointCloud & cloud = *lidarMeas->cloud;
    const size_t n = cloud.points_.size();
    if (n == 0 || lidarMeas->pointTime.size() != n) {return false;}
    uint64_t ticket = 0;
    {
      std::lock_guard<std::mutex> lk(shim::stagedSweepsMutex());
      shim::dropDeadSweeps(shim::stagedSweeps());   // measurements the caller discarded give their slots back first
    }
    const int rc = vgicp_sweep_stage(
      ctx_, n, reinterpret_cast<const double *>(cloud.points_.data()), lidarMeas->pointTime.data(), &ticket);
    if (rc != VGICP_OK) {return false;}
    std::lock_guard<std::mutex> lk(shim::stagedSweepsMutex());
    shim::stagedSweeps().push_back({ctx_, std::weak_ptr<const void>(lidarMeas), cloud.points_.data(), n, ticket});
    return true;
  }

  // reference src/CloudPreprocessor.cpp:8-23: extrinsic, deskew, down-sampling + covariances — here ONE enqueue
  // (vgicp_scan_prepare_async: 32 bytes per raw point go up, every step runs on the device) that leaves the
  // prepared scan resident, plus the host copy the configuration asks for.
  void process(const std::deque<State> & states, LidarMeasurementPtr lidarMeas) const
  {
    PointCloud & cloud = *lidarMeas->cloud;
    const size_t n = cloud.points_.size();
    std::vector<double> packed(states.size() * 8);
    size_t k = 0;
    for (const auto & state : states) {
      packed[k++] = state.timestamp;
      for (int a = 0; a < 3; ++a) {packed[k++] = state.position(a);}
      const double * q = shim::quatData(state.attitude);
      for (int a = 0; a < 4; ++a) {packed[k++] = q[a];}
    }
    static_assert(sizeof(Vector3d) == 3 * sizeof(double), "points must be packed xyz triples");
    uint64_t ticket = 0;   // staged when it arrived (stage())?
    {
      std::lock_guard<std::mutex> lk(shim::stagedSweepsMutex());
      auto & staged = shim::stagedSweeps();
      shim::dropDeadSweeps(staged);
      for (size_t i = 0; i < staged.size(); ++i) {
        if (staged[i].ctx == ctx_ && staged[i].meas.lock().get() == static_cast<const void *>(lidarMeas.get())) {
          if (staged[i].pointData == cloud.points_.data() && staged[i].n == n) {
            ticket = staged[i].ticket;
          } else {
            (void)vgicp_sweep_unstage(ctx_, staged[i].ticket);   // the cloud was replaced since: the staged bytes are stale
          }
          staged.erase(staged.begin() + static_cast<std::ptrdiff_t>(i));
          break;
        }
      }
    }
    int rc = VGICP_ERR_BAD_ARGUMENT;
    {
      shim::TraceScope ts(shim::Trace::ProcessEnqueue);
      if (ticket) {
        rc = vgicp_scan_prepare_staged_async(
          ctx_, ticket, states.size(), packed.data(), shim::poseData(T_il_), voxelSize_, knn_);
      }
      if (!ticket) {
        rc = vgicp_scan_prepare_async(
          ctx_, n, n ? reinterpret_cast<const double *>(cloud.points_.data()) : nullptr,
          lidarMeas->pointTime.data(), states.size(), packed.data(), shim::poseData(T_il_), voxelSize_, knn_);
      }
    }
    if (rc == VGICP_ERR_BAD_ARGUMENT && !states.empty() && n) {
      // where the reference's deskew would step off its state queue (undefined behaviour there)
      throw std::runtime_error(std::string("process: ") + vgicp_last_error(ctx_));
    }
    shim::check(ctx_, rc, "vgicp_scan_prepare_async");
    lidarMeas->pointTime.clear();          // as the reference: the capture times are consumed (copied when the call returned)
    lidarMeas->pointTime.shrink_to_fit();
    if (hostCopy_ == CloudPreprocessorConfig::HostCopy::Deferred) {
      cloud.covariances_.clear();
      shim::stampResident(ctx_, cloud, 0, false, residentCheck_);   // the host keeps the raw sweep; the prepared scan is on the device
      return;
    }
    // The eager host copy: one kernel behind the preparation writes the prepared scan into page-locked memory; the
    // count arrives first (the down-sampling knows it long before the neighbour search is through), the vectors are
    // sized while the device still works, and the pieces are copied out as they land (vgicp_scan_fetch_*).
    size_t kept = 0;
    {
      shim::TraceScope ts(shim::Trace::ProcessWait);
      shim::check(ctx_, vgicp_scan_fetch_begin(ctx_, &kept), "vgicp_scan_fetch_begin");
    }
    {
      shim::TraceScope ts(shim::Trace::ProcessResize);
      shim::adoptStorage(cloud.covariances_, shim::storagePool().covariances, kept);   // storage of an earlier frame's cloud
      cloud.points_.resize(kept);
      cloud.covariances_.resize(kept);
    }
    {
      shim::TraceScope ts(shim::Trace::ProcessDownload);
      shim::check(
        ctx_, vgicp_scan_fetch_end(
          ctx_, kept, kept ? reinterpret_cast<double *>(cloud.points_.data()) : nullptr,
          kept ? reinterpret_cast<double *>(cloud.covariances_.data()) : nullptr, &kept), "vgicp_scan_fetch_end");
    }
    // the stamp: the full hash of what was just delivered comes from the device's own sums over the bytes it wrote (no
    // second pass over 96 bytes per point here); the CHECKS in align / updateLocalMap still hash the host cloud itself
    shim::TraceScope ts(shim::Trace::ProcessStamp);
    uint64_t sums[64], known = 0;
    const bool summed = residentCheck_ == shim::ResidentCheck::FullHash && vgicp_scan_fetch_sums(ctx_, sums) == VGICP_OK;
    if (summed) {known = shim::fullHashFromSums(sums, kept);}
    shim::stampResident(ctx_, cloud, kept, true, residentCheck_, summed ? &known : nullptr);
  }

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

  // reference src/CloudPreprocessor.cpp:25-74 (private there; public here so that tests can reach it)
  void deskew(
    const std::deque<State> & states, const std::vector<double> & pointTime,
    std::vector<Vector3d> & points) const
  {
    std::vector<double> packed(states.size() * 8);
    size_t k = 0;
    for (const auto & state : states) {
      packed[k++] = state.timestamp;
      for (int a = 0; a < 3; ++a) {packed[k++] = state.position(a);}
      const double * q = shim::quatData(state.attitude);
      for (int a = 0; a < 4; ++a) {packed[k++] = q[a];}
    }
    int64_t moved = 0;
    shim::check(
      ctx_,
      vgicp_deskew(
        ctx_, points.size(), points.empty() ? nullptr : reinterpret_cast<double *>(points.data()),
        pointTime.data(), states.size(), packed.data(), &moved),
      "vgicp_deskew");
    if (moved < 0) {
      throw std::runtime_error("deskew: the IMU states do not bracket the end of the sweep");
    }
  }

private:
  CloudPreprocessor() = delete;

  double voxelSize_;
  int knn_;
  Isometry3d T_il_;
  CloudPreprocessorConfig::HostCopy hostCopy_ = CloudPreprocessorConfig::HostCopy::Eager;
  shim::ResidentCheck residentCheck_ = shim::ResidentCheck::FullHash;
  vgicp_ctx * ctx_;
};
}  // namespace ESKF_LIO

#endif  // ESKF_LIO_SHIM_CLOUD_PREPROCESSOR_HPP_
