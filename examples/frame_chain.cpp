// frame_chain.cpp — one LiDAR frame after the other through the WHOLE per-frame chain of the reference,
// in C++, twice:
//
//   A. with the drop-in classes, the way src/Odometry.cpp:73-87 is written:
//          cloudPreprocessor.process(states, lidarMeas);              // extrinsic, deskew, down-sampling + covariances
//          T = icp.align(*lidarMeas->cloud, localMap, guess);         // ErrorStateKF::update's registration
//          localMap.updateLocalMap(lidarMeas->cloud, T);
//      — with the grid on the device (LocalMapConfig::deviceResident) and the host copy of the prepared scan
//      deferred (CloudPreprocessorConfig::HostCopy::Deferred): the three classes pass the scan along ON the device;
//   A2. the same classes with the host-authoritative map of rounds 1-4 (LocalMap(voxelSize, maxNumPointsPerVoxel)), eager host copy (the host cloud holds
//      the prepared scan after process(), as in the reference);
//   B. straight on the C ABI with the scan resident on the GPU from the raw sweep to the map update:
//          vgicp_scan_prepare(...); vgicp_align_resident(...); vgicp_map_insert_resident(...);
//   C. the same through the calls that do not wait (vgicp_scan_prepare_async / _align_resident /
//      _map_insert_resident_async): one host synchronisation per frame.  Chain A must return C's bits.
//
// and compares the two trajectories with each other and with the motion that generated the sweeps. The world
// is a floor, two walls and a ceiling sampled at random; the sensor moves at constant velocity with a slow
// yaw, every point is seen from the pose at its own capture time (so the deskew matters), and the IMU state
// queue holds the true poses at 400 Hz (the filter itself is not part of this example: tools/replay.py has it).
//
// build:  make -C examples   (plain g++; links eskf_lio_amd/lib/libvgicp_hip.so)
#define ESKF_LIO_SHIM_FORCE_POD 1
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <vector>

#include "../include/eskf_lio_shim/CloudPreprocessor.hpp"
#include "../include/eskf_lio_shim/Registration.hpp"

using namespace ESKF_LIO;

namespace
{
uint64_t splitmix(uint64_t & s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double unit(uint64_t & s) {return (splitmix(s) >> 11) * (1.0 / 9007199254740992.0);}

// sensor pose at time tau: constant velocity along x/y, slow yaw
struct Pose {double x, y, z, yaw;};
Pose motion(double tau) {return Pose{1.2 * tau, 0.3 * tau, 0.0, 0.15 * tau};}

Isometry3d toIsometry(const Pose & p)
{
  Isometry3d T = Isometry3d::Identity();
  T.matrix()(0, 0) = std::cos(p.yaw); T.matrix()(0, 1) = -std::sin(p.yaw);
  T.matrix()(1, 0) = std::sin(p.yaw); T.matrix()(1, 1) = std::cos(p.yaw);
  T.matrix()(0, 3) = p.x; T.matrix()(1, 3) = p.y; T.matrix()(2, 3) = p.z;
  return T;
}

State stateAt(double t0, double tau)
{
  const Pose p = motion(tau);
  State s;
  s.timestamp = t0 + tau;
  s.position = Vector3d{{p.x, p.y, p.z}};
  s.attitude.c[0] = 0.0; s.attitude.c[1] = 0.0; s.attitude.c[2] = std::sin(0.5 * p.yaw); s.attitude.c[3] = std::cos(0.5 * p.yaw);
  return s;
}

// a room of 24 x 16 x 5 m around the origin, sampled at random on its six faces
Vector3d worldPoint(uint64_t & seed)
{
  const double u = unit(seed), a = unit(seed), b = unit(seed);
  const double X = 12.0, Y = 8.0, Z0 = -1.5, Z1 = 3.5;
  if (u < 0.40) {return Vector3d{{-X + 2 * X * a, -Y + 2 * Y * b, Z0}};}          // floor
  if (u < 0.55) {return Vector3d{{-X + 2 * X * a, -Y + 2 * Y * b, Z1}};}          // ceiling
  if (u < 0.70) {return Vector3d{{-X + 2 * X * a, -Y, Z0 + (Z1 - Z0) * b}};}      // walls
  if (u < 0.85) {return Vector3d{{-X + 2 * X * a, Y, Z0 + (Z1 - Z0) * b}};}
  if (u < 0.925) {return Vector3d{{-X, -Y + 2 * Y * a, Z0 + (Z1 - Z0) * b}};}
  return Vector3d{{X, -Y + 2 * Y * a, Z0 + (Z1 - Z0) * b}};
}

// one sweep of n points ending at time end (seconds after t0), 0.1 s long, seen from the moving sensor
LidarMeasurementPtr sweep(double t0, double end, size_t n, uint64_t seed, bool moving)
{
  auto meas = std::make_shared<LidarMeasurement>();
  meas->cloud = std::make_shared<PointCloud>();
  meas->cloud->points_.resize(n);
  meas->pointTime.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const double tau = end - 0.1 + 0.1 * (double)(i + 1) / (double)n;
    const Pose p = moving ? motion(tau) : motion(end);
    const Vector3d w = worldPoint(seed);
    const double dx = w(0) - p.x, dy = w(1) - p.y, dz = w(2) - p.z;
    const double c = std::cos(p.yaw), s = std::sin(p.yaw);
    Vector3d local{{c * dx + s * dy + 0.004 * (unit(seed) - 0.5), -s * dx + c * dy + 0.004 * (unit(seed) - 0.5),
      dz + 0.004 * (unit(seed) - 0.5)}};
    meas->cloud->points_[i] = local;
    meas->pointTime[i] = t0 + tau;
  }
  meas->startTime = meas->pointTime.front();
  meas->endTime = meas->pointTime.back();
  return meas;
}

double positionError(const double * pose16, const Pose & truth)
{
  const double ex = pose16[12] - truth.x, ey = pose16[13] - truth.y, ez = pose16[14] - truth.z;
  return std::sqrt(ex * ex + ey * ey + ez * ez);
}
}  // namespace

int main(int argc, char ** argv)
{
  try {
    const double t0 = 500.0, voxel = 0.3;
    const int frames = argc > 1 ? std::atoi(argv[1]) : 6;             // frame_chain [frames] [points per sweep]
    const size_t n = argc > 2 ? (size_t)std::atol(argv[2]) : 30000;
    // the state queue of the whole run: true poses at 400 Hz (what ErrorStateKF::getStates() would hold)
    std::deque<State> states;
    for (int k = -50; k <= (int)(400 * 0.1 * frames) + 20; ++k) {states.push_back(stateAt(t0, k / 400.0 + 0.00037));}

    CloudPreprocessorConfig pc;
    pc.voxelSize = voxel;
    pc.hostCopy = CloudPreprocessorConfig::HostCopy::Deferred;
    CloudPreprocessor preprocessor(pc);
    RegistrationConfig rc;
    rc.maxIteration = 100; rc.translationSquaredThreshold = 1e-6; rc.cosineThreshold = 0.9999;
    ICP icp(rc);
    LocalMapConfig mc;                       // chain A: every update inserts (no motion gate), nothing is evicted
    mc.voxelSize = voxel; mc.maxNumPointsPerVoxel = 20; mc.translationSquaredThreshold = -1.0; mc.cosineThreshold = 2.0;
    mc.removeDistantPoints = false; mc.deviceResident = true;
    LocalMap localMap(mc);
    // chain A2 owns a context of its own: the classes with the host-authoritative map
    vgicp_ctx * ctxA2 = nullptr;
    shim::check(nullptr, vgicp_create(0, &ctxA2), "vgicp_create");
    CloudPreprocessorConfig pc2;
    pc2.voxelSize = voxel;
    pc2.hostCopy = CloudPreprocessorConfig::HostCopy::Eager;
    CloudPreprocessor preprocessor2(pc2, ctxA2);
    ICP icp2(rc);
    LocalMap localMap2(voxel, 20, false, ctxA2);

    // chain B owns a second context: its own device-resident map
    vgicp_ctx * ctx = nullptr;
    shim::check(nullptr, vgicp_create(0, &ctx), "vgicp_create");
    shim::check(ctx, vgicp_map_reset(ctx, voxel, 0), "vgicp_map_reset");
    std::vector<double> packed(states.size() * 8);
    for (size_t s = 0; s < states.size(); ++s) {
      packed[8 * s] = states[s].timestamp;
      for (int a = 0; a < 3; ++a) {packed[8 * s + 1 + a] = states[s].position(a);}
      for (int a = 0; a < 4; ++a) {packed[8 * s + 4 + a] = states[s].attitude.c[a];}
    }
    // chain C: the same again through the calls that do not wait (one host synchronisation per frame)
    vgicp_ctx * ctxC = nullptr;
    shim::check(nullptr, vgicp_create(0, &ctxC), "vgicp_create");
    shim::check(ctxC, vgicp_map_reset(ctxC, voxel, 400000), "vgicp_map_reset");
    vgicp_params params{};
    params.max_iteration = 100; params.translation_sq_threshold = 1e-6; params.cosine_threshold = 0.9999;

    Isometry3d estimateA = Isometry3d::Identity(), estimateA2 = Isometry3d::Identity();
    double classesMs = 0.0, classesEagerMs = 0.0, stageMs[3] = {0.0, 0.0, 0.0};
    int residentAligns = 0;
    double estimateB[16];
    std::memcpy(estimateB, shim::poseData(estimateA), sizeof estimateB);
    double estimateC[16];
    std::memcpy(estimateC, shim::poseData(estimateA), sizeof estimateC);
    int bad = 0, residentFrames = 0;
    double residentMs = 0.0, asyncMs = 0.0;
    unsigned long long asyncSyncs = 0, asyncLaunches = 0;
    for (int f = 0; f < frames; ++f) {
      const double end = 0.1 * f;
      LidarMeasurementPtr meas = sweep(t0, end, n, 77 + f, f > 0);
      const std::vector<Vector3d> raw = meas->cloud->points_;
      const std::vector<double> times = meas->pointTime;
      const Pose truth = motion(end);

      // ---- A: the drop-in classes, the scan handed along on the device ----
      LidarMeasurementPtr meas2 = std::make_shared<LidarMeasurement>(*meas);   // chain A2's own copy of the measurement
      meas2->cloud = std::make_shared<PointCloud>(*meas->cloud);
      const auto a0 = std::chrono::steady_clock::now();
      if (f == 0) {
        preprocessor.process({}, meas);                                        // src/Odometry.cpp:60
        localMap.updateLocalMap(meas->cloud, Isometry3d::Identity());          // :61
      } else {
        preprocessor.process(states, meas);                                    // :74
        const auto a1 = std::chrono::steady_clock::now();
        estimateA = icp.align(*meas->cloud, localMap, toIsometry(motion(end - 0.1)));   // guess: last frame's pose
        const auto a2 = std::chrono::steady_clock::now();
        if (icp.lastUsedResidentScan()) {++residentAligns;}
        localMap.updateLocalMap(meas->cloud, estimateA);                       // :86
        if (std::getenv("FRAME_CHAIN_VERBOSE")) {
          vgicp_frame_stats vs{};
          (void)vgicp_get_frame_stats(shim::defaultContext(), &vs, 0);
          std::printf("  [chain A] frame %d stages: process %.3f align %.3f update %.3f ms; device spans (VGICP_STAGE_EVENTS=1): prepare %.1f us (head %.1f), align %.1f us\n", f,
                      std::chrono::duration<double, std::milli>(a1 - a0).count(), std::chrono::duration<double, std::milli>(a2 - a1).count(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a2).count(), vs.prepare_us, vs.prepare_head_us, vs.align_us);
        }
        if (f > 1) {
          stageMs[0] += std::chrono::duration<double, std::milli>(a1 - a0).count();
          stageMs[1] += std::chrono::duration<double, std::milli>(a2 - a1).count();
          stageMs[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a2).count();
        }
      }
      if (f > 1) {classesMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();}
      if (std::getenv("FRAME_CHAIN_VERBOSE")) {
        std::printf("  [chain A] frame %d: %.3f ms\n", f, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count());
      }
      // ---- A2: the same classes with the host-authoritative map and the eager host copy ----
      const auto a20 = std::chrono::steady_clock::now();
      if (f == 0) {
        preprocessor2.process({}, meas2);
        localMap2.updateLocalMap(meas2->cloud, Isometry3d::Identity());
      } else {
        preprocessor2.process(states, meas2);
        const auto e1 = std::chrono::steady_clock::now();
        estimateA2 = icp2.align(*meas2->cloud, localMap2, toIsometry(motion(end - 0.1)));
        const auto e2 = std::chrono::steady_clock::now();
        localMap2.updateLocalMap(meas2->cloud, estimateA2);
        if (std::getenv("FRAME_CHAIN_VERBOSE")) {
          std::printf("  [chain A2] frame %d stages: process %.3f align %.3f update %.3f ms\n", f,
                      std::chrono::duration<double, std::milli>(e1 - a20).count(), std::chrono::duration<double, std::milli>(e2 - e1).count(),
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e2).count());
        }
      }
      if (f > 1) {classesEagerMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a20).count();}

      // ---- B: the same frame with the scan resident on the device ----
      const auto b0 = std::chrono::steady_clock::now();
      size_t kept = 0;
      int64_t moved = 0;
      shim::check(ctx, vgicp_scan_prepare(ctx, raw.size(), raw[0].data(), times.data(), f == 0 ? 0 : states.size(),
                                         packed.data(), nullptr, voxel, 30, &kept, &moved), "vgicp_scan_prepare");
      if (f > 0) {
        const Isometry3d guess = toIsometry(motion(end - 0.1));
        shim::check(ctx, vgicp_align_resident(ctx, shim::poseData(guess), &params, estimateB, nullptr), "vgicp_align_resident");
      }
      size_t fresh = 0;
      shim::check(ctx, vgicp_map_insert_resident(ctx, estimateB, 20, &fresh), "vgicp_map_insert_resident");
      const double frameMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
      if (f > 1) {residentMs += frameMs; ++residentFrames;}   // frame 1 pays the process's one-time costs (code objects, first allocations)

      // ---- C: vgicp_scan_prepare_async -> vgicp_align_resident -> vgicp_map_insert_resident_async ----
      vgicp_frame_stats fs{};
      shim::check(ctxC, vgicp_get_frame_stats(ctxC, &fs, 1), "vgicp_get_frame_stats");
      const auto c0 = std::chrono::steady_clock::now();
      shim::check(ctxC, vgicp_scan_prepare_async(ctxC, raw.size(), raw[0].data(), times.data(), f == 0 ? 0 : states.size(),
                                               packed.data(), nullptr, voxel, 30), "vgicp_scan_prepare_async");
      if (f > 0) {
        const Isometry3d guess = toIsometry(motion(end - 0.1));
        shim::check(ctxC, vgicp_align_resident(ctxC, shim::poseData(guess), &params, estimateC, nullptr), "vgicp_align_resident");
      }
      shim::check(ctxC, vgicp_map_insert_resident_async(ctxC, estimateC, 20), "vgicp_map_insert_resident_async");
      const double asyncFrameMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
      shim::check(ctxC, vgicp_get_frame_stats(ctxC, &fs, 0), "vgicp_get_frame_stats");
      if (f > 1) {asyncMs += asyncFrameMs; asyncSyncs += fs.host_syncs; asyncLaunches += fs.kernel_launches;}
      for (int k = 0; k < 16; ++k) {if (estimateC[k] != estimateB[k]) {++bad;}}   // the same bits as chain B
      if (f > 0) {
        for (int k = 0; k < 16; ++k) {if (shim::poseData(estimateA)[k] != estimateC[k]) {++bad;}}   // the classes ride chain C
        double gap2 = 0.0;
        for (int k = 0; k < 16; ++k) {gap2 = std::fmax(gap2, std::fabs(shim::poseData(estimateA2)[k] - estimateB[k]));}
        if (!(gap2 < 1e-9)) {++bad;}
      }

      const double errA = positionError(shim::poseData(estimateA), truth), errB = positionError(estimateB, truth);
      double gap = 0.0;
      for (int k = 0; k < 16; ++k) {gap = std::fmax(gap, std::fabs(shim::poseData(estimateA)[k] - estimateB[k]));}
      std::printf("frame %d: %zu -> %zu points (%lld deskewed); classes: error %.2e m; resident chain: error %.2e m; "
        "difference between the two %.1e\n", f, raw.size(), kept, (long long)moved, errA, errB, gap);
      // chain A2's host cloud holds the prepared scan (moved into the world frame by updateLocalMap, as the reference does)
      if (kept != meas2->cloud->points_.size() || !(errA < 1e-2) || !(errB < 1e-2) || !(gap < 1e-9)) {++bad;}
    }
    if (residentFrames) {
      std::printf("drop-in classes, scan handed along on the device (process / align / updateLocalMap as src/Odometry.cpp:73-87): "
        "%.3f ms per frame, %d of %d aligns found their cloud resident, poses bit-equal to the chain that does not wait\n",
        classesMs / residentFrames, residentAligns, frames - 1);
      std::printf("  of which process() %.3f ms (enqueue only), align() %.3f ms (the frame's one synchronisation), updateLocalMap() %.3f ms (enqueue only)\n",
        stageMs[0] / residentFrames, stageMs[1] / residentFrames, stageMs[2] / residentFrames);
      std::printf("drop-in classes with the host-authoritative map (eager host copy): %.3f ms per frame\n", classesEagerMs / residentFrames);
      if (residentAligns != frames - 1) {++bad;}
    }
    if (residentFrames) {
      std::printf("resident chain (raw sweep in, pose out, map updated): %.3f ms per %zu-point frame on average over %d frames (from the third on)\n",
        residentMs / residentFrames, n, residentFrames);
    }
    if (residentFrames) {
      std::printf("the same without waiting (prepare_async / align / insert_async): %.3f ms per frame, %.1f kernel launches and "
        "%.1f host synchronisations per frame, poses bit-equal to the resident chain\n", asyncMs / residentFrames,
        (double)asyncLaunches / residentFrames, (double)asyncSyncs / residentFrames);
    }
    size_t voxelsB = 0, voxelsC = 0;
    shim::check(ctx, vgicp_map_size(ctx, &voxelsB, nullptr), "vgicp_map_size");
    shim::check(ctxC, vgicp_map_size(ctxC, &voxelsC, nullptr), "vgicp_map_size");
    if (voxelsB != voxelsC) {++bad;}
    size_t voxelsA = localMap.size();
    if (voxelsA != voxelsC) {++bad;}
    vgicp_destroy(ctx);
    vgicp_destroy(ctxC);
    vgicp_destroy(ctxA2);
    return bad == 0 ? 0 : 2;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "frame_chain: %s\n", e.what());
    return 1;   // e.g. no gfx950 device: there is no CPU fallback
  }
}
