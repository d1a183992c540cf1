// register_frames.cpp — the reference's per-frame sequence written against the drop-in shim, in C++,
// the way src/Odometry.cpp:55-87 and src/ErrorStateKF.cpp:127-130 use the two classes:
//
//     localMap->updateLocalMap(cloud0, Identity)              // first frame builds the map
//     for each later frame:
//         T = icp->align(*cloud, *localMap, guess)            // ErrorStateKF::update
//         localMap->updateLocalMap(cloud, T)                  // Odometry::run
//
// No Eigen / Open3D / ROS here: ShimTypes.hpp supplies layout-identical stand-ins; with those libraries
// installed the same source compiles against the reference's own types.  A synthetic world (random
// planar patches on a voxel lattice) is scanned from a sensor that moves a few centimetres per frame;
// the program prints the estimated and the true pose per frame and exits non-zero if tracking is lost.
//
// build:  make -C examples   (plain g++; links eskf_lio_amd/lib/libvgicp_hip.so)
#define ESKF_LIO_SHIM_FORCE_POD 1
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

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

Isometry3d poseFrom(double x, double y, double z, double yaw)
{
  Isometry3d T = Isometry3d::Identity();
  T.matrix()(0, 0) = std::cos(yaw); T.matrix()(0, 1) = -std::sin(yaw);
  T.matrix()(1, 0) = std::sin(yaw); T.matrix()(1, 1) = std::cos(yaw);
  T.matrix()(0, 3) = x; T.matrix()(1, 3) = y; T.matrix()(2, 3) = z;
  return T;
}

struct World
{
  std::vector<Vector3d> points;
  std::vector<Matrix3d> covs;
};

World makeWorld(size_t n, uint64_t seed)
{
  World w;
  w.points.resize(n);
  w.covs.resize(n);
  for (size_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) {w.points[i](a) = -9.0 + 18.0 * unit(seed);}
    // disc-shaped covariance I - 0.99 n n^T with a random unit normal
    double nx, ny, nz, s;
    do {
      nx = 2.0 * unit(seed) - 1.0; ny = 2.0 * unit(seed) - 1.0; nz = 2.0 * unit(seed) - 1.0;
      s = nx * nx + ny * ny + nz * nz;
    } while (s > 1.0 || s < 1e-6);
    const double inv = 1.0 / std::sqrt(s);
    const double nrm[3] = {nx * inv, ny * inv, nz * inv};
    for (int c = 0; c < 3; ++c) {
      for (int r = 0; r < 3; ++r) {w.covs[i](r, c) = (r == c ? 1.0 : 0.0) - 0.99 * nrm[r] * nrm[c];}
    }
  }
  return w;
}

// what the sensor at pose T sees of a random subset of the world: p_sensor = T^-1 p_world (+ noise)
PointCloudPtr scanFrom(const World & w, const Isometry3d & T, size_t n, uint64_t seed)
{
  auto cloud = std::make_shared<PointCloud>();
  const Isometry3d Tinv = T.inverse();
  cloud->points_.resize(n);
  cloud->covariances_.resize(n);
  for (size_t k = 0; k < n; ++k) {
    const size_t i = splitmix(seed) % w.points.size();
    cloud->points_[k] = w.points[i];
    for (int a = 0; a < 3; ++a) {cloud->points_[k](a) += 0.005 * (unit(seed) - 0.5);}
    cloud->covariances_[k] = w.covs[i];
  }
  cloud->Transform(Tinv.matrix());
  return cloud;
}
}  // namespace

int main()
{
  try {
    RegistrationConfig rc;          // registration.* of config/hilti_config.yaml:50-53
    rc.maxIteration = 100;
    rc.translationSquaredThreshold = 1e-6;
    rc.cosineThreshold = 0.9999;
    ICP icp(rc);
    LocalMap localMap(0.3, 20);     // the reference's test-friendly constructor (LocalMap.hpp:54-61)

    const World world = makeWorld(60000, 42);
    Isometry3d estimate = Isometry3d::Identity();
    int lost = 0;
    for (int f = 0; f < 6; ++f) {
      const Isometry3d truth = poseFrom(0.10 * f, 0.04 * f, 0.01 * f, 0.01 * f);
      PointCloudPtr cloud = scanFrom(world, truth, 8000, 1000 + f);
      if (f == 0) {
        localMap.updateLocalMap(cloud, Isometry3d::Identity());   // src/Odometry.cpp:61
        std::printf("frame 0: map initialised with %zu voxels\n", localMap.size());
        continue;
      }
      estimate = icp.align(*cloud, localMap, estimate);           // src/ErrorStateKF.cpp:130
      const double ex = estimate.matrix()(0, 3) - truth.matrix()(0, 3);
      const double ey = estimate.matrix()(1, 3) - truth.matrix()(1, 3);
      const double ez = estimate.matrix()(2, 3) - truth.matrix()(2, 3);
      const double err = std::sqrt(ex * ex + ey * ey + ez * ez);
      std::printf("frame %d: %d iterations, converged %d, position (%.4f %.4f %.4f), error %.2e m, map %zu voxels\n",
        f, icp.lastStats().iterations, (int)icp.lastStats().converged, estimate.matrix()(0, 3),
        estimate.matrix()(1, 3), estimate.matrix()(2, 3), err, localMap.size());
      if (!(err < 5e-3) || !icp.lastStats().converged) {++lost;}
      localMap.updateLocalMap(cloud, estimate);                   // src/Odometry.cpp:86
    }
    return lost == 0 ? 0 : 2;
  } catch (const std::exception & e) {
    std::fprintf(stderr, "register_frames: %s\n", e.what());
    return 1;   // e.g. no gfx950 device: there is no CPU fallback
  }
}
