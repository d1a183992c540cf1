// ShimTypes.hpp — boundary types of the drop-in shim.
//
// With Eigen and Open3D installed (the reference's own build environment) the shim uses the
// reference's types unchanged: ESKF_LIO::PointCloud = open3d::geometry::PointCloud and Eigen's
// Vector3d / Matrix3d / Isometry3d (reference include/ESKF_LIO/Types.hpp:6-12).  Where they are
// absent — this repository's build and GPU hosts have neither — it falls back to layout-identical
// plain structs exposing the few members the path touches (points_, covariances_, Transform(),
// matrix().data(), linear(), translation()), so the same LocalMap / ICP code compiles and is tested.
#ifndef ESKF_LIO_SHIM_SHIMTYPES_HPP_
#define ESKF_LIO_SHIM_SHIMTYPES_HPP_

#include <cstring>
#include <memory>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<open3d/Open3D.h>) && !defined(ESKF_LIO_SHIM_FORCE_POD)
#define ESKF_LIO_SHIM_NATIVE_TYPES 1
#endif
#endif

#if defined(ESKF_LIO_SHIM_NATIVE_TYPES)
#include <Eigen/Dense>
#include <open3d/Open3D.h>
#endif

namespace ESKF_LIO
{
#if defined(ESKF_LIO_SHIM_NATIVE_TYPES)

using PointCloud = open3d::geometry::PointCloud;
using Vector3d = Eigen::Vector3d;
using Vector3i = Eigen::Vector3i;
using Matrix3d = Eigen::Matrix3d;
using Isometry3d = Eigen::Isometry3d;
using Quaterniond = Eigen::Quaterniond;

namespace shim
{
inline const double * quatData(const Quaterniond & q) {return q.coeffs().data();}  // x y z w
inline const double * poseData(const Isometry3d & T) {return T.matrix().data();}
inline Isometry3d poseFromData(const double * m16)
{
  Isometry3d T;
  T.matrix() = Eigen::Map<const Eigen::Matrix4d>(m16);
  return T;
}
}  // namespace shim

#else  // ---- dependency-free stand-ins, byte-compatible with the Eigen/Open3D containers ----

struct Vector3d
{
  double v[3];
  double & operator()(int i) {return v[i];}
  double operator()(int i) const {return v[i];}
  const double * data() const {return v;}
  double * data() {return v;}
};
struct Vector3i
{
  int v[3];
  int & operator()(int i) {return v[i];}
  int operator()(int i) const {return v[i];}
};
struct Matrix3d
{
  double m[9];  // column-major, as Eigen::Matrix3d
  double & operator()(int r, int c) {return m[r + 3 * c];}
  double operator()(int r, int c) const {return m[r + 3 * c];}
  const double * data() const {return m;}
  double * data() {return m;}
};
static_assert(sizeof(Vector3d) == 24 && sizeof(Matrix3d) == 72, "must match Eigen's dense storage");

struct Matrix4d
{
  double m[16];  // column-major
  double & operator()(int r, int c) {return m[r + 4 * c];}
  double operator()(int r, int c) const {return m[r + 4 * c];}
  const double * data() const {return m;}
  double * data() {return m;}
};

class Isometry3d
{
public:
  Isometry3d() {*this = Identity();}
  static Isometry3d Identity()
  {
    Isometry3d T(0);
    std::memset(T.m_.m, 0, sizeof T.m_.m);
    T.m_(0, 0) = T.m_(1, 1) = T.m_(2, 2) = T.m_(3, 3) = 1.0;
    return T;
  }
  const Matrix4d & matrix() const {return m_;}
  Matrix4d & matrix() {return m_;}
  Matrix3d linear() const
  {
    Matrix3d R;
    for (int c = 0; c < 3; ++c) {
      for (int r = 0; r < 3; ++r) {R(r, c) = m_(r, c);}
    }
    return R;
  }
  Vector3d translation() const {return Vector3d{{m_(0, 3), m_(1, 3), m_(2, 3)}};}
  // T = this * rhs
  Isometry3d operator*(const Isometry3d & rhs) const
  {
    Isometry3d out = Identity();
    for (int c = 0; c < 4; ++c) {
      for (int r = 0; r < 3; ++r) {
        double s = m_(r, 0) * rhs.m_(0, c) + m_(r, 1) * rhs.m_(1, c) + m_(r, 2) * rhs.m_(2, c);
        if (c == 3) {s += m_(r, 3);}
        out.m_(r, c) = s;
      }
    }
    return out;
  }
  Isometry3d inverse() const
  {
    Isometry3d out = Identity();
    for (int c = 0; c < 3; ++c) {
      for (int r = 0; r < 3; ++r) {out.m_(r, c) = m_(c, r);}
    }
    for (int r = 0; r < 3; ++r) {
      out.m_(r, 3) = -(out.m_(r, 0) * m_(0, 3) + out.m_(r, 1) * m_(1, 3) + out.m_(r, 2) * m_(2, 3));
    }
    return out;
  }

private:
  explicit Isometry3d(int) {}
  Matrix4d m_;
};

// The members of open3d::geometry::PointCloud the path touches.
struct PointCloud
{
  std::vector<Vector3d> points_;
  std::vector<Matrix3d> covariances_;

  // Open3D semantics: p <- (T [p;1]).xyz / w;  C <- R C R^T  (serial, in place)
  PointCloud & Transform(const Matrix4d & T)
  {
    for (auto & p : points_) {
      double q[4];
      for (int r = 0; r < 4; ++r) {
        q[r] = T(r, 0) * p(0) + T(r, 1) * p(1) + T(r, 2) * p(2) + T(r, 3);
      }
      p = Vector3d{{q[0] / q[3], q[1] / q[3], q[2] / q[3]}};
    }
    for (auto & C : covariances_) {
      Matrix3d RC, out;
      for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) {
          RC(r, c) = T(r, 0) * C(0, c) + T(r, 1) * C(1, c) + T(r, 2) * C(2, c);
        }
      }
      for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) {
          out(r, c) = RC(r, 0) * T(c, 0) + RC(r, 1) * T(c, 1) + RC(r, 2) * T(c, 2);
        }
      }
      C = out;
    }
    return *this;
  }
};

namespace shim
{
inline const double * poseData(const Isometry3d & T) {return T.matrix().data();}
inline Isometry3d poseFromData(const double * m16)
{
  Isometry3d T;
  std::memcpy(T.matrix().data(), m16, 16 * sizeof(double));
  return T;
}
}  // namespace shim

// Eigen::Quaterniond's storage: coefficients x, y, z, w
struct Quaterniond
{
  double c[4] = {0.0, 0.0, 0.0, 1.0};
};

namespace shim
{
inline const double * quatData(const Quaterniond & q) {return q.c;}
}  // namespace shim

#endif

using PointCloudPtr = std::shared_ptr<PointCloud>;

#if !defined(ESKF_LIO_SHIM_NATIVE_TYPES)
// Stand-ins for the parts of the reference's include/ESKF_LIO/Types.hpp the scan preparation reads
// (State: Types.hpp:31-40, LidarMeasurement: Types.hpp:22-29). With the native types the reference's own
// Types.hpp provides them.
struct State
{
  double timestamp = 0.0;
  Vector3d position{};
  Vector3d velocity{};
  Quaterniond attitude{};
};

struct LidarMeasurement
{
  PointCloudPtr cloud;
  std::vector<double> pointTime;
  double startTime = 0.0;
  double endTime = 0.0;
};
using LidarMeasurementPtr = std::shared_ptr<LidarMeasurement>;
#endif

}  // namespace ESKF_LIO

#endif  // ESKF_LIO_SHIM_SHIMTYPES_HPP_
