// vgicp_oracle.cpp — CPU restatement of the reference's VGICP scan-to-map registration.
//
// TEST INFRASTRUCTURE ONLY (see vgicp_oracle.h).  PARITY UNPINNED: the reference holds no tests or
// golden vectors for this path and cannot be built here; this file is pinned by analytic
// known-answer tests and by an independent numpy restatement.
//
// What is restated, and from where (paths relative to the reference checkout):
//   align loop            src/Registration.cpp:7-35
//   convergence rule      src/Registration.cpp:37-50
//   normal equations      src/Registration.cpp:52-81 (loop + LDLT solve), :83-102 (one residual)
//   voxel lookup          src/LocalMap.cpp:78-118
//   voxel statistics      include/ESKF_LIO/LocalMap.hpp:63-89, insertion loop src/LocalMap.cpp:47-58
//   so(3)/se(3) helpers   src/Utils.cpp:5-11, 28-32, 40-63
// Third-party arithmetic the reference calls but does not vendor (no pinned version; Eigen must be
// >= 3.4, Open3D >= 0.13), restated from the published algorithms:
//   Eigen  Matrix3d::inverse (cofactors), LDLT<6x6> with diagonal pivoting + pseudo-inverse of D,
//          AngleAxisd::toRotationMatrix, normalized() (no-op on the zero vector), Isometry product
//   Open3D PointCloud::Transform (homogeneous product then /w; covariance R C R^T; serial loops),
//          utility::hash_eigen (boost-style combiner; affects iteration order only)
//
// Built by oracle/Makefile with the reference's flags: -O3, OpenMP, no -march (CMakeLists.txt:6,19,41),
// so there is no FMA contraction on x86-64 and every product/sum below rounds once, as in Eigen.
#include "vgicp_oracle.h"

#include <omp.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct V3 {
  double x, y, z;
};
struct M3 {
  double a[9];  // column-major: a[r + 3c]
  double& operator()(int r, int c) { return a[r + 3 * c]; }
  double operator()(int r, int c) const { return a[r + 3 * c]; }
};
struct M4 {
  double a[16];  // column-major: a[r + 4c]
  double& operator()(int r, int c) { return a[r + 4 * c]; }
  double operator()(int r, int c) const { return a[r + 4 * c]; }
};
using M6 = std::array<double, 36>;  // column-major 6x6
using V6 = std::array<double, 6>;

static_assert(sizeof(V3) == 24 && sizeof(M3) == 72, "layout must match Eigen's dense storage");

inline M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      double s = A(r, 0) * B(0, c);
      s += A(r, 1) * B(1, c);
      s += A(r, 2) * B(2, c);
      C(r, c) = s;
    }
  return C;
}
inline M3 transpose(const M3& A) {
  M3 T;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T(r, c) = A(c, r);
  return T;
}
inline M3 add(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] + B.a[i];
  return C;
}
inline V3 mul(const M3& A, const V3& v) {
  return {A(0, 0) * v.x + A(0, 1) * v.y + A(0, 2) * v.z,
          A(1, 0) * v.x + A(1, 1) * v.y + A(1, 2) * v.z,
          A(2, 0) * v.x + A(2, 1) * v.y + A(2, 2) * v.z};
}
inline M3 identity3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M4 identity4() {
  M4 I{};
  I(0, 0) = I(1, 1) = I(2, 2) = I(3, 3) = 1.0;
  return I;
}
inline M3 rotation_of(const M4& T) {
  M3 R;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) R(r, c) = T(r, c);
  return R;
}

// Utils::skewSymmetric — src/Utils.cpp:5-11
inline M3 hat(const V3& v) {
  M3 S{};
  S(0, 1) = -v.z; S(0, 2) = v.y;
  S(1, 0) = v.z;  S(1, 2) = -v.x;
  S(2, 0) = -v.y; S(2, 1) = v.x;
  return S;
}

// Eigen fixed-size 3x3 inverse: adjugate times 1/det, det expanded along column 0.
inline double cof(const M3& m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
}
inline M3 inverse(const M3& m) {
  const double k0 = cof(m, 0, 0), k1 = cof(m, 1, 0), k2 = cof(m, 2, 0);
  const double det = k0 * m(0, 0) + k1 * m(1, 0) + k2 * m(2, 0);
  const double invdet = 1.0 / det;
  M3 R;
  R(0, 0) = k0 * invdet; R(0, 1) = k1 * invdet; R(0, 2) = k2 * invdet;
  R(1, 0) = cof(m, 0, 1) * invdet; R(1, 1) = cof(m, 1, 1) * invdet; R(1, 2) = cof(m, 2, 1) * invdet;
  R(2, 0) = cof(m, 0, 2) * invdet; R(2, 1) = cof(m, 1, 2) * invdet; R(2, 2) = cof(m, 2, 2) * invdet;
  return R;
}

// Isometry3d * Isometry3d (affine-compact product, last row stays 0 0 0 1).
inline M4 compose(const M4& A, const M4& B) {
  M4 C = identity4();
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) {
      double s = A(r, 0) * B(0, c);
      s += A(r, 1) * B(1, c);
      s += A(r, 2) * B(2, c);
      if (c == 3) s += A(r, 3);
      C(r, c) = s;
    }
  return C;
}

// AngleAxisd(r.norm(), r.normalized()).toRotationMatrix() — src/Utils.cpp:28-32
inline M3 rotation_from_vector(const V3& r) {
  const double sq = r.x * r.x + r.y * r.y + r.z * r.z;
  const double angle = std::sqrt(sq);
  V3 k = r;
  if (sq > 0.0) k = {r.x / angle, r.y / angle, r.z / angle};
  const double s = std::sin(angle), c = std::cos(angle);
  const V3 sk = {s * k.x, s * k.y, s * k.z};
  const V3 ck = {(1.0 - c) * k.x, (1.0 - c) * k.y, (1.0 - c) * k.z};
  M3 R;
  double t;
  t = ck.x * k.y; R(0, 1) = t - sk.z; R(1, 0) = t + sk.z;
  t = ck.x * k.z; R(0, 2) = t + sk.y; R(2, 0) = t - sk.y;
  t = ck.y * k.z; R(1, 2) = t - sk.x; R(2, 1) = t + sk.x;
  R(0, 0) = ck.x * k.x + c;
  R(1, 1) = ck.y * k.y + c;
  R(2, 2) = ck.z * k.z + c;
  return R;
}

// Utils::computeJ — src/Utils.cpp:40-54
inline M3 so3_left_jacobian(const V3& r) {
  const double sq = r.x * r.x + r.y * r.y + r.z * r.z;
  const double angle = std::sqrt(sq);
  if (angle < 1e-6) return identity3();
  const V3 k = {r.x / angle, r.y / angle, r.z / angle};
  const double f1 = std::sin(angle) / angle;
  const double f2 = (1.0 - std::cos(angle)) / angle;
  const double kk[3] = {k.x, k.y, k.z};
  const M3 K = hat(k);
  M3 J;
  for (int c = 0; c < 3; ++c)
    for (int rr = 0; rr < 3; ++rr)
      J(rr, c) = f1 * (rr == c ? 1.0 : 0.0) + ((1.0 - f1) * kk[rr]) * kk[c] + f2 * K(rr, c);
  return J;
}

// Utils::se3ToSE3 — src/Utils.cpp:56-63; se3 = [rho; phi]
inline M4 exp_se3(const V6& xi) {
  const V3 rho = {xi[0], xi[1], xi[2]}, phi = {xi[3], xi[4], xi[5]};
  const M3 J = so3_left_jacobian(phi);
  const V3 t = mul(J, rho);
  const M3 R = rotation_from_vector(phi);
  M4 T = identity4();
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T(r, c) = R(r, c);
  T(0, 3) = t.x; T(1, 3) = t.y; T(2, 3) = t.z;
  return T;
}

// Eigen LDLT<Matrix<double,6,6>, Lower>::compute + solve (unblocked, diagonal pivoting).
V6 ldlt_solve(const M6& Ain, const V6& rhs) {
  constexpr int n = 6;
  double A[n][n];
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < n; ++r) A[r][c] = Ain[r + 6 * c];
  int tr[n];
  for (int i = 0; i < n; ++i) tr[i] = i;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double bigv = std::fabs(A[k][k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(A[i][i]) > bigv) { bigv = std::fabs(A[i][i]); big = i; }
    tr[k] = big;
    if (big != k) {
      for (int j = 0; j < k; ++j) std::swap(A[k][j], A[big][j]);
      for (int i = big + 1; i < n; ++i) std::swap(A[i][k], A[i][big]);
      std::swap(A[k][k], A[big][big]);
      for (int i = k + 1; i < big; ++i) std::swap(A[i][k], A[big][i]);
    }
    if (k > 0) {
      double temp[n];
      for (int j = 0; j < k; ++j) temp[j] = A[j][j] * A[k][j];
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += A[k][j] * temp[j];
      A[k][k] -= dot;
      for (int i = k + 1; i < n; ++i) {
        double d = 0.0;
        for (int j = 0; j < k; ++j) d += A[i][j] * temp[j];
        A[i][k] -= d;
      }
    }
    const double pivot = A[k][k];
    const bool ok = std::fabs(pivot) > 0.0;
    if (k == 0 && !ok) {
      for (int j = 0; j < n; ++j) tr[j] = j;
      break;
    }
    if (ok)
      for (int i = k + 1; i < n; ++i) A[i][k] /= pivot;
  }
  V6 x = rhs;
  for (int k = 0; k < n; ++k) std::swap(x[k], x[tr[k]]);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) x[i] -= A[i][j] * x[j];
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < n; ++i) x[i] = std::fabs(A[i][i]) > tol ? x[i] / A[i][i] : 0.0;
  for (int i = n - 1; i >= 0; --i)
    for (int j = i + 1; j < n; ++j) x[i] -= A[j][i] * x[j];
  for (int k = n - 1; k >= 0; --k) std::swap(x[k], x[tr[k]]);
  return x;
}

// ICP::computeJTJAndJTr — src/Registration.cpp:83-102.  J = [I | -hat(p)], dense products as Eigen
// evaluates them: JT = J^T * inv(cov) (6x3), JTJ = JT * J (6x6), JTr = JT * (p - mu).
void residual_blocks(const V3& p, const V3& mu, const M3& cov, M6& JTJ, V6& JTr) {
  double J[3][6] = {};
  const M3 S = hat(p);
  for (int r = 0; r < 3; ++r) {
    J[r][r] = 1.0;
    for (int c = 0; c < 3; ++c) J[r][3 + c] = -S(r, c);
  }
  const M3 W = inverse(cov);
  double JT[6][3];
  for (int i = 0; i < 6; ++i)
    for (int c = 0; c < 3; ++c) {
      double s = J[0][i] * W(0, c);
      s += J[1][i] * W(1, c);
      s += J[2][i] * W(2, c);
      JT[i][c] = s;
    }
  const double res[3] = {p.x - mu.x, p.y - mu.y, p.z - mu.z};
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      double s = JT[i][0] * J[0][j];
      s += JT[i][1] * J[1][j];
      s += JT[i][2] * J[2][j];
      JTJ[i + 6 * j] = s;
    }
    double s = JT[i][0] * res[0];
    s += JT[i][1] * res[1];
    s += JT[i][2] * res[2];
    JTr[i] = s;
  }
}

// ICP::convergenceCheck — src/Registration.cpp:37-50
bool step_converged(const M4& step, double cosine_threshold, double translation_sq_threshold) {
  const double cosine = 0.5 * ((step(0, 0) + step(1, 1) + step(2, 2)) - 1.0);
  if (cosine < cosine_threshold) return false;
  const double tsq = step(0, 3) * step(0, 3) + step(1, 3) * step(1, 3) + step(2, 3) * step(2, 3);
  if (tsq > translation_sq_threshold) return false;
  return true;
}

// Open3D PointCloud::Transform, serial as upstream.
void transform_cloud(std::vector<V3>& pts, std::vector<M3>& covs, const M4& T) {
  for (auto& p : pts) {
    double q[4];
    for (int r = 0; r < 4; ++r) q[r] = T(r, 0) * p.x + T(r, 1) * p.y + T(r, 2) * p.z + T(r, 3) * 1.0;
    p = {q[0] / q[3], q[1] / q[3], q[2] / q[3]};
  }
  const M3 R = rotation_of(T), Rt = transpose(R);
  for (auto& C : covs) C = mul(mul(R, C), Rt);
}

struct Key {
  int32_t i, j, k;
  bool operator==(const Key& o) const { return i == o.i && j == o.j && k == o.k; }
};
// open3d::utility::hash_eigen<Vector3i>: boost-style combine of std::hash<int> over the 3 entries.
struct KeyHash {
  size_t operator()(const Key& key) const {
    size_t seed = 0;
    const int32_t e[3] = {key.i, key.j, key.k};
    for (int n = 0; n < 3; ++n)
      seed ^= std::hash<int>()(e[n]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};

// LocalMap::Voxel — include/ESKF_LIO/LocalMap.hpp:63-89
struct Voxel {
  size_t cap, count;
  std::vector<V3> samples;
  V3 mean;
  M3 cov;
  Voxel(size_t cap_, const V3& p, const M3& C) : cap(cap_), count(1), mean(p), cov(C) {
    samples.reserve(cap);
    samples.push_back(p);
  }
  void add(const V3& p, const M3& C) {
    if (count >= cap) return;
    samples.push_back(p);
    const double n = static_cast<double>(count), n1 = static_cast<double>(count + 1);
    mean = {(n * mean.x + p.x) / n1, (n * mean.y + p.y) / n1, (n * mean.z + p.z) / n1};
    for (int e = 0; e < 9; ++e) cov.a[e] = (n * cov.a[e] + C.a[e]) / n1;
    ++count;
  }
};

inline Key voxel_key(const V3& p, double voxel_size) {
  return {static_cast<int32_t>(std::floor(p.x / voxel_size)),
          static_cast<int32_t>(std::floor(p.y / voxel_size)),
          static_cast<int32_t>(std::floor(p.z / voxel_size))};
}

}  // namespace

struct oracle_map {
  double voxel_size;
  size_t cap;
  std::unordered_map<Key, Voxel, KeyHash> grid;
};

namespace {

struct Matches {
  std::vector<V3> src_pts, map_pts;
  std::vector<M3> src_covs, map_covs;
};

// LocalMap::correspondenceMatching as the reference runs it (thread-private buffers, arrival-order
// concatenation under a critical section).
Matches match_faithful(const oracle_map& map, const std::vector<V3>& pts, const std::vector<M3>& covs) {
  Matches out;
  out.src_pts.reserve(pts.size());
  out.src_covs.reserve(pts.size());
  out.map_pts.reserve(pts.size());
  out.map_covs.reserve(pts.size());
#pragma omp parallel
  {
    Matches mine;
#pragma omp for nowait
    for (size_t i = 0; i < pts.size(); ++i) {
      auto it = map.grid.find(voxel_key(pts[i], map.voxel_size));
      if (it == map.grid.cend()) continue;
      mine.src_pts.push_back(pts[i]);
      mine.src_covs.push_back(covs[i]);
      mine.map_pts.push_back(it->second.mean);
      mine.map_covs.push_back(it->second.cov);
    }
#pragma omp critical
    {
      out.src_pts.insert(out.src_pts.end(), mine.src_pts.begin(), mine.src_pts.end());
      out.src_covs.insert(out.src_covs.end(), mine.src_covs.begin(), mine.src_covs.end());
      out.map_pts.insert(out.map_pts.end(), mine.map_pts.begin(), mine.map_pts.end());
      out.map_covs.insert(out.map_covs.end(), mine.map_covs.begin(), mine.map_covs.end());
    }
  }
  return out;
}

void sum_faithful(const Matches& m, M6& JTJ, V6& JTr) {
  JTJ.fill(0.0);
  JTr.fill(0.0);
  const size_t count = m.src_pts.size();
#pragma omp parallel
  {
    M6 accJ;
    V6 accr;
    accJ.fill(0.0);
    accr.fill(0.0);
#pragma omp for nowait
    for (size_t i = 0; i < count; ++i) {
      M6 Ji;
      V6 ri;
      residual_blocks(m.src_pts[i], m.map_pts[i], add(m.src_covs[i], m.map_covs[i]), Ji, ri);
      for (int e = 0; e < 36; ++e) accJ[e] += Ji[e];
      for (int e = 0; e < 6; ++e) accr[e] += ri[e];
    }
#pragma omp critical
    {
      for (int e = 0; e < 36; ++e) JTJ[e] += accJ[e];
      for (int e = 0; e < 6; ++e) JTr[e] += accr[e];
    }
  }
}

// Same arithmetic, one pass in ascending point order, nothing materialised.
size_t sum_deterministic(const oracle_map& map, const V3* pts, const M3* covs, size_t n, M6& JTJ,
                         V6& JTr) {
  JTJ.fill(0.0);
  JTr.fill(0.0);
  size_t hits = 0;
  for (size_t i = 0; i < n; ++i) {
    auto it = map.grid.find(voxel_key(pts[i], map.voxel_size));
    if (it == map.grid.cend()) continue;
    M6 Ji;
    V6 ri;
    residual_blocks(pts[i], it->second.mean, add(covs[i], it->second.cov), Ji, ri);
    for (int e = 0; e < 36; ++e) JTJ[e] += Ji[e];
    for (int e = 0; e < 6; ++e) JTr[e] += ri[e];
    ++hits;
  }
  return hits;
}

M4 solve_step(const M6& JTJ, const V6& JTr, V6* se3_out) {
  V6 neg;
  for (int e = 0; e < 6; ++e) neg[e] = -JTr[e];
  const V6 se3 = ldlt_solve(JTJ, neg);
  if (se3_out) *se3_out = se3;
  return exp_se3(se3);
}

M4 load4(const double* p) {
  M4 T;
  std::memcpy(T.a, p, sizeof T.a);
  return T;
}

}  // namespace

extern "C" {

oracle_map* oracle_map_create(double voxel_size, size_t max_points_per_voxel) {
  auto* m = new oracle_map;
  m->voxel_size = voxel_size;
  m->cap = max_points_per_voxel;
  return m;
}
void oracle_map_destroy(oracle_map* map) { delete map; }
size_t oracle_map_size(const oracle_map* map) { return map->grid.size(); }

void oracle_map_insert(oracle_map* map, size_t n, const double* points, const double* covs) {
  const V3* P = reinterpret_cast<const V3*>(points);
  const M3* C = reinterpret_cast<const M3*>(covs);
  for (size_t i = 0; i < n; ++i) {
    const Key key = voxel_key(P[i], map->voxel_size);
    auto it = map->grid.find(key);
    if (it == map->grid.end()) {
      map->grid.emplace(key, Voxel(map->cap, P[i], C[i]));
    } else {
      it->second.add(P[i], C[i]);
    }
  }
}

// The eviction loop of LocalMap::updateLocalMap (reference src/LocalMap.cpp:60-72) with needsPointRemoval (:149-154):
// every voxel whose centre (index + 0.5) * voxelSize is farther than distance_threshold from `position` is erased.
// (voxelCenter - currentPos).norm(): Eigen's norm() is sqrt(squaredNorm()), the squares added left to right.
size_t oracle_map_evict(oracle_map* map, const double position[3], double distance_threshold) {
  size_t removed = 0;
  for (auto it = map->grid.begin(); it != map->grid.end();) {
    const double cx = ((double)it->first.i + 0.5) * map->voxel_size - position[0];
    const double cy = ((double)it->first.j + 0.5) * map->voxel_size - position[1];
    const double cz = ((double)it->first.k + 0.5) * map->voxel_size - position[2];
    const double distance = std::sqrt(cx * cx + cy * cy + cz * cz);
    if (distance > distance_threshold) {
      it = map->grid.erase(it);
      ++removed;
    } else {
      ++it;
    }
  }
  return removed;
}

size_t oracle_map_export(const oracle_map* map, size_t capacity, int32_t* keys, double* means,
                         double* covs, uint64_t* counts) {
  size_t w = 0;
  for (const auto& kv : map->grid) {
    if (w == capacity) break;
    keys[3 * w + 0] = kv.first.i;
    keys[3 * w + 1] = kv.first.j;
    keys[3 * w + 2] = kv.first.k;
    std::memcpy(means + 3 * w, &kv.second.mean, 24);
    std::memcpy(covs + 9 * w, kv.second.cov.a, 72);
    if (counts) counts[w] = kv.second.count;
    ++w;
  }
  return w;
}

void oracle_voxel_index(double voxel_size, size_t n, const double* points, int32_t* keys) {
  const V3* P = reinterpret_cast<const V3*>(points);
  for (size_t i = 0; i < n; ++i) {
    const Key k = voxel_key(P[i], voxel_size);
    keys[3 * i] = k.i;
    keys[3 * i + 1] = k.j;
    keys[3 * i + 2] = k.k;
  }
}

size_t oracle_match(const oracle_map* map, size_t n, const double* points, const double* covs,
                    double* src_points, double* src_covs, double* map_points, double* map_covs,
                    uint64_t* src_index) {
  const V3* P = reinterpret_cast<const V3*>(points);
  const M3* C = reinterpret_cast<const M3*>(covs);
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    auto it = map->grid.find(voxel_key(P[i], map->voxel_size));
    if (it == map->grid.cend()) continue;
    std::memcpy(src_points + 3 * m, &P[i], 24);
    std::memcpy(src_covs + 9 * m, &C[i], 72);
    std::memcpy(map_points + 3 * m, &it->second.mean, 24);
    std::memcpy(map_covs + 9 * m, it->second.cov.a, 72);
    if (src_index) src_index[m] = i;
    ++m;
  }
  return m;
}

void oracle_jtj_jtr(const double src_point[3], const double map_point[3], const double cov[9],
                    double JTJ[36], double JTr[6]) {
  V3 p, mu;
  M3 C;
  std::memcpy(&p, src_point, 24);
  std::memcpy(&mu, map_point, 24);
  std::memcpy(C.a, cov, 72);
  M6 J;
  V6 r;
  residual_blocks(p, mu, C, J, r);
  std::memcpy(JTJ, J.data(), sizeof(double) * 36);
  std::memcpy(JTr, r.data(), sizeof(double) * 6);
}

size_t oracle_accumulate(const oracle_map* map, size_t n, const double* points, const double* covs,
                         double JTJ[36], double JTr[6]) {
  M6 J;
  V6 r;
  const size_t m = sum_deterministic(*map, reinterpret_cast<const V3*>(points),
                                     reinterpret_cast<const M3*>(covs), n, J, r);
  std::memcpy(JTJ, J.data(), sizeof(double) * 36);
  std::memcpy(JTr, r.data(), sizeof(double) * 6);
  return m;
}

void oracle_solve_step(const double JTJ[36], const double JTr[6], double se3[6], double step16[16]) {
  M6 J;
  V6 r, xi;
  std::memcpy(J.data(), JTJ, sizeof(double) * 36);
  std::memcpy(r.data(), JTr, sizeof(double) * 6);
  const M4 T = solve_step(J, r, &xi);
  if (se3) std::memcpy(se3, xi.data(), sizeof(double) * 6);
  std::memcpy(step16, T.a, sizeof T.a);
}

void oracle_se3_to_SE3(const double se3[6], double out16[16]) {
  V6 xi;
  std::memcpy(xi.data(), se3, sizeof(double) * 6);
  const M4 T = exp_se3(xi);
  std::memcpy(out16, T.a, sizeof T.a);
}

int oracle_convergence_check(const double step16[16], double cosine_threshold,
                             double translation_sq_threshold) {
  return step_converged(load4(step16), cosine_threshold, translation_sq_threshold) ? 1 : 0;
}

void oracle_transform(size_t n, double* points, double* covs, const double T16[16]) {
  std::vector<V3> P(n);
  std::vector<M3> C(n);
  std::memcpy(P.data(), points, n * 24);
  std::memcpy(C.data(), covs, n * 72);
  transform_cloud(P, C, load4(T16));
  std::memcpy(points, P.data(), n * 24);
  std::memcpy(covs, C.data(), n * 72);
}

int oracle_align(const oracle_map* map, size_t n, const double* points, const double* covs,
                 const double guess16[16], int max_iteration, double translation_sq_threshold,
                 double cosine_threshold, int mode, double out_pose16[16],
                 oracle_align_stats* stats) {
  const double t0 = omp_get_wtime();
  // working copy of the scan, moved by the guess first
  std::vector<V3> pts(n);
  std::vector<M3> cvs(n);
  std::memcpy(pts.data(), points, n * 24);
  std::memcpy(cvs.data(), covs, n * 72);
  M4 total = load4(guess16);
  transform_cloud(pts, cvs, total);

  bool done = false;
  int rounds = 0;
  for (int it = 0; it < max_iteration; ++it) {
    M6 JTJ;
    V6 JTr;
    size_t hits;
    if (mode == ORACLE_FAITHFUL) {
      Matches m = match_faithful(*map, pts, cvs);
      hits = m.src_pts.size();
      sum_faithful(m, JTJ, JTr);
    } else {
      hits = sum_deterministic(*map, pts.data(), cvs.data(), n, JTJ, JTr);
    }
    if (stats) {
      if (stats->corr_count) stats->corr_count[it] = hits;
      if (stats->JTJ) std::memcpy(stats->JTJ + 36 * it, JTJ.data(), sizeof(double) * 36);
      if (stats->JTr) std::memcpy(stats->JTr + 6 * it, JTr.data(), sizeof(double) * 6);
    }
    const M4 step = solve_step(JTJ, JTr, nullptr);
    total = compose(step, total);
    rounds = it + 1;
    if (step_converged(step, cosine_threshold, translation_sq_threshold)) {
      done = true;
      break;
    }
    transform_cloud(pts, cvs, step);
  }
  std::memcpy(out_pose16, total.a, sizeof total.a);
  if (stats) {
    stats->iterations = rounds;
    stats->converged = done ? 1 : 0;
    stats->threads = (mode == ORACLE_FAITHFUL) ? omp_get_max_threads() : 1;
    stats->seconds = omp_get_wtime() - t0;
  }
  return 0;
}

// ---- N2: CloudPreprocessor::voxelDownsampleAndEstimateCovariances (src/CloudPreprocessor.cpp:76-127) ----
namespace {

// Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV) restated in its published operation order
// (Eigen 3.4 src/SVD/JacobiSVD.h: compute(), real_2x2_jacobi_svd(); src/Jacobi/Jacobi.h: makeJacobi(),
// JacobiRotation product / transpose, apply_rotation_in_the_plane()).  Two-sided Jacobi on the square work
// matrix A / max|A|: sweeps over (p, q) = (1,0), (2,0), (2,1) until every off-diagonal pair is below
// 2 eps * max|diagonal|; each 2x2 block is first made symmetric by a left rotation (the identity while the
// block IS symmetric), then diagonalised; U collects the left rotations, V the right ones.  Afterwards the
// diagonal is made non-negative by negating the U column of every negative entry, and values and columns are
// sorted descending by selection (first maximum wins).
// What that means for the caller below (src/CloudPreprocessor.cpp:119-123): for a symmetric input the singular
// values are |eigenvalue| and a NEGATIVE eigenvalue leaves U.col(k) = -V.col(k), so U F V^T carries
// sign(eigenvalue_k) on its k-th term.  Open3D's cumulant covariance E[xx^T] - E[x]E[x]^T is not positive
// semi-definite in floating point: far from the origin an exactly planar, collinear or repeated neighbourhood has
// a smallest eigenvalue that is rounding noise of either sign, and the reference then emits an INDEFINITE
// "covariance" (-1e-2 on the normal).  This restatement reproduces that class of result; whether a particular
// noise-level eigenvalue comes out negative depends on every rounding of Eigen's compiled code and cannot be
// pinned without Eigen (parity unpinned, see the header).  `negated` receives the number of columns k with
// U.col(k) . V.col(k) < 0, i.e. of negative eigenvalues of a symmetric input (NOT the number of sign folds: a
// rounding-level asymmetry of a 2x2 block can turn it by ~180 degrees, which negates work-matrix entries and U
// columns together).
// Returns false for a non-finite input (Eigen 3.4: info() == InvalidInput, U and V left unset).
struct Rot {  // Eigen::JacobiRotation<double>
  double c, s;
};
// JacobiRotation::makeJacobi(x, y, z) for the symmetric 2x2 [[x, y], [y, z]]
inline Rot make_jacobi(double x, double y, double z) {
  const double deno = 2.0 * std::fabs(y);
  if (deno < std::numeric_limits<double>::min()) return {1.0, 0.0};
  const double tau = (x - z) / deno;
  const double w = std::sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / std::sqrt(t * t + 1.0);
  return {n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
// apply_rotation_in_the_plane(x, y, j): x_i <- c x_i + s y_i, y_i <- -s x_i + c y_i (skipped for the identity)
inline void rotate_pair(double& x, double& y, const Rot& j) {
  const double xi = x, yi = y;
  x = j.c * xi + j.s * yi;
  y = -j.s * xi + j.c * yi;
}
bool jacobi_svd3(const M3& Ain, M3& U, M3& V, double sv[3], int* negated) {
  const double precision = 2.0 * std::numeric_limits<double>::epsilon();
  const double consider_as_zero = std::numeric_limits<double>::min();
  double scale = 0.0;
  bool finite = true;
  for (int i = 0; i < 9; ++i) {
    const double v = std::fabs(Ain.a[i]);
    if (!(v - v == 0.0)) finite = false;
    if (v > scale) scale = v;
  }
  if (!finite) return false;
  if (scale == 0.0) scale = 1.0;
  M3 W;
  for (int i = 0; i < 9; ++i) W.a[i] = Ain.a[i] / scale;
  U = identity3();
  V = identity3();
  double max_diag = std::max(std::fabs(W(0, 0)), std::max(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
  bool finished = false;
  while (!finished) {
    finished = true;
    for (int p = 1; p < 3; ++p)
      for (int q = 0; q < p; ++q) {
        const double threshold = std::max(consider_as_zero, precision * max_diag);
        if (!(std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold)) continue;
        finished = false;
        // real_2x2_jacobi_svd on m = [[W(p,p), W(p,q)], [W(q,p), W(q,q)]]
        double m00 = W(p, p), m01 = W(p, q), m10 = W(q, p), m11 = W(q, q);
        Rot rot1;
        const double t = m00 + m11, d = m10 - m01;
        if (std::fabs(d) < std::numeric_limits<double>::min()) {
          rot1 = {1.0, 0.0};
        } else {
          const double u = t / d;
          const double tmp = std::sqrt(1.0 + u * u);
          rot1 = {u / tmp, 1.0 / tmp};
        }
        if (!(rot1.c == 1.0 && rot1.s == 0.0)) {  // m.applyOnTheLeft(0, 1, rot1)
          rotate_pair(m00, m10, rot1);
          rotate_pair(m01, m11, rot1);
        }
        const Rot jr = make_jacobi(m00, m01, m11);
        const Rot jrt = {jr.c, -jr.s};                                         // j_right.transpose()
        const Rot jl = {rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};  // rot1 * j_right^T
        const Rot jlt = {jl.c, -jl.s};
        // m_workMatrix.applyOnTheLeft(p, q, j_left): rows p and q
        if (!(jl.c == 1.0 && jl.s == 0.0))
          for (int k = 0; k < 3; ++k) rotate_pair(W(p, k), W(q, k), jl);
        // m_matrixU.applyOnTheRight(p, q, j_left.transpose()): columns p and q, rotated by (j_left^T)^T = j_left
        if (!(jl.c == 1.0 && -jlt.s == 0.0))
          for (int k = 0; k < 3; ++k) rotate_pair(U(k, p), U(k, q), jl);
        // m_workMatrix.applyOnTheRight(p, q, j_right): columns p and q, rotated by j_right^T
        if (!(jrt.c == 1.0 && jrt.s == 0.0)) {
          for (int k = 0; k < 3; ++k) rotate_pair(W(k, p), W(k, q), jrt);
          for (int k = 0; k < 3; ++k) rotate_pair(V(k, p), V(k, q), jrt);     // m_matrixV.applyOnTheRight
        }
        max_diag = std::max(max_diag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
      }
  }
  for (int i = 0; i < 3; ++i) {
    const double a = W(i, i);
    sv[i] = std::fabs(a);
    if (a < 0.0)
      for (int k = 0; k < 3; ++k) U(k, i) = -U(k, i);
  }
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  for (int i = 0; i < 3; ++i) {  // selection sort, descending, first maximum of the tail
    int pos = i;
    for (int k = i + 1; k < 3; ++k)
      if (sv[k] > sv[pos]) pos = k;
    if (sv[pos] == 0.0) break;
    if (pos != i) {
      std::swap(sv[i], sv[pos]);
      for (int k = 0; k < 3; ++k) {
        std::swap(U(k, i), U(k, pos));
        std::swap(V(k, i), V(k, pos));
      }
    }
  }
  if (negated) {
    int opposed = 0;
    for (int i = 0; i < 3; ++i)
      if (U(0, i) * V(0, i) + U(1, i) * V(1, i) + U(2, i) * V(2, i) < 0.0) ++opposed;
    *negated = opposed;
  }
  return true;
}

// svd.matrixU() * covarianceFactor_ * svd.matrixV().transpose() (src/CloudPreprocessor.cpp:119-123) with
// covarianceFactor_ = diag(1, 1, 1e-2) (include/ESKF_LIO/CloudPreprocessor.hpp:30-31).  U F has the columns
// u0, u1, 1e-2 u2 (its other terms are exact zeros).  A non-finite covariance gives NaNs (the reference: unset U, V).
inline M3 regularize(const M3& cov, int* negated) {
  M3 U, V, R;
  double sv[3];
  if (!jacobi_svd3(cov, U, V, sv, negated)) {
    if (negated) *negated = 0;
    for (double& v : R.a) v = std::numeric_limits<double>::quiet_NaN();
    return R;
  }
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc)
      R(r, cc) = U(r, 0) * V(cc, 0) + U(r, 1) * V(cc, 1) + (U(r, 2) * 1e-2) * V(cc, 2);
  return R;
}

}  // namespace

// Test hooks: the JacobiSVD restatement on ANY real 3x3 (column-major), and the regulariser alone.
extern "C" int oracle_jacobi_svd3(const double A[9], double U[9], double V[9], double sv[3]) {
  M3 a, u, v;
  std::memcpy(a.a, A, 72);
  int negated = 0;
  if (!jacobi_svd3(a, u, v, sv, &negated)) return -1;
  std::memcpy(U, u.a, 72);
  std::memcpy(V, v.a, 72);
  return negated;
}
extern "C" int oracle_regularize(const double cov[9], double out[9]) {
  M3 c;
  std::memcpy(c.a, cov, 72);
  int negated = 0;
  const M3 R = regularize(c, &negated);
  std::memcpy(out, R.a, 72);
  return negated;
}

extern "C" size_t oracle_preprocess(size_t n, const double* points, double voxel_size, int knn,
                                    double* out_points, double* out_covs, uint64_t* out_index) {
  return oracle_preprocess_ex(n, points, voxel_size, knn, out_points, out_covs, out_index, nullptr);
}

extern "C" size_t oracle_preprocess_ex(size_t n, const double* points, double voxel_size, int knn,
                                       double* out_points, double* out_covs, uint64_t* out_index,
                                       uint64_t* indefinite) {
  return oracle_preprocess_ordered(n, points, voxel_size, knn, ORACLE_ORDER_ASCENDING, out_points, out_covs, out_index, indefinite);
}

// order: ORACLE_ORDER_ASCENDING — the kept points in ascending input index (what the HIP path emits, order-independent
// by construction); ORACLE_ORDER_REFERENCE_HASH — in the iteration order of the reference's own container
// (src/CloudPreprocessor.cpp:85-99: std::unordered_map<Vector3i, int, open3d::utility::hash_eigen> filled in scan
// order, then iterated): libstdc++'s node order for that hash and that insertion sequence, i.e. the order on the
// reference's platform (Ubuntu 22.04 / GCC 11, README.md:13-14, the same libstdc++ as this container's), which the
// C++ standard leaves unspecified.  Exists to MEASURE what the documented deviation costs downstream (the map's
// running means depend on insertion order, include/ESKF_LIO/LocalMap.hpp:79-87), not as a parity target.
extern "C" size_t oracle_preprocess_ordered(size_t n, const double* points, double voxel_size, int knn, int order,
                                            double* out_points, double* out_covs, uint64_t* out_index,
                                            uint64_t* indefinite) {
  uint64_t flipped = 0;
  const V3* P = reinterpret_cast<const V3*>(points);
  // first point per voxel (src/CloudPreprocessor.cpp:87-92)
  std::unordered_map<Key, size_t, KeyHash> first;
  for (size_t i = 0; i < n; ++i) first.emplace(voxel_key(P[i], voxel_size), i);
  std::vector<size_t> kept;
  kept.reserve(first.size());
  for (const auto& kv : first) kept.push_back(kv.second);   // src/CloudPreprocessor.cpp:96-99, the container's own order
  if (order != ORACLE_ORDER_REFERENCE_HASH) std::sort(kept.begin(), kept.end());
  const size_t m = kept.size();
  const size_t K = std::min<size_t>(static_cast<size_t>(knn > 0 ? knn : 0), n);
#pragma omp parallel reduction(+ : flipped)
  {
    std::vector<std::pair<double, size_t>> dist(n);
#pragma omp for schedule(dynamic, 16)
    for (size_t o = 0; o < m; ++o) {
      const V3 q = P[kept[o]];
      for (size_t j = 0; j < n; ++j) {
        const double dx = P[j].x - q.x, dy = P[j].y - q.y, dz = P[j].z - q.z;
        dist[j] = {dx * dx + dy * dy + dz * dz, j};
      }
      std::partial_sort(dist.begin(), dist.begin() + K, dist.end());  // ascending distance, then index
      M3 cov = identity3();
      if (K >= 3) {
        // open3d::utility::ComputeCovariance: cumulants over the neighbours in search order
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = 0; k < K; ++k) {
          const V3& p = P[dist[k].second];
          c[0] += p.x; c[1] += p.y; c[2] += p.z;
          c[3] += p.x * p.x; c[4] += p.x * p.y; c[5] += p.x * p.z;
          c[6] += p.y * p.y; c[7] += p.y * p.z; c[8] += p.z * p.z;
        }
        for (double& v : c) v /= static_cast<double>(K);
        cov(0, 0) = c[3] - c[0] * c[0];
        cov(1, 1) = c[6] - c[1] * c[1];
        cov(2, 2) = c[8] - c[2] * c[2];
        cov(0, 1) = cov(1, 0) = c[4] - c[0] * c[1];
        cov(0, 2) = cov(2, 0) = c[5] - c[0] * c[2];
        cov(1, 2) = cov(2, 1) = c[7] - c[1] * c[2];
      }
      int negated = 0;
      const M3 R = regularize(cov, &negated);
      if (negated > 0) ++flipped;
      std::memcpy(out_points + 3 * o, &q, 24);
      std::memcpy(out_covs + 9 * o, R.a, 72);
      out_index[o] = kept[o];
    }
  }
  if (indefinite) *indefinite = flipped;
  return m;
}

// ---- CloudPreprocessor::deskew (src/CloudPreprocessor.cpp:25-74) ----------------------------------
namespace {
struct Pose34 {  // Eigen::Isometry3d as far as it is used here: linear part + translation
  M3 R;
  V3 t;
};
// Eigen::Quaterniond::toRotationMatrix, coefficients in Eigen's storage order (x, y, z, w)
inline M3 quat_to_matrix(const double q[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  M3 R;
  R(0, 0) = 1.0 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
  R(1, 0) = txy + twz; R(1, 1) = 1.0 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1.0 - (txx + tyy);
  return R;
}
// Eigen::QuaternionBase::slerp
inline void quat_slerp(const double a[4], const double b[4], double t, double out[4]) {
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = std::fabs(d);
  double scale0, scale1;
  if (absD >= one) {
    scale0 = 1.0 - t;
    scale1 = t;
  } else {
    const double theta = std::acos(absD);
    const double sinTheta = std::sin(theta);
    scale0 = std::sin((1.0 - t) * theta) / sinTheta;
    scale1 = std::sin(t * theta) / sinTheta;
  }
  if (d < 0.0) scale1 = -scale1;
  for (int k = 0; k < 4; ++k) out[k] = scale0 * a[k] + scale1 * b[k];
}
inline Pose34 pose_compose(const Pose34& A, const Pose34& B) {  // Isometry3d * Isometry3d
  Pose34 C;
  C.R = mul(A.R, B.R);
  const V3 rt = mul(A.R, B.t);
  C.t = {rt.x + A.t.x, rt.y + A.t.y, rt.z + A.t.z};
  return C;
}
inline Pose34 pose_inverse(const Pose34& A) {  // Isometry3d::inverse(): R^T, -(R^T t)
  Pose34 C;
  C.R = transpose(A.R);
  const V3 rt = mul(C.R, A.t);
  C.t = {-rt.x, -rt.y, -rt.z};
  return C;
}
}  // namespace

// states: num_states x 8 doubles (timestamp, position xyz, attitude quaternion x y z w), ascending time.
// Returns the number of leading points that were transformed (the rest are left as they are, exactly like
// the reference's loop), or -1 when the reference would run off its state queue (no state at or before
// the last point's time, or none after it): nothing is touched then.
int64_t oracle_deskew(size_t n, double* points, const double* point_time, size_t num_states,
                      const double* states) {
  if (n == 0 || num_states == 0) return 0;
  V3* P = reinterpret_cast<V3*>(points);
  const double t_end = point_time[n - 1];
  long b = static_cast<long>(num_states) - 1;
  while (b >= 0 && states[8 * b] > t_end) --b;  // src/CloudPreprocessor.cpp:35-42
  if (b < 0 || static_cast<size_t>(b) + 1 >= num_states) return -1;
  const size_t a = static_cast<size_t>(b) + 1;    // stateAfterLidarEnd (:44)
  // Utils::interpolateSE3 (src/Utils.cpp:65-75)
  const double* s1 = states + 8 * b;
  const double* s2 = states + 8 * a;
  const double factor = (t_end - s1[0]) / (s2[0] - s1[0] + 1e-6);
  double q[4];
  quat_slerp(s1 + 4, s2 + 4, factor, q);
  Pose34 end_pose;
  end_pose.R = quat_to_matrix(q);
  end_pose.t = {s1[1] + factor * (s2[1] - s1[1]), s1[2] + factor * (s2[2] - s1[2]), s1[3] + factor * (s2[3] - s1[3])};
  const Pose34 end_inv = pose_inverse(end_pose);
  size_t start = 0, end = 0;
  for (size_t s = 0; s <= a; ++s) {  // states.cbegin() .. stateAfterLidarEnd.base() (:51)
    start = end;
    size_t i = start;
    while (i < n) {
      if (point_time[i] < states[8 * s]) {
        ++i;
      } else {
        end = i;
        break;
      }
    }
    if (start == end) continue;
    Pose34 T;
    T.R = quat_to_matrix(states + 8 * s + 4);
    T.t = {states[8 * s + 1], states[8 * s + 2], states[8 * s + 3]};
    T = pose_compose(end_inv, T);
    for (size_t k = start; k < end; ++k) {  // Utils::transformPoints (src/Utils.cpp:13-20)
      const V3 r = mul(T.R, P[k]);
      P[k] = {r.x + T.t.x, r.y + T.t.y, r.z + T.t.z};
    }
  }
  return static_cast<int64_t>(end);
}

int oracle_max_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int threads) { if (threads > 0) omp_set_num_threads(threads); }

}  // extern "C"
