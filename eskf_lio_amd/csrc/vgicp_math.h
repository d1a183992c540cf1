// vgicp_math.h — fixed-size fp64 algebra shared by the HIP kernels and the C++ host side.
//
// Zero dependencies (Eigen is not available on the build or GPU hosts).  Every routine is
// `__host__ __device__` under hipcc and plain inline C++ under g++.  3x3 matrices are stored
// column-major (m[r + 3*c]) so that a `double[9]` here is byte-compatible with
// `Eigen::Matrix3d::data()`, which is what the reference hands over at the boundary
// (reference: include/ESKF_LIO/Registration.hpp:18-21).
//
// Semantics followed (the reference delegates these to Eigen >= 3.4; restated from Eigen's
// published algorithms, call sites cited):
//   inv3          Matrix3d::inverse(): cofactor / determinant, no pivoting, no invertibility
//                 check                                   (reference src/Registration.cpp:95)
//   ldlt6_solve   Matrix<6,6>::ldlt().solve(): LDL^T with symmetric diagonal pivoting reading the
//                 LOWER triangle only; D is pseudo-inverted  (reference src/Registration.cpp:78)
//   rodrigues     AngleAxisd(|r|, r.normalized()).toRotationMatrix()  (reference src/Utils.cpp:28-32)
//   left_jacobian Utils::computeJ incl. the `angle < 1e-6 -> I` branch (reference src/Utils.cpp:40-54)
//   se3_exp       Utils::se3ToSE3, state order [translation; rotation] (reference src/Utils.cpp:56-63)
//   pose_compose  Isometry3d * Isometry3d                  (reference src/Registration.cpp:20)
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VG_HD __host__ __device__ __forceinline__
#else
#define VG_HD inline
#endif

namespace vgicp {

// Rigid transform: rotation (column-major 3x3) + translation.
struct Pose {
  double R[9];
  double t[3];
};

VG_HD void pose_identity(Pose& T) {
  for (int i = 0; i < 9; ++i) T.R[i] = 0.0;
  T.R[0] = T.R[4] = T.R[8] = 1.0;
  T.t[0] = T.t[1] = T.t[2] = 0.0;
}

// Column-major 4x4 (Eigen::Isometry3d::matrix().data()) <-> Pose.
VG_HD void pose_from_mat4(const double* m16, Pose& T) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T.R[r + 3 * c] = m16[r + 4 * c];
  for (int r = 0; r < 3; ++r) T.t[r] = m16[r + 12];
}

VG_HD void pose_to_mat4(const Pose& T, double* m16) {
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) m16[r + 4 * c] = T.R[r + 3 * c];
    m16[3 + 4 * c] = 0.0;
  }
  for (int r = 0; r < 3; ++r) m16[r + 12] = T.t[r];
  m16[15] = 1.0;
}

// out = A * B  (column-major 3x3)
VG_HD void mat3_mul(const double* A, const double* B, double* out) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      out[r + 3 * c] = A[r] * B[3 * c] + A[r + 3] * B[1 + 3 * c] + A[r + 6] * B[2 + 3 * c];
}

// out = A * B^T
VG_HD void mat3_mul_bt(const double* A, const double* B, double* out) {
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      out[r + 3 * c] = A[r] * B[c] + A[r + 3] * B[c + 3] + A[r + 6] * B[c + 6];
}

VG_HD void mat3_vec(const double* A, const double* x, double* y) {
  for (int r = 0; r < 3; ++r) y[r] = A[r] * x[0] + A[r + 3] * x[1] + A[r + 6] * x[2];
}

// T = A * B  (apply B first, then A): R = Ra Rb, t = Ra tb + ta.
VG_HD void pose_compose(const Pose& A, const Pose& B, Pose& out) {
  Pose tmp;
  mat3_mul(A.R, B.R, tmp.R);
  mat3_vec(A.R, B.t, tmp.t);
  for (int r = 0; r < 3; ++r) tmp.t[r] += A.t[r];
  out = tmp;
}

// Cofactor inverse of a general 3x3 (no symmetry assumed, no singularity check).
VG_HD void inv3(const double* m, double* inv) {
  // cof(i,j) = m(i1,j1) m(i2,j2) - m(i1,j2) m(i2,j1), i1=(i+1)%3, i2=(i+2)%3
#define VG_M(r, c) m[(r) + 3 * (c)]
  const double c00 = VG_M(1, 1) * VG_M(2, 2) - VG_M(1, 2) * VG_M(2, 1);
  const double c10 = VG_M(2, 1) * VG_M(0, 2) - VG_M(2, 2) * VG_M(0, 1);
  const double c20 = VG_M(0, 1) * VG_M(1, 2) - VG_M(0, 2) * VG_M(1, 1);
  const double det = c00 * VG_M(0, 0) + c10 * VG_M(1, 0) + c20 * VG_M(2, 0);
  const double id = 1.0 / det;
  const double c01 = VG_M(1, 2) * VG_M(2, 0) - VG_M(1, 0) * VG_M(2, 2);
  const double c11 = VG_M(2, 2) * VG_M(0, 0) - VG_M(2, 0) * VG_M(0, 2);
  const double c21 = VG_M(0, 2) * VG_M(1, 0) - VG_M(0, 0) * VG_M(1, 2);
  const double c02 = VG_M(1, 0) * VG_M(2, 1) - VG_M(1, 1) * VG_M(2, 0);
  const double c12 = VG_M(2, 0) * VG_M(0, 1) - VG_M(2, 1) * VG_M(0, 0);
  const double c22 = VG_M(0, 0) * VG_M(1, 1) - VG_M(0, 1) * VG_M(1, 0);
#undef VG_M
  // inv(r,c) = cof(c,r) / det
  inv[0] = c00 * id; inv[3] = c10 * id; inv[6] = c20 * id;
  inv[1] = c01 * id; inv[4] = c11 * id; inv[7] = c21 * id;
  inv[2] = c02 * id; inv[5] = c12 * id; inv[8] = c22 * id;
}

// Index of entry (r, c), r >= c, in the packed lower triangle of a 6x6 (row by row):
// (0,0) (1,0) (1,1) (2,0) ... (5,5)  -> 21 values.
VG_HD constexpr int tri6(int r, int c) { return r * (r + 1) / 2 + c; }

// Solve A x = b for symmetric 6x6 A given by its packed lower triangle (21 values), with the
// pivoted LDL^T of Eigen's LDLT as published (unblocked in-place form): largest remaining |diagonal|
// first, symmetric swap inside the lower triangle, temp = D(0..k-1) .* row k, the diagonal and the
// column below it each reduced by ONE dot product with temp (dot first, then the subtraction), a zero
// pivot leaves its column undivided, D pseudo-inverted with tolerance = smallest normal double.
// A all-zero with b all-zero returns x = 0, as Eigen does.  No FMA contraction anywhere, so with the
// same operation order the CPU oracle returns the same bits on any system, rank-deficient ones with
// noise pivots included.  `work` holds >= kLdltWork doubles the caller provides (LDS on the device:
// pivoting makes the indices dynamic, and private arrays with dynamic indices would go to scratch).
VG_HD void ldlt6_solve(const double* lower21, const double* b, double* x, double* work) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double* A = work;        // 6x6, row-major A[6 * r + c], lower triangle live
  double* y = work + 36;   // 6
  double* tmp = work + 42; // 6
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c <= r; ++c) A[6 * r + c] = lower21[tri6(r, c)];
  int perm_packed = 0;     // 3 bits per pivot index
  bool zero_diag = false;
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(A[7 * k]);
    for (int i = k + 1; i < 6; ++i) {
      const double v = fabs(A[7 * i]);
      if (v > best) { best = v; p = i; }
    }
    perm_packed |= p << (3 * k);
    if (p != k) {
      for (int j = 0; j < k; ++j) { const double s = A[6 * k + j]; A[6 * k + j] = A[6 * p + j]; A[6 * p + j] = s; }
      for (int i = p + 1; i < 6; ++i) { const double s = A[6 * i + k]; A[6 * i + k] = A[6 * i + p]; A[6 * i + p] = s; }
      for (int i = k + 1; i < p; ++i) { const double s = A[6 * i + k]; A[6 * i + k] = A[6 * p + i]; A[6 * p + i] = s; }
      { const double s = A[7 * k]; A[7 * k] = A[7 * p]; A[7 * p] = s; }
    }
    if (k > 0) {
      for (int j = 0; j < k; ++j) tmp[j] = A[7 * j] * A[6 * k + j];
      double dot = 0.0;
      for (int j = 0; j < k; ++j) dot += A[6 * k + j] * tmp[j];
      A[7 * k] -= dot;
      for (int i = k + 1; i < 6; ++i) {
        double d = 0.0;
        for (int j = 0; j < k; ++j) d += A[6 * i + j] * tmp[j];
        A[6 * i + k] -= d;
      }
    }
    const double akk = A[7 * k];
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      zero_diag = true;    // the whole diagonal is zero: Eigen stops here with identity transpositions
      break;
    }
    if (valid)
      for (int i = k + 1; i < 6; ++i) A[6 * i + k] /= akk;
  }
  for (int i = 0; i < 6; ++i) y[i] = b[i];
  if (!zero_diag)
    for (int k = 0; k < 6; ++k) {
      const int p = (perm_packed >> (3 * k)) & 7;
      if (p != k) { const double s = y[k]; y[k] = y[p]; y[p] = s; }
    }
  for (int i = 1; i < 6; ++i)            // L y = P b (unit lower), one term at a time
    for (int j = 0; j < i; ++j) y[i] -= A[6 * i + j] * y[j];
  const double tol = 2.2250738585072014e-308;
  for (int i = 0; i < 6; ++i) y[i] = (fabs(A[7 * i]) > tol) ? y[i] / A[7 * i] : 0.0;
  for (int i = 4; i >= 0; --i)           // L^T z = y
    for (int j = i + 1; j < 6; ++j) y[i] -= A[6 * j + i] * y[j];
  if (!zero_diag)
    for (int k = 5; k >= 0; --k) {
      const int p = (perm_packed >> (3 * k)) & 7;
      if (p != k) { const double s = y[k]; y[k] = y[p]; y[p] = s; }
    }
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}
constexpr int kLdltWork = 48;

// Rotation vector -> rotation matrix, AngleAxis(|r|, r/|r|) with r left as is when |r| == 0.
VG_HD void rodrigues(const double* r, double* R) {
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double angle = sqrt(n2);
  double ax[3] = {r[0], r[1], r[2]};
  if (n2 > 0.0) { ax[0] = r[0] / angle; ax[1] = r[1] / angle; ax[2] = r[2] / angle; }
  const double s = sin(angle), c = cos(angle);
  const double sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
  const double cx = (1.0 - c) * ax[0], cy = (1.0 - c) * ax[1], cz = (1.0 - c) * ax[2];
  double tmp;
  tmp = cx * ax[1]; R[0 + 3 * 1] = tmp - sz; R[1 + 3 * 0] = tmp + sz;
  tmp = cx * ax[2]; R[0 + 3 * 2] = tmp + sy; R[2 + 3 * 0] = tmp - sy;
  tmp = cy * ax[2]; R[1 + 3 * 2] = tmp - sx; R[2 + 3 * 1] = tmp + sx;
  R[0] = cx * ax[0] + c;
  R[4] = cy * ax[1] + c;
  R[8] = cz * ax[2] + c;
}

// SO(3) left Jacobian as the reference writes it (identity below 1e-6 rad).
VG_HD void left_jacobian(const double* r, double* J) {
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double angle = sqrt(n2);
  for (int i = 0; i < 9; ++i) J[i] = 0.0;
  if (angle < 1e-6) {
    J[0] = J[4] = J[8] = 1.0;
    return;
  }
  const double k[3] = {r[0] / angle, r[1] / angle, r[2] / angle};
  const double f1 = sin(angle) / angle;
  const double f2 = (1.0 - cos(angle)) / angle;
  for (int c = 0; c < 3; ++c)
    for (int rr = 0; rr < 3; ++rr) J[rr + 3 * c] = (1.0 - f1) * k[rr] * k[c];
  J[0] += f1; J[4] += f1; J[8] += f1;
  // + f2 * [k]x
  J[0 + 3 * 1] += -f2 * k[2]; J[0 + 3 * 2] += f2 * k[1];
  J[1 + 3 * 0] += f2 * k[2];  J[1 + 3 * 2] += -f2 * k[0];
  J[2 + 3 * 0] += -f2 * k[1]; J[2 + 3 * 1] += f2 * k[0];
}

// se(3) -> SE(3), xi = [rho(3); phi(3)]:  R = exp([phi]x), t = J_l(phi) rho.
VG_HD void se3_exp(const double* xi, Pose& T) {
  double J[9];
  left_jacobian(xi + 3, J);
  mat3_vec(J, xi, T.t);
  rodrigues(xi + 3, T.R);
}

// Reference convergence rule (src/Registration.cpp:37-50): both tests are non-strict on the
// "converged" side, i.e. equality with a threshold counts as converged.
VG_HD bool converged(const Pose& step, double cosine_threshold, double translation_sq_threshold) {
  const double cosine = 0.5 * (step.R[0] + step.R[4] + step.R[8] - 1.0);
  if (cosine < cosine_threshold) return false;
  const double t2 = step.t[0] * step.t[0] + step.t[1] * step.t[1] + step.t[2] * step.t[2];
  if (t2 > translation_sq_threshold) return false;
  return true;
}

}  // namespace vgicp
