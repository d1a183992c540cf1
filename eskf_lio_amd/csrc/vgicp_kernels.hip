// vgicp_kernels.hip — hand-written CDNA4 (gfx950) kernels of the VGICP registration path.
//
// Reference behaviour each kernel reproduces (paths relative to the reference checkout):
//   iterate_kernel  one round of the loop in ICP::align (src/Registration.cpp:15-28). Prologue: the
//                   merge of the previous round's partial sums, the LDLT solve, se3ToSE3, pose
//                   composition and convergence test (:71-79, :20-25, :37-50). Body:
//                   Open3D PointCloud::Transform of point + covariance (call sites :13,27),
//                   LocalMap::getVoxelIndex + voxelGrid_.find (src/LocalMap.cpp:94-100,114-118),
//                   ICP::computeJTJAndJTr (src/Registration.cpp:83-102) and the accumulation of
//                   ICP::computeTransform (:60-70)
//   upsert/erase    the effect of LocalMap::updateLocalMap's insert and evict loops on the data the
//                   path reads (src/LocalMap.cpp:47-72), mirrored from host-computed voxel values
//   match kernels   LocalMap::correspondenceMatching with materialised output (src/LocalMap.cpp:78-112)
//
// Design notes (DESIGN.md has the numbers):
//   * The scan is never written back: the TOTAL pose is applied to the original point in registers
//     each round (the reference transforms a copy incrementally), so a point costs 96 B of reads.
//   * MFMA is not used: this is a gather plus a 27-value reduction at ~2 fp64 flop per byte.
//   * Reduction: 32-slot halving butterfly inside a 64-lane wave (32 exchanges instead of 6 x 28),
//     LDS across the waves of a workgroup, one plain 256-byte partial row per workgroup.  Nothing in a
//     launch waits for another workgroup: the NEXT launch's prologue folds the rows (every workgroup,
//     redundantly, in one fixed order — bit-reproducible) while its own scan loads are in flight, so
//     the kernel boundary is the only grid-wide synchronisation and no workgroup runs a serial tail
//     alone (the first version's ticket + write-through + last-workgroup tail cost 13 us per round).
#include "vgicp_device.h"
#include "vgicp_device_fn.h"

namespace vgicp {
namespace {

// Lookup + payload: probe the key word(s), then fetch the 96-byte payload of the matching record as
// six 16-byte loads (same 128-byte line as the key, so they are L1/L2 hits).  Requesting key and
// payload together was measured slower: every lane addresses its own line, so each extra wave-level
// load costs 64 tag lookups whether or not the lane ends up matching.
__device__ __forceinline__ bool find_and_load(const VoxelRecord* table, uint32_t mask, int32_t kx,
                                              int32_t ky, int32_t kz, double (&mu)[3],
                                              double (&S)[9]) {
  const VoxelRecord* rec = find_voxel(table, mask, kx, ky, kz);
  if (rec == nullptr) return false;
  const double2* pay = reinterpret_cast<const double2*>(rec->mean);  // 16-byte aligned
  const double2 a0 = pay[0], a1 = pay[1], a2 = pay[2], a3 = pay[3], a4 = pay[4], a5 = pay[5];
  mu[0] = a0.x; mu[1] = a0.y; mu[2] = a1.x;
  S[0] = a1.y; S[1] = a2.x; S[2] = a2.y; S[3] = a3.x; S[4] = a3.y; S[5] = a4.x;
  S[6] = a4.y; S[7] = a5.x; S[8] = a5.y;
  return true;
}

// Value of lane ^ MASK for MASK in {8, 4, 2, 1}, by DPP inside the 16-lane row (VALU moves; ds_bpermute
// would send each value through the LDS crossbar and put its latency on the reduction's dependent path):
// 8 = rotate the row by 8; 4 = shift by 4 towards the lower lanes for banks 0 and 2, towards the upper ones
// for banks 1 and 3; 2 and 1 = quad permutes.
template <int MASK>
__device__ __forceinline__ int xor_lane_i32(int x) {
  static_assert(MASK == 8 || MASK == 4 || MASK == 2 || MASK == 1, "row-local masks only");
  if constexpr (MASK == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
  if constexpr (MASK == 4) {
    const int lower = __builtin_amdgcn_update_dpp(x, x, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
    return __builtin_amdgcn_update_dpp(lower, x, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
  }
  if constexpr (MASK == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, false);
  return __builtin_amdgcn_update_dpp(x, x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
}
template <int MASK>
__device__ __forceinline__ double xor_lane_f64(double v) {
  return __hiloint2double(xor_lane_i32<MASK>(__double2hiint(v)), xor_lane_i32<MASK>(__double2loint(v)));
}

// One halving step of the wave butterfly: N live values -> N/2, exchanging with lane ^ MASK.
template <int N, int MASK>
__device__ __forceinline__ void fold(double (&v)[kSlots], bool upper) {
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const double keep = upper ? v[j + N / 2] : v[j];
    const double send = upper ? v[j] : v[j + N / 2];
    v[j] = keep + xor_lane_f64<MASK>(send);
  }
}

// The two widest halving steps without LDS and without selects: v_permlane32_swap / v_permlane16_swap
// (gfx950) exchange the upper half (odd 16-lane rows) of one register with the lower half (even rows)
// of another, which is exactly "keep my half of the slots, receive the partner's": 3 instructions per
// pair of slots instead of 7.
template <int N, bool ROW16>
__device__ __forceinline__ void fold_swap(double (&v)[kSlots]) {
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const unsigned alo = (unsigned)__double2loint(v[j]), ahi = (unsigned)__double2hiint(v[j]);
    const unsigned blo = (unsigned)__double2loint(v[j + N / 2]), bhi = (unsigned)__double2hiint(v[j + N / 2]);
    const auto lo = ROW16 ? __builtin_amdgcn_permlane16_swap(alo, blo, false, false)
                          : __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = ROW16 ? __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false)
                          : __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    v[j] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
}

// 1 / d for a normal, finite d by v_rcp_f64 (good to 2^-24) + ONE third-order step y (1 + e + e^2), e = 1 - d y: three
// dependent FMAs, error 2^-72 before the last rounding — the correctly rounded quotient for all of 4 M random
// arguments (tools/micro/rcp_accuracy.hip), as the two Newton steps of rounds 1-4 were, with one instruction less on
// the chain; a quarter of the dependent instructions of the IEEE division.
__device__ __forceinline__ double rcp_newton(double d) {
  const double y = __builtin_amdgcn_rcp(d);
#if defined(VGICP_RCP_TWO_STEPS)   // developer A/B: two second-order steps (what rounds 1-4 shipped): 4 dependent operations
  double e = fma(-d, y, 1.0);
  const double y1 = fma(y, e, y);
  e = fma(-d, y1, 1.0);
  return fma(y1, e, y1);
#else
  const double e = fma(-d, y, 1.0);
  return fma(y, fma(e, e, e), y);
#endif
}

// Fallback of the 6x6 solve: vgicp_math.h's ldlt6_solve — Eigen's pivoted LDLT with pseudo-inverted D, the
// operation order of the published algorithm, no contraction — run by lane 0 on an LDS work array (the
// pivoting makes every index dynamic), the solution then read by every lane.  It is taken only when the
// natural-order factorisation below meets a pivot that is not safely positive (singular or indefinite
// normal equations, "no correspondences" included), so its ~10 us of dependent LDS traffic are off the
// hot path.  packed: LDS, 21 lower-triangle entries then the 6 of J^T r; work: LDS, kSolveWork doubles.
// All 64 lanes of the wave must call it.
constexpr int kSolveWork = kLdltWork + 16;
__device__ __forceinline__ void ldlt6_solve_pivoted(const double* packed, double* work, uint32_t lane,
                                                    double (&x)[6]) {
  double* rhs = work + kLdltWork;
  double* sol = work + kLdltWork + 8;
  if (lane == 0) {
    for (int k = 0; k < 6; ++k) rhs[k] = -packed[21 + k];
    ldlt6_solve(packed, rhs, sol, work);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = sol[k];
}

// sin and cos of a SMALL angle (|a| <= 0.5 rad; Gauss-Newton steps are far smaller) by their Taylor series in
// Horner form: truncation below 5e-17 relative, no argument reduction, ~20 dependent FMAs instead of the
// library's ~100 instructions.  Larger angles take the library call.
__device__ __forceinline__ void sincos_step(double a, double* s, double* c) {
  if (fabs(a) > 0.5) {  // uniform on the solver wave
    sincos(a, s, c);
    return;
  }
  const double z = a * a;
  double ps = -1.0 / 1307674368000.0;   // -1/15!
  ps = fma(ps, z, 1.0 / 6227020800.0);  //  1/13!
  ps = fma(ps, z, -1.0 / 39916800.0);   // -1/11!
  ps = fma(ps, z, 1.0 / 362880.0);      //  1/9!
  ps = fma(ps, z, -1.0 / 5040.0);       // -1/7!
  ps = fma(ps, z, 1.0 / 120.0);         //  1/5!
  ps = fma(ps, z, -1.0 / 6.0);          // -1/3!
  *s = fma(a * z, ps, a);
  double pc = 1.0 / 20922789888000.0;   //  1/16!
  pc = fma(pc, z, -1.0 / 87178291200.0);  // -1/14!
  pc = fma(pc, z, 1.0 / 479001600.0);   //  1/12!
  pc = fma(pc, z, -1.0 / 3628800.0);    // -1/10!
  pc = fma(pc, z, 1.0 / 40320.0);       //  1/8!
  pc = fma(pc, z, -1.0 / 720.0);        // -1/6!
  pc = fma(pc, z, 1.0 / 24.0);          //  1/4!
  pc = fma(pc, z, -0.5);                // -1/2!
  *c = fma(pc, z, 1.0);
}

// Fast path of the 6x6 solve: LDL^T in natural order, fully unrolled, everything in registers with
// static indices (about 130 fp64 operations and 6 reciprocals; every lane of the wave runs the same
// scalar program).  For a symmetric positive definite system any elimination order is backward
// stable, so x differs from the pivoted solve only by rounding (~cond * 1e-16).  Returns false when a
// pivot is not safely positive — singular or indefinite normal equations, the all-zero system of
// "no correspondences" included — and the caller then runs the pivoted, Eigen-faithful solve above.
// A: 21 lower-triangle entries row by row; g: right-hand side (already negated J^T r).
__device__ __forceinline__ bool ldlt6_solve_spd(const double (&A)[21], const double (&g)[6],
                                                double (&x)[6]) {
  double L[6][6], W[6][6], inv[6];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[tri6(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * W[j][k];
    ok = ok && (d > 1e-13 * A[tri6(j, j)]) && (d < 1.0e300) && (d > 1.0e-290);  // also false for NaN
    inv[j] = rcp_newton(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[tri6(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i][k] * W[j][k];
      W[i][j] = t;           // L(i,j) * D(j)
      L[i][j] = t * inv[j];
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) t -= L[i][k] * z[k];
    z[i] = t;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double t = z[i] * inv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t -= L[k][i] * x[k];
    x[i] = t;
  }
  return ok;
}

// se(3) exponential as vgicp_math.h's se3_exp (reference src/Utils.cpp:28-32,40-63) with the sine
// and cosine of the one angle taken from a single sincos call.
__device__ __forceinline__ void se3_exp_angle(const double* xi, Pose& T) {
  const double* r = xi + 3;
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double angle = sqrt(n2);
  double s, c;
  sincos_step(angle, &s, &c);
  double k[3] = {r[0], r[1], r[2]};
  const double inv_angle = (n2 > 1.0e-280) ? rcp_newton(angle) : 1.0 / angle;
  if (n2 > 0.0) { k[0] = r[0] * inv_angle; k[1] = r[1] * inv_angle; k[2] = r[2] * inv_angle; }
  const double sx = s * k[0], sy = s * k[1], sz = s * k[2];
  const double cx = (1.0 - c) * k[0], cy = (1.0 - c) * k[1], cz = (1.0 - c) * k[2];
  double tmp;
  tmp = cx * k[1]; T.R[3] = tmp - sz; T.R[1] = tmp + sz;
  tmp = cx * k[2]; T.R[6] = tmp + sy; T.R[2] = tmp - sy;
  tmp = cy * k[2]; T.R[7] = tmp - sx; T.R[5] = tmp + sx;
  T.R[0] = cx * k[0] + c;
  T.R[4] = cy * k[1] + c;
  T.R[8] = cz * k[2] + c;
  if (angle < 1e-6) {
    T.t[0] = xi[0]; T.t[1] = xi[1]; T.t[2] = xi[2];
    return;
  }
  const double f1 = s * inv_angle, f2 = (1.0 - c) * inv_angle;
  double J[9];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) J[rr + 3 * cc] = (1.0 - f1) * k[rr] * k[cc];
  J[0] += f1; J[4] += f1; J[8] += f1;
  J[3] += -f2 * k[2]; J[6] += f2 * k[1];
  J[1] += f2 * k[2];  J[7] += -f2 * k[0];
  J[2] += -f2 * k[1]; J[5] += f2 * k[0];
  mat3_vec(J, xi, T.t);
}

// The same map for the steps Gauss-Newton actually takes (|phi| <= 0.5 rad), written in phi itself instead of
// (angle, axis): with n2 = |phi|^2,
//   A = sin a / a = 1 - n2 C,   B = (1 - cos a) / a^2,   C = (a - sin a) / a^3      (power series in n2)
//   R = (1 - n2 B) I + B phi phi^T + A [phi]x              (= cos a I + (1 - cos a) k k^T + sin a [k]x, reference src/Utils.cpp:28-32)
//   t = A rho + C phi (phi . rho) + B (phi x rho)          (= J_l(phi) rho, reference src/Utils.cpp:40-54)
// No square root, no reciprocal, no normalisation and no sine / cosine on the round's serial chain: the two series in
// Estrin form are 4 dependent operations deep after n2 (the angle-axis form: sqrt ~10, Taylor sincos 10, rcp 5, products
// 4).  Truncation < 1e-17 relative at |phi| = 0.5; the results agree with the angle-axis form to rounding (~1e-16), not bit for bit.
// The reference's branch "angle < 1e-6 -> t = rho" is kept: kSmallAngle2 is the smallest double whose square root rounds
// to >= 1e-6 (tests/test_capi_cpu.py checks the constant against sqrt).
constexpr double kSmallAngle2 = 0x1.19799812dea10p-40;   // 9.999999999999998e-13
__device__ __forceinline__ void se3_exp_device(const double* xi, Pose& T) {
  const double px = xi[0], py = xi[1], pz = xi[2], rx = xi[3], ry = xi[4], rz = xi[5];
  const double n2 = rx * rx + ry * ry + rz * rz;
#if defined(VGICP_EXP_ANGLE_AXIS)   // developer A/B (tools/ab_build.sh): the angle-axis form for every step
  if (true) {
#else
  if (n2 > 0.25) {  // uniform on the solver wave
#endif
    se3_exp_angle(xi, T);
    return;
  }
  const double z2 = n2 * n2, z4 = z2 * z2;
  // B = sum (-1)^k n2^k / (2k + 2)!,  C = sum (-1)^k n2^k / (2k + 3)!,  k = 0 .. 7
  const double b01 = fma(n2, -1.0 / 24.0, 0.5), b23 = fma(n2, -1.0 / 40320.0, 1.0 / 720.0);
  const double b45 = fma(n2, -1.0 / 479001600.0, 1.0 / 3628800.0), b67 = fma(n2, -1.0 / 20922789888000.0, 1.0 / 87178291200.0);
  const double c01 = fma(n2, -1.0 / 120.0, 1.0 / 6.0), c23 = fma(n2, -1.0 / 362880.0, 1.0 / 5040.0);
  const double c45 = fma(n2, -1.0 / 6227020800.0, 1.0 / 39916800.0), c67 = fma(n2, -1.0 / 355687428096000.0, 1.0 / 1307674368000.0);
  const double B = fma(z4, fma(z2, b67, b45), fma(z2, b23, b01));
  const double C = fma(z4, fma(z2, c67, c45), fma(z2, c23, c01));
  const double A = fma(-n2, C, 1.0), c = fma(-n2, B, 1.0);
  const double bx = B * rx, by = B * ry, bz = B * rz, ax = A * rx, ay = A * ry, az = A * rz;
  T.R[0] = fma(bx, rx, c); T.R[4] = fma(by, ry, c); T.R[8] = fma(bz, rz, c);
  T.R[3] = fma(bx, ry, -az); T.R[1] = fma(bx, ry, az);
  T.R[6] = fma(bx, rz, ay);  T.R[2] = fma(bx, rz, -ay);
  T.R[7] = fma(by, rz, -ax); T.R[5] = fma(by, rz, ax);
  if (n2 < kSmallAngle2) {
    T.t[0] = px; T.t[1] = py; T.t[2] = pz;
    return;
  }
  const double cd = C * (rx * px + ry * py + rz * pz);
  T.t[0] = fma(A, px, fma(cd, rx, B * (ry * pz - rz * py)));
  T.t[1] = fma(A, py, fma(cd, ry, B * (rz * px - rx * pz)));
  T.t[2] = fma(A, pz, fma(cd, rz, B * (rx * py - ry * px)));
}

// Fixed-shape pairwise sum of N values: (first half) + (second half), recursively.  The order every fold of the
// partial sums uses, in both loop variants (hence the same bits), instead of a left-to-right chain: a dependent
// fp64 add costs ~32 cycles on one wave (tools/micro/valu_chain.hip), so 16 values take 4 steps instead of 15.
template <int N>
__device__ __forceinline__ double tree_sum(const double* v) {
  if constexpr (N == 1) return v[0];
  else return tree_sum<N / 2>(v) + tree_sum<N - N / 2>(v + N / 2);
}

// What every wave needs to run a round: the total pose, or the news that the loop has ended.
struct RoundHead {
  Pose total;
  bool stop;
};

// LDS a workgroup of BLOCK threads needs for the prologue.
template <int BLOCK>
struct PrologueShared {
  double fin[BLOCK / kSlots][kSlots];
  double totals[kSlots];
  double work[kSolveWork];
  double pose[12];
  int stop;
};

// The two halves of a round's prologue, run by EVERY workgroup (identical arithmetic on identical
// inputs, so every workgroup derives the same pose).  Reference: the merge, solve, compose and
// convergence test of src/Registration.cpp:71-79, :20-25.
//
// prologue_fold: issue every global load (previous rows, state) before the first wait, fold the rows
// in a fixed order into LDS, return the OLD pose.  prologue_solve (wave 0 only): finish the fold,
// pivoted LDLT, se(3) exponential, compose, convergence test; the new pose goes to LDS, and from
// workgroup 0 to the state / log the next launch and the host read.
template <int BLOCK>
__device__ __forceinline__ RoundHead prologue_fold(const IterArgs& a, PrologueShared<BLOCK>& sh,
                                                   int& it, int& max_it, double& cos_thr,
                                                   double& tsq_thr) {
  constexpr int kGroups = BLOCK / kSlots;
  constexpr int kBatch = 16;  // independent loads in flight per thread; the add order stays fixed
  const uint32_t tid = threadIdx.x;
  const AlignState* in = a.state_in;

  const uint32_t slot = tid & (kSlots - 1), group = tid / kSlots;
  double row[kBatch];
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const uint32_t b = group + u * kGroups;
    row[u] = b < a.prev_rows ? a.prev[(size_t)b * kSlots + slot] : 0.0;
  }
  RoundHead head;
#pragma unroll
  for (int k = 0; k < 9; ++k) head.total.R[k] = in->pose[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) head.total.t[k] = in->pose[9 + k];
  it = in->iteration;
  max_it = in->max_iteration;
  cos_thr = in->cosine_threshold;
  tsq_thr = in->translation_sq_threshold;
  head.stop = in->done != 0;
  if (head.stop || a.prev_rows == 0) {
    // nothing to fold: either the loop ended in an earlier launch, or this is the first round
    if (blockIdx.x == 0 && tid == 0) *a.state_out = *in;
    return head;
  }
  double s = tree_sum<kBatch>(row);  // rows group, group + kGroups, ...: the 16 rows a folder of the persistent launch adds
  for (uint32_t b0 = group + kBatch * kGroups; b0 < a.prev_rows; b0 += kGroups * kBatch) {
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const uint32_t b = b0 + u * kGroups;
      row[u] = b < a.prev_rows ? a.prev[(size_t)b * kSlots + slot] : 0.0;
    }
    s += tree_sum<kBatch>(row);
  }
  sh.fin[group][slot] = s;
  return head;
}

template <int BLOCK>
__device__ __forceinline__ void prologue_solve(const IterArgs& a, PrologueShared<BLOCK>& sh,
                                               const Pose& old_total, uint32_t lane, int it,
                                               int max_it, double cos_thr, double tsq_thr) {
  constexpr int kGroups = BLOCK / kSlots;
  double tot = 0.0;
  if (lane < kSlots) {
    double part[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) part[g] = sh.fin[g][lane];
    tot = tree_sum<kGroups>(part);
  }
  // the 27 sums to every lane (same wave wrote them: LDS is in order within a wave)
  if (lane < kSlots) sh.totals[lane] = tot;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double A[21], g[6], xi[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) A[k] = sh.totals[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = -sh.totals[21 + k];
  if (!ldlt6_solve_spd(A, g, xi)) ldlt6_solve_pivoted(sh.totals, sh.work, lane, xi);  // uniform branch
  Pose step, next;
  se3_exp_device(xi, step);
  pose_compose(step, old_total, next);
  const bool conv = converged(step, cos_thr, tsq_thr);
  const bool stop = conv || (it + 1 >= max_it);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) sh.pose[k] = next.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) sh.pose[9 + k] = next.t[k];
    sh.stop = stop ? 1 : 0;
  }
  if (blockIdx.x == 0) {
    AlignState* out = a.state_out;
    if (lane < kSlots) a.log[(size_t)it * kSlots + lane] = tot;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) { out->pose[k] = next.R[k]; out->step[k] = step.R[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { out->pose[9 + k] = next.t[k]; out->step[9 + k] = step.t[k]; }
      out->cosine_threshold = cos_thr;
      out->translation_sq_threshold = tsq_thr;
      out->max_iteration = max_it;
      out->iteration = it + 1;
      out->converged = conv ? 1 : 0;
      out->done = stop ? 1 : 0;
    }
  }
}

__device__ __forceinline__ void load_point(const double* scan, uint64_t stride, uint32_t i,
                                           double (&q)[kScanPlanes]) {
  const double* s = scan + i;
#pragma unroll
  for (int k = 0; k < kScanPlanes; ++k) q[k] = s[k * stride];
}

// The same for a scan whose covariances are ALL bitwise symmetric (pack_scan_kernel found no exception): the three
// planes above the diagonal are copies of the three below it and are not read — 72 instead of 96 bytes per point.
__device__ __forceinline__ void load_point_sym(const double* scan, uint64_t stride, uint32_t i,
                                               double (&q)[kScanPlanes], bool symmetric) {
  if (!symmetric) {
    load_point(scan, stride, i, q);
    return;
  }
  const double* s = scan + i;
  // planes: x y z c00 c10 c20 c01 c11 c21 c02 c12 c22
  q[0] = s[0]; q[1] = s[stride]; q[2] = s[2 * stride];
  q[3] = s[3 * stride]; q[4] = s[4 * stride]; q[5] = s[5 * stride];
  q[7] = s[7 * stride]; q[8] = s[8 * stride]; q[11] = s[11 * stride];
  q[6] = q[4]; q[9] = q[5]; q[10] = q[8];
}

// One correspondence: p (already in the map frame), scan covariance C, voxel mean / covariance.
// ICP::computeJTJAndJTr in structured form (J = [I | -[p]x]); S holds C_voxel on entry.
template <bool B>
struct Flag { static constexpr bool value = B; };

// FIRST: v holds nothing yet (the thread's first match of the round): the 28 values are stored, not added to
// zeros — 28 dependent-latency adds less per round in the one-point-per-thread case.  Both loop variants make the
// same choice for the same point, so they still agree bit for bit.
template <bool FIRST>
__device__ __forceinline__ void accumulate_match(const double* R, const double (&p)[3],
                                                 const double (&C)[9], const double (&mu)[3],
                                                 double (&S)[9], double (&v)[kSlots]) {
  // S = R C R^T + C_voxel
  double RC[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      RC[r + 3 * c] = R[r] * C[3 * c] + R[r + 3] * C[1 + 3 * c] + R[r + 6] * C[2 + 3 * c];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      S[r + 3 * c] += RC[r] * R[c] + RC[r + 3] * R[c + 3] + RC[r + 6] * R[c + 6];

  // W = S^-1 by cofactors as Eigen's fixed-size inverse (vgicp_math.h inv3), the one division replaced by
  // v_rcp_f64 + two Newton steps (~1 ulp; a third of the dependent instructions)
  double W[9];
  {
    const double c00 = S[4] * S[8] - S[7] * S[5], c10 = S[5] * S[6] - S[8] * S[3], c20 = S[3] * S[7] - S[6] * S[4];
    const double det = c00 * S[0] + c10 * S[1] + c20 * S[2];
    const double id = rcp_newton(det);
    const double c01 = S[7] * S[2] - S[1] * S[8], c11 = S[8] * S[0] - S[2] * S[6], c21 = S[6] * S[1] - S[0] * S[7];
    const double c02 = S[1] * S[5] - S[4] * S[2], c12 = S[2] * S[3] - S[5] * S[0], c22 = S[0] * S[4] - S[3] * S[1];
    W[0] = c00 * id; W[3] = c10 * id; W[6] = c20 * id;
    W[1] = c01 * id; W[4] = c11 * id; W[7] = c21 * id;
    W[2] = c02 * id; W[5] = c12 * id; W[8] = c22 * id;
  }
  auto acc = [](double& dst, double x) { if (FIRST) dst = x; else dst += x; };
  const double e0 = p[0] - mu[0], e1 = p[1] - mu[1], e2 = p[2] - mu[2];
  // Q = [p]x W  (rows 3..5, columns 0..2 of J^T Sigma^-1 J)
  double Q[9];  // Q[r + 3c]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Q[0 + 3 * c] = p[1] * W[2 + 3 * c] - p[2] * W[1 + 3 * c];
    Q[1 + 3 * c] = p[2] * W[0 + 3 * c] - p[0] * W[2 + 3 * c];
    Q[2 + 3 * c] = p[0] * W[1 + 3 * c] - p[1] * W[0 + 3 * c];
  }
  // lower triangle of J^T Sigma^-1 J, row by row
  acc(v[0], W[0]);
  acc(v[1], W[1]); acc(v[2], W[4]);
  acc(v[3], W[2]); acc(v[4], W[5]); acc(v[5], W[8]);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int base = (3 + r) * (4 + r) / 2;
    const double q0 = Q[r], q1 = Q[r + 3], q2 = Q[r + 6];
    acc(v[base + 0], q0); acc(v[base + 1], q1); acc(v[base + 2], q2);
    acc(v[base + 3], q2 * p[1] - q1 * p[2]);
    if (r >= 1) acc(v[base + 4], q0 * p[2] - q2 * p[0]);
    if (r >= 2) acc(v[base + 5], q1 * p[0] - q0 * p[1]);
  }
  // J^T Sigma^-1 r
  acc(v[21], W[0] * e0 + W[3] * e1 + W[6] * e2);
  acc(v[22], W[1] * e0 + W[4] * e1 + W[7] * e2);
  acc(v[23], W[2] * e0 + W[5] * e1 + W[8] * e2);
  acc(v[24], Q[0] * e0 + Q[3] * e1 + Q[6] * e2);
  acc(v[25], Q[1] * e0 + Q[4] * e1 + Q[7] * e2);
  acc(v[26], Q[2] * e0 + Q[5] * e1 + Q[8] * e2);
  acc(v[kCountSlot], 1.0);
}

constexpr uint32_t kMemoMiss = 0xFFFFFFFFu;  // memo.w of a point whose voxel is not in the map
constexpr uint32_t kMemoNone = 0xFFFFFFFEu;  // nothing looked up

// Payload of a record whose slot is known (an unchanged key that hit last round): one round trip.
__device__ __forceinline__ void load_payload(const VoxelRecord* rec, double (&mu)[3], double (&S)[9]) {
  const double2* pay = reinterpret_cast<const double2*>(rec->mean);
  const double2 a0 = pay[0], a1 = pay[1], a2 = pay[2], a3 = pay[3], a4 = pay[4], a5 = pay[5];
  mu[0] = a0.x; mu[1] = a0.y; mu[2] = a1.x;
  S[0] = a1.y; S[1] = a2.x; S[2] = a2.y; S[3] = a3.x; S[4] = a3.y; S[5] = a4.x;
  S[6] = a4.y; S[7] = a5.x; S[8] = a5.y;
}

// One VGICP round.  Launch j reads state j&1 and the rows launch j-1 wrote, writes state (j+1)&1
// and its own rows; the host alternates the buffers; the kernel boundary is the only synchronisation
// between workgroups.  Inside a workgroup wave 0 is the SOLVER (it owns no points); waves 1.. are
// workers, BLOCK-64 points per workgroup pass:
//   all      issue the loads of the previous rows, the state and (workers) the first point AND ITS MEMO; fold rows
//   solver   LDLT solve, exponential, compose, convergence -> new pose in LDS       } concurrently,
//   workers  fetch the payload of the voxel the memo names (nothing to compute)     } between two barriers
//   workers  transform with the NEW pose; "still inside last round's voxel?" (same_voxel_coord, the persistent launch's
//            test) — almost always yes: the payload is in registers, or the point is known to have no voxel and costs
//            nothing; a changed key is probed and remembered
//   workers  accumulate, grid-stride over further points, butterfly, one 256-byte row per workgroup.
// Round 6 brought the loop to the persistent launch's data path (it is what an RCCL communicator, a multi-device
// context after a give-up and every VGICP_FLAG_NO_PERSISTENT align run): a 16-byte MEMO per point in HBM {key, slot or
// index in the dense record copy} — the map is immutable during an align (src/LocalMap.cpp:94-100), so an unchanged
// key that missed costs 16 bytes instead of a 128-byte line, and one that hit goes straight to its payload (rounds
// 1-5 transformed with the OLD pose, hashed and probed every point in every launch: 1.66 x the algorithmic bytes at C2);
// nine instead of twelve planes of a scan whose covariances are all bitwise symmetric; the dense record copy of tables
// beyond the caches' reach.  The arithmetic per match is unchanged, so the loop still returns the persistent launch's bits.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void iterate_kernel(IterArgs a) {
  constexpr int kWaves = BLOCK / 64;
  constexpr int kWorkers = BLOCK - 64;
  __shared__ PrologueShared<BLOCK> sh;
  __shared__ double red[kWaves][kSlots];

  const uint64_t t_begin = a.stamps ? wall_clock64() : 0;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool worker = wave != 0;
  const uint32_t stride_pts = gridDim.x * kWorkers;
  const double inv_voxel = 1.0 / a.voxel_size;
  const double same_margin = 0x1p-20 * a.voxel_size;  // same_voxel_coord
  const bool cov_sym = a.asym_dev != nullptr && *a.asym_dev != a.scan_seq;  // uniform
  const bool remembered = a.memo != nullptr && a.memo_valid != 0;           // uniform: an earlier launch of this align wrote the memos
  const VoxelRecord* pay_base = a.dense ? a.dense : a.table;                // uniform: where a remembered record's payload is read from
  uint32_t i = worker ? blockIdx.x * kWorkers + (tid - 64) : a.n;

  double q[kScanPlanes];
#pragma unroll
  for (int k = 0; k < kScanPlanes; ++k) q[k] = 0.0;
  int4 m = make_int4(0, 0, 0, (int32_t)kMemoNone);
  if (i < a.n) {
    load_point_sym(a.scan, a.stride, i, q, cov_sym);
    if (remembered) m = a.memo[i];
  }

  int it, max_it;
  double cos_thr, tsq_thr;
  RoundHead head = prologue_fold<BLOCK>(a, sh, it, max_it, cos_thr, tsq_thr);
  if (head.stop) return;  // uniform across the grid: the loop ended in an earlier launch
  const bool solving = a.prev_rows != 0;  // uniform
  if (solving) __syncthreads();
  const uint64_t t_fold = a.stamps ? wall_clock64() : 0;

  // the payload the memo names (workers) || solve (wave 0)
  double mu[3] = {0.0, 0.0, 0.0}, S[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) S[k] = 0.0;
  bool have_payload = false;
  if (worker) {
    if (i < a.n && remembered && (uint32_t)m.w < kMemoNone) {
      load_payload(pay_base + (uint32_t)m.w, mu, S);
      have_payload = true;
    }
  } else if (solving) {
    prologue_solve<BLOCK>(a, sh, head.total, lane, it, max_it, cos_thr, tsq_thr);
    if (a.stamps && blockIdx.x == 0 && tid == 0)
      atomicAdd((unsigned long long*)&a.stamps[6], (unsigned long long)(wall_clock64() - t_fold));
  }
  if (solving) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) head.total.R[k] = sh.pose[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) head.total.t[k] = sh.pose[9 + k];
    head.stop = sh.stop != 0;
    if (head.stop) return;  // uniform: converged, or max_iteration reached
  }
  const uint64_t t_head = a.stamps ? wall_clock64() : 0;
  const double* R = head.total.R;
  const double* t = head.total.t;

  double v[kSlots];
#pragma unroll
  for (int k = 0; k < kSlots; ++k) v[k] = 0.0;

  bool first = true;
  while (i < a.n) {
    const double x = q[0], y = q[1], z = q[2];
    double C[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) C[k] = q[3 + k];
    const uint32_t icur = i, inext = i + stride_pts;
    const int4 mcur = m;
    if (inext < a.n) {  // prefetch the next point (and its memo) under this one's gather
      load_point_sym(a.scan, a.stride, inext, q, cov_sym);
      if (remembered) m = a.memo[inext];
    }
    i = inext;

    double p[3];
    transform_point(R, t, x, y, z, p);
    bool hit;
    // still inside last round's voxel?  (true implies an unchanged key; the key itself is made only otherwise)
    const bool stayed = remembered && (uint32_t)mcur.w != kMemoNone &&
                        (((int)same_voxel_coord(p[0], mcur.x, a.voxel_size, same_margin) &
                          (int)same_voxel_coord(p[1], mcur.y, a.voxel_size, same_margin) &
                          (int)same_voxel_coord(p[2], mcur.z, a.voxel_size, same_margin)) != 0);
    if (stayed) {
      hit = (uint32_t)mcur.w != kMemoMiss;
      if (hit && !(first && have_payload)) load_payload(pay_base + (uint32_t)mcur.w, mu, S);
    } else {
      const int32_t kx = voxel_coord_fast(p[0], a.voxel_size, inv_voxel);
      const int32_t ky = voxel_coord_fast(p[1], a.voxel_size, inv_voxel);
      const int32_t kz = voxel_coord_fast(p[2], a.voxel_size, inv_voxel);
      if (remembered && (uint32_t)mcur.w != kMemoNone && mcur.x == kx && mcur.y == ky && mcur.z == kz) {
        hit = (uint32_t)mcur.w != kMemoMiss;   // a point right below a face: key unchanged after all
        if (hit && !(first && have_payload)) load_payload(pay_base + (uint32_t)mcur.w, mu, S);
      } else {
        const VoxelRecord* rec = find_voxel(a.table, a.mask, kx, ky, kz);
        hit = rec != nullptr;
        if (a.memo) {
          uint32_t where = kMemoMiss;
          if (hit) where = a.dense ? reinterpret_cast<const uint32_t*>(&rec->reserved)[1] : (uint32_t)(rec - a.table);
          a.memo[icur] = make_int4(kx, ky, kz, (int32_t)where);
        }
        if (hit) load_payload(rec, mu, S);
      }
    }
    if (hit) {
      if (first) accumulate_match<true>(R, p, C, mu, S, v);  // the thread's first point of the round
      else accumulate_match<false>(R, p, C, mu, S, v);
    }
    first = false;
  }
  const uint64_t t_loop = a.stamps ? wall_clock64() : 0;

  // ---- wave: 32-slot halving butterfly; lane l ends with slot (l >> 1) summed over 64 lanes ----
  if (worker) {
    fold_swap<32, false>(v);
    fold_swap<16, true>(v);
    fold<8, 8>(v, (lane & 8) != 0);
    fold<4, 4>(v, (lane & 4) != 0);
    fold<2, 2>(v, (lane & 2) != 0);
    const double wsum = v[0] + xor_lane_f64<1>(v[0]);
    if ((lane & 1) == 0) red[wave][lane >> 1] = wsum;
  }
  __syncthreads();
  // ---- workgroup: fixed-order sum over the worker waves, one plain 256-byte row per workgroup ----
  if (tid < kSlots) {
    double w_sum[kWaves - 1];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) w_sum[w - 1] = red[w][tid];
    a.rows[(size_t)blockIdx.x * kSlots + tid] = tree_sum<kWaves - 1>(w_sum);
  }
  if (a.stamps && blockIdx.x == 0 && tid == 64) {
    const uint64_t t_end = wall_clock64();
    atomicAdd((unsigned long long*)&a.stamps[0], (unsigned long long)(t_fold - t_begin));
    atomicAdd((unsigned long long*)&a.stamps[5], (unsigned long long)(t_head - t_fold));
    atomicAdd((unsigned long long*)&a.stamps[1], (unsigned long long)(t_loop - t_head));
    atomicAdd((unsigned long long*)&a.stamps[2], (unsigned long long)(t_end - t_loop));
    atomicAdd((unsigned long long*)&a.stamps[4], 1ull);
  }
}

// The launch after the last round: prologue only (fold the last rows, solve, publish the final
// state), one workgroup.  A kernel of its own so per-kernel profiles of iterate_kernel hold rounds only.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void close_kernel(IterArgs a) {
  __shared__ PrologueShared<BLOCK> sh;
  int it, max_it;
  double cos_thr, tsq_thr;
  const RoundHead head = prologue_fold<BLOCK>(a, sh, it, max_it, cos_thr, tsq_thr);
  if (head.stop || a.prev_rows == 0) return;
  __syncthreads();
  if (threadIdx.x < 64) prologue_solve<BLOCK>(a, sh, head.total, threadIdx.x, it, max_it, cos_thr, tsq_thr);
}

// Test hook (vgicp_solve_step): the tail of one round on given normal equations, run by ONE wave with
// the very device functions the loop kernels inline — JTJ.ldlt().solve(-JTr), Utils::se3ToSE3 and
// ICP::convergenceCheck (src/Registration.cpp:78-79,37-50, src/Utils.cpp:40-63).
// packed: 21 lower-triangle entries row by row, then the 6 entries of JTr.
// out: [0..5] se3, [6..14] step R (column-major), [15..17] step t, [18] 1.0 when the pivoted
// (Eigen-faithful) solve produced the result, [19] 1.0 when the step passes the convergence test.
__global__ __launch_bounds__(64) void solve_step_kernel(const double* __restrict__ packed, double cos_thr,
                                                        double tsq_thr, int force_pivoted,
                                                        double* __restrict__ out) {
  __shared__ double totals[kSlots];
  __shared__ double work[kSolveWork];
  const uint32_t lane = threadIdx.x;
  if (lane < kSlots) totals[lane] = lane < kNormalEq ? packed[lane] : 0.0;
  __syncthreads();
  double A[21], g[6], xi[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) A[k] = packed[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = -packed[21 + k];
  bool pivoted = force_pivoted != 0;
  if (pivoted || !ldlt6_solve_spd(A, g, xi)) {  // uniform branch
    ldlt6_solve_pivoted(totals, work, lane, xi);
    pivoted = true;
  }
  Pose step;
  se3_exp_device(xi, step);
  const bool conv = converged(step, cos_thr, tsq_thr);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = xi[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) out[6 + k] = step.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[15 + k] = step.t[k];
    out[18] = pivoted ? 1.0 : 0.0;
    out[19] = conv ? 1.0 : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// Persistent variant: the whole loop of ICP::align (src/Registration.cpp:15-28) in ONE launch.
//
// Same arithmetic, same fixed summation orders and the same workgroup geometry as iterate_kernel<512>
// (so both variants return the same bits); what changes is how a round's partial rows travel.  There is
// no flag, no counter and no drain on the path: the DATA is the signal.  Every 8-byte word of the exchange
// buffers holds kRowUnset until its producer stores the value with ONE write-through (sc1) 8-byte store —
// a naturally aligned granule written by one store is never seen torn (MI355X_MICROARCH.md, "R2's granule")
// — and consumers poll the words they need with sc1 loads until none is kRowUnset:
//   level 1  wave 0 of workgroup b stores its row (28 words).  Folder g (= workgroup g < 16) polls the rows
//            b = g, g + 16, g + 32 ... (lane = slot, up to 16 loads in flight), adds them in ascending b and
//            stores the part
//   level 2  wave 0 of EVERY workgroup polls the <= 16 parts, adds them in ascending g, solves
// which is the order in which prologue_fold / prologue_solve add the rows (thread (g, slot) adds rows
// g + 16u, then the 16 group sums are added in ascending g).  Two one-way hops instead of store + drain +
// atomic + poll + row read, 57 KB less to read per workgroup and round, and two workgroup barriers per
// round instead of five.
// Re-arming (three buffers by round % 3): at the end of round j (all parts of round j seen) a workgroup
// stores kRowUnset over its row and, if a folder, its part of buffer (j - 1) % 3.  Safe: every part of
// round j exists, so every row of round j was published, so every workgroup had finished reading round
// j - 1.  Ordered: the producer's next publication (round j + 1) waits for its own stores to complete
// first (s_waitcnt vmcnt(0)), and a consumer polls buffer (j - 1) % 3 again only in round j + 2, after it
// has seen a part of round j + 1 that depends on that publication.  At the end of the launch every
// workgroup re-arms its row of the last round, arrives at the exit counter, and the LAST arriver (everyone
// has read the last parts by then) re-arms the parts: the buffers are all kRowUnset between launches.
// What stays on chip across rounds: a thread's first point with the voxel record it used (registers);
// for scans larger than the grid, per further point its last key and table slot (16-byte memo in LDS: an
// unchanged key that missed costs no table access at all — the map is immutable during align,
// src/LocalMap.cpp:94-100 — and one that hit loads its payload straight from the slot), and up to
// stash_points whole points (LDS).
// Needs every workgroup resident (grid <= CUs, one 512-thread workgroup per CU); a poll that exceeds
// spin_limit ends the kernel without the final state, and the host re-runs the align with iterate_kernel.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void store_through_bits(double* p, unsigned long long bits) {
  __hip_atomic_store((gu64*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_through_bits(const double* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Mailbox words live in fine-grained memory of this or another GPU: system scope (sc0 sc1) both ways.
__device__ __forceinline__ void store_system_bits(double* p, unsigned long long bits) {
  __hip_atomic_store((gu64*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_system_bits(const double* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A sum as it is published: never the "unset" pattern (a NaN with that payload can only come from
// non-finite input; it is published as the canonical NaN and the align ends as VGICP_ERR_DEGENERATE).
__device__ __forceinline__ unsigned long long publishable(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return b == kRowUnset ? kRowNaN : b;
}

// Lanes <= kCountSlot (lane = slot): wait until the 16 consecutive rows at src are published and return
// the sum of their words as the pairwise tree over ascending rows (tree_sum<16>, as the folds of iterate_kernel
// do; rows that belong to no workgroup hold +0.0 for good).  All 64 lanes of the wave call it; false when
// spin_limit polls did not suffice.  The 16 loads are in flight together (one address, immediate offsets).
#ifndef VGICP_L2_DELAY
#define VGICP_L2_DELAY 64      // s_sleep units (64 clocks each) a workgroup that is no folder waits before its first poll of the parts
#endif
// Lanes <= kCountSlot (lane = slot) end with the sum over the 16 consecutive rows at src, as the pairwise tree over
// ascending rows (tree_sum<16>, as the folds of iterate_kernel do; rows that belong to no workgroup hold +0.0 for
// good).  All 64 lanes of the wave call it; false when spin_limit polls did not suffice.
// The two halves of the wave share the work (round 6): lane l < 32 polls rows 0..7 of slot l, lane l + 32 rows 8..15, and
// tree_sum<16> IS tree_sum<8>(first half) + tree_sum<8>(second half), so one half-swap gives the same bits with half the
// registers and half the loads per lane.  ONE poll of the eight words is in flight: two, three or four of them, issued a
// fraction of a memory round trip apart so that a word which lands just after a poll has passed need not wait a whole
// round trip for the next, made every round SLOWER (C2 +0.3 / +0.9 / +1.5 us: the polls of 256 workgroups queue at the
// memory side, profiles/NOTES_dropped_experiments.md) — which is what led to VGICP_L2_DELAY: poll LESS.
template <bool SYSTEM>
__device__ __forceinline__ bool poll_and_sum(const double* src, uint32_t lane, uint32_t spin_limit, double& sum) {
  constexpr int H = kFolders / 2;
  const uint32_t slot = lane & 31u, half = lane >> 5;
  const bool active = slot <= (uint32_t)kCountSlot;
  const double* mine = src + (size_t)half * H * kSlots + (active ? slot : 0u);
  unsigned long long w[H];
#pragma unroll
  for (int k = 0; k < H; ++k)
    w[k] = active ? (SYSTEM ? load_system_bits(mine + k * kSlots) : load_through_bits(mine + k * kSlots)) : 0ull;
  for (uint32_t spins = 0;; ++spins) {
    bool missing = false;
#pragma unroll
    for (int k = 0; k < H; ++k) missing = missing || (w[k] == kRowUnset);
    if (!__any(missing)) break;
    if (spins >= spin_limit) return false;
    __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int k = 0; k < H; ++k)
      if (w[k] == kRowUnset) w[k] = SYSTEM ? load_system_bits(mine + k * kSlots) : load_through_bits(mine + k * kSlots);
  }
  double x[H];
#pragma unroll
  for (int k = 0; k < H; ++k) x[k] = __longlong_as_double((long long)w[k]);
  const double part = tree_sum<H>(x);
  // lanes < 32 receive the sum of lane + 32 (rows 8..15 of the same slot)
  const unsigned plo = (unsigned)__double2loint(part), phi = (unsigned)__double2hiint(part);
  const auto lo = __builtin_amdgcn_permlane32_swap(plo, plo, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(phi, phi, false, false);
  sum = part + __hiloint2double((int)hi[1], (int)lo[1]);
  return true;
}

// STAMPS builds only: a clock reading that cannot move across the values it is given (they pass through an empty asm)
template <int N>
__device__ __forceinline__ uint64_t pinned_clock(double (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(x[k]));
  const uint64_t t = wall_clock64();
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(x[k]));
  return t;
}

// MULTI: several GPUs (the rank totals cross xGMI through mailboxes); STAMPS: in-kernel phase clocks
// (VGICP_DEBUG_STAMPS=1).  Separate instantiations: the single-GPU production kernel carries neither.
// MANY: a thread owns several points (scan larger than grid x 448), wave 0 included; every point has a memo in LDS, the
// thread's first point stays in registers (its voxel record does not), some more are parked in LDS.  !MANY: one point
// per thread (waves 1..7), kept in registers together with the
// voxel record it used, plus the neighbour prefetch.  Separate instantiations keep both within 256 VGPRs.
template <int BLOCK, bool MULTI, bool STAMPS, bool MANY>
__device__ __forceinline__ void persistent_body(const PersistArgs& a) {
  static_assert(BLOCK / kSlots == kFolders, "the exchange reproduces the fold order of iterate_kernel<512>");
  constexpr int kWaves = BLOCK / 64;
  constexpr int kWorkers = BLOCK - 64;
  __shared__ double red[kWaves][kSlots];
  __shared__ double totals[kSlots];
  __shared__ double work[kSolveWork];
  __shared__ double pose_sh[12];
  __shared__ int stop_sh;
  // Scans larger than the grid: a thread owns several points. The first stays in registers; of the
  // others, the first a.memo_points have a 16-byte memo {key, slot} and the first a.stash_points are
  // parked whole after round 0 (plane-major per point: conflict-free), so only the rest is re-read.
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  // Scans that fit the grid (one point per thread, no memo / stash): the same LDS holds, per thread, the voxel
  // BEHIND THE NEAREST FACE of the point's voxel, looked up while the workers wait for the exchange: key + slot
  // (int4) and the 96-byte payload (6 planes of double2).  A point that changes voxel next round usually
  // enters exactly that one and finds its record (or the news that there is none) here.
  int4* pf_key = reinterpret_cast<int4*>(dyn_lds);
  double2* pf_pay = reinterpret_cast<double2*>(dyn_lds + (size_t)kWorkers * sizeof(int4));
  const bool prefetch = !MANY && a.prefetch_margin > 0.0;  // uniform

  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a.n bounds the count the device holds (a scan preparation that gave up must not send this launch out of range)
  const uint32_t n_pts = a.n_dev ? (*a.n_dev < a.n ? *a.n_dev : a.n) : a.n;  // uniform
  const bool cov_sym = MANY && a.asym_dev != nullptr && *a.asym_dev != a.scan_seq;  // uniform
  const uint32_t grid = gridDim.x, blk = blockIdx.x;
  // MANY (the scan, or the upper bound the launch plan was made from, has more points than grid x 448): a thread owns
  // several points anyway, so wave 0 owns points as well and turns solver after the accumulate phase — 512 points
  // per workgroup pass instead of 448, two point-carrying waves on every SIMD instead of 2/2/2/1.
  constexpr uint32_t W = MANY ? (uint32_t)BLOCK : (uint32_t)kWorkers;  // point-carrying threads of a workgroup
  const bool worker = MANY || wave != 0;
  const uint32_t wt = MANY ? tid : tid - 64u;  // index among them (workers only)
  // Which points a thread owns.  One point per thread (!MANY): point blk x 448 + its index among the workers.  MANY: the
  // scan is dealt out in UNITS of 64 consecutive points (one wave's worth), every workgroup a contiguous run of
  // floor(U / grid) units and the first U mod grid workgroups one more, taken eight units (512 points) per pass —
  // so all workgroups hold the same number of points to within one unit.  (Dealing whole passes of grid x 512
  // points out instead left the last, partial pass to the first workgroups only: at C5 161 workgroups of 8 points
  // per thread and 95 of 7, 23 against 20 us to the barrier, and the round waits for the slowest.)
  uint32_t stride_pts = grid * W, first = worker ? blk * W + wt : n_pts, end_pts = n_pts;
  if constexpr (MANY) {
    const uint32_t units = (n_pts + 63u) / 64u, share = units / grid, extra = units % grid;  // uniform
    const uint32_t u0 = blk * share + (blk < extra ? blk : extra), u1 = u0 + share + (blk < extra ? 1u : 0u);
    stride_pts = W;
    first = u0 * 64u + wt;
    end_pts = u1 * 64u < n_pts ? u1 * 64u : n_pts;
  }
  const VoxelRecord* pay_base = a.dense ? a.dense : a.table;  // uniform: where a remembered record's payload is read from
  int4* memo = reinterpret_cast<int4*>(dyn_lds);
  double* stash = reinterpret_cast<double*>(dyn_lds + (size_t)a.memo_points * W * sizeof(int4));
  // parked points hold the planes that are read: 9 of a scan whose covariances are all bitwise symmetric, else 12
  const uint32_t park_planes = cov_sym ? 9u : (uint32_t)kScanPlanes;
  const uint32_t fit = a.stash_bytes / (park_planes * W * (uint32_t)sizeof(double));
  const uint32_t parked = MANY ? (fit < a.stash_points ? fit : a.stash_points) : 0u;  // uniform
  const bool folder = blk < (uint32_t)kFolders;  // folder g adds the rows of workgroups g, g + 16, g + 32 ...
  // row of workgroup b inside a buffer: the 16 rows of a folder are consecutive
  const uint32_t my_row = (blk % kFolders) * kFolders + blk / kFolders;

  Pose total;
#pragma unroll
  for (int k = 0; k < 9; ++k) total.R[k] = a.pose0[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) total.t[k] = a.pose0[9 + k];
  const double cos_thr = a.cosine_threshold, tsq_thr = a.translation_sq_threshold;
  const int max_it = a.max_iteration;
  const double inv_voxel = 1.0 / a.voxel_size;
  const double same_margin = 0x1p-20 * a.voxel_size;  // same_voxel_coord

  double q0[kScanPlanes];
#pragma unroll
  for (int k = 0; k < kScanPlanes; ++k) q0[k] = 0.0;
  const bool have = first < end_pts;
  if (have) load_point(a.scan, a.stride, first, q0);

  // what the first point used last round: key, hit flag, voxel payload (raw)
  bool spec = false, hit = false;
  int32_t okx = 0, oky = 0, okz = 0;
  double mu[3] = {0.0, 0.0, 0.0}, Sv[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Sv[k] = 0.0;

  uint64_t t_mark = STAMPS ? wall_clock64() : 0;
  uint64_t acc_body = 0, acc_l1 = 0, acc_l2 = 0, acc_solve = 0;
  uint64_t fine[4] = {0, 0, 0, 0};   // STAMPS: inside "solve + broadcast": totals to registers, LDL^T, exp + compose, publication
  bool conv_keep = false;            // wave 0: the last round's convergence verdict, for the final state
  int it = 0;
  bool gave_up = false;
  for (;;) {
    const double* R = total.R;
    const double* t = total.t;
    if (worker) {
      double v[kSlots];
#pragma unroll
      for (int k = 0; k < kSlots; ++k) v[k] = 0.0;
      if constexpr (!MANY) {
        if (have) {
          double p[3], C[9], S[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) C[k] = q0[3 + k];
          transform_point(R, t, q0[0], q0[1], q0[2], p);
          const int32_t kx = voxel_coord_fast(p[0], a.voxel_size, inv_voxel);
          const int32_t ky = voxel_coord_fast(p[1], a.voxel_size, inv_voxel);
          const int32_t kz = voxel_coord_fast(p[2], a.voxel_size, inv_voxel);
          if (!spec || kx != okx || ky != oky || kz != okz) {
            bool served = false;
            if (spec && prefetch) {
              const int4 m = pf_key[tid - 64];
              if (m.x == kx && m.y == ky && m.z == kz && (uint32_t)m.w != kMemoNone) {
                served = true;
                hit = (uint32_t)m.w != kMemoMiss;
                if (hit) {
                  const double2* pay = pf_pay + (tid - 64);
                  const double2 a0 = pay[0 * kWorkers], a1 = pay[1 * kWorkers], a2 = pay[2 * kWorkers],
                                a3 = pay[3 * kWorkers], a4 = pay[4 * kWorkers], a5 = pay[5 * kWorkers];
                  mu[0] = a0.x; mu[1] = a0.y; mu[2] = a1.x;
                  Sv[0] = a1.y; Sv[1] = a2.x; Sv[2] = a2.y; Sv[3] = a3.x; Sv[4] = a3.y; Sv[5] = a4.x;
                  Sv[6] = a4.y; Sv[7] = a5.x; Sv[8] = a5.y;
                }
              }
            }
            if (!served) hit = find_and_load(a.table, a.mask, kx, ky, kz, mu, Sv);
            okx = kx; oky = ky; okz = kz;
            spec = true;
          }
          if (hit) {
#pragma unroll
            for (int k = 0; k < 9; ++k) S[k] = Sv[k];
            accumulate_match<true>(R, p, C, mu, S, v);
          }
        }
      } else {
        // one point: transform, key, the memo of point e (if it has one), accumulate
        auto one_point = [&](const double (&q)[kScanPlanes], uint32_t e, auto first_of_round) {
          double p[3], C[9], m2[3], S[9];
          transform_point(R, t, q[0], q[1], q[2], p);
          bool got;
          if (e < a.memo_points) {
            int4* mslot = memo + (size_t)e * W + wt;
            const int4 m = *mslot;
            // still inside last round's voxel?  (true implies an unchanged key; the key itself is made only otherwise)
            const bool stayed = ((int)(it != 0) & (int)same_voxel_coord(p[0], m.x, a.voxel_size, same_margin) &
                                 (int)same_voxel_coord(p[1], m.y, a.voxel_size, same_margin) &
                                 (int)same_voxel_coord(p[2], m.z, a.voxel_size, same_margin)) != 0;
            if (stayed) {
              got = (uint32_t)m.w != kMemoMiss;
              if (got) load_payload(pay_base + (uint32_t)m.w, m2, S);
            } else {
              const int32_t kx = voxel_coord_fast(p[0], a.voxel_size, inv_voxel);
              const int32_t ky = voxel_coord_fast(p[1], a.voxel_size, inv_voxel);
              const int32_t kz = voxel_coord_fast(p[2], a.voxel_size, inv_voxel);
              if (it != 0 && m.x == kx && m.y == ky && m.z == kz) {  // a point right below a face: key unchanged after all
                got = (uint32_t)m.w != kMemoMiss;
                if (got) load_payload(pay_base + (uint32_t)m.w, m2, S);
              } else {
                const VoxelRecord* rec = find_voxel(a.table, a.mask, kx, ky, kz);
                got = rec != nullptr;
                // what the memo remembers: the record's slot, or (dense copy in use) its index there, which the
                // compaction left in the upper half of the record's spare word — the same 128-byte line as the key
                uint32_t where = kMemoMiss;
                if (got) where = a.dense ? reinterpret_cast<const uint32_t*>(&rec->reserved)[1] : (uint32_t)(rec - a.table);
                *mslot = make_int4(kx, ky, kz, (int32_t)where);
                if (got) load_payload(rec, m2, S);
              }
            }
          } else {
            const int32_t kx = voxel_coord_fast(p[0], a.voxel_size, inv_voxel);
            const int32_t ky = voxel_coord_fast(p[1], a.voxel_size, inv_voxel);
            const int32_t kz = voxel_coord_fast(p[2], a.voxel_size, inv_voxel);
            got = find_and_load(a.table, a.mask, kx, ky, kz, m2, S);
          }
          if (got) {
#pragma unroll
            for (int k = 0; k < 9; ++k) C[k] = q[3 + k];
            accumulate_match<decltype(first_of_round)::value>(R, p, C, m2, S, v);
          }
        };
        // the thread's first point stays in registers (its voxel record does not: the memo finds it), the next
        // a.stash_points are parked in LDS after round 0, the rest is re-read from HBM every round
        if (have) one_point(q0, 0u, Flag<true>{});
        uint32_t e = 1;
        for (uint32_t i = first + stride_pts; i < end_pts; i += stride_pts, ++e) {
          double q[kScanPlanes];
          if (e <= parked) {
            double* slot = stash + (size_t)(e - 1) * park_planes * W + wt;
            if (it == 0) {
              load_point_sym(a.scan, a.stride, i, q, cov_sym);
              if (cov_sym) {
                constexpr int kUsed[9] = {0, 1, 2, 3, 4, 5, 7, 8, 11};
#pragma unroll
                for (int j = 0; j < 9; ++j) slot[j * W] = q[kUsed[j]];
              } else {
#pragma unroll
                for (int k = 0; k < kScanPlanes; ++k) slot[k * W] = q[k];
              }
            } else if (cov_sym) {
              constexpr int kUsed[9] = {0, 1, 2, 3, 4, 5, 7, 8, 11};
#pragma unroll
              for (int j = 0; j < 9; ++j) q[kUsed[j]] = slot[j * W];
              q[6] = q[4]; q[9] = q[5]; q[10] = q[8];
            } else {
#pragma unroll
              for (int k = 0; k < kScanPlanes; ++k) q[k] = slot[k * W];
            }
          } else {
            load_point_sym(a.scan, a.stride, i, q, cov_sym);
          }
          one_point(q, e, Flag<false>{});
        }
      }
      fold_swap<32, false>(v);
      fold_swap<16, true>(v);
      fold<8, 8>(v, (lane & 8) != 0);
      fold<4, 4>(v, (lane & 4) != 0);
      fold<2, 2>(v, (lane & 2) != 0);
      const double wsum = v[0] + xor_lane_f64<1>(v[0]);
      if ((lane & 1) == 0) red[wave][lane >> 1] = wsum;
    }
    __syncthreads();
    if (STAMPS) { const uint64_t n = wall_clock64(); acc_body += n - t_mark; t_mark = n; }

    if (!MANY && worker && prefetch && have) {
      // While wave 0 exchanges and solves: where is this point inside its voxel?  If a face is nearer than
      // prefetch_margin voxels, look the voxel behind it up now (full probe; 2 dependent round trips that
      // nobody waits for) so that next round's key change finds the record in LDS.
      double p[3], rx, ry, rz;  // r*: distance of the point from the lower face of its voxel, in [0, voxel)
      transform_point(R, t, q0[0], q0[1], q0[2], p);
      (void)voxel_coord_fast(p[0], a.voxel_size, inv_voxel, &rx);
      (void)voxel_coord_fast(p[1], a.voxel_size, inv_voxel, &ry);
      (void)voxel_coord_fast(p[2], a.voxel_size, inv_voxel, &rz);
      const double half = 0.5 * a.voxel_size;
      const double dx = fmin(rx, a.voxel_size - rx), dy = fmin(ry, a.voxel_size - ry), dz = fmin(rz, a.voxel_size - rz);
      int32_t nx = okx, ny = oky, nz = okz;
      double d;
      if (dx <= dy && dx <= dz) { d = dx; nx += rx < half ? -1 : 1; }
      else if (dy <= dz) { d = dy; ny += ry < half ? -1 : 1; }
      else { d = dz; nz += rz < half ? -1 : 1; }
      int4 m = make_int4(nx, ny, nz, (int32_t)kMemoNone);
      if (d < a.prefetch_margin * a.voxel_size) {
        const VoxelRecord* rec = find_voxel(a.table, a.mask, nx, ny, nz);
        m.w = rec ? (int32_t)(uint32_t)(rec - a.table) : (int32_t)kMemoMiss;
        if (rec) {
          const double2* pay = reinterpret_cast<const double2*>(rec->mean);
          double2* dst = pf_pay + (tid - 64);
#pragma unroll
          for (int j = 0; j < 6; ++j) dst[j * kWorkers] = pay[j];
        }
      }
      pf_key[tid - 64] = m;
    }
    // STAMPS: how long the look-ahead kept this wave (reported in the worker lane's "level-1" column)
    if (STAMPS && !MANY && wave != 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const uint64_t n = wall_clock64(); acc_l1 += n - t_mark; t_mark = n; }

    if (wave == 0) {
      // Addresses that depend on the lane are re-made here every round: hoisted out of the loop they would sit in
      // registers through the whole accumulate phase (the MANY instantiations spilled them to scratch).
      uint32_t lane_here = tid & 63;
      asm volatile("" : "+v"(lane_here));
      const uint32_t lane = lane_here;
      // the buffers rotate with a round number that runs on from launch to launch (a.round0: rounds executed on
      // this context before): nothing has to be tidied up when a launch ends
      const uint32_t buf = (a.round0 + (uint32_t)it) % 3u;
      double* rows = a.rows + (size_t)buf * kExchangeRows * kSlots;
      double* parts = a.parts + (size_t)buf * kFolders * kSlots;
      // ---- level 1: publish this workgroup's row (every re-arming store of mine has completed) ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane <= (uint32_t)kCountSlot) {
        double w_sum[kWaves];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) w_sum[w] = red[w][lane];
        // all eight waves when wave 0 carries points too, else waves 1..7 (iterate_kernel<512>'s order)
        const double row_sum = MANY ? tree_sum<kWaves>(w_sum) : tree_sum<kWaves - 1>(w_sum + 1);
        store_through_bits(rows + (size_t)my_row * kSlots + lane, publishable(row_sum));
      }
      bool ok = true;
      if (folder) {  // uniform
        double part = 0.0;
        ok = poll_and_sum<false>(rows + (size_t)blk * kFolders * kSlots, lane, a.spin_limit, part);
        if (ok && lane <= (uint32_t)kCountSlot) store_through_bits(parts + (size_t)blk * kSlots + lane, publishable(part));
      }
      if (STAMPS) { const uint64_t n = wall_clock64(); acc_l1 += n - t_mark; t_mark = n; }
      // ---- level 2: every workgroup adds the parts ----
      double tot = 0.0;
      if (!MULTI) {
        // A workgroup that is no folder has nothing to find before the folders have gathered their rows (a hop of
        // ~1.5 us): polling the parts from the start only puts 240 workgroups' loads into the queues the 16 folders'
        // polls and stores go through.  One point per thread (C2): 6.82 -> 6.55 us per round with 64 units (50: 6.63,
        // 60 - 70: 6.55, 85: 6.9, 100: 7.1); several points per thread (C5): no gain, not delayed.
        if (VGICP_L2_DELAY > 0 && !MANY && !folder) __builtin_amdgcn_s_sleep(VGICP_L2_DELAY);
        if (ok) ok = poll_and_sum<false>(parts, lane, a.spin_limit, tot);
      } else {
        // ---- several GPUs: workgroup 0 adds the parts to this rank's total and stores it into the mailbox
        // of every rank (its own included); every workgroup then adds the ranks' totals in rank order ----
        const uint32_t mbuf = (a.mail_round0 + (uint32_t)it) % 3u;
        if (blk == 0) {
          double mine = 0.0;
          if (ok) ok = poll_and_sum<false>(parts, lane, a.spin_limit, mine);
          if (ok && lane <= (uint32_t)kCountSlot) {
            const unsigned long long bits = publishable(mine);
            for (uint32_t r = 0; r < a.world; ++r)
              store_system_bits(a.mail[r] + ((size_t)mbuf * kMaxRanks + a.rank) * kSlots + lane, bits);
          }
        }
        if (ok) ok = poll_and_sum<true>(a.mail[a.rank] + (size_t)mbuf * kMaxRanks * kSlots, lane, a.spin_limit, tot);
        // (the mailbox buffer everyone finished with a round ago is re-armed behind the round's last barrier, with the
        // rows and parts)
      }
      if (STAMPS) { const uint64_t n = wall_clock64(); acc_l2 += n - t_mark; t_mark = n; }
      if (!ok) {
        if (lane == 0) stop_sh = 2;
      } else {
        // (what was consumed a round ago is re-armed, and the round's row of the log written, BEHIND the barrier below:
        // a release fence — the one in front of the LDS hand-over here, the one inside __syncthreads — makes the wave
        // wait for every store it has in flight, and a write-through store takes its time; this wave's part of the
        // round is the one everybody waits for)
        if (lane < kSlots) totals[lane] = lane <= (uint32_t)kCountSlot ? tot : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double A[21], g[6], xi[6];
#pragma unroll
        for (int k = 0; k < 21; ++k) A[k] = totals[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = -totals[21 + k];
        uint64_t f0 = 0, f1 = 0, f2 = 0;
        if (STAMPS) { f0 = pinned_clock(A); fine[0] += f0 - t_mark; }
        if (!ldlt6_solve_spd(A, g, xi)) ldlt6_solve_pivoted(totals, work, lane, xi);  // uniform branch
        if (STAMPS) { f1 = pinned_clock(xi); fine[1] += f1 - f0; }
        Pose next, step;
        se3_exp_device(xi, step);
        pose_compose(step, total, next);
        conv_keep = converged(step, cos_thr, tsq_thr);
        if (STAMPS) { f2 = pinned_clock(next.R); fine[2] += f2 - f1; }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) pose_sh[k] = next.R[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) pose_sh[9 + k] = next.t[k];
          stop_sh = (conv_keep || (it + 1 >= max_it)) ? 1 : 0;
        }
        if (STAMPS) fine[3] += wall_clock64() - f2;
      }
    }
    __syncthreads();
    const int stop = stop_sh;
    if (stop == 2) { gave_up = true; break; }  // uniform: a workgroup never published (not all resident?)
    if (wave == 0) {
      // Off the round's critical path (the other waves are already transforming their points with the new pose): re-arm
      // what was consumed a round ago.  In a launch's first round that is what the launch BEFORE published last:
      // everyone has published this round, so everyone has left that launch and its reads behind.  (Before the very
      // first round of a context the buffer is unset already; storing "unset" again is harmless.)  The stores are
      // complete long before this wave publishes again (s_waitcnt vmcnt(0) in front of the publication).
      uint32_t lane_here = tid & 63;
      asm volatile("" : "+v"(lane_here));
      const uint32_t rearm = (a.round0 + (uint32_t)it + 2u) % 3u;
      if (lane_here <= (uint32_t)kCountSlot) {
        store_through_bits(a.rows + ((size_t)rearm * kExchangeRows + my_row) * kSlots + lane_here, kRowUnset);
        if (folder) store_through_bits(a.parts + ((size_t)rearm * kFolders + blk) * kSlots + lane_here, kRowUnset);
      }
      if (MULTI && blk == 0 && lane_here <= (uint32_t)kCountSlot) {
        // ... and the mailbox buffer everyone (on every rank) finished with a round ago; the running round number
        // carries over from launch to launch, so there is no clean-up at the end of a launch
        const uint32_t old = (a.mail_round0 + (uint32_t)it + 2u) % 3u;
        for (uint32_t r = 0; r < a.world; ++r)
          store_system_bits(a.mail[a.rank] + ((size_t)old * kMaxRanks + r) * kSlots + lane_here, kRowUnset);
      }
      // the round's row of the log: still in LDS (written again only after the next round's barrier)
      if (blk == 0 && lane_here < kSlots) a.log[(size_t)it * kSlots + lane_here] = totals[lane_here];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) total.R[k] = pose_sh[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) total.t[k] = pose_sh[9 + k];
    ++it;
    if (STAMPS) { const uint64_t n = wall_clock64(); acc_solve += n - t_mark; t_mark = n; }
    if (stop) break;
  }
  if (gave_up) {
    // The host sees state->seq != seq (workgroup 0) or state->abort_seq == seq (any other workgroup: it may have
    // timed out although workgroup 0 found everything in place), resets the exchange buffers and uses launches.
    if (tid == 0) a.state->abort_seq = a.seq;
    if (MULTI && blk == 0 && tid < a.world)   // the peers learn it from the verdict word, at the latest at their end
      store_system_bits(a.mail[tid] + kMailRowWords + a.rank, (unsigned long long)a.mail_seq << 1);
    return;
  }

  if (wave == 0 && blk == 0) {
    uint32_t outcome = kOutcomeCommitted;
    if (MULTI) {
      // All ranks commit this align or none does: every rank's workgroup 0 tells every rank how its loop ended and
      // waits for everybody's word.  (A rank that gave up alone would otherwise re-run the align through the host
      // collective while a late peer, finding that rank's last row already in its mailbox, returns success.)
      const unsigned long long mine = ((unsigned long long)a.mail_seq << 1) | 1ull;
      if (lane < a.world) store_system_bits(a.mail[lane] + kMailRowWords + a.rank, mine);
      const double* src = a.mail[a.rank] + kMailRowWords + (lane < a.world ? lane : 0u);
      unsigned long long w = lane < a.world ? load_system_bits(src) : mine;
      bool late = false;
      for (uint32_t spins = 0; __any((w >> 1) < a.mail_seq); ++spins) {
        if (spins >= a.spin_limit) { late = true; break; }
        __builtin_amdgcn_s_sleep(1);
        if ((w >> 1) < a.mail_seq) w = load_system_bits(src);
      }
      if (late) outcome = kOutcomeNoAgreement;
      else if (__any(w != mine)) outcome = kOutcomeAgreedAbort;
    }
    if (lane == 0) {
      AlignState* out = a.state;
      out->outcome = outcome;
      if (outcome == kOutcomeCommitted) {
#pragma unroll
        for (int k = 0; k < 9; ++k) out->pose[k] = total.R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) out->pose[9 + k] = total.t[k];
        out->cosine_threshold = cos_thr;
        out->translation_sq_threshold = tsq_thr;
        out->max_iteration = max_it;
        out->iteration = it;
        out->converged = conv_keep ? 1 : 0;   // (the last increment itself, AlignState::step, is not reported by this launch)
        out->done = 1;
        out->pad = 0;
        out->seq = a.seq;
      }
    }
  }
  if (STAMPS && tid == 0) atomicAdd((unsigned long long*)&a.stamps[32 + blk], (unsigned long long)acc_body);
  if (STAMPS && blk == 0 && (tid == 0 || tid == 64)) {
    const int o = tid == 0 ? 8 : 16;  // solver wave / first worker lane
    atomicAdd((unsigned long long*)&a.stamps[o + 0], (unsigned long long)acc_body);
    atomicAdd((unsigned long long*)&a.stamps[o + 1], (unsigned long long)acc_l1);
    atomicAdd((unsigned long long*)&a.stamps[o + 2], (unsigned long long)acc_l2);
    atomicAdd((unsigned long long*)&a.stamps[o + 3], (unsigned long long)acc_solve);
    atomicAdd((unsigned long long*)&a.stamps[o + 5], (unsigned long long)it);
    if (tid == 0)
      for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)&a.stamps[24 + k], (unsigned long long)fine[k]);
  }
}

// The launch.  A MANY launch is planned from a.n, which is an UPPER BOUND for a scan whose size is still on the device
// (a scan prepared there and aligned without waiting for its count): when the real size turns out to fit the grid
// after all, the workgroups run the one-point-per-thread body — the mapping, the sums and hence the bits of the launch
// that a settled scan of that size gets (without the neighbour prefetch, which changes no result) — so the chain that
// does not wait returns the bits of the one that does, whatever the sweep's raw size.
template <int BLOCK, bool MULTI, bool STAMPS, bool MANY>
__global__ __launch_bounds__(BLOCK) void persistent_kernel(PersistArgs a) {
  if constexpr (MANY) {
    const uint32_t n_pts = a.n_dev ? (*a.n_dev < a.n ? *a.n_dev : a.n) : a.n;  // uniform
    if (n_pts <= gridDim.x * (uint32_t)(BLOCK - 64)) {
      persistent_body<BLOCK, MULTI, STAMPS, false>(a);
      return;
    }
  }
  persistent_body<BLOCK, MULTI, STAMPS, MANY>(a);
}

// Multi-GPU only: fold this rank's rows into one row (fixed order) so the all-reduce moves 256 B.
__global__ __launch_bounds__(1024) void fold_rows_kernel(const double* __restrict__ rows,
                                                         uint32_t nrows, const AlignState* state,
                                                         double* __restrict__ sums) {
  constexpr int kGroups = 1024 / kSlots;
  __shared__ double fin[kGroups][kSlots];
  if (state->done) return;  // rows are stale once the loop has ended; the value is never used
  const uint32_t tid = threadIdx.x, slot = tid & (kSlots - 1), group = tid / kSlots;
  double s = 0.0;
  for (uint32_t b = group; b < nrows; b += kGroups) s += rows[(size_t)b * kSlots + slot];
  fin[group][slot] = s;
  __syncthreads();
  if (tid < kSlots) {
    double tot = fin[0][tid];
#pragma unroll
    for (int g = 1; g < kGroups; ++g) tot += fin[g][tid];
    sums[tid] = tot;
  }
}

// out[o] = in[perm[o]] for the AoS arrays of a scan (points 3, covariances 9, optionally the input indices): the
// reordering of VGICP_OPTION_REFERENCE_ORDER (one thread per kept point; the planes are re-made by pack_scan_kernel).
__global__ void gather_scan_kernel(const uint32_t* __restrict__ perm, uint32_t m, const double* __restrict__ pts,
                                   const double* __restrict__ covs, const unsigned long long* __restrict__ idx,
                                   double* __restrict__ out_pts, double* __restrict__ out_covs,
                                   unsigned long long* __restrict__ out_idx) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= m) return;
  const uint32_t from = perm[o];
#pragma unroll
  for (int k = 0; k < 3; ++k) out_pts[3 * (size_t)o + k] = pts[3 * (size_t)from + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) out_covs[9 * (size_t)o + k] = covs[9 * (size_t)from + k];
  if (idx) out_idx[o] = idx[from];
}

// AoS (the caller's Eigen memory) -> 12 SoA planes.
__global__ void pack_scan_kernel(const double* __restrict__ pts, const double* __restrict__ covs,
                                 uint32_t n, double* __restrict__ soa, uint64_t stride, uint32_t* asym, uint32_t seq) {
  // *asym = seq as soon as one covariance is not bitwise symmetric (the word needs no clearing: seq differs per upload)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) soa[k * stride + i] = pts[3 * (size_t)i + k];
  double c[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c[k] = covs[9 * (size_t)i + k];
    soa[(3 + k) * stride + i] = c[k];
  }
  const bool same = __double_as_longlong(c[1]) == __double_as_longlong(c[3]) &&
                    __double_as_longlong(c[2]) == __double_as_longlong(c[6]) &&
                    __double_as_longlong(c[5]) == __double_as_longlong(c[7]);
  if (!same) *asym = seq;
}

// The same packing for a scan that is still ARRIVING in page-locked host memory: the upload of vgicp_align /
// vgicp_scan_upload without the runtime's copy engine.  Host threads (plain copies, no HIP call: CopyCrew,
// vgicp_context.h) fill `apts` (n x 24 B) and `acov` in units of `unit` points and publish each unit by storing `seq`
// into flags[16 * u] (one 64-byte line per flag, written after the unit's bytes).  This ONE launch reads the staging
// memory over PCIe itself — 16-byte loads of consecutive lanes, a block's bytes through LDS — while the threads are
// still copying the units behind (tools/micro/stage_crew_probe.hip: a kernel gets 54 GB/s out of the link; staging +
// copy commands + pack took 0.45 ms).
// Round 6: the link is what bounds the upload, so a unit whose covariances are ALL bitwise symmetric — every covariance
// the reference produces is (src/CloudPreprocessor.cpp:119-123 apart from its indefinite corner) — crosses
// it as six doubles per point (c00 c10 c20 c11 c21 c22: 48 instead of 72 bytes, 72 instead of 96 per point in all), and
// the three mirrored entries are made here.  The host thread that copies a unit checks every covariance while it reads
// it anyway and says which form it wrote in flags[16 * u + 1] (kArenaCompact / kArenaFull, stored before the flag); a
// unit with ONE asymmetric covariance travels whole.  A unit's covariances start at acov + 72 * (first point) in both
// forms.  What lands on the device is the caller's scan bit for bit either way.
// Every block also leaves the AoS copy on the device (map insertion and download read it) and reports an asymmetric
// covariance like pack_scan_kernel.  wait == 0: everything is there already (the forms are still read from the flags).
// A wait that exceeds spin_limit polls (each a PCIe round trip, >= 1 us) ends the block's workgroup: the host, which
// knows how long its copy threads took, then repeats the packing without waiting behind this launch.
constexpr int kPackArenaBlock = 256;
__global__ __launch_bounds__(kPackArenaBlock) void pack_arena_kernel(
    const char* __restrict__ apts, const char* __restrict__ acov, uint32_t n, uint32_t unit, const uint32_t* flags,
    uint32_t wait, uint32_t seq, uint32_t spin_limit, double* __restrict__ aos_pts, double* __restrict__ aos_cov,
    double* __restrict__ soa, uint64_t stride, uint32_t* asym) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  constexpr uint32_t B = kPackArenaBlock;
  __shared__ __attribute__((aligned(16))) double lds[12 * B];
  __shared__ __attribute__((aligned(16))) double lds_compact[6 * B];
  __shared__ uint32_t ok, form_sh;
  const uint32_t t = threadIdx.x, blocks = (n + B - 1) / B;
  uint32_t have_unit = 0xFFFFFFFFu;
  for (uint32_t b = blockIdx.x; b < blocks; b += gridDim.x) {
    const uint32_t p0 = b * B, cnt = min(B, n - p0);
    const uint32_t u = p0 / unit;   // unit is a multiple of the block: a block never straddles two units
    if (u != have_unit) {
      if (t == 0) {
        uint32_t good = 1;
        if (wait)
          for (uint32_t spins = 0; __hip_atomic_load(flags + 16 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq; ++spins) {
            if (spins >= spin_limit) { good = 0; break; }
            __builtin_amdgcn_s_sleep(20);
          }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = good;
        form_sh = good ? __hip_atomic_load(flags + 16 * u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : kArenaFull;
      }
      __syncthreads();
      if (!ok) return;   // uniform
      have_unit = u;
    }
    const bool compact = form_sh == kArenaCompact;   // uniform
    const v4i* sp = reinterpret_cast<const v4i*>(apts + (size_t)p0 * 24);
    v4i* lp = reinterpret_cast<v4i*>(lds);
    v4i* lc = reinterpret_cast<v4i*>(lds + 3 * B);
    // an odd count (the scan's last block only) leaves half a 16-byte chunk at the end of the points AND of the full
    // covariances: read whole (the staging areas are padded), stored to the device copy as 8 bytes
    const uint32_t np = (cnt * 24 + 15) / 16, np_whole = (cnt * 24) / 16, nc = (cnt * 72 + 15) / 16, nc_whole = (cnt * 72) / 16;
    v4i* dp = reinterpret_cast<v4i*>(aos_pts + 3 * (size_t)p0);
    v4i* dc = reinterpret_cast<v4i*>(aos_cov + 9 * (size_t)p0);
    for (uint32_t k = t; k < np; k += B) {
      const v4i w = __builtin_nontemporal_load(sp + k);
      lp[k] = w;
      if (k < np_whole) dp[k] = w;
    }
    if (!compact) {
      const v4i* sc = reinterpret_cast<const v4i*>(acov + (size_t)p0 * 72);
      for (uint32_t k = t; k < nc; k += B) {
        const v4i w = __builtin_nontemporal_load(sc + k);
        lc[k] = w;
        if (k < nc_whole) dc[k] = w;
      }
    } else {
      // the unit's compact records are contiguous from the unit's start: this block's begin (p0 - unit start) * 48 in
      const uint32_t u0 = u * unit;
      const v4i* sc = reinterpret_cast<const v4i*>(acov + (size_t)u0 * 72 + (size_t)(p0 - u0) * 48);
      v4i* lq = reinterpret_cast<v4i*>(lds_compact);
      for (uint32_t k = t; k < cnt * 3; k += B) lq[k] = __builtin_nontemporal_load(sc + k);   // 48 B = three chunks per point
    }
    __syncthreads();
    if (t == 0 && np != np_whole) {
      aos_pts[3 * (size_t)(p0 + cnt) - 1] = lds[3 * cnt - 1];
      if (!compact) aos_cov[9 * (size_t)(p0 + cnt) - 1] = lds[3 * B + 9 * cnt - 1];
    }
    if (t < cnt) {
      const size_t i = p0 + t;
#pragma unroll
      for (int k = 0; k < 3; ++k) soa[k * stride + i] = lds[3 * t + k];
      double c[9];
      if (compact) {
        const double a0 = lds_compact[6 * t], a1 = lds_compact[6 * t + 1], a2 = lds_compact[6 * t + 2],
                     a3 = lds_compact[6 * t + 3], a4 = lds_compact[6 * t + 4], a5 = lds_compact[6 * t + 5];
        c[0] = a0; c[1] = a1; c[2] = a2; c[3] = a1; c[4] = a3; c[5] = a4; c[6] = a2; c[7] = a4; c[8] = a5;
#pragma unroll
        for (int k = 0; k < 9; ++k) lds[3 * B + 9 * t + k] = c[k];   // the whole record for the device copy below
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = lds[3 * B + 9 * t + k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) soa[(3 + k) * stride + i] = c[k];
      const bool same = __double_as_longlong(c[1]) == __double_as_longlong(c[3]) &&
                        __double_as_longlong(c[2]) == __double_as_longlong(c[6]) &&
                        __double_as_longlong(c[5]) == __double_as_longlong(c[7]);
      if (!same) *asym = seq;
    }
    __syncthreads();
    if (compact) {   // the device's AoS copy of the covariances, whole records, coalesced out of LDS
      const uint32_t words = cnt * 9;   // doubles
      for (uint32_t k = t; k < words / 2; k += B) dc[k] = lc[k];
      if (t == 0 && (words & 1u)) aos_cov[9 * (size_t)p0 + words - 1] = lds[3 * B + words - 1];
      __syncthreads();
    }
  }
}

__global__ void table_clear_kernel(VoxelRecord* table, uint64_t slots) {
  // 8 threads per record, 16 B each
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= slots * 8) return;
  reinterpret_cast<int4*>(table)[g] = make_int4(0, 0, 0, 0);
}

// Batched upsert / rebuild in TWO launches, so that nothing inside a launch depends on what another workgroup of the
// same launch wrote except the one state word that is claimed by compare-and-swap:
//   claim   one thread per record: walk the probe sequence with ONE compare-and-swap per slot (EMPTY -> LOCKED).
//           Won: the slot is this record's (fresh).  Lost to LOCKED: another key of this batch (keys are unique within
//           a batch) — next slot.  Lost to FULL: a record from an earlier launch, its key is plain memory by now —
//           mine (update in place) or next slot.  The slot index (and the fresh bit) goes to a scratch word per record.
//   write   eight lanes per record, 16 bytes each (a record is one 128-byte line: {key, state} | mean + covariance =
//           six 16-byte pieces | {count, spare}): plain coalesced stores, FULL included — the kernel boundary publishes.
// One thread per record with a CAS, a release store and twelve scalar 8-byte stores into a random line measured 1.8 ms
// per million voxels (profiles/r08_c2_kernel_stats.csv); eight lanes per record in ONE launch needs the key visible
// before the state inside the launch: write-through key stores + wait cost 1.45 ms, a release fence per wave (it writes
// the XCD's L2 back) 5.5 ms.
constexpr uint32_t kClaimFresh = 0x80000000u;   // bit 31 of the scratch word: the record is new
constexpr uint32_t kClaimFailed = 0x7FFFFFFFu;  // probe sequence exhausted

__device__ __forceinline__ uint32_t claim_slot(VoxelRecord* table, uint32_t mask, int32_t kx, int32_t ky, int32_t kz) {
  uint32_t slot = voxel_hash(kx, ky, kz) & mask;
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    VoxelRecord* rec = table + slot;
    int32_t seen = SLOT_EMPTY;
    if (__hip_atomic_compare_exchange_strong(&rec->state, &seen, SLOT_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT))
      return slot | kClaimFresh;
    if (seen == SLOT_FULL && rec->key[0] == kx && rec->key[1] == ky && rec->key[2] == kz) return slot;
    slot = (slot + 1) & mask;
  }
  return kClaimFailed;
}

__global__ __launch_bounds__(256) void upsert_claim_kernel(VoxelRecord* table, uint32_t mask, uint32_t n,
                                                           const int32_t* __restrict__ keys, uint32_t* __restrict__ claimed,
                                                           uint32_t* counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;
  if (i < n) {
    c = claim_slot(table, mask, keys[3 * (size_t)i], keys[3 * (size_t)i + 1], keys[3 * (size_t)i + 2]);
    claimed[i] = c;
  }
  wave_count(&counters[0], i < n && c != kClaimFailed && (c & kClaimFresh) != 0);
  wave_count(&counters[1], i < n && c == kClaimFailed);
}

__global__ __launch_bounds__(256) void upsert_write_kernel(VoxelRecord* table, uint32_t n, const uint32_t* __restrict__ claimed,
                                                           const int32_t* __restrict__ keys, const double* __restrict__ means,
                                                           const double* __restrict__ covs) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = t >> 3, sub = t & 7u;
  if (i >= n) return;
  const uint32_t c = claimed[i];
  if (c == kClaimFailed) return;
  const bool fresh = (c & kClaimFresh) != 0;
  double2* line = reinterpret_cast<double2*>(table + (c & ~kClaimFresh));
  if (sub == 0) {
    if (fresh) reinterpret_cast<int4*>(line)[0] = make_int4(keys[3 * (size_t)i], keys[3 * (size_t)i + 1], keys[3 * (size_t)i + 2], SLOT_FULL);
  } else if (sub == 7) {
    // numPoints of a voxel mirrored from the host is not part of the batch: a new record starts at 1
    if (fresh) reinterpret_cast<ulonglong2*>(line)[7] = make_ulonglong2(1ull, 0ull);
  } else {
    // piece `sub` of the payload: doubles 2 (sub - 1), 2 (sub - 1) + 1 of {mean[3], cov[9]}
    const uint32_t p = 2u * (sub - 1u);
    double2 piece;
    piece.x = p < 3u ? means[3 * (size_t)i + p] : covs[9 * (size_t)i + (p - 3u)];
    piece.y = p + 1u < 3u ? means[3 * (size_t)i + p + 1u] : covs[9 * (size_t)i + (p + 1u - 3u)];
    line[sub] = piece;
  }
}

// The rebuild into a larger table: one thread per OLD slot claims (the new table holds nothing else, every claim wins a
// fresh slot), eight lanes per old slot then copy the 128-byte line.
__global__ __launch_bounds__(256) void rehash_claim_kernel(const VoxelRecord* __restrict__ old_table, uint64_t old_slots,
                                                           VoxelRecord* table, uint32_t mask, uint32_t* __restrict__ claimed,
                                                           uint32_t* counters) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;
  bool full = false;
  if (i < old_slots) {
    const int4 head = *reinterpret_cast<const int4*>(old_table + i);
    full = head.w == SLOT_FULL;
    c = full ? claim_slot(table, mask, head.x, head.y, head.z) : kClaimFailed;
    claimed[i] = c;
  }
  wave_count(&counters[0], full && c != kClaimFailed);
  wave_count(&counters[1], full && c == kClaimFailed);
}

__global__ __launch_bounds__(256) void rehash_write_kernel(const VoxelRecord* __restrict__ old_table, uint64_t old_slots,
                                                           VoxelRecord* table, const uint32_t* __restrict__ claimed) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i = t >> 3;
  const uint32_t sub = (uint32_t)(t & 7u);
  if (i >= old_slots) return;
  const uint32_t c = claimed[i];
  if (c == kClaimFailed) return;
  double2 piece = reinterpret_cast<const double2*>(old_table + i)[sub];
  if (sub == 7) piece.y = 0.0;   // the spare word (per-voxel list head of an insertion) does not travel
  reinterpret_cast<double2*>(table + (c & ~kClaimFresh))[sub] = piece;
}

__global__ void erase_kernel(VoxelRecord* table, uint32_t mask, uint32_t n,
                             const int32_t* __restrict__ keys, uint32_t* counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t kx = keys[3 * (size_t)i], ky = keys[3 * (size_t)i + 1], kz = keys[3 * (size_t)i + 2];
  uint32_t slot = voxel_hash(kx, ky, kz) & mask;
  bool erased = false;
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    VoxelRecord* rec = table + slot;
    const int32_t state = rec->state;
    if (state == SLOT_EMPTY) break;
    if (state == SLOT_FULL && rec->key[0] == kx && rec->key[1] == ky && rec->key[2] == kz) {
      // duplicates in the batch race here: exactly one CAS wins and counts
      int32_t expected = SLOT_FULL;
      erased = __hip_atomic_compare_exchange_strong(&rec->state, &expected, SLOT_TOMB, __ATOMIC_RELAXED,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    slot = (slot + 1) & mask;
  }
  wave_count(&counters[0], erased);
}

__global__ void voxel_index_kernel(const double* __restrict__ pts, uint32_t n, double voxel_size,
                                   int32_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // the division-free form the loop kernels use (exact, see voxel_coord_fast): this hook is how tests pin it
  // ... and the persistent launch's shortcut "still in last round's voxel?" (same_voxel_coord): it may say no for
  // the true key (the caller then makes the key), but a yes for any OTHER key would be a wrong correspondence — the
  // hook poisons the key it returns in that case, so the same tests pin both functions.
  const double inv_voxel = 1.0 / voxel_size;
  const double same_margin = 0x1p-20 * voxel_size;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double x = pts[3 * (size_t)i + k];
    int32_t key = voxel_coord_fast(x, voxel_size, inv_voxel);
    bool wrong_yes = false;
#pragma unroll
    for (int d = 1; d <= 2; ++d) {
      if (key <= INT32_MAX - d) wrong_yes = wrong_yes || same_voxel_coord(x, key + d, voxel_size, same_margin);
      if (key >= INT32_MIN + d) wrong_yes = wrong_yes || same_voxel_coord(x, key - d, voxel_size, same_margin);
    }
    keys[3 * (size_t)i + k] = wrong_yes ? INT32_MIN : key;
  }
}

// ---- correspondence materialisation: count per block, scan block counts, compact ----
constexpr int kMatchBlock = 256;

__device__ __forceinline__ const VoxelRecord* match_point(const double* pts, uint32_t i, uint32_t n,
                                                          const VoxelRecord* table, uint32_t mask,
                                                          double voxel_size) {
  if (i >= n) return nullptr;
  const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  return find_voxel(table, mask, voxel_coord(x, voxel_size), voxel_coord(y, voxel_size),
                    voxel_coord(z, voxel_size));
}

__global__ __launch_bounds__(kMatchBlock) void match_count_kernel(
    const double* __restrict__ pts, uint32_t n, const VoxelRecord* __restrict__ table, uint32_t mask,
    double voxel_size, uint32_t* __restrict__ block_counts) {
  const uint32_t i = blockIdx.x * kMatchBlock + threadIdx.x;
  const bool hit = match_point(pts, i, n, table, mask, voxel_size) != nullptr;
  const int total = __syncthreads_count(hit ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint32_t)total;
}

// Exclusive scan of block_counts in place by one workgroup; total -> *total_out.
__global__ __launch_bounds__(1024) void match_scan_kernel(uint32_t* counts, uint32_t nb,
                                                          uint32_t* total_out) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t idx = base + tid;
    const uint32_t val = idx < nb ? counts[idx] : 0u;
    uint32_t incl = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
    if (idx < nb) counts[idx] = before + incl - val;
    __syncthreads();
    if (tid == 1023) carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) *total_out = carry;
}

__global__ __launch_bounds__(kMatchBlock) void match_write_kernel(
    const double* __restrict__ pts, const double* __restrict__ covs, uint32_t n,
    const VoxelRecord* __restrict__ table, uint32_t mask, double voxel_size,
    const uint32_t* __restrict__ block_offsets, double* __restrict__ src_points,
    double* __restrict__ src_covs, double* __restrict__ map_points, double* __restrict__ map_covs,
    uint64_t* __restrict__ src_index) {
  __shared__ uint32_t wave_hits[kMatchBlock / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t i = blockIdx.x * kMatchBlock + tid;
  const VoxelRecord* rec = match_point(pts, i, n, table, mask, voxel_size);
  const unsigned long long ballot = __ballot(rec != nullptr);
  if (lane == 0) wave_hits[wave] = (uint32_t)__popcll(ballot);
  __syncthreads();
  if (rec == nullptr) return;
  uint32_t pos = block_offsets[blockIdx.x] + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
  for (uint32_t w = 0; w < wave; ++w) pos += wave_hits[w];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    src_points[3 * (size_t)pos + k] = pts[3 * (size_t)i + k];
    map_points[3 * (size_t)pos + k] = rec->mean[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    src_covs[9 * (size_t)pos + k] = covs[9 * (size_t)i + k];
    map_covs[9 * (size_t)pos + k] = rec->cov[k];
  }
  if (src_index) src_index[pos] = i;
}

// ---- dense copy of the FULL records (count per block -> scan -> copy) ----
constexpr int kDenseBlock = 256;
__global__ __launch_bounds__(kDenseBlock) void dense_count_kernel(const VoxelRecord* __restrict__ table, uint64_t slots,
                                                                  uint32_t* __restrict__ block_counts) {
  const uint64_t i = (uint64_t)blockIdx.x * kDenseBlock + threadIdx.x;
  const bool full = i < slots && table[i].state == SLOT_FULL;
  const int total = __syncthreads_count(full ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint32_t)total;
}
__global__ __launch_bounds__(kDenseBlock) void dense_write_kernel(VoxelRecord* table, uint64_t slots,
                                                                  const uint32_t* __restrict__ block_offsets,
                                                                  VoxelRecord* __restrict__ dense, uint64_t capacity) {
  __shared__ uint32_t wave_hits[kDenseBlock / 64];
  __shared__ uint32_t src[kDenseBlock];   // local slot numbers of the block's FULL records, in slot order
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * kDenseBlock + tid;
  const bool full = i < slots && table[i].state == SLOT_FULL;
  const unsigned long long ballot = __ballot(full);
  if (lane == 0) wave_hits[wave] = (uint32_t)__popcll(ballot);
  __syncthreads();
  uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull)), total = 0;
  for (uint32_t w = 0; w < kDenseBlock / 64; ++w) {
    if (w < wave) rank += wave_hits[w];
    total += wave_hits[w];
  }
  const uint32_t base = block_offsets[blockIdx.x];
  if (full) {
    src[rank] = tid;
    reinterpret_cast<uint32_t*>(&table[i].reserved)[1] = base + rank;   // where this record's copy lives
  }
  __syncthreads();
  // eight lanes per record, 16 bytes each: coalesced 128-byte lines both ways
  // never beyond the allocation (the host sizes it for every record a pending insertion may still add; this is the belt)
  for (uint32_t r = tid >> 3; r < total && (uint64_t)base + r < capacity; r += kDenseBlock / 8) {
    const double2 piece = reinterpret_cast<const double2*>(table + ((uint64_t)blockIdx.x * kDenseBlock + src[r]))[tid & 7u];
    reinterpret_cast<double2*>(dense + base + r)[tid & 7u] = piece;
  }
}

inline uint32_t blocks_for(uint64_t work, uint32_t block) { return (uint32_t)((work + block - 1) / block); }

}  // namespace

hipError_t launch_iterate(hipStream_t s, const IterArgs& args, uint32_t grid, int block) {
  switch (block) {
    case 256: ++g_kernel_launches; hipLaunchKernelGGL(iterate_kernel<256>, dim3(grid), dim3(256), 0, s, args); break;
    case 512: ++g_kernel_launches; hipLaunchKernelGGL(iterate_kernel<512>, dim3(grid), dim3(512), 0, s, args); break;
    case 1024: ++g_kernel_launches; hipLaunchKernelGGL(iterate_kernel<1024>, dim3(grid), dim3(1024), 0, s, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

namespace {
constexpr uint32_t kPersistWorkers = 512 - 64;
constexpr uint32_t kPersistWide = 512;  // point-carrying threads when a thread owns several points (wave 0 included)
constexpr uint32_t kMemoBytesPerPoint = kPersistWide * sizeof(int4);  // 8 192
// dynamic LDS of the persistent launch: the CU's 160 KB minus the kernel's static use and a margin
constexpr uint32_t kPersistDynLds = 150 * 1024;
constexpr uint32_t kMaxMemoPoints = 12;  // beyond that a thread's points are looked up every round
constexpr uint32_t kPrefetchBytes = kPersistWorkers * (sizeof(int4) + 6 * sizeof(double2));  // 50 176
}  // namespace

// The plan is made from n, which may be an upper bound of the scan's size (a scan prepared on the device): it is
// sized for 512 point-carrying threads; a launch that finds fewer points than grid x 448 uses 448 and less of it.
void persistent_lds_plan(uint32_t n, uint32_t grid, uint32_t* memo_points, uint32_t* stash_points, uint32_t* stash_bytes,
                         uint32_t lds_budget) {
  *memo_points = *stash_points = *stash_bytes = 0;
  // lds_budget: 0 = the whole CU (kPersistDynLds); contexts that SHARE a device take less, so that two of their
  // workgroups fit one CU (what is not kept in LDS is re-read: same arithmetic, same bits)
  const uint32_t budget = (lds_budget == 0 || lds_budget > kPersistDynLds) ? kPersistDynLds : lds_budget;
  if ((uint64_t)n <= (uint64_t)grid * kPersistWorkers) return;  // one point per thread: registers (and the neighbour prefetch area)
  // units of 64 points, dealt out evenly: a workgroup holds at most ceil(units / grid), taken 8 units per pass
  const uint32_t units = (n + 63u) / 64u, most = (units + grid - 1) / grid;
  const uint32_t per_thread = (most + kPersistWide / 64 - 1) / (kPersistWide / 64);  // points per thread (upper bound)
  // the memo first (it saves the table access, the larger term), the rest of the LDS parks whole points
  uint32_t memo = per_thread < kMaxMemoPoints ? per_thread : kMaxMemoPoints;
  if (memo * kMemoBytesPerPoint > budget) memo = budget / kMemoBytesPerPoint;
  *memo_points = memo;
  *stash_points = per_thread - 1;  // the first point of a thread lives in registers
  *stash_bytes = *stash_points ? budget - memo * kMemoBytesPerPoint : 0u;
}

uint32_t persistent_dyn_lds_bytes(uint32_t memo_points, uint32_t stash_bytes) {
  return memo_points * kMemoBytesPerPoint + stash_bytes;
}
uint32_t persistent_max_dyn_lds_bytes() { return kPersistDynLds; }

size_t persistent_rows_words() { return 3 * (size_t)kExchangeRows * kSlots; }
size_t persistent_parts_words() { return 3 * (size_t)kFolders * kSlots; }

// Initial / between-launch content of the exchange buffers for a launch of `grid` workgroups: the words a
// workgroup (rows) or a folder (parts) publishes are unset, every other word is +0.0 for good, so that
// every folder adds 16 rows and every workgroup adds 16 parts whatever the grid.
void persistent_exchange_image(uint32_t grid, unsigned long long* rows_words, unsigned long long* parts_words) {
  for (int buf = 0; buf < 3; ++buf) {
    unsigned long long* rw = rows_words + (size_t)buf * kExchangeRows * kSlots;
    for (size_t w = 0; w < (size_t)kExchangeRows * kSlots; ++w) rw[w] = 0ull;
    for (uint32_t b = 0; b < grid && b < (uint32_t)kExchangeRows; ++b) {
      const uint32_t row = (b % kFolders) * kFolders + b / kFolders;
      for (int sl = 0; sl <= kCountSlot; ++sl) rw[(size_t)row * kSlots + sl] = kRowUnset;
    }
    unsigned long long* pw = parts_words + (size_t)buf * kFolders * kSlots;
    for (size_t w = 0; w < (size_t)kFolders * kSlots; ++w) pw[w] = 0ull;
    for (uint32_t g = 0; g < grid && g < (uint32_t)kFolders; ++g)
      for (int sl = 0; sl <= kCountSlot; ++sl) pw[(size_t)g * kSlots + sl] = kRowUnset;
  }
}

namespace {
template <bool MULTI, bool STAMPS, bool MANY>
hipError_t launch_persistent_as(hipStream_t s, const PersistArgs& args, uint32_t grid, size_t dyn, int device) {
  (void)device;
  ++g_kernel_launches; hipLaunchKernelGGL((persistent_kernel<512, MULTI, STAMPS, MANY>), dim3(grid), dim3(512), dyn, s, args);
  return hipGetLastError();
}
template <bool MULTI, bool STAMPS, bool MANY>
hipError_t raise_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&persistent_kernel<512, MULTI, STAMPS, MANY>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistDynLds);
}
}  // namespace

// LDS beyond the default 64 KB per workgroup has to be asked for, per device and per instantiation.  Done when a
// context is created, NOT at the first launch: the launch path of a persistent kernel must not contain a runtime
// call that may wait for the device — with several sub-contexts on one device another sub-context's launch is
// already running (and waiting for this one) by then.
hipError_t persistent_prepare_device() {
  hipError_t e = raise_lds_limit<false, false, false>();
  if (e == hipSuccess) e = raise_lds_limit<false, false, true>();
  if (e == hipSuccess) e = raise_lds_limit<false, true, false>();
  if (e == hipSuccess) e = raise_lds_limit<false, true, true>();
  if (e == hipSuccess) e = raise_lds_limit<true, false, false>();
  if (e == hipSuccess) e = raise_lds_limit<true, false, true>();
  if (e == hipSuccess) e = raise_lds_limit<true, true, false>();
  if (e == hipSuccess) e = raise_lds_limit<true, true, true>();
  return e;
}

hipError_t launch_persistent(hipStream_t s, const PersistArgs& args, uint32_t grid) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  if (device < 0 || device >= 64) return hipErrorInvalidDevice;
  size_t dyn = (size_t)args.memo_points * kMemoBytesPerPoint + (size_t)args.stash_bytes;
  if (args.prefetch_margin > 0.0) {
    if (dyn != 0) return hipErrorInvalidValue;  // the prefetch area shares the LDS of memo / stash
    dyn = kPrefetchBytes;
  }
  if (dyn > kPersistDynLds) return hipErrorInvalidValue;
  const bool multi = args.world > 1, stamps = args.stamps != nullptr;
  const bool many = (uint64_t)args.n > (uint64_t)grid * kPersistWorkers;
  if (many) {
    if (multi) return stamps ? launch_persistent_as<true, true, true>(s, args, grid, dyn, device)
                             : launch_persistent_as<true, false, true>(s, args, grid, dyn, device);
    return stamps ? launch_persistent_as<false, true, true>(s, args, grid, dyn, device)
                  : launch_persistent_as<false, false, true>(s, args, grid, dyn, device);
  }
  if (multi) return stamps ? launch_persistent_as<true, true, false>(s, args, grid, dyn, device)
                           : launch_persistent_as<true, false, false>(s, args, grid, dyn, device);
  return stamps ? launch_persistent_as<false, true, false>(s, args, grid, dyn, device)
                : launch_persistent_as<false, false, false>(s, args, grid, dyn, device);
}

// Whether `grid` 512-thread workgroups of the persistent kernel with this much dynamic LDS can all be
// resident at once on the current device (the in-kernel exchange requires it).
hipError_t persistent_max_resident(uint32_t dyn_lds_bytes, int cu_count, uint32_t* max_grid) {
  int per_cu = 0;
  const void* fn = reinterpret_cast<const void*>(&persistent_kernel<512, true, false, true>);
  // more than the default 64 KB of LDS per workgroup has to be asked for before the occupancy query can say yes
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistDynLds);
  if (e != hipSuccess) return e;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, dyn_lds_bytes);
  if (e != hipSuccess) return e;
  *max_grid = per_cu > 0 ? (uint32_t)cu_count : 0u;  // one workgroup per CU is what the design uses
  return hipSuccess;
}

// The closing launch folds the last round's rows with the workgroup size of the body launches, so that
// its sums are added in the same order as every other round's (and as the persistent launch adds them).
hipError_t launch_close(hipStream_t s, const IterArgs& args, int block) {
  switch (block) {
    case 256: ++g_kernel_launches; hipLaunchKernelGGL(close_kernel<256>, dim3(1), dim3(256), 0, s, args); break;
    case 512: ++g_kernel_launches; hipLaunchKernelGGL(close_kernel<512>, dim3(1), dim3(512), 0, s, args); break;
    case 1024: ++g_kernel_launches; hipLaunchKernelGGL(close_kernel<1024>, dim3(1), dim3(1024), 0, s, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_solve_step(hipStream_t s, const double* packed27, double cosine_threshold,
                             double translation_sq_threshold, int force_pivoted, double* out20) {
  ++g_kernel_launches; hipLaunchKernelGGL(solve_step_kernel, dim3(1), dim3(64), 0, s, packed27, cosine_threshold,
                     translation_sq_threshold, force_pivoted, out20);
  return hipGetLastError();
}

hipError_t launch_fold_rows(hipStream_t s, const double* rows, uint32_t nrows, const AlignState* state,
                            double* sums) {
  ++g_kernel_launches; hipLaunchKernelGGL(fold_rows_kernel, dim3(1), dim3(1024), 0, s, rows, nrows, state, sums);
  return hipGetLastError();
}

hipError_t launch_gather_scan(hipStream_t s, const uint32_t* perm, uint32_t m, const double* pts, const double* covs,
                              const unsigned long long* idx, double* out_pts, double* out_covs, unsigned long long* out_idx) {
  if (m == 0) return hipSuccess;
  ++g_kernel_launches; hipLaunchKernelGGL(gather_scan_kernel, dim3(blocks_for(m, 256)), dim3(256), 0, s, perm, m, pts, covs, idx,
                                          out_pts, out_covs, out_idx);
  return hipGetLastError();
}

hipError_t launch_pack_scan(hipStream_t s, const double* points_aos, const double* covs_aos,
                            uint32_t n, double* soa, uint64_t stride, uint32_t* asym, uint32_t seq) {
  if (n == 0) return hipSuccess;
  ++g_kernel_launches; hipLaunchKernelGGL(pack_scan_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, points_aos,
                     covs_aos, n, soa, stride, asym, seq);
  return hipGetLastError();
}

uint32_t pack_arena_unit() { return 8u * kPackArenaBlock; }   // 2 048 points = 196 KB per unit
hipError_t launch_pack_arena(hipStream_t s, const void* arena_points, const void* arena_covs, uint32_t n,
                             const uint32_t* flags, bool wait, uint32_t seq, uint32_t spin_limit, double* aos_pts, double* aos_cov,
                             double* soa, uint64_t stride, uint32_t* asym) {
  if (n == 0) return hipSuccess;
  // few enough workgroups that the ones still waiting for their unit are a trickle of PCIe reads, enough of them that
  // 3 MB of loads are in flight; small LDS and 256 threads so that they find room beside other contexts' launches
  const uint32_t grid = std::min<uint32_t>(blocks_for(n, kPackArenaBlock), 128u);
  ++g_kernel_launches; hipLaunchKernelGGL(pack_arena_kernel, dim3(grid), dim3(kPackArenaBlock), 0, s,
                                          static_cast<const char*>(arena_points), static_cast<const char*>(arena_covs), n,
                                          pack_arena_unit(), flags, wait ? 1u : 0u, seq, spin_limit, aos_pts, aos_cov, soa, stride, asym);
  return hipGetLastError();
}

hipError_t launch_table_clear(hipStream_t s, VoxelRecord* table, uint64_t slots) {
  ++g_kernel_launches; hipLaunchKernelGGL(table_clear_kernel, dim3(blocks_for(slots * 8, 256)), dim3(256), 0, s, table,
                     slots);
  return hipGetLastError();
}

uint32_t table_dense_blocks(uint64_t slots) { return blocks_for(slots, kDenseBlock); }

hipError_t launch_table_dense(hipStream_t s, VoxelRecord* table, uint64_t slots, VoxelRecord* dense, uint64_t dense_capacity,
                              uint32_t* block_counts) {
  const uint32_t nb = table_dense_blocks(slots);
  ++g_kernel_launches; hipLaunchKernelGGL(dense_count_kernel, dim3(nb), dim3(kDenseBlock), 0, s, table, slots, block_counts);
  ++g_kernel_launches; hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nb, block_counts + nb);
  ++g_kernel_launches; hipLaunchKernelGGL(dense_write_kernel, dim3(nb), dim3(kDenseBlock), 0, s, table, slots, block_counts, dense, dense_capacity);
  return hipGetLastError();
}

hipError_t launch_upsert(hipStream_t s, VoxelRecord* table, uint32_t mask, uint32_t n,
                         const int32_t* keys, const double* means, const double* covs,
                         uint32_t* counters, uint32_t* claimed) {
  if (n == 0) return hipSuccess;
  ++g_kernel_launches; hipLaunchKernelGGL(upsert_claim_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, table, mask, n, keys, claimed, counters);
  ++g_kernel_launches; hipLaunchKernelGGL(upsert_write_kernel, dim3(blocks_for((uint64_t)n * 8, 256)), dim3(256), 0, s, table, n, claimed, keys,
                     means, covs);
  return hipGetLastError();
}

hipError_t launch_erase(hipStream_t s, VoxelRecord* table, uint32_t mask, uint32_t n,
                        const int32_t* keys, uint32_t* counters) {
  if (n == 0) return hipSuccess;
  ++g_kernel_launches; hipLaunchKernelGGL(erase_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, table, mask, n, keys,
                     counters);
  return hipGetLastError();
}

hipError_t launch_rehash(hipStream_t s, const VoxelRecord* old_table, uint64_t old_slots,
                         VoxelRecord* table, uint32_t mask, uint32_t* counters, uint32_t* claimed) {
  ++g_kernel_launches; hipLaunchKernelGGL(rehash_claim_kernel, dim3(blocks_for(old_slots, 256)), dim3(256), 0, s, old_table,
                     old_slots, table, mask, claimed, counters);
  ++g_kernel_launches; hipLaunchKernelGGL(rehash_write_kernel, dim3(blocks_for(old_slots * 8, 256)), dim3(256), 0, s, old_table,
                     old_slots, table, claimed);
  return hipGetLastError();
}

hipError_t launch_voxel_index(hipStream_t s, const double* points_aos, uint32_t n, double voxel_size,
                              int32_t* keys) {
  if (n == 0) return hipSuccess;
  ++g_kernel_launches; hipLaunchKernelGGL(voxel_index_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, points_aos, n,
                     voxel_size, keys);
  return hipGetLastError();
}

uint32_t match_blocks(uint32_t n) { return blocks_for(n, kMatchBlock); }

hipError_t launch_match(hipStream_t s, const double* points_aos, const double* covs_aos, uint32_t n,
                        const VoxelRecord* table, uint32_t mask, double voxel_size,
                        uint32_t* block_counts, uint32_t* total, double* src_points,
                        double* src_covs, double* map_points, double* map_covs, uint64_t* src_index) {
  const uint32_t nb = match_blocks(n);
  ++g_kernel_launches; hipLaunchKernelGGL(match_count_kernel, dim3(nb), dim3(kMatchBlock), 0, s, points_aos, n, table,
                     mask, voxel_size, block_counts);
  ++g_kernel_launches; hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nb, total);
  ++g_kernel_launches; hipLaunchKernelGGL(match_write_kernel, dim3(nb), dim3(kMatchBlock), 0, s, points_aos, covs_aos, n,
                     table, mask, voxel_size, block_counts, src_points, src_covs, map_points,
                     map_covs, src_index);
  return hipGetLastError();
}

}  // namespace vgicp
