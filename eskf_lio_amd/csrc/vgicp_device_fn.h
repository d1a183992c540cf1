// vgicp_device_fn.h — small device functions shared by the kernel translation units.
#pragma once

#include "vgicp_device.h"

namespace vgicp {
namespace {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t voxel_hash(int32_t x, int32_t y, int32_t z) {
  uint32_t h = fmix32((uint32_t)x * 0x9E3779B1u + 0x7F4A7C15u);
  h = fmix32(h ^ ((uint32_t)y * 0x85EBCA77u));
  h = fmix32(h ^ ((uint32_t)z * 0xC2B2AE3Du));
  return h;
}

// Counting with ONE atomic per wave on the same word, spelled out (the compiler's atomic optimizer merges the lanes
// of a wave as well; measured: no difference). What hurts is many WAVES on one word: device-scope atomics are
// carried out at the memory side (eight XCDs, eight L2s), one after the other per address -- ~12 ns each; 3 000
// of them made a 5 us kernel of the scan preparation take 38 us, so that one takes its counts from a scan instead.
__device__ __forceinline__ void wave_count(uint32_t* counter, bool flag) {
  const unsigned long long votes = __ballot(flag);
  if (votes && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(votes)) atomicAdd(counter, (uint32_t)__builtin_popcountll(votes));
}
// Every active lane gets its own position in a list whose length is *counter: one atomic per wave.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter) {
  const unsigned long long active = __ballot(true);
  const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(active);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(active));
  base = (uint32_t)__shfl((int)base, (int)leader, 64);
  return base + (uint32_t)__builtin_popcountll(active & ((1ull << lane) - 1ull));
}

// LocalMap::getVoxelIndex: IEEE division, floor, double -> int32.
__device__ __forceinline__ int32_t voxel_coord(double x, double voxel_size) {
  return (int32_t)floor(x / voxel_size);
}

// The same key without the division on the common path.  k = floor(x * fl(1/h)) can differ from the
// reference's floor(fl(x / h)) only when the exact quotient lies within a few ulps of an integer.  The
// remainder r = x - k h (one FMA, so its sign is exact) locates the quotient inside [k, k + 1): outside
// (r < 0 or r >= h) the guess was off by one; close to the upper end, (h - r) / h <= 2^-53 (|k| + 1), the
// correctly rounded division may round UP to k + 1 and the reference then takes k + 1.  All of these go to
// the real division (rare, and bit for bit the reference's value); everywhere else the guess IS
// floor(fl(x / h)).  NaN / infinite / huge x fail the tests and take the division, too.  *rem receives r.
__device__ __forceinline__ int32_t voxel_coord_fast(double x, double h, double inv_h, double* rem) {
  const double q = x * inv_h;
  const double kf = floor(q);
  const double r = fma(-kf, h, x);
  const double margin = (fabs(kf) + 4.0) * 0x1p-52 * h;
  if (fabs(q) < 0x1p30 && r >= 0.0 && (h - r) > margin) {
    *rem = r;
    return (int32_t)kf;
  }
  const double ks = floor(x / h);
  *rem = fma(-ks, h, x);
  return (int32_t)ks;
}
__device__ __forceinline__ int32_t voxel_coord_fast(double x, double h, double inv_h) {
  double r;
  return voxel_coord_fast(x, h, inv_h, &r);
}

// "Is floor(fl(x / h)) still k?" without computing the key: voxel_coord_fast's acceptance test applied to a GIVEN
// candidate (the key of the round before).  r = x - k h by one FMA has the exact sign; 0 <= r and h - r > margin put
// the exact quotient into [k, k + 1 - margin / h), and a margin of 2^-20 h exceeds voxel_coord_fast's
// (|k| + 4) 2^-52 h for every int32 k, so the correctly rounded division cannot round up to k + 1 either.  True
// therefore IMPLIES an unchanged key; false (a changed key, a point within a millionth of a voxel of its upper face,
// NaN) sends the caller to the full computation.  Five instructions per coordinate, no branch.
__device__ __forceinline__ bool same_voxel_coord(double x, int32_t k, double h, double margin) {
  const double r = fma(-(double)k, h, x);
  return ((int)(r >= 0.0) & (int)((h - r) > margin)) != 0;  // no short circuit: no branch
}

// q = R p + t evaluated as Open3D's homogeneous product does (left to right, no FMA contraction),
// so the first round reproduces the CPU path's voxel keys bit for bit.
__device__ __forceinline__ void transform_point(const double* R, const double* t, double x, double y,
                                                double z, double* q) {
#pragma clang fp contract(off)
  q[0] = ((R[0] * x + R[3] * y) + R[6] * z) + t[0];
  q[1] = ((R[1] * x + R[4] * y) + R[7] * z) + t[1];
  q[2] = ((R[2] * x + R[5] * y) + R[8] * z) + t[2];
}

// Probe for the voxel that contains the key. Returns the record or nullptr.
__device__ __forceinline__ const VoxelRecord* find_voxel(const VoxelRecord* table, uint32_t mask,
                                                         int32_t kx, int32_t ky, int32_t kz) {
  uint32_t slot = voxel_hash(kx, ky, kz) & mask;
  for (;;) {
    const VoxelRecord* rec = table + slot;
    const int4 ks = *reinterpret_cast<const int4*>(rec);
    if (ks.w == SLOT_EMPTY) return nullptr;
    if (ks.w == SLOT_FULL && ks.x == kx && ks.y == ky && ks.z == kz) return rec;
    slot = (slot + 1) & mask;
  }
}

}  // namespace
}  // namespace vgicp
