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

// LocalMap::getVoxelIndex: IEEE division, floor, double -> int32.
__device__ __forceinline__ int32_t voxel_coord(double x, double voxel_size) {
  return (int32_t)floor(x / voxel_size);
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
