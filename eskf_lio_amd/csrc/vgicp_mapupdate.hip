// vgicp_mapupdate.hip — LocalMap::updateLocalMap's insert and evict loops on the device
// (reference src/LocalMap.cpp:44-72, include/ESKF_LIO/LocalMap.hpp:63-89).
//
// The reference inserts serially: for every point in scan order, find its voxel; a missing voxel is
// constructed from the point (mean = p, covariance = C, numPoints = 1), an existing one takes
// Voxel::addPoint — mean <- (n*mean + p)/(n+1), covariance likewise, while numPoints < maxNumPoints.
// The result depends on the ORDER of the points inside a voxel, so the device version keeps it:
//   1. prepare   one thread per point: transform point and covariance exactly as Open3D does (no FMA
//                contraction, same evaluation order), voxel key, find-or-CLAIM the voxel's slot
//                (claims race through a CAS; same-key racers spin on the LOCKED word until the winner,
//                which finishes its critical section inside the same loop iteration, publishes FULL)
//   2. sort      stable radix sort of (slot, point index) by slot (hipCUB): the points of one voxel
//                become one segment, still in scan order
//   3. apply     the thread at the head of a segment walks it and applies the constructor / addPoint
//                arithmetic of the reference sequentially, in registers, one record write at the end
// Arithmetic is bit-exact with the CPU path (tests compare means, covariances and counts with `==`).
//
// A scan that the device has down-sampled itself (vgicp_scan_prepare: one point per voxel of its grid) puts only a
// handful of points into any voxel of the map, and for those the sort is most of the insertion's time (five launches
// for 27 000 pairs).  launch_map_insert(..., short_lists = true) keeps the order without it:
//   1. prepare   as above, and every point pushes itself onto its voxel's list (the record's spare word is the head:
//                atomic exchange, the old head becomes the point's `next`); the point that found the list empty is the
//                voxel's LEADER
//   2. apply     the leader walks the list, takes the points in ascending index = scan order (a register buffer of
//                kListChunk indices per walk; longer lists take several walks: correct for any length, quadratic
//                beyond the buffer, which is why arbitrary scans keep the sort), applies them, empties the list.
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include "vgicp_device.h"
#include "vgicp_device_fn.h"

namespace vgicp {
namespace {

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

// C <- R C R^T evaluated as Eigen evaluates (R * C) * R^T: left to right, every product and sum
// rounded once.
__device__ __forceinline__ void rotate_cov_exact(const double* R, const double* C, double* out) {
#pragma clang fp contract(off)
  double RC[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      RC[r + 3 * c] = (R[r] * C[3 * c] + R[r + 3] * C[1 + 3 * c]) + R[r + 6] * C[2 + 3 * c];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      out[r + 3 * c] = (RC[r] * R[c] + RC[r + 3] * R[c + 3]) + RC[r + 6] * R[c + 6];
}

struct InsertScratch {
  double* wpts;        // n x 3 world points
  double* wcovs;       // n x 9 world covariances
  uint32_t* slot_in;   // n
  uint32_t* slot_out;  // n (sorted)
  uint32_t* idx_in;    // n
  uint32_t* idx_out;   // n (sorted)
  void* cub;           // radix sort temp storage
  size_t cub_bytes;
};

__host__ inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// (slot, point index) pairs, stable. rocPRIM's merge sort with 2 048-item blocks: at scan sizes a launch costs as
// much as the work it does, and hipCUB's radix-sort front end picks the same sort with half the block and one more
// merge pass (see vgicp_preprocess.hip).
__host__ inline hipError_t sort_slots(void* temp, size_t& temp_bytes, const uint32_t* slot_in, uint32_t* slot_out,
                                      const uint32_t* idx_in, uint32_t* idx_out, uint32_t n, hipStream_t s) {
  using Config = rocprim::merge_sort_config<512, 512, 4>;
  return rocprim::merge_sort<Config>(temp, temp_bytes, slot_in, slot_out, idx_in, idx_out, (size_t)n,
                                     rocprim::less<uint32_t>(), s);
}
__host__ inline size_t sort_temp_bytes(uint32_t n) {
  size_t bytes = 0;
  (void)sort_slots(nullptr, bytes, nullptr, nullptr, nullptr, nullptr, n, nullptr);
  return bytes;
}

__host__ inline InsertScratch carve(void* base, uint32_t n, size_t cub_bytes) {
  char* p = static_cast<char*>(base);
  InsertScratch s;
  s.wpts = reinterpret_cast<double*>(p); p += align256((size_t)n * 3 * sizeof(double));
  s.wcovs = reinterpret_cast<double*>(p); p += align256((size_t)n * 9 * sizeof(double));
  s.slot_in = reinterpret_cast<uint32_t*>(p); p += align256((size_t)n * sizeof(uint32_t));
  s.slot_out = reinterpret_cast<uint32_t*>(p); p += align256((size_t)n * sizeof(uint32_t));
  s.idx_in = reinterpret_cast<uint32_t*>(p); p += align256((size_t)n * sizeof(uint32_t));
  s.idx_out = reinterpret_cast<uint32_t*>(p); p += align256((size_t)n * sizeof(uint32_t));
  s.cub = p;
  s.cub_bytes = cub_bytes;
  return s;
}

struct Pose12 {
  double v[12];
};

// One point of insert_prepare_kernel: transform, key, find-or-claim the slot. fresh: this thread created the
// voxel; returns the slot or kNoSlot (table full).
__device__ __forceinline__ uint32_t prepare_point(VoxelRecord* table, uint32_t mask, double voxel_size,
                                                  const double* __restrict__ pts, const double* __restrict__ covs,
                                                  uint32_t i, const Pose12& pose, double* __restrict__ wpts,
                                                  double* __restrict__ wcovs, bool& fresh) {
  double p[3], C[9], W[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = covs[9 * (size_t)i + k];
  transform_point(pose.v, pose.v + 9, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], p);
  rotate_cov_exact(pose.v, C, W);
#pragma unroll
  for (int k = 0; k < 3; ++k) wpts[3 * (size_t)i + k] = p[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) wcovs[9 * (size_t)i + k] = W[k];
  const int32_t kx = voxel_coord(p[0], voxel_size);
  const int32_t ky = voxel_coord(p[1], voxel_size);
  const int32_t kz = voxel_coord(p[2], voxel_size);

  uint32_t slot = voxel_hash(kx, ky, kz) & mask;
  uint32_t found = kNoSlot;
  // `spins` bounds the waits on LOCKED words so that a bug can only fail the call, never hang the GPU
  for (uint32_t probes = 0, spins = 0; probes <= mask && spins < (1u << 22); ++spins) {
    VoxelRecord* rec = table + slot;
    int32_t state = __hip_atomic_load(&rec->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (state == SLOT_EMPTY) {
      int32_t expected = SLOT_EMPTY;
      if (__hip_atomic_compare_exchange_strong(&rec->state, &expected, SLOT_LOCKED, __ATOMIC_RELAXED,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        // new voxel: key now, statistics in the apply pass (count 0 marks "not constructed yet")
        rec->key[0] = kx; rec->key[1] = ky; rec->key[2] = kz;
        rec->count = 0;
        rec->reserved = 0;
        __hip_atomic_store(&rec->state, SLOT_FULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        fresh = true;
        found = slot;
        break;
      }
      continue;  // lost the race: look at the same slot again
    }
    if (state == SLOT_LOCKED) continue;  // a claim in flight (perhaps of this very key): wait for it
    if (state == SLOT_FULL) {
      // keys are written before FULL is released; read them through the L2 (another CU may own them)
      const int32_t a = __hip_atomic_load(&rec->key[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t b = __hip_atomic_load(&rec->key[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int32_t c = __hip_atomic_load(&rec->key[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a == kx && b == ky && c == kz) { found = slot; break; }
    }
    slot = (slot + 1) & mask;  // other key or tombstone
    ++probes;
  }
  return found;
}

// LISTS: idx_of receives the point's `next` on its voxel's list (index + 1 of the point that headed the list before;
// 0: this point found the list empty and is the voxel's leader) instead of the point's own index for the sort.
template <bool LISTS>
__global__ void insert_prepare_kernel(VoxelRecord* table, uint32_t mask, double voxel_size,
                                      const double* __restrict__ pts, const double* __restrict__ covs,
                                      uint32_t n, Pose12 pose, double* __restrict__ wpts,
                                      double* __restrict__ wcovs, uint32_t* __restrict__ slot_of,
                                      uint32_t* __restrict__ idx_of, uint32_t* counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool fresh = false, lost = false;
  if (i < n) {
    const uint32_t found = prepare_point(table, mask, voxel_size, pts, covs, i, pose, wpts, wcovs, fresh);
    lost = found == kNoSlot;
    slot_of[i] = found;
    if constexpr (LISTS) {
      uint32_t next = 0xFFFFFFFFu;  // on no list
      if (!lost) next = atomicExch(reinterpret_cast<uint32_t*>(&table[found].reserved), i + 1u);
      idx_of[i] = next;
    } else {
      idx_of[i] = i;
    }
  }
  // one atomic per workgroup and counter (every wave of a scan that opens new ground creates some voxel: one
  // atomic per wave on the same word was half of this kernel's time)
  const int created = __syncthreads_count(fresh ? 1 : 0);
  const int failed = __syncthreads_count(lost ? 1 : 0);
  if (threadIdx.x == 0) {
    if (created) atomicAdd(&counters[0], (uint32_t)created);
    if (failed) atomicAdd(&counters[1], (uint32_t)failed);
  }
}

// Voxel(max, p, C) / Voxel::addPoint applied to the segment that starts at sorted position j.
__global__ void insert_apply_kernel(VoxelRecord* table, const uint32_t* __restrict__ slot_sorted,
                                    const uint32_t* __restrict__ idx_sorted, uint32_t n,
                                    const double* __restrict__ wpts, const double* __restrict__ wcovs,
                                    uint64_t max_points) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t slot = slot_sorted[j];
  if (slot == kNoSlot) return;
  if (j > 0 && slot_sorted[j - 1] == slot) return;  // not the head of its segment
  VoxelRecord* rec = table + slot;
  uint64_t count = rec->count;
  double mean[3], cov[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) mean[k] = rec->mean[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) cov[k] = rec->cov[k];
  for (uint32_t q = j; q < n && slot_sorted[q] == slot; ++q) {
    const uint32_t i = idx_sorted[q];
    if (count == 0) {  // constructor
#pragma unroll
      for (int k = 0; k < 3; ++k) mean[k] = wpts[3 * (size_t)i + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) cov[k] = wcovs[9 * (size_t)i + k];
      count = 1;
    } else if (count < max_points) {  // addPoint
#pragma clang fp contract(off)
      const double nn = (double)count, n1 = (double)(count + 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) mean[k] = (nn * mean[k] + wpts[3 * (size_t)i + k]) / n1;
#pragma unroll
      for (int k = 0; k < 9; ++k) cov[k] = (nn * cov[k] + wcovs[9 * (size_t)i + k]) / n1;
      ++count;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rec->mean[k] = mean[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) rec->cov[k] = cov[k];
  rec->count = count;
}

// The same for a voxel whose points hang on its list (short_lists): run by the voxel's leader.
constexpr int kListChunk = 8;
__global__ void insert_apply_list_kernel(VoxelRecord* table, const uint32_t* __restrict__ slot_of,
                                         const uint32_t* __restrict__ next_of, uint32_t n,
                                         const double* __restrict__ wpts, const double* __restrict__ wcovs,
                                         uint64_t max_points) {
  const uint32_t me = blockIdx.x * blockDim.x + threadIdx.x;
  if (me >= n || next_of[me] != 0u) return;  // on no list (table full), or not the leader
  VoxelRecord* rec = table + slot_of[me];
  const uint32_t head = (uint32_t)rec->reserved;  // index + 1 of the point that pushed itself last
  uint64_t count = rec->count;
  double mean[3], cov[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) mean[k] = rec->mean[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) cov[k] = rec->cov[k];
  auto apply = [&](uint32_t i) {
    if (count == 0) {  // constructor
#pragma unroll
      for (int k = 0; k < 3; ++k) mean[k] = wpts[3 * (size_t)i + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) cov[k] = wcovs[9 * (size_t)i + k];
      count = 1;
    } else if (count < max_points) {  // addPoint
#pragma clang fp contract(off)
      const double nn = (double)count, n1 = (double)(count + 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) mean[k] = (nn * mean[k] + wpts[3 * (size_t)i + k]) / n1;
#pragma unroll
      for (int k = 0; k < 9; ++k) cov[k] = (nn * cov[k] + wcovs[9 * (size_t)i + k]) / n1;
      ++count;
    }
  };
  if (head == me + 1u) {
    apply(me);  // the usual case: the voxel received this one point
  } else {
    // ascending index, kListChunk at a time: every walk keeps the smallest indices above the last one applied
    long long done = -1;  // the largest index applied so far
    for (;;) {
      uint32_t buf[kListChunk];
#pragma unroll
      for (int k = 0; k < kListChunk; ++k) buf[k] = 0xFFFFFFFFu;
      int held = 0;
      for (uint32_t cur = head; cur != 0u; cur = next_of[cur - 1u]) {
        uint32_t x = cur - 1u;
        if ((long long)x <= done) continue;
        // insert x into the ascending buffer, dropping the largest when it is full (static indices: registers)
#pragma unroll
        for (int k = 0; k < kListChunk; ++k) {
          const uint32_t lo = x < buf[k] ? x : buf[k], hi = x < buf[k] ? buf[k] : x;
          buf[k] = lo;
          x = hi;
        }
        if (held < kListChunk) ++held;
      }
#pragma unroll
      for (int k = 0; k < kListChunk; ++k)
        if (k < held) { apply(buf[k]); done = (long long)buf[k]; }
      if (held < kListChunk) break;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rec->mean[k] = mean[k];
#pragma unroll
  for (int k = 0; k < 9; ++k) rec->cov[k] = cov[k];
  rec->count = count;
  rec->reserved = 0;  // the list is empty again
}

// LocalMap::needsPointRemoval (src/LocalMap.cpp:149-154): |(index + 0.5) * voxelSize - position| > d
__global__ void evict_kernel(VoxelRecord* table, uint64_t slots, double voxel_size, double px,
                             double py, double pz, double distance, uint32_t* counters) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool far = false;
  if (i < slots && table[i].state == SLOT_FULL) {
#pragma clang fp contract(off)
    VoxelRecord* rec = table + i;
    const double dx = ((double)rec->key[0] + 0.5) * voxel_size - px;
    const double dy = ((double)rec->key[1] + 0.5) * voxel_size - py;
    const double dz = ((double)rec->key[2] + 0.5) * voxel_size - pz;
    far = sqrt((dx * dx + dy * dy) + dz * dz) > distance;
    if (far) rec->state = SLOT_TOMB;
  }
  // one atomic per workgroup: a mass eviction (788k of 1M voxels) was bound by 16 000 waves adding to one word
  const int gone = __syncthreads_count(far ? 1 : 0);
  if (threadIdx.x == 0 && gone) atomicAdd(&counters[0], (uint32_t)gone);
}

__global__ void export_kernel(const VoxelRecord* __restrict__ table, uint64_t slots, uint32_t capacity,
                              int32_t* __restrict__ keys, double* __restrict__ means,
                              double* __restrict__ covs, uint64_t* __restrict__ counts,
                              uint32_t* counters) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slots) return;
  const VoxelRecord* rec = table + i;
  if (rec->state != SLOT_FULL) return;
  const uint32_t pos = wave_append(&counters[0]);
  if (pos >= capacity) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    keys[3 * (size_t)pos + k] = rec->key[k];
    means[3 * (size_t)pos + k] = rec->mean[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) covs[9 * (size_t)pos + k] = rec->cov[k];
  counts[pos] = rec->count;
}

inline uint32_t blocks_for(uint64_t work, uint32_t block) { return (uint32_t)((work + block - 1) / block); }

}  // namespace

size_t map_insert_scratch_bytes(uint32_t n) {
  const uint32_t m = n ? n : 1;
  return align256((size_t)m * 3 * sizeof(double)) + align256((size_t)m * 9 * sizeof(double)) +
         4 * align256((size_t)m * sizeof(uint32_t)) + align256(sort_temp_bytes(m)) + 256;
}

hipError_t launch_map_insert(hipStream_t s, VoxelRecord* table, uint32_t mask, double voxel_size,
                             const double* points_aos, const double* covs_aos, uint32_t n,
                             const double pose12[12], uint64_t max_points, void* scratch,
                             size_t scratch_bytes, uint32_t* counters, bool short_lists) {
  if (n == 0) return hipSuccess;
  if (scratch_bytes < map_insert_scratch_bytes(n)) return hipErrorInvalidValue;
  size_t cub_bytes = sort_temp_bytes(n);
  InsertScratch w = carve(scratch, n, cub_bytes);
  Pose12 pose;
  for (int k = 0; k < 12; ++k) pose.v[k] = pose12[k];
  if (short_lists) {
    ++g_kernel_launches; hipLaunchKernelGGL(insert_prepare_kernel<true>, dim3(blocks_for(n, 256)), dim3(256), 0, s, table,
                       mask, voxel_size, points_aos, covs_aos, n, pose, w.wpts, w.wcovs, w.slot_in, w.idx_in, counters);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    ++g_kernel_launches; hipLaunchKernelGGL(insert_apply_list_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, table,
                       w.slot_in, w.idx_in, n, w.wpts, w.wcovs, max_points);
    return hipGetLastError();
  }
  ++g_kernel_launches; hipLaunchKernelGGL(insert_prepare_kernel<false>, dim3(blocks_for(n, 256)), dim3(256), 0, s, table, mask,
                     voxel_size, points_aos, covs_aos, n, pose, w.wpts, w.wcovs, w.slot_in, w.idx_in,
                     counters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // stable: equal slots keep ascending point index = scan order
  e = sort_slots(w.cub, cub_bytes, w.slot_in, w.slot_out, w.idx_in, w.idx_out, n, s);
  if (e != hipSuccess) return e;
  for (uint64_t run = 2048; ; run <<= 1) {  // rocPRIM's merge sort: one block sort + one merge launch per doubling
    ++g_kernel_launches;
    if (run >= n) break;
  }
  ++g_kernel_launches; hipLaunchKernelGGL(insert_apply_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, table, w.slot_out,
                     w.idx_out, n, w.wpts, w.wcovs, max_points);
  return hipGetLastError();
}

hipError_t launch_map_evict(hipStream_t s, VoxelRecord* table, uint64_t slots, double voxel_size,
                            const double position[3], double distance, uint32_t* counters) {
  ++g_kernel_launches; hipLaunchKernelGGL(evict_kernel, dim3(blocks_for(slots, 256)), dim3(256), 0, s, table, slots,
                     voxel_size, position[0], position[1], position[2], distance, counters);
  return hipGetLastError();
}

hipError_t launch_map_export(hipStream_t s, const VoxelRecord* table, uint64_t slots, uint32_t capacity,
                             int32_t* keys, double* means, double* covs, uint64_t* counts,
                             uint32_t* counters) {
  ++g_kernel_launches; hipLaunchKernelGGL(export_kernel, dim3(blocks_for(slots, 256)), dim3(256), 0, s, table, slots,
                     capacity, keys, means, covs, counts, counters);
  return hipGetLastError();
}

}  // namespace vgicp
