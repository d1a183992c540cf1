// vgicp_preprocess.hip — CloudPreprocessor::voxelDownsampleAndEstimateCovariances on the device
// (reference src/CloudPreprocessor.cpp:76-127).
//
// The reference builds an Open3D KD-tree on the whole scan, keeps the first point of every voxel, and
// for each kept point takes its 30 nearest neighbours in the whole scan (the point itself included),
// Open3D's cumulant covariance over them and the regularisation U diag(1, 1, 1e-2) V^T of a JacobiSVD.
// Here:
//   1. every point gets the 60-bit Morton code of its cell on a grid of voxel_size / 4 (so that
//      code >> 6 is the Morton code of the voxel, the same floor(p / h) as the map keys); ONE stable
//      sort (rocPRIM merge sort) of (code, index) orders the scan so that the cells of every octree
//      level — h/4, h/2, h, 2h, ... — are contiguous runs
//   2. the lowest index of every voxel-level run is the first point of its voxel: that is the
//      down-sampling. One segmented scan over the sorted order yields, at the last point of every run,
//      the voxel's rank, its kept point and that point's sorted position (the query list in Morton
//      order), and in its total the number of kept points and of octree cells; a prefix sum over scan
//      order gives every kept point its output slot (ascending index)
//   3. a hash table of (level, cell) -> [start, end) over the sorted order is built for all levels:
//      an octree whose nodes are contiguous runs of the sorted points (by the workgroups behind the scan's tiles
//      in the same launch); query_split_body (keep_and_split_kernel) then puts the queries of sparse neighbourhoods -- the expensive
//      ones -- at the head of the search's launch
//   4. ONE WAVE per kept point (control flow is uniform, lanes share the work): the k-list starts
//      full with sorted-order neighbours of the query; the finest own cell with >= k points ("home")
//      is measured first, and the k-th distance after it picks the level whose 27-cell block covers
//      the ball; those cells seed a pool; the wave repeatedly takes the nearest cell of the pool
//      (best-first), and either lets its lanes look up the 64 cells two levels below (one each) or,
//      for a short run, lets them measure its points (one each); passing points are inserted into the
//      sorted k-list the lanes hold in registers. It stops when the nearest cell left is farther than
//      the k-th distance: the result is the EXACT k nearest neighbours, ties broken by index, for
//      work proportional to what lies inside that ball (DESIGN.md §9 has the measurements behind
//      every choice in this kernel)
//   5. one THREAD per kept point: cumulants in ascending-distance order, Open3D's cumulant covariance, then
//      svd.matrixU() diag(1,1,1e-2) svd.matrixV()^T with Eigen 3.4's two-sided JacobiSVD restated in its
//      published operation order (U != V where an eigenvalue of the — in floating point possibly indefinite —
//      covariance is negative: the reference then emits an indefinite matrix, and so does this kernel).
// Output order is ascending original index (the reference's is unordered_map iteration order).
// This file is compiled without FMA contraction and keeps the oracle's operation order: the
// distances, sums and rotations round as they do on the CPU.
#include <cstring>
#include <string.h>

#ifdef VGICP_SORT_ROCPRIM
#include <rocprim/rocprim.hpp>
#endif

#include "vgicp_device.h"
#include "vgicp_device_fn.h"
#include "vgicp_sort.h"

namespace vgicp {
namespace {

constexpr int kMaxKnn = 32;
constexpr int kFineShift = 2;          // the finest cells are voxel_size / 4
constexpr int kLevels = 12;            // cells of h/4, h/2, h, ... 512 h
constexpr int kCoordBits = 20;         // per axis: 60-bit Morton codes, the top 4 bits of a key hold the level
constexpr int kCoordOffset = 1 << (kCoordBits - 1);  // cell indices are offset to be non-negative
constexpr int kCoordMax = (1 << kCoordBits) - 1;
constexpr int kSearchBlock = 64;       // one wave = one query per workgroup
constexpr int kPool = 256;             // cells waiting per query
// a waiting cell with at most this many points is measured, not opened.  Opening costs a turn per child that has to be
// taken afterwards (a trip to memory each, about 1 us for a wave on its own), measuring 64 points costs about as much:
// where a dense surface is seen from a sparse place (a query metres from a wall) the small threshold made the search
// take a hundred tiny cells one by one, and the kernel ended on a handful of such queries running alone
// (frame sweep, 29 151 queries: 128 -> 203 us, longest query 168 us; 256 -> 160 / 126; 512 -> 148 / 99; 1024 -> 153 / 98;
// 64 -> +13 % on 128.  profiles/r10_knn_leaf.txt).  Round 6, with the candidates kept unordered (a point that passes costs
// an append, not an insertion): 256 -> 115 us, 512 -> 95.5, **1024 -> 89.8**, 1536 -> 90.4, 2048 -> 91.5, 4096 -> 95.0.
#ifndef VGICP_LEAF_POINTS
#define VGICP_LEAF_POINTS 1024
#endif
constexpr uint32_t kLeafPoints = VGICP_LEAF_POINTS;
// The neighbour search keeps its candidates unordered and makes the order once (round 6, knn_search_kernel);
// -DVGICP_KNN_EAGER builds rounds 2-5's sorted k-list with one insertion per candidate instead (A/B: 104.9 -> 101.3 us
// per 60 000-point sweep, the same bits; tools/soak_preprocess.py: 817 random scans against the oracle, no mismatch).
#ifndef VGICP_KNN_EAGER
#define VGICP_KNN_LAZY 1
#endif
#ifndef VGICP_KNN_TIGHTEN_AT
#define VGICP_KNN_TIGHTEN_AT 40   // entries in the candidate buffer (of 64) at which the bound is tightened
#endif
#ifndef VGICP_KNN_SELECT_LOW
#define VGICP_KNN_SELECT_LOW 15   // the select resolves bits 30 .. this one of the single-precision key (16 steps)
#endif
#ifndef VGICP_HOME_POINTS
#define VGICP_HOME_POINTS 128
#endif
constexpr uint32_t kHomePoints = VGICP_HOME_POINTS;  // the query's own cell is measured whole when it holds at most this many
constexpr int kCovBlock = 128;
constexpr unsigned long long kEmptyCell = ~0ull;
constexpr unsigned long long kKeyMask = (1ull << 60) - 1;

struct CellEntry {
  unsigned long long key;  // (level << 60) | (morton >> 3 level)
  uint32_t start, end;
};

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ uint32_t compact21(unsigned long long v) {
  v &= 0x1249249249249249ull;
  v = (v ^ (v >> 2)) & 0x10C30C30C30C30C3ull;
  v = (v ^ (v >> 4)) & 0x100F00F00F00F00Full;
  v = (v ^ (v >> 8)) & 0x1F0000FF0000FFull;
  v = (v ^ (v >> 16)) & 0x1F00000000FFFFull;
  v = (v ^ (v >> 32)) & 0x1FFFFFull;
  return (uint32_t)v;
}
__device__ __forceinline__ unsigned long long morton3(uint32_t x, uint32_t y, uint32_t z) {
  return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}
// cell index on the finest grid (cell = fine), offset to be non-negative
__device__ __forceinline__ uint32_t cell_coord(double v, double fine) {
  long long c = (long long)floor(v / fine) + kCoordOffset;
  c = c < 0 ? 0 : (c > kCoordMax ? kCoordMax : c);
  return (uint32_t)c;
}
__device__ __forceinline__ unsigned long long cell_key(unsigned long long morton_at_level, int level) {
  return ((unsigned long long)level << 60) | morton_at_level;
}
// 32-bit multiplies only (three; a 64-bit mixer costs eight, and integer multiplies run at quarter rate)
__device__ __forceinline__ uint32_t cell_hash(unsigned long long key) {
  uint32_t h = (uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  return h;
}

// A finite coordinate whose finest cell lies outside the 2^20 cells per axis the Morton codes span (+-2^17
// voxel sizes): the clamped code would merge distinct far voxels and break the bounds of the neighbour search,
// so such a scan is refused (counters[3] flags it; vgicp_preprocess returns VGICP_ERR_BAD_ARGUMENT).
__device__ __forceinline__ bool beyond_grid(double v, double fine) {
  const double c = floor(v / fine);
  return c < -(double)kCoordOffset || c > (double)(kCoordMax - kCoordOffset);  // false for NaN / infinity
}

// write-through stores / loads for words that workgroups of ONE launch hand to each other (MI355X_MICROARCH.md)
typedef __attribute__((address_space(1))) unsigned long long pg64;
typedef __attribute__((address_space(1))) unsigned int pg32;
__device__ __forceinline__ void st_through64(void* p, unsigned long long v) {
  __hip_atomic_store((pg64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_through32(void* p, uint32_t v) {
  __hip_atomic_store((pg32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_through64(const void* p) {
  return __hip_atomic_load((pg64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_through32(const void* p) {
  return __hip_atomic_load((pg32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr uint32_t kLookBackSpinLimit = 1u << 22;

// First kernel of a preparation, one thread per raw point (CloudPreprocessor::process, src/CloudPreprocessor.cpp:8-23,
// up to the point where the neighbour search starts):
//   extrinsic   Open3D PointCloud::Transform on points: p <- (T [p;1]).xyz / w              (:14)
//   deskew      p <- T_s p for the IMU state s whose segment holds the point (:25-74); the segment ends are the
//               minimum over the blocks of deskew_first_hit_kernel, taken here by every workgroup for itself
//               (a few thousand words) instead of by a launch of its own
//   Morton code of the point's finest cell, identity index, kept flag cleared
// and, on the side, what later kernels of the sequence need cleared: the octree's cell table (grid-stride over the
// entries), the tile tickets of the scans and the indefinite-covariance count (workgroup 0).
struct Mat16 { double m[16]; };
struct PrologueArgs {
  double* pts;
  uint32_t n;
  int has_T;
  Mat16 T;
  uint32_t states;          // 0: no deskew
  uint32_t parts;           // blocks of deskew_first_hit_kernel; 0: `ends` already hold the segment ends
  const uint32_t* part;     // [parts][states] first hits
  uint32_t* ends;           // [states] segment ends (written by workgroup 0 when parts != 0)
  const double* poses;      // 12 doubles per state
  double fine;
  unsigned long long* codes;
  uint32_t* idx;
  uint32_t* keep_by_index;
  uint32_t* counters;
  uint32_t epoch;
  unsigned long long* table;  // CellEntry[entries] as 16-byte pairs
  unsigned long long entries;
  const char* src;            // nullptr, or the raw points in page-locked HOST memory (PrepareArgs::src_points)
  const uint32_t* src_flags;
  uint32_t src_seq, src_unit, src_spin;
  uint32_t src_step, src_off[3];   // != 0: the sensor's own records, float32 x y z at these byte offsets (PrepareArgs::src_step)
  // fused != 0: the segment of every point is found HERE, without a launch for the bounds (see the kernel)
  uint32_t fused, max_hits;
  const double* point_time;
  const double* state_time;
  unsigned long long* slots;       // [kFusedBoundsBlocksMax] (epoch << 32 | largest count of a workgroup)
};
// The deskew's segments without a kernel of their own (ordered state times).  Let a(j) = the number of states whose
// timestamp is not above point j's capture time ("hits": !(t_j < ts_s); they nest because the timestamps ascend).  The
// reference's walk gives point i to the first state s with i < ends[s], ends[s] = the first j with a(j) > s — that is
// s = M(i) = max of a(j) over j <= i, a running maximum; and the states from A = max over ALL points on have no hit
// at all, keep the last bound found and move nothing: a point with M(i) = A stays as it is.  A comes from the host
// (deskew_table walks the capture times anyway: the count for the latest time, every state if a time is NaN).  So a
// workgroup needs its own points' counts (binary search in LDS), their running maximum (wave scan) and the largest
// count of every workgroup BEFORE it: each publishes that one number at once, tagged with the call's epoch, and reads
// the earlier ones (look-back over slots, no chain: nobody waits for more than one publication).  Workgroup numbers
// are drawn from a ticket, so a workgroup only ever waits for workgroups that started before it.
__global__ __launch_bounds__(256) void sweep_prologue_kernel(PrologueArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char prologue_lds[];
  uint32_t* ends_sh = reinterpret_cast<uint32_t*>(prologue_lds);
  __shared__ uint32_t last_found;
  __shared__ __attribute__((aligned(16))) double staged[256 * 64 / 8];   // 256 points of 24 bytes (doubles) or of up to 64 (sensor records)
  __shared__ uint32_t staged_ok;
  __shared__ uint32_t bid_sh, wave_max_sh[4], before_sh, first_stay_sh, gave_up_sh;
  const uint32_t tid = threadIdx.x;
  uint32_t bid = blockIdx.x;
  uint32_t run_max = 0, block_max = 0;   // fused: running maximum of the counts up to this thread's point (inside the workgroup), the workgroup's largest
  if (a.fused) {
    if (tid == 0) {
      const uint32_t ticket = atomicAdd(&a.counters[kTicketP], 1u);
      if (ticket == gridDim.x - 1) st_through32(&a.counters[kTicketP], 0u);   // all handed out: zero for the next launch
      bid_sh = ticket;
      before_sh = 0u;
      first_stay_sh = 0xFFFFFFFFu;
      gave_up_sh = 0u;
    }
    double* ts = reinterpret_cast<double*>(prologue_lds);
    for (uint32_t s = tid; s < a.states; s += blockDim.x) ts[s] = a.state_time[s];
    __syncthreads();
    bid = bid_sh;
  }
  if (a.src) {
    // this block's 256 points out of the staging memory: wait for their unit (one lane polls over PCIe), then 16-byte
    // loads of consecutive lanes (every 64-byte request of the link is used whole), handed out through LDS
    typedef int v4i __attribute__((ext_vector_type(4)));
    const uint32_t p0 = bid * 256u;
    if (a.src_flags && p0 < a.n) {
      if (tid == 0) {
        const uint32_t* flag = a.src_flags + 16 * (size_t)(p0 / a.src_unit);
        uint32_t good = 1;
        for (uint32_t spins = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.src_seq; ++spins) {
          if (spins >= a.src_spin) { good = 0; break; }
          __builtin_amdgcn_s_sleep(20);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        staged_ok = good;
      }
      __syncthreads();
      if (!staged_ok) {   // uniform: the host's copy threads never delivered; reported like a scan that gave up
        if (tid == 0) {
          a.counters[kScanTimeout] = a.epoch;
          // (the workgroups behind this one look back at its slot: tell them not to wait)
          if (a.fused) st_through64(&a.slots[bid], ((unsigned long long)a.epoch << 32) | 0xFFFFFFFFull);
        }
        return;
      }
    }
    if (p0 < a.n) {
      const uint32_t cnt = a.n - p0 < 256u ? a.n - p0 : 256u;
      const uint32_t step = a.src_step ? a.src_step : 24u;
      const v4i* sp = reinterpret_cast<const v4i*>(a.src + (size_t)p0 * step);   // p0 * step is a multiple of 16 (p0 of 256, step of 4)
      v4i* lp = reinterpret_cast<v4i*>(staged);
      const uint32_t chunks = (cnt * step + 15u) / 16u;   // a last chunk may reach past the points (the staging memory is padded)
      for (uint32_t c = tid; c < chunks; c += 256u) lp[c] = __builtin_nontemporal_load(sp + c);
    }
    __syncthreads();
  }
  if (a.fused) {
    // (behind the wait for this workgroup's unit: the capture times may be staged with the points)
    double* ts = reinterpret_cast<double*>(prologue_lds);
    const uint32_t i0 = bid * 256u + tid;
    uint32_t cnt = 0;
    if (i0 < a.n) {
      const double t = a.point_time[i0];
      uint32_t hi = a.states;   // NaN is a hit for every state, like in the walk
      while (cnt < hi) {
        const uint32_t mid = (cnt + hi) >> 1;
        if (!(t < ts[mid])) cnt = mid + 1; else hi = mid;
      }
    }
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    run_max = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)run_max, o, 64);
      if (lane >= (uint32_t)o) run_max = v > run_max ? v : run_max;
    }
    if (lane == 63u) wave_max_sh[wave] = run_max;
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      const uint32_t v = wave_max_sh[k];
      block_max = v > block_max ? v : block_max;
      if (k < wave) carry = v > carry ? v : carry;
    }
    run_max = carry > run_max ? carry : run_max;
    if (tid == 0) st_through64(&a.slots[bid], ((unsigned long long)a.epoch << 32) | block_max);
    uint32_t before = 0;
    bool gave_up = false;
    for (uint32_t b = tid; b < bid; b += blockDim.x) {
      unsigned long long v = ld_through64(&a.slots[b]);
      for (uint32_t spins = 0; (uint32_t)(v >> 32) != a.epoch; ++spins) {
        if (spins >= kLookBackSpinLimit) { gave_up = true; break; }
        __builtin_amdgcn_s_sleep(2);
        v = ld_through64(&a.slots[b]);
      }
      const uint32_t c = (uint32_t)v;
      if (c == 0xFFFFFFFFu) { gave_up = true; break; }   // that workgroup gave up waiting for its unit
      before = c > before ? c : before;
    }
    if (before) atomicMax(&before_sh, before);
    if (gave_up) gave_up_sh = 1u;
    __syncthreads();
    if (gave_up_sh) {   // uniform; reported like a scan that gave up (the preparation is refused)
      if (tid == 0) a.counters[kScanTimeout] = a.epoch;
      return;
    }
    run_max = before_sh > run_max ? before_sh : run_max;
    // what the deskew reports: the points that a state moved = the first point whose running maximum is A
    if (before_sh < a.max_hits && block_max >= a.max_hits) {   // uniform: the one workgroup in which A is reached (A > 0)
      const uint32_t i0 = bid * 256u + tid;
      if (i0 < a.n && run_max >= a.max_hits) atomicMin(&first_stay_sh, i0);
      __syncthreads();
      if (tid == 0) a.counters[kDeskewedCounter] = first_stay_sh;
    }
    if (a.max_hits == 0u && bid == 0u && tid == 0) a.counters[kDeskewedCounter] = 0u;
  } else if (a.states) {
    if (a.parts) {
      if (tid == 0) last_found = 0u;
      for (uint32_t s = tid; s < a.states; s += blockDim.x) ends_sh[s] = 0xFFFFFFFFu;
      __syncthreads();
      for (uint32_t k = tid; k < a.parts * a.states; k += blockDim.x) {
        const uint32_t v = a.part[k];
        if (v != 0xFFFFFFFFu) atomicMin(&ends_sh[k % a.states], v);
      }
      __syncthreads();
      // first hits are non-decreasing in s and the states without one form a tail: they keep the last bound found
      for (uint32_t s = tid; s < a.states; s += blockDim.x)
        if (ends_sh[s] != 0xFFFFFFFFu) atomicMax(&last_found, ends_sh[s]);
      __syncthreads();
      const uint32_t keep = last_found;
      for (uint32_t s = tid; s < a.states; s += blockDim.x)
        if (ends_sh[s] == 0xFFFFFFFFu) ends_sh[s] = keep;
      __syncthreads();
      if (blockIdx.x == 0)
        for (uint32_t s = tid; s < a.states; s += blockDim.x) a.ends[s] = ends_sh[s];
    } else {
      for (uint32_t s = tid; s < a.states; s += blockDim.x) ends_sh[s] = a.ends[s];
      __syncthreads();
    }
  }
  if (bid == 0 && tid == 0) {
    a.counters[kTicketA] = 0u;
    a.counters[kTicketB] = 0u;
    a.counters[kHeavyQueries] = 0u;
    a.counters[kLightQueries] = 0u;
    a.counters[kIndefiniteCounter] = 0u;
    if (!a.fused) a.counters[kDeskewedCounter] = a.states ? ends_sh[a.states - 1] : 0u;
  }
  // the cell table: key = empty, bounds 0 (16-byte entries, two 8-byte words each)
  {
    const unsigned long long total = (unsigned long long)gridDim.x * blockDim.x;
    ulonglong2* t2 = reinterpret_cast<ulonglong2*>(a.table);
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + tid; e < a.entries; e += total)
      t2[e] = make_ulonglong2(kEmptyCell, 0ull);
  }
  const uint32_t i = bid * blockDim.x + tid;
  if (i >= a.n) return;
  double x, y, z;
  bool moved = a.src != nullptr;   // points that came from the host are written to the device whether or not they move
  if (a.src && a.src_step) {
    const char* rec = reinterpret_cast<const char*>(staged) + tid * a.src_step;
    x = (double)*reinterpret_cast<const float*>(rec + a.src_off[0]);
    y = (double)*reinterpret_cast<const float*>(rec + a.src_off[1]);
    z = (double)*reinterpret_cast<const float*>(rec + a.src_off[2]);
  } else if (a.src) {
    x = staged[3 * tid]; y = staged[3 * tid + 1]; z = staged[3 * tid + 2];
  } else {
    x = a.pts[3 * (size_t)i]; y = a.pts[3 * (size_t)i + 1]; z = a.pts[3 * (size_t)i + 2];
  }
  if (a.has_T) {
    double q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = a.T.m[r] * x + a.T.m[r + 4] * y + a.T.m[r + 8] * z + a.T.m[r + 12];
    x = q[0] / q[3];
    y = q[1] / q[3];
    z = q[2] / q[3];
    moved = true;
  }
  if (a.states) {
    uint32_t lo = run_max, hi = a.fused ? lo : a.states;  // first s with i < ends[s] (fused: the running maximum IS that state)
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (i < ends_sh[mid]) hi = mid; else lo = mid + 1;
    }
    if (lo < (a.fused ? a.max_hits : a.states)) {  // (after the last segment: left as it is)
      const double* T = a.poses + 12 * (size_t)lo;
      const double rx = T[0] * x + T[3] * y + T[6] * z;
      const double ry = T[1] * x + T[4] * y + T[7] * z;
      const double rz = T[2] * x + T[5] * y + T[8] * z;
      x = rx + T[9];
      y = ry + T[10];
      z = rz + T[11];
      moved = true;
    }
  }
  if (moved) {
    a.pts[3 * (size_t)i] = x;
    a.pts[3 * (size_t)i + 1] = y;
    a.pts[3 * (size_t)i + 2] = z;
  }
  a.keep_by_index[i] = 0u;
  a.codes[i] = morton3(cell_coord(x, a.fine), cell_coord(y, a.fine), cell_coord(z, a.fine));
  a.idx[i] = i;
  const bool is_finite = x - x == 0.0 && y - y == 0.0 && z - z == 0.0;
  if (is_finite && (beyond_grid(x, a.fine) || beyond_grid(y, a.fine) || beyond_grid(z, a.fine))) a.counters[kBeyondGrid] = a.epoch;
}

// Element and operator of the segmented scan over the sorted order: count = voxel runs started so far, packed =
// (original index << 32 | sorted position) of the LOWEST original index since the last run start. At the last
// point of a voxel's run that is the voxel's rank (+1) and its kept point -- the first point of the voxel in scan
// order -- with its sorted position (inside a run the sort orders by the finer cells, not by index).
struct RunMin {
  unsigned long long packed;
  uint32_t count;
  uint32_t runs;  // cells opened over all levels (plain sum: its total sizes the cell table)
};
struct RunMinOp {
  __host__ __device__ __forceinline__ RunMin operator()(const RunMin& a, const RunMin& b) const {
    RunMin r;
    r.count = a.count + b.count;
    r.packed = b.count ? b.packed : (a.packed < b.packed ? a.packed : b.packed);
    r.runs = a.runs + b.runs;
    return r;
  }
};

// ---- device-wide scans in ONE launch each ---------------------------------------------------------------------
// A scan is 10^4 .. 10^6 points: 5 .. 500 tiles of 2 048.  Every tile publishes its aggregate and folds the aggregates
// of ALL tiles before it (they are few, the fold is integer arithmetic in tile order): no second pass, no prefix
// status.  Tile numbers are drawn from a ticket, so a tile only ever waits for tiles that started before it (never a
// cycle, whatever the dispatch order or residency); waits are bounded all the same and a timeout is reported through
// the counter block instead of hanging the device.  Hand-off per MI355X_MICROARCH.md ("who signals ... ONE lane"):
// write-through stores of the aggregate, s_waitcnt vmcnt(0), write-through store of the flag (= the call's epoch, so
// the slots need no clearing between calls); the consumer polls the flag and then reads the aggregate write-through.
constexpr int kScanThreads = 256;
// consecutive items per thread: 4 (tiles of 1 024) while the scan has room in the tile slots, 8 (tiles of 2 048) beyond
// 4 M points.  At scan sizes a tile more is a workgroup more on an idle CU, and four items are half the chain of
// eight (run_scan_kernel at 60 000 points: 20.2 us with 8, 17.6 with 4, 16.7 with 2; keep_scan_kernel 5.5 / 5.1 / 5.1).
constexpr int kScanItemsSmall = 4, kScanItemsLarge = 8;
inline int scan_items_for(uint32_t n) {
#ifdef VGICP_SCAN_ITEMS_LARGE_ONLY   // developer builds: the > 4 M-point variant at test sizes (tools/ab_build.sh items8 -DVGICP_SCAN_ITEMS_LARGE_ONLY)
  (void)n;
  return kScanItemsLarge;
#else
  return (uint64_t)n <= (uint64_t)kMaxScanTiles * kScanThreads * kScanItemsSmall ? kScanItemsSmall : kScanItemsLarge;
#endif
}
constexpr uint32_t kScanSpinLimit = 1u << 22;

struct TileSlot {  // 32 bytes
  unsigned long long packed;
  uint32_t count, runs;
  uint32_t flag;
  uint32_t pad[3];
};
static_assert(sizeof(TileSlot) == 32, "tile slot layout");

__device__ __forceinline__ RunMin runmin_identity() {
  RunMin r;
  r.packed = ~0ull;
  r.count = 0;
  r.runs = 0;
  return r;
}
__device__ __forceinline__ RunMin runmin_up(const RunMin& v, int d) {
  RunMin r;
  const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v.packed >> 32), d, 64);
  const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v.packed, d, 64);
  r.packed = ((unsigned long long)hi << 32) | lo;
  r.count = (uint32_t)__shfl_up((int)v.count, d, 64);
  r.runs = (uint32_t)__shfl_up((int)v.runs, d, 64);
  return r;
}
__device__ __forceinline__ RunMin runmin_lane(const RunMin& v, int src) {
  RunMin r;
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v.packed >> 32), src, 64);
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v.packed, src, 64);
  r.packed = ((unsigned long long)hi << 32) | lo;
  r.count = (uint32_t)__shfl((int)v.count, src, 64);
  r.runs = (uint32_t)__shfl((int)v.runs, src, 64);
  return r;
}
// inclusive scan over the 64 lanes of a wave, in lane order
__device__ __forceinline__ RunMin runmin_wave_scan(RunMin v, uint32_t lane) {
  const RunMinOp op;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const RunMin o = runmin_up(v, d);
    if ((int)lane >= d) v = op(o, v);
  }
  return v;
}

// The pass over the sorted codes, fused: the points are copied into sorted order (32-byte records {x, y, z, index}),
// voxel-level run starts feed the segmented scan, and the last point of every voxel's run -- which then knows the
// voxel's rank, its kept point (lowest original index of the run) and that point's sorted position -- writes the
// query list (Morton order of the voxels) and the kept flag.  The scan's total is the number of kept points and of
// octree cells over all levels (counters[0], [1]); no atomics on shared words ("a note on atomics").
// The launch carries the octree's cell table along: the workgroups behind the scan's tiles run cell_build_body
// (they need the sorted codes only and fill the CUs the thirty-odd tiles of a sweep leave idle; as a launch of its own
// the table cost 13 us between the scans and the search).
__device__ __forceinline__ void cell_build_body(const unsigned long long* __restrict__ codes, uint32_t n, CellEntry* table,
                                                uint32_t mask, uint32_t block_x, int level);
template <int kScanItems>
__global__ __launch_bounds__(kScanThreads) void run_scan_kernel(
    const double* __restrict__ pts, const unsigned long long* __restrict__ codes, const uint32_t* __restrict__ idx,
    uint32_t n, double* __restrict__ sorted_pts, uint32_t* __restrict__ queries, uint32_t* __restrict__ keep_by_index,
    uint32_t* counters, TileSlot* tiles, uint32_t epoch, uint32_t scan_tiles, CellEntry* table, uint32_t mask,
    uint32_t cell_blocks_x, unsigned long long* host_kept) {
  if (blockIdx.x >= scan_tiles) {  // (whole workgroups; before anything synchronises)
    const uint32_t v = blockIdx.x - scan_tiles;
    cell_build_body(codes, n, table, mask, v % cell_blocks_x, (int)(v / cell_blocks_x));
    return;
  }
  constexpr uint32_t kScanTile = kScanThreads * kScanItems;
  __shared__ uint32_t tile_sh;
  __shared__ RunMin wave_tot[kScanThreads / 64];
  __shared__ RunMin prefix_sh;
  __shared__ uint32_t abort_sh;
  const RunMinOp op;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) tile_sh = atomicAdd(&counters[kTicketA], 1u);
  __syncthreads();
  const uint32_t tile = tile_sh;
  const uint32_t j0 = tile * kScanTile + tid * kScanItems;  // this thread's kScanItems consecutive sorted positions
  // Every load below is unconditional, from a clamped place, and masked afterwards: a load under a condition becomes a
  // branch, and the eight items' chains (index -> point -> store) then run one after the other — thirty tiles are one
  // wave per SIMD, nothing else hides a round trip (round 6: 24 -> see profiles; the sort's header has the same note).
  unsigned long long c[kScanItems + 2];                     // codes j0 - 1 .. j0 + kScanItems
#pragma unroll
  for (int k = 0; k < kScanItems + 2; ++k) {
    const long long j = (long long)j0 + k - 1;
    const bool there = j >= 0 && j < (long long)n;
    const unsigned long long got = codes[there ? j : 0];
    c[k] = there ? got : 0ull;
  }
  uint32_t src[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const uint32_t j = j0 + (uint32_t)k;
    src[k] = idx[j < n ? j : 0u];
  }
  double px[kScanItems], py[kScanItems], pz[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const size_t i = j0 + (uint32_t)k < n ? src[k] : 0u;
    px[k] = pts[3 * i];
    py[k] = pts[3 * i + 1];
    pz[k] = pts[3 * i + 2];
  }
  RunMin e[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const uint32_t j = j0 + (uint32_t)k;
    e[k] = runmin_identity();
    if (j < n) {
      const unsigned long long cur = c[k + 1], prev = c[k];
      const uint32_t i = src[k];
      double2* rec = reinterpret_cast<double2*>(sorted_pts + 4 * (size_t)j);
      rec[0] = make_double2(px[k], py[k]);
      rec[1] = make_double2(pz[k], __longlong_as_double((long long)i));
      // cells this point opens: the levels, from the finest, on which its code differs from its predecessor's — all
      // levels up to the one that holds the highest differing bit
      const unsigned long long diff = cur ^ prev;
      uint32_t runs = (uint32_t)kLevels;
      if (j != 0) {
        const uint32_t upto = diff ? (63u - (uint32_t)__builtin_clzll(diff)) / 3u + 1u : 0u;
        runs = upto < (uint32_t)kLevels ? upto : (uint32_t)kLevels;
      }
      e[k].packed = ((unsigned long long)i << 32) | j;
      e[k].count = (j == 0 || (cur >> (3 * kFineShift)) != (prev >> (3 * kFineShift))) ? 1u : 0u;
      e[k].runs = runs;
    }
  }
#pragma unroll
  for (int k = 1; k < kScanItems; ++k) e[k] = op(e[k - 1], e[k]);
  const RunMin incl = runmin_wave_scan(e[kScanItems - 1], lane);
  if (lane == 63u) wave_tot[wave] = incl;
  RunMin excl = runmin_up(incl, 1);
  if (lane == 0u) excl = runmin_identity();
  __syncthreads();
  if (wave == 0) {
    // publish this tile's aggregate, then fold the aggregates of all tiles before it, in tile order
    RunMin agg = wave_tot[0];
#pragma unroll
    for (int w = 1; w < kScanThreads / 64; ++w) agg = op(agg, wave_tot[w]);
    if (lane == 0u) {
      TileSlot* mine = tiles + tile;
      st_through64(&mine->packed, agg.packed);
      st_through64(&mine->count, ((unsigned long long)agg.runs << 32) | agg.count);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st_through32(&mine->flag, epoch);
    }
    RunMin pre = runmin_identity();
    bool late = false;
    for (uint32_t base = 0; base < tile; base += 64u) {
      const uint32_t t = base + lane;
      RunMin v = runmin_identity();
      if (t < tile) {
        const TileSlot* src = tiles + t;
        uint32_t spins = 0;
        while (ld_through32(&src->flag) != epoch) {
          if (++spins > kScanSpinLimit) { late = true; break; }
          __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        v.packed = ld_through64(&src->packed);
        const unsigned long long cr = ld_through64(&src->count);
        v.count = (uint32_t)cr;
        v.runs = (uint32_t)(cr >> 32);
      }
      const RunMin sc = runmin_wave_scan(v, lane);
      pre = op(pre, runmin_lane(sc, 63));
    }
    const bool gave_up = __any(late);
    if (gave_up && lane == 0u) counters[kScanTimeout] = epoch;
    if (lane == 0u) {
      prefix_sh = pre;
      abort_sh = gave_up ? 1u : 0u;
    }
  }
  __syncthreads();
  if (abort_sh != 0u) {
    // The prefix was folded from slots that were never published (stale values of an older call, possibly of a larger
    // scan): nothing this tile would store is trustworthy, and the indices it would store THROUGH are not even in
    // range.  No store at all; the tile that owns the scan's total leaves an empty scan behind, so that a kernel
    // enqueued behind the preparation (the align that does not wait for the host) finds nothing to read.
    if (tid == 0 && (size_t)(tile + 1) * kScanTile >= n) {
      counters[0] = counters[1] = 0u;
      if (host_kept) __hip_atomic_store(host_kept, (unsigned long long)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  RunMin base_prefix = prefix_sh;
  for (uint32_t w = 0; w < wave; ++w) base_prefix = op(base_prefix, wave_tot[w]);
  base_prefix = op(base_prefix, excl);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const uint32_t j = j0 + (uint32_t)k;
    if (j >= n) break;
    // the last point of a voxel's run reads the scan
    if (j + 1 < n && (c[k + 2] >> (3 * kFineShift)) == (c[k + 1] >> (3 * kFineShift))) continue;
    const RunMin v = op(base_prefix, e[k]);
    if (j + 1 == n) {
      counters[0] = v.count;
      counters[1] = v.runs;
      // the host may be waiting for exactly this number (vgicp_scan_fetch_begin: it sizes its vectors while the
      // neighbour search runs): one posted write into page-locked memory, tagged with the call's epoch
      if (host_kept) __hip_atomic_store(host_kept, ((unsigned long long)epoch << 32) | v.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    queries[v.count - 1u] = (uint32_t)v.packed;
    keep_by_index[(uint32_t)(v.packed >> 32)] = 1u;
  }
}

// Exclusive prefix sum of the kept flags over SCAN order: every kept point's output slot (ascending original
// index). Same single-launch scheme; a tile's aggregate travels as one 8-byte granule {epoch, sum}: the data is
// the signal.
// (round 6) Run by SUB-BLOCKS of 256 threads inside the 1 024-thread workgroups of keep_and_split_kernel: `sub` is the
// sub-block, `tid` the thread inside it; every thread of the workgroup calls this (the barriers are the workgroup's), a
// sub-block whose ticket lies beyond the scan's tiles only keeps the others company.
template <int kScanItems>
__device__ __forceinline__ void keep_scan_body(const uint32_t* __restrict__ keep_by_index, uint32_t n,
                                               uint32_t* __restrict__ rank_of_index, uint32_t* counters,
                                               unsigned long long* tiles, uint32_t epoch, uint32_t scan_tiles) {
  constexpr uint32_t kScanTile = kScanThreads * kScanItems;
  constexpr int kSubs = 4;   // kSplitBlock / kScanThreads
  __shared__ uint32_t tile_sh[kSubs];
  __shared__ uint32_t wave_tot[kSubs][kScanThreads / 64];
  __shared__ uint32_t prefix_sh[kSubs];
  __shared__ uint32_t abort_sh[kSubs];
  const uint32_t sub = threadIdx.x / kScanThreads, tid = threadIdx.x % kScanThreads, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) tile_sh[sub] = atomicAdd(&counters[kTicketB], 1u);
  __syncthreads();
  const uint32_t tile = tile_sh[sub];
  const bool idle = tile >= scan_tiles;   // (sub-block-uniform)
  const uint32_t i0 = tile * kScanTile + tid * kScanItems;
  uint32_t f[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {   // (unconditional loads from a clamped place: see run_scan_kernel)
    const bool there = !idle && i0 + (uint32_t)k < n;
    const uint32_t got = keep_by_index[there ? i0 + (uint32_t)k : 0u];
    f[k] = there ? got : 0u;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sum += f[k];
  uint32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    if ((int)lane >= d) incl += o;
  }
  if (lane == 63u) wave_tot[sub][wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t pre = 0;
    bool late = false;
    if (!idle) {
      uint32_t agg = 0;
#pragma unroll
      for (int w = 0; w < kScanThreads / 64; ++w) agg += wave_tot[sub][w];
      if (lane == 0u) st_through64(tiles + tile, ((unsigned long long)epoch << 32) | agg);
      for (uint32_t base = 0; base < tile; base += 64u) {
        const uint32_t t = base + lane;
        uint32_t v = 0;
        if (t < tile) {
          uint32_t spins = 0;
          unsigned long long w = ld_through64(tiles + t);
          while ((uint32_t)(w >> 32) != epoch) {
            if (++spins > kScanSpinLimit) { late = true; break; }
            __builtin_amdgcn_s_sleep(2);
            w = ld_through64(tiles + t);
          }
          v = (uint32_t)w;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
        pre += v;
      }
    }
    const bool gave_up = __any(late);
    if (gave_up && lane == 0u) counters[kScanTimeout] = epoch;
    if (lane == 0u) {
      prefix_sh[sub] = pre;
      abort_sh[sub] = gave_up ? 1u : 0u;
    }
  }
  __syncthreads();
  if (idle || abort_sh[sub] != 0u) return;  // a prefix made of stale slots: no output slot is written (every reader checks the timeout word)
  uint32_t run = prefix_sh[sub] + (incl - sum);
  for (uint32_t w = 0; w < wave; ++w) run += wave_tot[sub][w];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (i0 + (uint32_t)k < n) rank_of_index[i0 + (uint32_t)k] = run;
    run += f[k];
  }
}

// Runs of every level in one pass: point j opens the cells whose run starts at j (its code differs from its
// predecessor's at that level) and closes those whose run ends at j (differs from its successor's). Whichever of
// the two comes first claims the cell's entry (CAS on the key); each stores its own field.
__device__ __forceinline__ CellEntry* claim_cell(CellEntry* table, uint32_t mask, unsigned long long key) {
  uint32_t slot = cell_hash(key) & mask;
  for (;;) {  // cells are unique per (level, key) and the table holds at least eight times their number
    const unsigned long long seen = atomicCAS(&table[slot].key, kEmptyCell, key);
    if (seen == kEmptyCell || seen == key) return table + slot;
    slot = (slot + 1) & mask;
  }
}
__device__ __forceinline__ void cell_build_body(const unsigned long long* __restrict__ codes, uint32_t n, CellEntry* table,
                                                uint32_t mask, uint32_t block_x, int l) {
  // one thread per (point, level): a point on a coarse boundary opens and closes a cell on every level, and one
  // thread doing those two dozen atomic round trips in sequence set the pace
  const uint32_t j = block_x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  // (the three codes are requested together: a load behind `||` is a branch and a round trip of its own)
  const unsigned long long c_here = codes[j], c_before = codes[j ? j - 1 : 0u], c_after = codes[j + 1 < n ? j + 1 : j];
  const unsigned long long c = c_here >> (3 * l);
  const bool opens = j == 0 || c != (c_before >> (3 * l));
  const bool closes = j + 1 == n || c != (c_after >> (3 * l));
  if (!opens && !closes) return;
  CellEntry* e = claim_cell(table, mask, cell_key(c, l));
  if (opens) e->start = j;
  if (closes) e->end = j + 1;
}
__device__ __forceinline__ const CellEntry* find_cell(const CellEntry* table, uint32_t mask,
                                                      unsigned long long key) {
  uint32_t slot = cell_hash(key) & mask;
  for (;;) {
    const unsigned long long seen = table[slot].key;
    if (seen == key) return table + slot;
    if (seen == kEmptyCell) return nullptr;
    slot = (slot + 1) & mask;
  }
}


// ---- wave-level helpers ---------------------------------------------------------------------------
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  return ((unsigned long long)uniform_u32((uint32_t)(v >> 32)) << 32) | uniform_u32((uint32_t)v);
}

// Which queries start first.  A query in a sparse neighbourhood takes several times the average (it starts high up in
// the octree and takes many cells); dispatched in Morton order, the ones that happen to come last keep the kernel
// waiting 40 - 50 us for a handful of waves.  So the search starts them FIRST: a query whose own level-4 cell (1.2 m at
// 0.3 m voxels) holds fewer than kHeavyBelow points goes to the head of the launch (list scheduling of the measured
// durations: frame sweep 147 -> 136 us, 100k-point sweep 244 -> 216 us, profiles/r10_knn_leaf.txt), the others follow in
// (nearly) Morton order -- a workgroup's 1 024 queries stay together, the workgroups' blocks land in the order their atomics do.
// The order of the queries changes nothing in what the search writes: every query owns its output slot.
#ifndef VGICP_HEAVY_BELOW
#define VGICP_HEAVY_BELOW 4
#endif
#ifndef VGICP_HEAVY_LEVEL
#define VGICP_HEAVY_LEVEL 4
#endif
constexpr int kHeavyLevel = VGICP_HEAVY_LEVEL;
constexpr uint32_t kHeavyBelow = VGICP_HEAVY_BELOW;
constexpr int kSplitBlock = 1024;  // one atomic per list and workgroup: same-address atomics cost ~20 ns apiece
__device__ __forceinline__ void query_split_body(uint32_t block, const double* __restrict__ spts, uint32_t n, double h,
                                                 const CellEntry* __restrict__ table, uint32_t mask,
                                                 const uint32_t* __restrict__ queries, uint32_t epoch,
                                                 uint32_t* __restrict__ heavy, uint32_t* __restrict__ light,
                                                 uint32_t* counters) {
  __shared__ uint32_t wave_heavy[kSplitBlock / 64], wave_light[kSplitBlock / 64], base_sh[2];
  const uint32_t m = counters[0];
  if (counters[kBeyondGrid] == epoch || counters[kScanTimeout] == epoch) return;
  const uint32_t r = block * blockDim.x + threadIdx.x;
  const bool valid = r < m && r < n;
  uint32_t qj = 0;
  bool is_heavy = false;
  if (valid) {
    qj = queries[r];
    const double fine = h / (double)(1 << kFineShift);
    const double2* rec = reinterpret_cast<const double2*>(spts + 4 * (size_t)qj);
    const double2 xy = rec[0];
    const double z = rec[1].x;
    const unsigned long long mq = morton3(cell_coord(xy.x, fine), cell_coord(xy.y, fine), cell_coord(z, fine));
    const CellEntry* e = find_cell(table, mask, cell_key(mq >> (3 * kHeavyLevel), kHeavyLevel));
    is_heavy = e == nullptr || e->end - e->start < kHeavyBelow;
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long hm = __ballot(valid && is_heavy), lm = __ballot(valid && !is_heavy);
  if (lane == 0) {
    wave_heavy[wave] = (uint32_t)__builtin_popcountll(hm);
    wave_light[wave] = (uint32_t)__builtin_popcountll(lm);
  }
  __syncthreads();
  if (threadIdx.x < 2) {   // thread 0: the heavy list, thread 1: the light one -- exclusive prefix over the waves, one atomic
    uint32_t* per_wave = threadIdx.x == 0 ? wave_heavy : wave_light;
    uint32_t total = 0;
    for (int k = 0; k < kSplitBlock / 64; ++k) {
      const uint32_t c = per_wave[k];
      per_wave[k] = total;
      total += c;
    }
    base_sh[threadIdx.x] = total ? atomicAdd(&counters[threadIdx.x == 0 ? kHeavyQueries : kLightQueries], total) : 0u;
  }
  __syncthreads();
  if (valid) {
    if (is_heavy) heavy[base_sh[0] + wave_heavy[wave] + (uint32_t)__builtin_popcountll(hm & below)] = qj;
    else light[base_sh[1] + wave_light[wave] + (uint32_t)__builtin_popcountll(lm & below)] = qj;
  }
}

// The two small passes between the scans and the search in ONE launch (round 6; they need the first scan's results and
// nothing of each other): the first keep_blocks workgroups give every kept point its output slot (keep_scan_body, four
// tiles of 256 threads per workgroup), the others sort the queries into the two start lists (query_split_body).  As
// launches of their own they took 5.0 + 5.6 us and a launch boundary.
static_assert(kSplitBlock == 4 * kScanThreads, "keep_scan_body runs four sub-blocks per workgroup");
template <int kScanItems>
__global__ __launch_bounds__(kSplitBlock) void keep_and_split_kernel(const uint32_t* __restrict__ keep_by_index, uint32_t n,
                                                                    uint32_t* __restrict__ rank_of_index, uint32_t* counters,
                                                                    unsigned long long* tiles, uint32_t epoch, uint32_t scan_tiles,
                                                                    uint32_t keep_blocks, const double* __restrict__ spts, double h,
                                                                    const CellEntry* __restrict__ table, uint32_t mask,
                                                                    const uint32_t* __restrict__ queries, uint32_t* __restrict__ heavy,
                                                                    uint32_t* __restrict__ light) {
  if (blockIdx.x < keep_blocks) keep_scan_body<kScanItems>(keep_by_index, n, rank_of_index, counters, tiles, epoch, scan_tiles);
  else query_split_body(blockIdx.x - keep_blocks, spts, n, h, table, mask, queries, epoch, heavy, light, counters);
}

// Exact k nearest neighbours of one kept point per wave; writes the neighbours' original indices (ascending
// distance, ties by index) to nbr[slot * kMaxKnn + k] and the point itself to the output.
__global__ __launch_bounds__(kSearchBlock) void knn_search_kernel(
    const double* __restrict__ spts, const uint32_t* __restrict__ sorted_idx, uint32_t n, double h, int knn,
    const CellEntry* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ heavy, const uint32_t* __restrict__ light,
    const uint32_t* __restrict__ slot_of_index, uint32_t epoch, uint32_t* __restrict__ nbr,
    double* __restrict__ out_pts, unsigned long long* __restrict__ out_idx, double* __restrict__ soa, uint64_t soa_stride,
    uint32_t* counters, int debug) {
  __shared__ float pool_d[kSearchBlock / 64][kPool];   // cell distances rounded DOWN: ordering and pruning stay safe
  __shared__ unsigned long long pool_key[kSearchBlock / 64][kPool];
  __shared__ uint32_t pool_start[kSearchBlock / 64][kPool];
  __shared__ uint32_t pool_end[kSearchBlock / 64][kPool];
#ifdef VGICP_KNN_LAZY
  __shared__ double cand_d[kSearchBlock / 64][64];     // the candidates still in the race (at most 64), unordered
  __shared__ uint32_t cand_i[kSearchBlock / 64][64];
#endif
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give every XCD one contiguous
  // eighth of the Morton-ordered queries, so that neighbouring queries (same cells, same points) share an L2.
  constexpr uint32_t kXcds = 8;
  // the number of kept points is still on the device (the grid was sized from the raw scan): surplus waves leave;
  // so does everyone when the scan was refused (a point beyond the search grid) or a device-wide scan timed out
  const uint32_t m = uniform_u32(counters[0]);
  if (uniform_u32(counters[kBeyondGrid]) == epoch || uniform_u32(counters[kScanTimeout]) == epoch) return;
  // the queries of sparse neighbourhoods first (query_split_body), one per wave in dispatch order; then the others,
  // every XCD one contiguous eighth of their list
  const uint32_t n_heavy = uniform_u32(counters[kHeavyQueries]);
  const uint32_t n_light = m - n_heavy;
  const uint32_t w = blockIdx.x * (kSearchBlock / 64) + wave;     // this wave in dispatch order
  const uint32_t head = (n_heavy + kXcds * (kSearchBlock / 64) - 1) / (kXcds * (kSearchBlock / 64)) * (kXcds * (kSearchBlock / 64));
  uint32_t qj_pick;
  if (w < head) {
    if (w >= n_heavy) return;  // whole waves leave; nothing below synchronises across waves
    qj_pick = heavy[w];
  } else {
    const uint32_t per_xcd = (n_light + kXcds - 1) / kXcds;
    const uint32_t slot_in_xcd = ((blockIdx.x - head / (kSearchBlock / 64)) / kXcds) * (kSearchBlock / 64) + wave;
    const uint32_t qrank = (blockIdx.x % kXcds) * per_xcd + slot_in_xcd;
    if (slot_in_xcd >= per_xcd || qrank >= n_light) return;
    qj_pick = light[qrank];
  }
  // The pool is this wave's own: its lanes exchange entries through it, and a wave's LDS instructions execute in
  // order. Plain LDS pointers (ds_read / ds_write) with a wave-scope fence wherever one lane reads what another
  // wrote -- NOT volatile: that turns every access into a flat load with its own full wait.
  float* pd = pool_d[wave];
  unsigned long long* pk = pool_key[wave];
  uint32_t* ps = pool_start[wave];
  uint32_t* pe = pool_end[wave];
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const double fine = h / (double)(1 << kFineShift);
  const uint32_t qj = uniform_u32(qj_pick);
  const double qx = uniform_f64(spts[4 * (size_t)qj]), qy = uniform_f64(spts[4 * (size_t)qj + 1]),
               qz = uniform_f64(spts[4 * (size_t)qj + 2]);
  const int K = knn < (int)n ? knn : (int)n;
  const uint64_t t_begin = debug >= 2 ? wall_clock64() : 0;
#ifdef VGICP_PREP_TRACE
  const uint64_t t_trace = wall_clock64();
#endif
  const unsigned long long lanes_below = (1ull << lane) - 1ull;

  uint32_t batches = 0, pops = 0, spills = 0, inserts = 0;
#ifdef VGICP_PREP_TRACE
  uint32_t opens = 0, home_batches = 0;
#endif
  [[maybe_unused]] const unsigned long long list_lanes = K >= 64 ? ~0ull : (1ull << K) - 1ull;

  // squared distance to sorted point j and, from the same record, its original index
  auto dist2 = [&](uint32_t j, uint32_t& id) {
    const double2* rec = reinterpret_cast<const double2*>(spts + 4 * (size_t)j);
    const double2 xy = rec[0], zi = rec[1];
    id = (uint32_t)__double_as_longlong(zi.y);
    const double dx = xy.x - qx, dy = xy.y - qy, dz = zi.x - qz;
    return dx * dx + dy * dy + dz * dz;
  };
  const uint32_t cx = uniform_u32(cell_coord(qx, fine)), cy = uniform_u32(cell_coord(qy, fine)),
                 cz = uniform_u32(cell_coord(qz, fine));
  const unsigned long long mq = morton3(cx, cy, cz);  // wave-uniform: scalar unit
  const double fx = qx / fine - floor(qx / fine), fy = qy / fine - floor(qy / fine), fz = qz / fine - floor(qz / fine);
  const float ffx = (float)fx, ffy = (float)fy, ffz = (float)fz;
  const float fine2_low = __double2float_rd(fine * fine) * (1.0f - 0x1p-20f);
  // The query's own cell on every level (lane l asks for level l: the query's Morton code, made once on the scalar
  // unit, shifted). "Home" is the finest of them that holds at least K points: its points are the search's first
  // candidates, and the K-th distance among them is already close to the final one.
  uint32_t own_start = 0, own_end = 0;
  if (lane < kLevels) {
    const CellEntry* e = find_cell(table, mask, cell_key(mq >> (3 * lane), lane));
    if (e) { own_start = e->start; own_end = e->end; }
  }
  const unsigned long long enough = __ballot(own_end - own_start >= (uint32_t)K);
  const int home_level = enough ? __builtin_ctzll(enough) : 0;
  const uint32_t home_start = (uint32_t)__builtin_amdgcn_readlane((int)own_start, home_level),
                 home_end = (uint32_t)__builtin_amdgcn_readlane((int)own_end, home_level);
  // (not when it is crowded -- thousands of returns in one finest cell next to the sensor: measuring all of them in
  // sorted order costs more insertions than the best-first search needs; 64: 270 us, 128 / 256: 250 us, no cap: 308 us)
  const bool has_home = enough != 0 && home_end - home_start <= kHomePoints;

  // The k-list: lane l < K holds the l-th nearest so far, by (distance, index). It starts FULL, from consecutive points
  // of the sorted order around the query (inside home when there is one): real points and usually near ones. The
  // search skips the points it has measured here when it meets them again (sorted positions skip_start ..
  // skip_start + skip_count - 1).  (-DVGICP_KNN_SEEDS_ONLY keeps the opening of rounds 2-5 for A/B runs: K seeds.)
#ifdef VGICP_KNN_LAZY
  // ---- round 6: the candidates are kept UNORDERED until the end ---------------------------------------------------
  // What the search needs while it runs is one number, an upper bound of the K-th distance so far (it filters points
  // and prunes cells); the ORDER of the K nearest is needed once, at the end.  Rounds 2-5 kept a sorted list and paid
  // ~15 instructions per candidate that passed (27 per query) plus an all-pairs ranking of the 64-point opening window
  // (~200).  Here the candidates that pass the bound are appended to a buffer of at most 64 (lane l mirrors entry l),
  // and after every batch that brought any the bound is tightened by a RADIX SELECT over single-precision keys rounded
  // UP (16 wave ballots: one vector compare each, the counting on the scalar unit) and the buffer compacted to the
  // entries below it; the exact order by (distance, index) is made once from the ~K survivors.  Exact: a point is only
  // ever dropped for lying above a bound that at least K seen points lie under.
  double* cdv = cand_d[wave];
  uint32_t* civ = cand_i[wave];
  const uint32_t span_lo = has_home ? home_start : 0u, span_hi = has_home ? home_end : n;
  const uint32_t W = span_hi - span_lo < 64u ? span_hi - span_lo : 64u;   // >= K: home holds K points, and K <= n
  uint32_t w0 = qj > W / 2 ? qj - W / 2 : 0;
  if (w0 < span_lo) w0 = span_lo;
  if (w0 + W > span_hi) w0 = span_hi - W;
  uint32_t skip_start = w0, skip_count = W;
  double ld = INFINITY;           // lane < cnt: entry `lane` of the buffer
  uint32_t li = 0xFFFFFFFFu;
  uint32_t cnt = 0;               // uniform
  double kth = INFINITY;          // uniform: an upper bound of the K-th smallest distance among the points seen so far
  // K-th smallest single-precision key (distances rounded UP; non-negative floats order as unsigned integers) among the
  // cnt entries, to its 16 leading bits (VGICP_KNN_SELECT_LOW), rounded up: the new bound.  Entries above it leave the buffer.
  auto tighten = [&]() {
    const uint32_t fkey = (uint32_t)lane < cnt ? __builtin_bit_cast(uint32_t, __double2float_ru(ld)) : 0xFFFFFFFFu;
    uint32_t prefix = 0;
#pragma unroll
    for (int bit = 30; bit >= VGICP_KNN_SELECT_LOW; --bit) {
      const uint32_t trial = prefix | (1u << bit);
      const uint32_t below = (uint32_t)__builtin_popcountll(__ballot(fkey < trial));
      if (below < (uint32_t)K) prefix = trial;   // the K-th key has this bit set
    }
    const uint32_t top = prefix | ((1u << VGICP_KNN_SELECT_LOW) - 1u);
    const bool keep = fkey <= top;
    const unsigned long long who = __ballot(keep);
    const uint32_t kept = (uint32_t)__builtin_popcountll(who);
    if (kept < cnt) {
      const uint32_t at = (uint32_t)__builtin_popcountll(who & lanes_below);
      wave_sync();   // every lane holds its entry in registers: the slots can be rewritten
      if (keep) { cdv[at] = ld; civ[at] = li; }
      wave_sync();
      cnt = kept;
      if ((uint32_t)lane < cnt) { ld = cdv[lane]; li = civ[lane]; }
    }
    kth = top >= 0x7F800000u ? INFINITY : (double)__builtin_bit_cast(float, top);
    if (cnt > 56u) {
      // hardly any room left (dozens of points within a 2^-12 band of the K-th distance): keep exactly the K nearest by
      // (distance, index) — all pairs, the one place besides the end where the order is made
      uint32_t rank = 0;
      const unsigned long long key = (unsigned long long)__double_as_longlong(ld);
      for (uint32_t e = 0; e < cnt; ++e) {
        const unsigned long long ke = (unsigned long long)__double_as_longlong(cdv[e]);
        const uint32_t ie = civ[e];
        rank += ((ke < key) | ((ke == key) & (ie < li))) ? 1u : 0u;
      }
      const bool stay = (uint32_t)lane < cnt && rank < (uint32_t)K;
      wave_sync();
      if (stay) { cdv[rank] = ld; civ[rank] = li; }
      wave_sync();
      cnt = (uint32_t)K;
      if ((uint32_t)lane < cnt) { ld = cdv[lane]; li = civ[lane]; }
      kth = readlane_f64(ld, K - 1);
    }
  };
  {
    const bool mine = (uint32_t)lane < W;
    uint32_t id = 0xFFFFFFFFu;
    double d = INFINITY;
    if (mine) d = dist2(w0 + (uint32_t)lane, id);
    if (!(d < INFINITY)) d = INFINITY;  // a NaN / infinite point is a placeholder that everything finite displaces
    ld = d;
    li = id;
    cdv[lane] = d;
    civ[lane] = id;
    wave_sync();
    cnt = W;
    if (cnt > (uint32_t)K) tighten();
    else if (cnt == (uint32_t)K) {   // exactly K points in reach so far: their largest distance is the bound
      double mx = (uint32_t)lane < cnt ? ld : 0.0;
      for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
      kth = uniform_f64(mx);
    }
  }
  uint32_t kth_id = 0xFFFFFFFFu;   // (ties at the bound are admitted: the order is settled at the end)
  double bound = kth;
  // every lane brings one candidate (valid, d, id, j); the ones at or below the bound join the buffer
  auto offer = [&](bool valid, double d, uint32_t id, uint32_t j) {
    ++batches;
    bool pass = valid & (j - skip_start >= skip_count) & (d <= kth);
    unsigned long long todo = __ballot(pass);
    while (todo) {
      const uint32_t room = 64u - cnt;
      const uint32_t ahead = (uint32_t)__builtin_popcountll(todo & lanes_below);
      const bool now = pass && ahead < room;
      if (now) { cdv[cnt + ahead] = d; civ[cnt + ahead] = id; }
      const uint32_t took = (uint32_t)__builtin_popcountll(__ballot(now));
      inserts += took;
      cnt += took;
      // ... but not after every batch: a select costs ~35 vector instructions, a candidate that slips in under a stale
      // bound ~2.  The bound is tightened when the buffer holds VGICP_KNN_TIGHTEN_AT entries, or when candidates of this
      // batch are still waiting for room (A/B on the 60 000-point frame sweep: after every batch 100.4 us; at 40 / 48 /
      // 56 / 64 entries 95.8 / 96.6 / 95.1 / 96.3; at 40 with 16 instead of 20 bits of the key: 94.7)
      if (cnt > (uint32_t)K && (cnt >= (uint32_t)VGICP_KNN_TIGHTEN_AT || (todo & ~__ballot(now)) != 0ull)) {
        wave_sync();   // (the lanes' mirror of the buffer is brought up to date only where somebody reads it)
        if ((uint32_t)lane < cnt) { ld = cdv[lane]; li = civ[lane]; }
        tighten();
      }
      pass = pass && !now && (d <= kth);
      todo = __ballot(pass);
    }
  };
  (void)kth_id;
#else
#ifndef VGICP_KNN_SEEDS_ONLY
  // The opening window: up to 64 consecutive points of the sorted order around the query (inside home when there is
  // one), one per lane.  All of them are ranked against each other at once -- every lane counts the keys below its
  // own, the keys read one after the other from this wave's (still empty) pool: one LDS broadcast read, one compare
  // and one add per key -- and the K lowest are the list.  (Rounds 2-5 ranked K seeds with three v_readlane per key
  // and inserted the rest of home one candidate at a time: 38 insertions of ~16 instructions per query.)
  const uint32_t span_lo = has_home ? home_start : 0u, span_hi = has_home ? home_end : n;
  const uint32_t W = span_hi - span_lo < 64u ? span_hi - span_lo : 64u;   // >= K: home holds K points, and K <= n
  uint32_t w0 = qj > W / 2 ? qj - W / 2 : 0;
  if (w0 < span_lo) w0 = span_lo;
  if (w0 + W > span_hi) w0 = span_hi - W;
  uint32_t skip_start = w0, skip_count = W;
  double ld = INFINITY;
  uint32_t li = 0xFFFFFFFFu;
  {
    const bool mine = (uint32_t)lane < W;
    uint32_t id = 0xFFFFFFFFu;
    double d = INFINITY;
    if (mine) d = dist2(w0 + (uint32_t)lane, id);
    if (!(d < INFINITY)) d = INFINITY;  // a NaN / infinite point is a placeholder that everything finite displaces
    // squared distances are non-negative: their bit patterns order as unsigned integers
    const unsigned long long key = (unsigned long long)__double_as_longlong(d);
    pk[lane] = key;
    ps[lane] = id;
    wave_sync();
    uint32_t rank = 0;
#pragma unroll 16
    for (int s = 0; s < 64; ++s) rank += pk[s] < key ? 1u : 0u;
    // equal distances (rare) give equal ranks: found out by two lanes claiming one slot, settled by the index
    if (mine) pe[rank] = (uint32_t)lane;
    wave_sync();
    const bool clash = mine && pe[rank] != (uint32_t)lane;
    if (__ballot(clash)) {
      rank = 0;
      for (int s = 0; s < 64; ++s) {
        const unsigned long long ks = pk[s];
        const uint32_t is = ps[s];
        rank += ((ks < key) | ((ks == key) & (is < id))) ? 1u : 0u;
      }
    }
    wave_sync();   // every lane has read all keys: the slots can be rewritten in rank order
    if (mine) {
      pk[rank] = key;
      ps[rank] = id;
    }
    wave_sync();
    if (lane < K) {
      ld = __longlong_as_double((long long)pk[lane]);
      li = ps[lane];
    }
    wave_sync();
  }
#else
  const uint32_t half = (uint32_t)K / 2;
  uint32_t w0 = qj > half ? qj - half : 0;
  if (has_home) {
    if (w0 < home_start) w0 = home_start;
    if (w0 + (uint32_t)K > home_end) w0 = home_end - (uint32_t)K;
  } else if (w0 + (uint32_t)K > n) {
    w0 = n - (uint32_t)K;
  }
  uint32_t skip_start = w0, skip_count = (uint32_t)K;
  double ld = INFINITY;
  uint32_t li = 0xFFFFFFFFu;
  {
    const bool mine = lane < K;
    uint32_t id = 0xFFFFFFFFu;
    double d = INFINITY;
    if (mine) d = dist2(w0 + (uint32_t)lane, id);
    if (!(d < INFINITY)) d = INFINITY;  // a NaN / infinite point is a placeholder that everything finite displaces
    // rank of every seed among the seeds (all distinct by index), then one trip through this wave's (still
    // empty) pool to put lane l's entry into lane rank(l)
    uint32_t rank = 0;
    for (int s = 0; s < K; ++s) {
      const double sd = readlane_f64(d, s);
      const uint32_t si = (uint32_t)__builtin_amdgcn_readlane((int)id, s);
      rank += ((sd < d) | ((sd == d) & (si < id))) ? 1u : 0u;
    }
    if (mine) {
      pk[rank] = (unsigned long long)__double_as_longlong(d);
      ps[rank] = id;
    }
    wave_sync();
    if (mine) {
      ld = __longlong_as_double((long long)pk[lane]);
      li = ps[lane];
    }
    wave_sync();
  }
#endif
  double kth = readlane_f64(ld, K - 1);
  uint32_t kth_id = (uint32_t)__builtin_amdgcn_readlane((int)li, K - 1);
  double bound = kth;

  // every lane brings one candidate (valid, d, id, j); the passing ones are inserted one after the other
  auto offer = [&](bool valid, double d, uint32_t id, uint32_t j) {
    ++batches;
    const bool pass = valid & (j - skip_start >= skip_count) & (d <= bound) & ((d < kth) | ((d == kth) & (id < kth_id)));
    unsigned long long todo = __ballot(pass);
    bool moved = false;
    while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const double cd = readlane_f64(d, src);
      const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)id, src);
      // how many entries come before the candidate (scalar mask arithmetic; equal distances are rare and take
      // the uniform branch). Lanes >= K hold whatever was pushed out and are masked off.
      unsigned long long before = __ballot(ld < cd);
      const unsigned long long same = __ballot(ld == cd) & list_lanes;
      if (same) before |= same & __ballot(li < ci);
      const int p = __builtin_popcountll(before & list_lanes);
      if (p >= K) continue;  // the list moved on since the batch was filtered
      ++inserts;
      // lanes above p take the entry of the lane below (one DPP select per dword, lanes <= p keep theirs), then
      // lane p is overwritten from the scalar registers (v_writelane): 7 VALU instructions for the whole
      // shift-and-insert. (A DPP operand written by the instruction before needs two wait states: s_nop 1.)
      uint32_t lo = (uint32_t)__double_as_longlong(ld), hi = (uint32_t)((unsigned long long)__double_as_longlong(ld) >> 32);
      const unsigned long long cbits = (unsigned long long)__double_as_longlong(cd);
      uint32_t m0_saved;
      asm volatile(
          "s_nop 1\n\t"
          "v_cmp_ge_u32_e32 vcc, %4, %5\n\t"
          "s_mov_b32 %3, m0\n\t"
          "s_mov_b32 m0, %4\n\t"
          "v_cndmask_b32_dpp %0, %0, %0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_dpp %1, %1, %1, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_dpp %2, %2, %2, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_writelane_b32 %0, %6, m0\n\t"
          "v_writelane_b32 %1, %7, m0\n\t"
          "v_writelane_b32 %2, %8, m0\n\t"
          "s_mov_b32 m0, %3"
          : "+v"(lo), "+v"(hi), "+v"(li), "=&s"(m0_saved)
          : "s"(p), "v"(lane), "s"((uint32_t)cbits), "s"((uint32_t)(cbits >> 32)), "s"(ci)
          : "vcc");
      ld = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      moved = true;
    }
    if (moved) {  // the k-th entry is read back once per batch: inside the loop the list itself decides (p >= K)
      kth = readlane_f64(ld, K - 1);
      kth_id = (uint32_t)__builtin_amdgcn_readlane((int)li, K - 1);
    }
  };
#endif  // VGICP_KNN_LAZY
  if (has_home) {
    // the rest of home, then home as a whole is what the search skips; all K are now within home's diagonal
#ifndef VGICP_KNN_SEEDS_ONLY
    const uint32_t rest = (home_end - home_start) - skip_count;   // home's points outside the opening window
    for (uint32_t base = 0; base < rest; base += 64u) {
      const uint32_t t = base + (uint32_t)lane;
      const bool valid = t < rest;
      uint32_t j = home_start + t;
      if (j >= skip_start) j += skip_count;
      uint32_t id = 0xFFFFFFFFu;
      double d = INFINITY;
      if (valid) d = dist2(j, id);
      offer(valid, d, id, j);
    }
#else
    for (uint32_t base = home_start; base < home_end; base += 64u) {
      const uint32_t j = base + (uint32_t)lane;
      const bool valid = j < home_end;
      uint32_t id = 0xFFFFFFFFu;
      double d = INFINITY;
      if (valid) d = dist2(j, id);
      offer(valid, d, id, j);
    }
#endif
#ifdef VGICP_PREP_TRACE
    home_batches = batches;
#endif
    skip_start = home_start;
    skip_count = home_end - home_start;
    bound = fmin(bound, kth);
  }
  // Lower bound of the squared distance from the query to the cell (x, y, z) of level l, in single precision and
  // in units of the finest cell: the query is (cx + fx, ...) with the SAME quotient floor(q / fine) that binned the
  // points, so the integer part of every difference is exact; what single precision loses (2^-23 relative,
  // 2^-24 absolute per axis) is taken off eight times over (2^-20 relative and absolute, and 2^-20 relative on the
  // sum). The bound only orders and prunes cells; a bound that is low by a millionth costs nothing.
  auto cell_d2 = [&](int l, uint32_t x, uint32_t y, uint32_t z) {
    const float size = (float)(1u << l);
    constexpr float kOff = 0x1p-20f, kKeep = 1.0f - 0x1p-20f;
    const float ax = (float)(int)((x << l) - cx) - ffx, ay = (float)(int)((y << l) - cy) - ffy,
                az = (float)(int)((z << l) - cz) - ffz;
    const float ex = fmaxf(fmaf(fmaxf(fmaxf(ax, -ax - size), 0.0f), kKeep, -kOff), 0.0f);
    const float ey = fmaxf(fmaf(fmaxf(fmaxf(ay, -ay - size), 0.0f), kKeep, -kOff), 0.0f);
    const float ez = fmaxf(fmaf(fmaxf(fmaxf(az, -az - size), 0.0f), kKeep, -kOff), 0.0f);
    return fmaf(ex, ex, fmaf(ey, ey, ez * ez)) * fine2_low;
  };
  // where lane's child sits inside its parent: the lane number is the child's Morton digits (z y x per level)
  const uint32_t ox1 = (uint32_t)lane & 1u, oy1 = ((uint32_t)lane >> 1) & 1u, oz1 = ((uint32_t)lane >> 2) & 1u;
  const uint32_t ox2 = ox1 | (((uint32_t)lane >> 2) & 2u), oy2 = oy1 | (((uint32_t)lane >> 3) & 2u),
                 oz2 = oz1 | (((uint32_t)lane >> 4) & 2u);
  int waiting = 0;
  // drop the waiting cells that the k-th distance has overtaken since they were pushed
  auto compact = [&]() {
    const double limit = fmin(bound, kth);
    int kept = 0;
    for (int base = 0; base < waiting; base += 64) {
      const int i = base + lane;
      const bool valid = i < waiting;
      const float d = valid ? pd[i] : 0.0f;
      const unsigned long long key = valid ? pk[i] : 0ull;
      const uint32_t st = valid ? ps[i] : 0u, en = valid ? pe[i] : 0u;
      const bool stay = valid && (double)d <= limit;
      const unsigned long long who = __ballot(stay);
      const int at = kept + __builtin_popcountll(who & lanes_below);
      if (stay) { pd[at] = d; pk[at] = key; ps[at] = st; pe[at] = en; }  // at <= i: never ahead of the reads
      kept += __builtin_popcountll(who);
    }
    waiting = kept;
  };
  // the caller has made sure that the pool has room
  auto push = [&](bool want, float d2, unsigned long long key, uint32_t start, uint32_t end) {
    const unsigned long long who = __ballot(want);
    const int at = waiting + __builtin_popcountll(who & lanes_below);
    if (want) {
      pd[at] = d2;
      pk[at] = key;
      ps[at] = start;
      pe[at] = end;
    }
    waiting += __builtin_popcountll(who);
  };

  // the level whose 27-cell block covers the ball of the seed radius: everything that can be among the K
  // nearest lies inside it. Lane l works out level l: the distance from the query to the faces of that block
  int level;
  {
    const int l = lane < kLevels ? lane : kLevels - 1;
    const double sub = (double)(1u << l);
    const double px = (double)(cx & ((1u << l) - 1u)) + fx, py = (double)(cy & ((1u << l) - 1u)) + fy,
                 pz = (double)(cz & ((1u << l) - 1u)) + fz;
    const double margin = fmin(fmin(fmin(px, sub - px), fmin(py, sub - py)), fmin(pz, sub - pz));
    const double r = (sub + margin) * fine * (1.0 - 1e-12);
    const unsigned long long covers = __ballot((lane < kLevels) & (bound <= r * r));
    level = covers ? __builtin_ctzll(covers) : kLevels;
  }
  if (level < kLevels) {
    {
      const int top = kCoordMax >> level;
      const int x = (int)(cx >> level) + lane % 3 - 1, y = (int)(cy >> level) + (lane / 3) % 3 - 1,
                z = (int)(cz >> level) + lane / 9 - 1;
      bool want = lane < 27 && x >= 0 && y >= 0 && z >= 0 && x <= top && y <= top && z <= top;
      // Morton codes of the 27 cells without spreading any bits: the block's centre is the query's code shifted
      // (scalar), a step of +-1 along an axis is an increment / decrement of that axis' dilated digits (scalar
      // too), and every lane picks one of three per axis
      constexpr unsigned long long kMx = 0x1249249249249249ull, kMy = kMx << 1, kMz = kMx << 2;
      const unsigned long long centre = mq >> (3 * level);
      const unsigned long long bx = centre & kMx, by = centre & kMy, bz = centre & kMz;
      const unsigned long long xi = ((bx | ~kMx) + 1ull) & kMx, xd = (bx - 1ull) & kMx;
      const unsigned long long yi = ((by | ~kMy) + 1ull) & kMy, yd = (by - 1ull) & kMy;
      const unsigned long long zi = ((bz | ~kMz) + 1ull) & kMz, zd = (bz - 1ull) & kMz;
      const int sx = lane % 3, sy = (lane / 3) % 3, sz = lane / 9;
      const unsigned long long mk = ((sx == 0 ? xd : (sx == 1 ? bx : xi)) | (sy == 0 ? yd : (sy == 1 ? by : yi)) |
                                     (sz == 0 ? zd : (sz == 1 ? bz : zi))) & kKeyMask;
      float d2 = 0.0f;
      uint32_t start = 0, end = 0;
      if (want) {
        const CellEntry* e = find_cell(table, mask, cell_key(mk, level));
        want = e != nullptr;
        if (e) {
          start = e->start; end = e->end;
          d2 = cell_d2(level, (uint32_t)x, (uint32_t)y, (uint32_t)z);
          // (a cell whose whole run has been measured already has nothing to add)
          want = (d2 <= __double2float_ru(bound)) & !((start >= skip_start) & (end <= skip_start + skip_count));
        }
      }
      push(want, d2, cell_key(mk, level), start, end);
    }
    while (waiting > 0) {
      wave_sync();  // what the lanes pushed, moved or compacted in the turn before
      // the nearest waiting cell
      float mine = INFINITY;
      int at = 0;
      for (int i = lane; i < waiting; i += 64) {
        const float d = pd[i];
        if (d < mine) { mine = d; at = i; }
      }
      // the distances are non-negative floats: their bit patterns order as unsigned integers. Minimum of every
      // row of 16 lanes with four DPP rotations, then of the four rows on the scalar unit: no LDS-crossbar shuffles
      uint32_t mbits = __builtin_bit_cast(uint32_t, mine);
      asm volatile(  // a DPP operand written by the instruction before needs two wait states
          "s_nop 1\n\t"
          "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0"
          : "+v"(mbits));
      const uint32_t least_bits =
          min(min((uint32_t)__builtin_amdgcn_readlane((int)mbits, 0), (uint32_t)__builtin_amdgcn_readlane((int)mbits, 16)),
              min((uint32_t)__builtin_amdgcn_readlane((int)mbits, 32), (uint32_t)__builtin_amdgcn_readlane((int)mbits, 48)));
      const float least = __builtin_bit_cast(float, least_bits);
      const int owner = __builtin_ctzll(__ballot(mine == least));   // the lowest lane holding the minimum
      at = __builtin_amdgcn_readlane(at, owner);
      const double best = (double)least;
      const double limit = fmin(bound, kth);
      if (best > limit) break;  // nothing left can hold one of the K nearest
      ++pops;
      // every lane reads the same entry; telling the compiler so (readfirstlane) keeps the list length, the k-th
      // distance and all the loop control below in scalar registers instead of under exec masks
      const unsigned long long key = uniform_u64(pk[at]);
      const uint32_t start = uniform_u32(ps[at]), end = uniform_u32(pe[at]);
      --waiting;
      if (at != waiting) {  // every lane moves the same entry: a benign same-value write
        pd[at] = pd[waiting];
        pk[at] = pk[waiting];
        ps[at] = ps[waiting];
        pe[at] = pe[waiting];
      }
      wave_sync();
      const int l = (int)(key >> 60);
      const unsigned long long mk = key & kKeyMask;
            bool measure = l == 0 || end - start <= kLeafPoints;
      if (!measure) {  // one lane per cell two levels down (one level above the finest)
#ifdef VGICP_PREP_TRACE
        ++opens;
#endif
        const int step = l >= 2 ? 2 : 1;
        const int fan = 1 << (3 * step);
        const unsigned long long cm = (mk << (3 * step)) | (unsigned long long)lane;
        // the parent's coordinates come out of its (wave-uniform) key on the scalar unit
        const uint32_t px = compact21(mk), py = compact21(mk >> 1), pz = compact21(mk >> 2);
        const uint32_t chx = (px << step) | (step == 2 ? ox2 : ox1), chy = (py << step) | (step == 2 ? oy2 : oy1),
                       chz = (pz << step) | (step == 2 ? oz2 : oz1);
        bool want = lane < fan;
        float d2 = 0.0f;
        const float limit_up = __double2float_ru(limit);
        uint32_t cs = 0, ce = 0;
        if (want) {
          const CellEntry* e = find_cell(table, mask, cell_key(cm, l - step));
          want = e != nullptr;
          if (e) {
            cs = e->start;
            ce = e->end;
            d2 = cell_d2(l - step, chx, chy, chz);
            want = (d2 <= limit_up) & !((cs >= skip_start) & (ce <= skip_start + skip_count));
          }
        }
        const int incoming = __builtin_popcountll(__ballot(want));
        if (waiting + incoming > kPool) compact();
        if (waiting + incoming > kPool) {  // no room even so: measure this cell's points instead (still exact)
          measure = true;
          ++spills;
        } else {
          push(want, d2, cell_key(cm, l - step), cs, ce);
        }
      }
      if (measure) {
        for (uint32_t base = start; base < end; base += 64u) {
          const uint32_t j = base + (uint32_t)lane;
          const bool valid = j < end;
          uint32_t id = 0xFFFFFFFFu;
          double d = INFINITY;
          if (valid) d = dist2(j, id);
          offer(valid, d, id, j);
        }
      }
    }
  } else {  // sparser than the coarsest level resolves: every point of the scan
    for (uint32_t base = 0; base < n; base += 64u) {
      const uint32_t j = base + (uint32_t)lane;
      const bool valid = j < n;
      uint32_t id = 0xFFFFFFFFu;
      double d = INFINITY;
      if (valid) d = dist2(j, id);
      offer(valid, d, id, j);
    }
    spills += 1000000u;
  }
  if (lane == 0 && spills && debug) atomicAdd(&counters[2], 1u);
  const uint32_t qi = uniform_u32(sorted_idx[qj]);
  const uint32_t o = uniform_u32(slot_of_index[qi]);
#ifdef VGICP_KNN_LAZY
  {
    // the order, once: rank of every survivor by (distance, index) among the survivors (all distinct by index)
    wave_sync();
    if ((uint32_t)lane < cnt) { ld = cdv[lane]; li = civ[lane]; }
    uint32_t rank = 0;
    const unsigned long long key = (unsigned long long)__double_as_longlong(ld);
    for (uint32_t e = 0; e < cnt; ++e) {
      const unsigned long long ke = (unsigned long long)__double_as_longlong(cdv[e]);
      const uint32_t ie = civ[e];
      rank += ((ke < key) | ((ke == key) & (ie < li))) ? 1u : 0u;
    }
    if ((uint32_t)lane < cnt && rank < (uint32_t)K) nbr[(size_t)o * kMaxKnn + rank] = li;
  }
#else
  if (lane < kMaxKnn) nbr[(size_t)o * kMaxKnn + lane] = li;  // original indices: the list carries nothing else
#endif
  if (lane == 0) {
    out_pts[3 * (size_t)o] = qx; out_pts[3 * (size_t)o + 1] = qy; out_pts[3 * (size_t)o + 2] = qz;
    if (soa) { soa[o] = qx; soa[soa_stride + o] = qy; soa[2 * soa_stride + o] = qz; }  // the planes the registration reads
    out_idx[o] = qi;
#ifdef VGICP_PREP_TRACE  // developer build only (tools/ab_build.sh trace -DVGICP_PREP_TRACE): the index output carries a trace record
    {
      // the 10-bit field: cells taken (1), insertions (2), points in the query's own level-5 / level-4 cell (4 / 6), the
      // finest level whose own cell holds K points (5)
      const uint32_t trace_field = VGICP_PREP_TRACE == 2 ? inserts : VGICP_PREP_TRACE == 7 ? batches : VGICP_PREP_TRACE == 8 ? opens
                                 : VGICP_PREP_TRACE == 9 ? home_batches
                                 : VGICP_PREP_TRACE == 4 ? (uint32_t)__builtin_amdgcn_readlane((int)(own_end - own_start), 5)
                                 : VGICP_PREP_TRACE == 6 ? (uint32_t)__builtin_amdgcn_readlane((int)(own_end - own_start), 4)
                                 : VGICP_PREP_TRACE == 5 ? (enough ? (uint32_t)home_level : 15u) : pops;
      const uint64_t t_end = wall_clock64();
      uint64_t dt = t_end - t_trace; if (dt > 0xFFFFFu) dt = 0xFFFFFu;
      out_idx[o] = (dt << 44) | ((uint64_t)(trace_field > 1023u ? 1023u : trace_field) << 34) | ((uint64_t)((VGICP_PREP_TRACE == 3 ? (int)(blockIdx.x % 8u) : level) & 15) << 30) | ((t_trace / 10u) & 0x3FFFFFFFu);
    }
#endif
    if (debug) {
      atomicAdd(&counters[3], batches);
      atomicMax(&counters[4], batches);
      atomicAdd(&counters[5], pops);
      atomicMax(&counters[6], pops);
      if (level > kFineShift) atomicAdd(&counters[7], 1u);
      if (debug >= 2) {  // developer histograms: cells taken, and wall time of this query (100 MHz clock)
        atomicAdd(&counters[8 + (pops / 8u < 31u ? pops / 8u : 31u)], 1u);
        const uint32_t us8 = (uint32_t)((wall_clock64() - t_begin) / 800u);
        atomicAdd(&counters[40 + (us8 < 31u ? us8 : 31u)], 1u);
      }
    }
  }
}

// The prepared scan out to the host without a copy command and without a synchronisation: see launch_fetch
// (vgicp_device.h).  A copy command costs ~20 - 40 us apiece on this platform and the host can neither start its own copy
// before the command has completed nor learn the size without a round trip; a kernel's posted writes run at the link's
// rate and the host reads every piece the moment its flag arrives.
constexpr int kFetchBlock = 256;
// sums (device, 64 words + a ticket word, zero between launches): [buffer 0 = points, 1 = covariances][A = 0 / B = 1][16 lanes].
// The 8-byte words of each array are dealt to 16 lanes by their index (lane = index mod 16, k = index / 16): A_lane = sum of
// the lane's words, B_lane = sum of (m_lane - k) x word with m_lane = the lane's word count, both mod 2^64 — what a host
// that runs  s1 += w; s2 += s1  over the lane's words ends with, up to its start values (the drop-in's change detector,
// include/eskf_lio_shim/LocalMap.hpp: bufferHash).  The last block to finish posts the 64 sums to the host and zeroes them.
__global__ __launch_bounds__(kFetchBlock) void fetch_kernel(const double* __restrict__ aos_pts, const double* __restrict__ aos_cov,
                                                            const uint32_t* __restrict__ counters, uint32_t epoch, uint32_t n_cap,
                                                            char* __restrict__ stage, uint32_t* flags, unsigned long long* hdr_done,
                                                            uint32_t seq, uint32_t piece_bytes, unsigned long long* sums,
                                                            unsigned long long* host_sums) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  __shared__ unsigned long long part[kFetchBlock / 64][64];
  const bool refused = counters[kBeyondGrid] == epoch || counters[kScanTimeout] == epoch;
  const uint32_t kept = refused ? 0u : (counters[0] < n_cap ? counters[0] : n_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(hdr_done, ((unsigned long long)seq << 32) | (refused ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const size_t pb = (size_t)kept * 24u, pb_pad = (pb + 255u) & ~size_t(255), cb = (size_t)kept * 72u, total = pb_pad + cb;
  const uint32_t pieces = (uint32_t)((total + piece_bytes - 1) / piece_bytes);
  const char* sp = reinterpret_cast<const char*>(aos_pts);
  const char* sc = reinterpret_cast<const char*>(aos_cov);
  // this thread's chunks are 16 bytes at multiples of 16 x 256 from 16 x threadIdx.x, both parts start on a multiple of
  // 256 bytes: its two words always fall into lanes 2 (t & 7) and 2 (t & 7) + 1 of either array
  unsigned long long a0[2] = {0, 0}, a1[2] = {0, 0}, b0[2] = {0, 0}, b1[2] = {0, 0};   // [array]: A / B of the even and the odd lane
  const unsigned long long words_p = pb / 8u, words_c = cb / 8u;
  const uint32_t lane_even = 2u * (threadIdx.x & 7u);
  for (uint32_t piece = blockIdx.x; piece < pieces; piece += gridDim.x) {
    const size_t off = (size_t)piece * piece_bytes;
    const size_t len = total - off < piece_bytes ? total - off : piece_bytes;
    for (size_t k = (size_t)threadIdx.x * 16u; k < len; k += (size_t)kFetchBlock * 16u) {
      const size_t pos = off + k;   // both parts start on a multiple of 16 bytes; a chunk never straddles them
      const bool in_points = pos < pb_pad;
      const v4i w = in_points ? *reinterpret_cast<const v4i*>(sp + pos) : *reinterpret_cast<const v4i*>(sc + (pos - pb_pad));
      *reinterpret_cast<v4i*>(stage + pos) = w;
      // the sums: word index inside its array, the lane's word count, the weight m - k
      const int arr = in_points ? 0 : 1;
      const unsigned long long words = in_points ? words_p : words_c;
      const unsigned long long i0 = (in_points ? pos : pos - pb_pad) / 8u;
      const unsigned long long w0 = ((unsigned long long)(uint32_t)w.y << 32) | (uint32_t)w.x;
      const unsigned long long w1 = ((unsigned long long)(uint32_t)w.w << 32) | (uint32_t)w.z;
      if (i0 < words) {
        const unsigned long long m = (words - lane_even + 15u) >> 4;
        a0[arr] += w0;
        b0[arr] += (m - (i0 >> 4)) * w0;
      }
      if (i0 + 1u < words) {
        const unsigned long long m = (words - (lane_even + 1u) + 15u) >> 4;
        a1[arr] += w1;
        b1[arr] += (m - ((i0 + 1u) >> 4)) * w1;
      }
    }
    __threadfence_system();   // every thread: its stores have reached the host before the flag may
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + 16u * piece, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (sums == nullptr) return;
  // fold the eight values a thread holds over the threads that share its lanes (t & 7 equal), then one atomic per sum
  // and block; the block whose ticket is the last one posts the totals
  unsigned long long v[8] = {a0[0], b0[0], a1[0], b1[0], a0[1], b0[1], a1[1], b1[1]};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned long long x = v[q];
    x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);   // lanes with the same t & 7 inside the wave
    v[q] = x;
  }
  const uint32_t wave = threadIdx.x >> 6, wl = threadIdx.x & 63u;
  if (wl < 8u) {
#pragma unroll
    for (int q = 0; q < 8; ++q) part[wave][8 * q + wl] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 64u) {
    unsigned long long x = 0;
    for (uint32_t wv = 0; wv < kFetchBlock / 64; ++wv) x += part[wv][threadIdx.x];
    // part index 8 q + r: q = (array, odd lane?, B?) as in v[], r = t & 7  ->  sums[array][A/B][lane]
    const uint32_t q = threadIdx.x >> 3, r = threadIdx.x & 7u;
    const uint32_t arr = q >> 2, odd = (q >> 1) & 1u, isb = q & 1u;
    if (x) atomicAdd(&sums[arr * 32u + isb * 16u + 2u * r + odd], x);
  }
  __threadfence();
  __syncthreads();
  __shared__ uint32_t last_sh;
  if (threadIdx.x == 0) last_sh = atomicAdd(reinterpret_cast<unsigned int*>(sums + 64), 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (last_sh && threadIdx.x < 64u) {
    const unsigned long long x = __hip_atomic_load(&sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(host_sums + threadIdx.x, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    sums[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) *reinterpret_cast<unsigned int*>(sums + 64) = 0u;
  }
}
// Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU | ComputeFullV) in its published operation order (Eigen 3.4
// src/SVD/JacobiSVD.h compute() / real_2x2_jacobi_svd(), src/Jacobi/Jacobi.h makeJacobi(), rotation product and
// apply_rotation_in_the_plane()): what src/CloudPreprocessor.cpp:119-123 runs on every covariance.  Two-sided
// Jacobi on A / max|A| over (p, q) = (1,0), (2,0), (2,1) until all off-diagonal pairs are below 2 eps max|diag|;
// negative diagonal entries are folded into U, then values and columns are sorted descending (first maximum
// wins).  For a symmetric input that leaves U.col(k) = sign(eigenvalue_k) V.col(k): a cumulant covariance with a
// rounding-level NEGATIVE smallest eigenvalue (exact planes, lines, repeated points far from the origin) comes
// back indefinite, as from the reference.  This file is compiled without contraction, and the tests' CPU checker
// restates the same operations in the same order, so the two agree bit for bit.
struct Rot { double c, s; };
__device__ __forceinline__ double max_first(double a, double b) { return a < b ? b : a; }  // std::max
__device__ __forceinline__ Rot make_jacobi(double x, double y, double z) {
  const double deno = 2.0 * fabs(y);
  if (deno < 2.2250738585072014e-308) return {1.0, 0.0};
  const double tau = (x - z) / deno;
  const double w = sqrt(tau * tau + 1.0);
  const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
  const double sign_t = t > 0.0 ? 1.0 : -1.0;
  const double n = 1.0 / sqrt(t * t + 1.0);
  return {n, -sign_t * (y / fabs(y)) * fabs(t) * n};
}
__device__ __forceinline__ void rotate_pair(double& x, double& y, const Rot& j) {
  const double xi = x, yi = y;
  x = j.c * xi + j.s * yi;
  y = -j.s * xi + j.c * yi;
}
template <int P, int Q>
__device__ __forceinline__ bool jacobi_svd_step(double (&W)[3][3], double (&U)[3][3], double (&V)[3][3], double& max_diag) {
  const double threshold = max_first(2.2250738585072014e-308, 4.440892098500626e-16 * max_diag);
  if (!(fabs(W[P][Q]) > threshold || fabs(W[Q][P]) > threshold)) return false;
  double m00 = W[P][P], m01 = W[P][Q], m10 = W[Q][P], m11 = W[Q][Q];
  Rot rot1;
  const double t = m00 + m11, d = m10 - m01;
  if (fabs(d) < 2.2250738585072014e-308) {
    rot1 = {1.0, 0.0};
  } else {
    const double u = t / d;
    const double tmp = sqrt(1.0 + u * u);
    rot1 = {u / tmp, 1.0 / tmp};
  }
  if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
    rotate_pair(m00, m10, rot1);
    rotate_pair(m01, m11, rot1);
  }
  const Rot jr = make_jacobi(m00, m01, m11);
  const Rot jrt = {jr.c, -jr.s};
  const Rot jl = {rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
  if (!(jl.c == 1.0 && jl.s == 0.0)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rotate_pair(W[P][k], W[Q][k], jl);
#pragma unroll
    for (int k = 0; k < 3; ++k) rotate_pair(U[k][P], U[k][Q], jl);
  }
  if (!(jrt.c == 1.0 && jrt.s == 0.0)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rotate_pair(W[k][P], W[k][Q], jrt);
#pragma unroll
    for (int k = 0; k < 3; ++k) rotate_pair(V[k][P], V[k][Q], jrt);
  }
  max_diag = max_first(max_diag, max_first(fabs(W[P][P]), fabs(W[Q][Q])));
  return true;
}
// -> false for a non-finite input (U, V unset). `opposed` = columns with U.col(k) . V.col(k) < 0.
__device__ __forceinline__ bool jacobi_svd3(const double (&A)[3][3], double (&U)[3][3], double (&V)[3][3], int& opposed) {
  double scale = 0.0;
  bool finite = true;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v = fabs(A[r][c]);
      if (!(v - v == 0.0)) finite = false;
      if (v > scale) scale = v;
    }
  opposed = 0;
  if (!finite) return false;
  if (scale == 0.0) scale = 1.0;
  double W[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      W[r][c] = A[r][c] / scale;
      U[r][c] = V[r][c] = r == c ? 1.0 : 0.0;
    }
  double max_diag = max_first(fabs(W[0][0]), max_first(fabs(W[1][1]), fabs(W[2][2])));
  bool finished = false;
  while (!finished) {
    finished = true;
    if (jacobi_svd_step<1, 0>(W, U, V, max_diag)) finished = false;
    if (jacobi_svd_step<2, 0>(W, U, V, max_diag)) finished = false;
    if (jacobi_svd_step<2, 1>(W, U, V, max_diag)) finished = false;
  }
  double sv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a = W[i][i];
    sv[i] = fabs(a);
    if (a < 0.0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) U[k][i] = -U[k][i];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  // selection sort, descending, first maximum of the tail; stops at an all-zero tail (static indices)
#define VG_SWAP_COLS(a, b)                                                                     \
  {                                                                                            \
    const double tw = sv[a]; sv[a] = sv[b]; sv[b] = tw;                                        \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                            \
      const double tu = U[k][a]; U[k][a] = U[k][b]; U[k][b] = tu;                              \
      const double tv = V[k][a]; V[k][a] = V[k][b]; V[k][b] = tv;                              \
    }                                                                                          \
  }
  {
    const int pos = sv[2] > (sv[1] > sv[0] ? sv[1] : sv[0]) ? 2 : (sv[1] > sv[0] ? 1 : 0);
    const bool stop = sv[pos] == 0.0;
    if (!stop) {
      if (pos == 1) VG_SWAP_COLS(0, 1)
      if (pos == 2) VG_SWAP_COLS(0, 2)
      if (sv[2] > sv[1]) VG_SWAP_COLS(1, 2)   // (an all-zero tail ends Eigen's loop: nothing to swap then either)
    }
  }
#undef VG_SWAP_COLS
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (U[0][i] * V[0][i] + U[1][i] * V[1][i] + U[2][i] * V[2][i] < 0.0) ++opposed;
  return true;
}

// Covariance of the neighbours + regularisation, one thread per kept point.
__global__ __launch_bounds__(kCovBlock) void cov_kernel(const double* __restrict__ pts,
                                                        const uint32_t* __restrict__ nbr, int found,
                                                        double* __restrict__ out_covs, double* __restrict__ soa,
                                                        uint64_t soa_stride, uint32_t* __restrict__ counters, uint32_t epoch) {
  const uint32_t o = blockIdx.x * kCovBlock + threadIdx.x;
  if (o >= counters[0] || counters[kBeyondGrid] == epoch || counters[kScanTimeout] == epoch) return;
  uint32_t* indefinite = counters + kIndefiniteCounter;
  double cov[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (found >= 3) {
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // sixteen neighbours at a time: their indices, then their points, are requested together (one thread per point
    // is less than a wave per SIMD -- nothing else hides a load, and one neighbour after the other was 60
    // dependent round trips; 8 / 16 / 32 at a time: 19.6 / 18.9 / 20.2 us per 60 000-point sweep); the sums stay in
    // ascending-distance order.  (The rest of the kernel is the Jacobi sweeps' chain of fp64 divisions and roots:
    // half-filled waves -- twice as many -- take 26 us, not less.)
    constexpr int kChunk = 16;
    const uint4* row = reinterpret_cast<const uint4*>(nbr + (size_t)o * kMaxKnn);
    for (int k0 = 0; k0 < found; k0 += kChunk) {
      uint32_t j[kChunk];
#pragma unroll
      for (int q = 0; q < kChunk / 4; ++q) {
        const uint4 jq = k0 + 4 * q < kMaxKnn ? row[k0 / 4 + q] : make_uint4(0u, 0u, 0u, 0u);
        j[4 * q] = jq.x; j[4 * q + 1] = jq.y; j[4 * q + 2] = jq.z; j[4 * q + 3] = jq.w;
      }
      double px[kChunk], py[kChunk], pz[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const double* rec = pts + 3 * (size_t)(k0 + u < found ? j[u] : 0u);  // the scan as it came: original indices
        px[u] = rec[0];
        py[u] = rec[1];
        pz[u] = rec[2];
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (k0 + u < found) {
          const double x = px[u], y = py[u], z = pz[u];
          c[0] += x; c[1] += y; c[2] += z;
          c[3] += x * x; c[4] += x * y; c[5] += x * z;
          c[6] += y * y; c[7] += y * z; c[8] += z * z;
        }
      }
    }
    const double inv = (double)found;
#pragma unroll
    for (int k = 0; k < 9; ++k) c[k] /= inv;
    cov[0][0] = c[3] - c[0] * c[0];
    cov[1][1] = c[6] - c[1] * c[1];
    cov[2][2] = c[8] - c[2] * c[2];
    cov[0][1] = cov[1][0] = c[4] - c[0] * c[1];
    cov[0][2] = cov[2][0] = c[5] - c[0] * c[2];
    cov[1][2] = cov[2][1] = c[7] - c[1] * c[2];
  }
  // svd.matrixU() * diag(1, 1, 1e-2) * svd.matrixV()^T (src/CloudPreprocessor.cpp:119-123)
  double U[3][3], V[3][3];
  int opposed = 0;
  const bool ok = jacobi_svd3(cov, U, V, opposed);
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double v = ok ? U[r][0] * V[cc][0] + U[r][1] * V[cc][1] + (U[r][2] * 1e-2) * V[cc][2] : __builtin_nan("");
      out_covs[9 * (size_t)o + r + 3 * cc] = v;
      if (soa) soa[(size_t)(3 + r + 3 * cc) * soa_stride + o] = v;  // plane order x y z c00 c10 c20 c01 ...
    }
  if (opposed > 0) atomicAdd(indefinite, 1u);   // rare: a negative eigenvalue, the matrix written is indefinite
}

// ---- CloudPreprocessor::deskew (src/CloudPreprocessor.cpp:25-74) -----------------------------------
// The reference walks the IMU states in order and, for each, advances over the points taken before the
// state's timestamp, starting where the previous state stopped; a state whose search reaches the end of
// the scan moves nothing. ends[s] is where state s stopped (one workgroup, states in sequence, the search
// itself parallel), so the segments are the reference's even when the point times are not monotonic.
constexpr int kBoundsBlock = 1024;
__global__ __launch_bounds__(kBoundsBlock) void deskew_bounds_kernel(const double* __restrict__ point_time,
                                                                     uint32_t n,
                                                                     const double* __restrict__ state_time,
                                                                     uint32_t states, uint32_t* __restrict__ ends) {
  __shared__ uint32_t first_hit;
  uint32_t start = 0;
  for (uint32_t s = 0; s < states; ++s) {
    if (threadIdx.x == 0) first_hit = 0xFFFFFFFFu;
    __syncthreads();
    const double ts = state_time[s];
    for (uint32_t i = start + threadIdx.x; i < n; i += kBoundsBlock) {
      if (!(point_time[i] < ts)) {  // this thread's first point at or after the state's time
        atomicMin(&first_hit, i);
        break;
      }
      if (i > *(volatile uint32_t*)&first_hit) break;  // someone found an earlier one already
    }
    __syncthreads();
    const uint32_t hit = first_hit;
    if (hit != 0xFFFFFFFFu) start = hit;  // else: the search ran to the end and the reference keeps its old bound
    if (threadIdx.x == 0) ends[s] = start;
    __syncthreads();
  }
}

// The same bounds without the walk, for state times that are finite and non-decreasing (the host checks; any IMU
// queue is): "hit" = !(point_time < state_time) is then nested -- a point that is a hit for state s is one for
// every earlier state -- and by induction the walk's result for s is simply the FIRST hit of s in the whole scan
// (it cannot lie before where s-1 stopped), or, once a state has no hit at all, the last bound found. Every point
// counts the states it is a hit for (binary search in LDS); a thread that sees its count rise at point j lowers
// the first-hit of the states in between to j (LDS atomicMin); blocks cover contiguous parts of the scan and a
// second small kernel takes the minimum over the blocks. 163 us -> two launches of a few us per 60k-point sweep.
constexpr uint32_t kDeskewParts = 512;      // at most this many blocks share the scan (one load per thread: the loop is a chain)
constexpr uint32_t kProloguePartsMax = 64;  // ... and this many when every workgroup of the prologue merges them for itself
constexpr int kFirstHitBlock = 1024;   // a block covers its ~1 000 points in ONE pass: the loop over passes is a chain of trips to memory (256 threads: 9.6 us per 60k-point sweep)
__global__ __launch_bounds__(kFirstHitBlock) void deskew_first_hit_kernel(const double* __restrict__ point_time, uint32_t n,
                                                               const double* __restrict__ state_time, uint32_t states,
                                                               uint32_t per_block, uint32_t* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char deskew_lds[];
  double* ts = reinterpret_cast<double*>(deskew_lds);
  uint32_t* first = reinterpret_cast<uint32_t*>(deskew_lds + (size_t)states * sizeof(double));
  for (uint32_t s = threadIdx.x; s < states; s += blockDim.x) { ts[s] = state_time[s]; first[s] = 0xFFFFFFFFu; }
  __syncthreads();
  const uint32_t lo = blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t seen = 0;  // wave-uniform: states whose first hit this wave has already lowered (its later points come later)
  for (uint32_t base = lo + (threadIdx.x & ~63u); base < hi; base += blockDim.x) {  // whole waves stay together
    const uint32_t j = base + lane;
    uint32_t a = 0;
    if (j < hi) {
      const double t = point_time[j];
      // hits = number of states with !(t < ts[s]); NaN is a hit for all of them, like in the walk
      uint32_t b = states;
      while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (!(t < ts[mid])) a = mid + 1; else b = mid;
      }
    }
    uint32_t most = a;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)most, o, 64); most = v > most ? v : most; }
    // for every state not lowered yet: the first lane (= lowest j of this wave) that hits it; one lane stores
    for (; seen < most; ++seen) {
      const unsigned long long hit = __ballot(a > seen);
      const uint32_t jmin = base + (uint32_t)__builtin_ctzll(hit);
      if (lane == 0 && first[seen] > jmin) atomicMin(&first[seen], jmin);
    }
  }
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < states; s += blockDim.x) part[(size_t)blockIdx.x * states + s] = first[s];
}
__global__ __launch_bounds__(1024) void deskew_merge_kernel(const uint32_t* __restrict__ part, uint32_t parts,
                                                            uint32_t states, uint32_t* __restrict__ ends) {
  extern __shared__ __attribute__((aligned(16))) unsigned char deskew_lds[];
  uint32_t* least = reinterpret_cast<uint32_t*>(deskew_lds);
  __shared__ uint32_t last_found;
  if (threadIdx.x == 0) last_found = 0u;
  for (uint32_t s = threadIdx.x; s < states; s += blockDim.x) least[s] = 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < parts * states; k += blockDim.x) {  // all loads independent
    const uint32_t v = part[k];
    if (v != 0xFFFFFFFFu) atomicMin(&least[k % states], v);
  }
  __syncthreads();
  // first hits are non-decreasing in s and the states without one form a tail: they keep the last bound found
  for (uint32_t s = threadIdx.x; s < states; s += blockDim.x)
    if (least[s] != 0xFFFFFFFFu) atomicMax(&last_found, least[s]);
  __syncthreads();
  const uint32_t keep = last_found;
  for (uint32_t s = threadIdx.x; s < states; s += blockDim.x) ends[s] = least[s] != 0xFFFFFFFFu ? least[s] : keep;
}

// p <- T_s p for the state s whose segment holds the point; poses: 12 doubles per state, R column-major
// then t. Evaluated in the order of Eigen's Isometry3d * Vector3d (this file has no FMA contraction).
__global__ void deskew_apply_kernel(double* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ ends,
                                    uint32_t states, const double* __restrict__ poses) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = states;  // first s with i < ends[s]
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (i < ends[mid]) hi = mid; else lo = mid + 1;
  }
  if (lo >= states) return;  // after the last segment: left as it is
  const double* T = poses + 12 * (size_t)lo;
  const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  const double rx = T[0] * x + T[3] * y + T[6] * z;
  const double ry = T[1] * x + T[4] * y + T[7] * z;
  const double rz = T[2] * x + T[5] * y + T[8] * z;
  pts[3 * (size_t)i] = rx + T[9];
  pts[3 * (size_t)i + 1] = ry + T[10];
  pts[3 * (size_t)i + 2] = rz + T[11];
}

inline uint32_t blocks_for(uint64_t work, uint32_t block) { return (uint32_t)((work + block - 1) / block); }
__host__ inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
__host__ inline uint64_t pow2_at_least(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

// The one sort: (Morton code, index) pairs, stable — vgicp_sort.h (every wave sorts 256 pairs in registers, then groups
// of four runs are merged per launch: five launches, 34 us for a sweep of 60 000 points).  Until round 6 this was
// rocPRIM's merge sort, one launch per doubling of the run length (42.5 us there; launches of one preparation's sort,
// us, profiles/r10_knn_leaf.txt):
//   points           28k   33k   60k   65k   66k  100k  130k  150k  250k  300k
//   2 048 per block  35.5  50.8  50.9  51.3  48.7  50.1  52.7  70.8 108.1 115.8
//   4 096 per block  43.8  42.3  43.7  43.1  59.1  60.2  62.0  60.5  93.9 119.3
// -DVGICP_SORT_ROCPRIM builds that one again (A/B measurements only).
#ifdef VGICP_SORT_ROCPRIM
inline bool sort_in_large_blocks(uint32_t n) { return (n > 32768u && n <= 65536u) || (n > 131072u && n <= 262144u); }
inline hipError_t sort_codes_rocprim(void* temp, size_t& temp_bytes, const unsigned long long* codes_in,
                                     unsigned long long* codes_out, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n,
                                     hipStream_t s) {
  if (sort_in_large_blocks(n)) {
    using Large = rocprim::merge_sort_config<512, 1024, 4>;
    return rocprim::merge_sort<Large>(temp, temp_bytes, codes_in, codes_out, idx_in, idx_out, (size_t)n,
                                      rocprim::less<unsigned long long>(), s);
  }
  using Config = rocprim::merge_sort_config<512, 512, 4>;
  return rocprim::merge_sort<Config>(temp, temp_bytes, codes_in, codes_out, idx_in, idx_out, (size_t)n,
                                     rocprim::less<unsigned long long>(), s);
}
#endif

struct Layout {
  size_t codes_in, codes_out, idx_in, idx_out, spts, keep_i, rank_i, queries, nbr, cub, total;
  size_t cub_bytes;
};

__host__ inline Layout layout_for(uint32_t n) {
  Layout L;
  size_t sort_pairs = 0;
#ifdef VGICP_SORT_ROCPRIM
  (void)sort_codes_rocprim(nullptr, sort_pairs, nullptr, nullptr, nullptr, nullptr, n, nullptr);
#else
  sort_pairs = sortk::split_bytes(n, sizeof(unsigned long long));   // the runs' splitters; the pairs alternate between the in and out buffers
#endif
  L.cub_bytes = sort_pairs;
  size_t off = 0;
  L.codes_in = off; off += align256((size_t)n * 8);
  L.codes_out = off; off += align256((size_t)n * 8);
  L.idx_in = off; off += align256((size_t)n * 4);
  L.idx_out = off; off += align256((size_t)n * 4);
  L.spts = off; off += align256((size_t)n * 32);
  L.keep_i = off; off += align256((size_t)n * 4);
  L.rank_i = off; off += align256((size_t)n * 4);
  L.queries = off; off += align256((size_t)n * 4);
  L.nbr = off; off += align256((size_t)n * kMaxKnn * 4);
  L.cub = off; off += align256(L.cub_bytes);
  L.total = off + 256;
  return L;
}

}  // namespace

thread_local uint64_t g_kernel_launches = 0;

size_t preprocess_scratch_bytes(uint32_t n) { return layout_for(n ? n : 1).total; }
int preprocess_max_knn() { return kMaxKnn; }
// Load factor 1/8 .. 1/16: most of the search's lookups are for children that do not exist and end at the first
// empty slot; with a fuller table the longest probe sequence among 64 lanes sets the pace (x2: 411 us, x4: 350,
// x8: 337 for the search alone, and the kernels that build the table gain as well).
uint64_t preprocess_cell_entries(uint32_t cells) { return pow2_at_least((uint64_t)cells * 8 + 64); }
// Sized from the number of POINTS, so that no host round trip is needed for the cell count: a point opens at most
// kLevels = 12 cells, a real sweep 2.5 - 3 per point (260 654 cells for 100 000 points), so 32 entries per point
// keep the load at 1/10 - 1/13 there (the same table sizes the measured cell count gave) and below 3/8 whatever
// the scan looks like.
uint64_t preprocess_cell_entries_for(uint32_t n) { return pow2_at_least((uint64_t)(n ? n : 1) * 32 + 64); }
size_t preprocess_cell_bytes(uint64_t entries) { return entries * sizeof(CellEntry); }
size_t preprocess_tile_bytes() {   // tile slots of the two scans + the prologue's look-back slots
  return (size_t)kMaxScanTiles * (sizeof(TileSlot) + sizeof(unsigned long long)) + (size_t)kFusedBoundsBlocksMax * sizeof(unsigned long long);
}

namespace {
// launches of the sort
uint32_t merge_sort_launches(uint32_t n) {
#ifdef VGICP_SORT_ROCPRIM
  uint32_t launches = 1;
  for (uint64_t run = sort_in_large_blocks(n) ? 4096 : 2048; run < n; run <<= 1) ++launches;
  return launches;
#else
  return sortk::launches_for(n);
#endif
}
}  // namespace

hipError_t launch_fetch(hipStream_t s, const double* aos_pts, const double* aos_cov, const uint32_t* counters, uint32_t epoch,
                        uint32_t n_cap, char* stage, uint32_t* flags, unsigned long long* hdr_done, uint32_t seq,
                        uint32_t piece_bytes, unsigned long long* sums, unsigned long long* host_sums) {
  const size_t worst = (((size_t)n_cap * 24u + 255u) & ~size_t(255)) + (size_t)n_cap * 72u;
  const uint32_t grid = (uint32_t)std::min<size_t>(std::max<size_t>((worst + piece_bytes - 1) / piece_bytes, 1), 48);
  ++g_kernel_launches;
  hipLaunchKernelGGL(fetch_kernel, dim3(grid), dim3(kFetchBlock), 0, s, aos_pts, aos_cov, counters, epoch, n_cap, stage, flags,
                     hdr_done, seq, piece_bytes, sums, host_sums);
  return hipGetLastError();
}

hipError_t launch_prepare(hipStream_t s, const PrepareArgs& a) {
  const hipError_t e = launch_prepare_head(s, a);
  return e != hipSuccess ? e : launch_prepare_tail(s, a);
}

namespace {
struct PrepareBuffers {
  unsigned long long *codes_in, *codes_out;
  uint32_t *idx_in, *idx_out, *keep_i, *rank_i, *queries, *nbr;
  double* spts;
  CellEntry* table;
  uint32_t mask;
  TileSlot* tiles_a;
  unsigned long long* tiles_b;
  char* cub;
  size_t cub_bytes;
};
PrepareBuffers prepare_buffers(const PrepareArgs& a) {
  const Layout L = layout_for(a.n);
  char* b = static_cast<char*>(a.scratch);
  PrepareBuffers B;
  B.codes_in = reinterpret_cast<unsigned long long*>(b + L.codes_in);
  B.codes_out = reinterpret_cast<unsigned long long*>(b + L.codes_out);
  B.idx_in = reinterpret_cast<uint32_t*>(b + L.idx_in);
  B.idx_out = reinterpret_cast<uint32_t*>(b + L.idx_out);
  B.spts = reinterpret_cast<double*>(b + L.spts);
  B.keep_i = reinterpret_cast<uint32_t*>(b + L.keep_i);
  B.rank_i = reinterpret_cast<uint32_t*>(b + L.rank_i);
  B.queries = reinterpret_cast<uint32_t*>(b + L.queries);
  B.nbr = reinterpret_cast<uint32_t*>(b + L.nbr);
  B.table = static_cast<CellEntry*>(a.cell_table);
  B.mask = (uint32_t)(a.table_entries - 1);
  B.tiles_a = static_cast<TileSlot*>(a.tiles);
  B.tiles_b = reinterpret_cast<unsigned long long*>(static_cast<char*>(a.tiles) + (size_t)kMaxScanTiles * sizeof(TileSlot));
  B.cub = b + L.cub;
  B.cub_bytes = L.cub_bytes;
  return B;
}
}  // namespace

bool prepare_bounds_fused(uint32_t n, uint32_t states, bool ordered_states) {
  return states != 0 && ordered_states && states <= kFusedBoundsStatesMax && blocks_for(n, 256) <= kFusedBoundsBlocksMax;
}

hipError_t launch_prepare_head(hipStream_t s, const PrepareArgs& a) {
  const uint32_t n = a.n;
  if (n == 0) return hipSuccess;
  if ((uint64_t)blocks_for(n, kScanThreads * (uint32_t)scan_items_for(n)) > kMaxScanTiles) return hipErrorInvalidValue;
  const PrepareBuffers B = prepare_buffers(a);
  unsigned long long* codes_in = B.codes_in;
  uint32_t *idx_in = B.idx_in, *keep_i = B.keep_i;
  CellEntry* table = B.table;

  // ---- deskew bounds (times only), then the prologue: extrinsic + deskew + codes + clears ----
  PrologueArgs pa;
  std::memset(&pa, 0, sizeof pa);
  pa.pts = a.pts;
  pa.n = n;
  pa.has_T = a.extrinsic16 != nullptr;
  if (a.extrinsic16) for (int k = 0; k < 16; ++k) pa.T.m[k] = a.extrinsic16[k];
  pa.states = a.states;
  pa.ends = a.ends;
  pa.poses = a.poses;
  const bool fused = a.max_hits_known && prepare_bounds_fused(n, a.states, a.ordered_states);
  if (fused) {
    pa.fused = 1u;
    pa.max_hits = a.max_hits;
    pa.point_time = a.point_time;
    pa.state_time = a.state_time;
    pa.slots = reinterpret_cast<unsigned long long*>(static_cast<char*>(a.tiles) +
                                                     (size_t)kMaxScanTiles * (sizeof(TileSlot) + sizeof(unsigned long long)));
  } else if (a.states) {
    if (a.ordered_states && a.states <= kDeskewMaxStates) {
      // few blocks: every workgroup of the prologue merges their first hits for itself
      uint32_t parts = blocks_for(n, 1024);
      if (parts > kProloguePartsMax) parts = kProloguePartsMax;
      const uint32_t per_block = blocks_for(n, parts);
      uint32_t* part = a.ends + a.states;
      hipLaunchKernelGGL(deskew_first_hit_kernel, dim3(parts), dim3(kFirstHitBlock), (size_t)a.states * 12, s, a.point_time, n,
                         a.state_time, a.states, per_block, part);
      ++g_kernel_launches;
      pa.parts = parts;
      pa.part = part;
    } else {
      hipLaunchKernelGGL(deskew_bounds_kernel, dim3(1), dim3(kBoundsBlock), 0, s, a.point_time, n, a.state_time, a.states,
                         a.ends);
      ++g_kernel_launches;
      pa.parts = 0;
    }
  }
  pa.fine = a.voxel_size / (double)(1 << kFineShift);
  pa.codes = codes_in;
  pa.idx = idx_in;
  pa.keep_by_index = keep_i;
  pa.counters = a.counters;
  pa.epoch = a.epoch;
  pa.table = reinterpret_cast<unsigned long long*>(table);
  pa.entries = a.table_entries;
  pa.src = a.src_points;
  pa.src_flags = a.src_flags;
  pa.src_seq = a.src_seq;
  pa.src_unit = a.src_unit ? a.src_unit : 256u;
  pa.src_spin = a.src_spin;
  pa.src_step = a.src_step;
  for (int c = 0; c < 3; ++c) pa.src_off[c] = a.src_off[c];
  hipLaunchKernelGGL(sweep_prologue_kernel, dim3(blocks_for(n, 256)), dim3(256),
                     (size_t)a.states * (fused ? sizeof(double) : sizeof(uint32_t)), s, pa);
  ++g_kernel_launches;
  if (a.ev_after_prologue) {
    const hipError_t ee = hipEventRecord(a.ev_after_prologue, s);
    if (ee != hipSuccess) return ee;
  }
  return hipGetLastError();
}

hipError_t launch_prepare_tail(hipStream_t s, const PrepareArgs& a) {
  const uint32_t n = a.n;
  if (n == 0) return hipSuccess;
  const PrepareBuffers B = prepare_buffers(a);
  unsigned long long *codes_in = B.codes_in, *codes_out = B.codes_out;
  uint32_t *idx_in = B.idx_in, *idx_out = B.idx_out, *keep_i = B.keep_i, *rank_i = B.rank_i, *queries = B.queries, *nbr = B.nbr;
  double* spts = B.spts;
  CellEntry* table = B.table;
  const uint32_t mask = B.mask;
  TileSlot* tiles_a = B.tiles_a;
  unsigned long long* tiles_b = B.tiles_b;

  // ---- the one sort ----
#ifdef VGICP_SORT_ROCPRIM
  size_t cub_bytes = B.cub_bytes;
  hipError_t e = sort_codes_rocprim(B.cub, cub_bytes, codes_in, codes_out, idx_in, idx_out, n, s);
#else
  hipError_t e = sortk::sort_pairs(codes_in, idx_in, codes_out, idx_out, B.cub, n, s);   // the inputs are scratch from here on
#endif
  if (e != hipSuccess) return e;
  g_kernel_launches += merge_sort_launches(n);

  // ---- runs, kept points, query list (one launch); output slots in scan order (one launch) ----
  const uint32_t items = (uint32_t)scan_items_for(n);
  const uint32_t scan_tiles = blocks_for(n, kScanThreads * items), cell_blocks_x = blocks_for(n, kScanThreads);
  // the sort is done: its input buffers hold the two query lists now
  uint32_t* q_heavy = idx_in;
  uint32_t* q_light = reinterpret_cast<uint32_t*>(codes_in);
  const uint32_t keep_blocks = (scan_tiles + 3u) / 4u, split_blocks = blocks_for(n, kSplitBlock);
  if (items == (uint32_t)kScanItemsSmall) {
    hipLaunchKernelGGL(run_scan_kernel<kScanItemsSmall>, dim3(scan_tiles + cell_blocks_x * kLevels), dim3(kScanThreads), 0, s, a.pts,
                       codes_out, idx_out, n, spts, queries, keep_i, a.counters, tiles_a, a.epoch, scan_tiles, table, mask,
                       cell_blocks_x, a.host_kept);
    hipLaunchKernelGGL(keep_and_split_kernel<kScanItemsSmall>, dim3(keep_blocks + split_blocks), dim3(kSplitBlock), 0, s, keep_i, n,
                       rank_i, a.counters, tiles_b, a.epoch, scan_tiles, keep_blocks, spts, a.voxel_size, table, mask, queries, q_heavy,
                       q_light);
  } else {
    hipLaunchKernelGGL(run_scan_kernel<kScanItemsLarge>, dim3(scan_tiles + cell_blocks_x * kLevels), dim3(kScanThreads), 0, s, a.pts,
                       codes_out, idx_out, n, spts, queries, keep_i, a.counters, tiles_a, a.epoch, scan_tiles, table, mask,
                       cell_blocks_x, a.host_kept);
    hipLaunchKernelGGL(keep_and_split_kernel<kScanItemsLarge>, dim3(keep_blocks + split_blocks), dim3(kSplitBlock), 0, s, keep_i, n,
                       rank_i, a.counters, tiles_b, a.epoch, scan_tiles, keep_blocks, spts, a.voxel_size, table, mask, queries, q_heavy,
                       q_light);
  }
  // ---- octree cells of all levels, the exact search (one wave per kept point; the grid covers every raw point,
  //      the waves beyond the kept count leave at once), covariances ----
  hipLaunchKernelGGL(knn_search_kernel, dim3(8 * (blocks_for((n + 7) / 8, kSearchBlock / 64) + 2)), dim3(kSearchBlock), 0, s, spts,
                     idx_out, n, a.voxel_size, a.knn, table, mask, q_heavy, q_light, rank_i, a.epoch, nbr, a.out_pts, a.out_idx, a.soa,
                     a.soa_stride, a.counters, a.debug);
  const int found = a.knn < (int)n ? a.knn : (int)n;
  hipLaunchKernelGGL(cov_kernel, dim3(blocks_for(n, kCovBlock)), dim3(kCovBlock), 0, s, a.pts, nbr, found, a.out_covs, a.soa,
                     a.soa_stride, a.counters, a.epoch);
  g_kernel_launches += 4;
  return hipGetLastError();
}

size_t deskew_scratch_words(uint32_t states) { return (size_t)states * (1 + kDeskewParts); }

hipError_t launch_deskew(hipStream_t s, double* pts, uint32_t n, const double* point_time, const double* state_time,
                         uint32_t states, const double* poses, uint32_t* ends, bool ordered_states) {
  if (n == 0 || states == 0) return hipSuccess;
  if (ordered_states && states <= kDeskewMaxStates) {
    uint32_t parts = blocks_for(n, 256);
    if (parts > kDeskewParts) parts = kDeskewParts;
    const uint32_t per_block = blocks_for(n, parts);
    uint32_t* part = ends + states;
    hipLaunchKernelGGL(deskew_first_hit_kernel, dim3(parts), dim3(kFirstHitBlock), (size_t)states * 12, s, point_time, n,
                       state_time, states, per_block, part);
    hipLaunchKernelGGL(deskew_merge_kernel, dim3(1), dim3(1024), (size_t)states * 4, s, part, parts, states, ends);
    g_kernel_launches += 2;
  } else {
    hipLaunchKernelGGL(deskew_bounds_kernel, dim3(1), dim3(kBoundsBlock), 0, s, point_time, n, state_time, states, ends);
    ++g_kernel_launches;
  }
  hipLaunchKernelGGL(deskew_apply_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, pts, n, ends, states, poses);
  ++g_kernel_launches;
  return hipGetLastError();
}

}  // namespace vgicp
