// vgicp_preprocess.hip — CloudPreprocessor::voxelDownsampleAndEstimateCovariances on the device
// (reference src/CloudPreprocessor.cpp:76-127).
//
// The reference builds an Open3D KD-tree on the whole scan, keeps the first point of every voxel, and
// for each kept point takes its 30 nearest neighbours in the whole scan (the point itself included),
// Open3D's cumulant covariance over them and the regularisation U diag(1, 1, 1e-2) V^T of a JacobiSVD.
// Here:
//   1. every point gets the 63-bit Morton code of its voxel (cell = voxel_size, the same floor(p / h)
//      as the map keys); ONE stable radix sort (hipCUB) of (code, index) orders the scan so that the
//      cells of every octree level — h, 2h, 4h, ... — are contiguous runs
//   2. the first entry of every level-0 run is the first point of its voxel (stable sort keeps scan
//      order inside a run): that is the down-sampling; the kept indices are then sorted ascending
//   3. a hash table of (level, cell) -> [start, end) over the sorted order is built for all levels
//   4. one thread per kept point searches the 27 cells around it at the finest level whose block can
//      hold k points, keeps the k best (distance, index) pairs sorted in LDS, and accepts the result
//      only if the k-th distance is within the distance to the block's boundary — otherwise the next
//      coarser level (twice the cell) is searched; the result is therefore the EXACT k nearest
//      neighbours, ties broken by index
//   5. cumulants in ascending-distance order, covariance, cyclic-Jacobi eigen-decomposition (for a
//      symmetric positive semi-definite matrix U = V = eigenvectors), U diag(1,1,1e-2) U^T.
// Output order is ascending original index (the reference's is unordered_map iteration order).
#include <hipcub/hipcub.hpp>

#include "vgicp_device.h"
#include "vgicp_device_fn.h"

namespace vgicp {
namespace {

constexpr int kMaxKnn = 32;
constexpr int kLevels = 10;          // cells of h, 2h, ... 512h
constexpr int kCoordOffset = 1 << 20;  // voxel indices are offset to be non-negative (21 bits per axis)
constexpr int kKnnBlock = 128;
constexpr unsigned long long kEmptyCell = ~0ull;

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
__device__ __forceinline__ unsigned long long morton3(uint32_t x, uint32_t y, uint32_t z) {
  return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}
__device__ __forceinline__ uint32_t cell_coord(double v, double h) {
  long long c = (long long)floor(v / h) + kCoordOffset;
  c = c < 0 ? 0 : (c > 0x1FFFFF ? 0x1FFFFF : c);
  return (uint32_t)c;
}
__device__ __forceinline__ unsigned long long cell_key(unsigned long long morton_at_level, int level) {
  return ((unsigned long long)level << 60) | morton_at_level;
}
__device__ __forceinline__ uint32_t cell_hash(unsigned long long key) {
  key ^= key >> 33; key *= 0xFF51AFD7ED558CCDull; key ^= key >> 33; key *= 0xC4CEB9FE1A85EC53ull; key ^= key >> 33;
  return (uint32_t)key;
}

__global__ void morton_kernel(const double* __restrict__ pts, uint32_t n, double h,
                              unsigned long long* __restrict__ codes, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  codes[i] = morton3(cell_coord(pts[3 * (size_t)i], h), cell_coord(pts[3 * (size_t)i + 1], h),
                     cell_coord(pts[3 * (size_t)i + 2], h));
  idx[i] = i;
}

// one pass over the sorted codes: the first entry of a level-0 run is the first point of its voxel
// (appended to the kept list), and the number of runs over all levels sizes the cell table
__global__ void run_count_kernel(const unsigned long long* __restrict__ codes, const uint32_t* __restrict__ idx,
                                 uint32_t n, uint32_t* kept, uint32_t* counters) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long c = codes[j];
  const unsigned long long prev = j ? codes[j - 1] : 0ull;
  uint32_t runs = 0;
#pragma unroll 1
  for (int l = 0; l < kLevels; ++l) {
    if (j != 0 && (c >> (3 * l)) == (prev >> (3 * l))) break;
    if (l == 0) kept[atomicAdd(&counters[0], 1u)] = idx[j];
    ++runs;
  }
  if (runs) atomicAdd(&counters[1], runs);
}

// starts of the runs of every level: claim the cell's entry and store the start
__global__ void cell_start_kernel(const unsigned long long* __restrict__ codes, uint32_t n, CellEntry* table,
                                  uint32_t mask) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long c = codes[j];
  const unsigned long long prev = j ? codes[j - 1] : 0ull;
#pragma unroll 1
  for (int l = 0; l < kLevels; ++l) {
    const unsigned long long m = c >> (3 * l);
    if (j != 0 && m == (prev >> (3 * l))) break;  // same cell as the predecessor here and at every coarser level
    const unsigned long long key = cell_key(m, l);
    uint32_t slot = cell_hash(key) & mask;
    for (;;) {  // cells are unique per (level, key) and the table holds twice their number
      const unsigned long long seen = atomicCAS(&table[slot].key, kEmptyCell, key);
      if (seen == kEmptyCell || seen == key) break;
      slot = (slot + 1) & mask;
    }
    table[slot].start = j;
  }
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

__global__ void cell_end_kernel(const unsigned long long* __restrict__ codes, uint32_t n, CellEntry* table,
                                uint32_t mask) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const unsigned long long c = codes[j];
  const unsigned long long next = j + 1 < n ? codes[j + 1] : 0ull;
#pragma unroll 1
  for (int l = 0; l < kLevels; ++l) {
    const unsigned long long m = c >> (3 * l);
    if (j + 1 < n && m == (next >> (3 * l))) break;  // not the last of its run (nor of any coarser one)
    CellEntry* e = const_cast<CellEntry*>(find_cell(table, mask, cell_key(m, l)));
    if (e) e->end = j + 1;
  }
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi; eigenvalues descending, U's columns the vectors.
__device__ __forceinline__ void symmetric_eigen3(double (&A)[3][3], double (&U)[3][3]) {
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  // order columns by descending eigenvalue (static index swaps)
  double w0 = A[0][0], w1 = A[1][1], w2 = A[2][2];
#define VG_SWAP_COL(a, b, wa, wb)                                                   \
  if (wb > wa) {                                                                    \
    const double tw = wa; wa = wb; wb = tw;                                         \
    for (int k = 0; k < 3; ++k) { const double tv = V[k][a]; V[k][a] = V[k][b]; V[k][b] = tv; } \
  }
  VG_SWAP_COL(0, 1, w0, w1)
  VG_SWAP_COL(0, 2, w0, w2)
  VG_SWAP_COL(1, 2, w1, w2)
#undef VG_SWAP_COL
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) U[r][c] = V[r][c];
}

// Exact k nearest neighbours + covariance + regularisation for the kept points (one thread each).
__global__ __launch_bounds__(kKnnBlock) void knn_cov_kernel(
    const double* __restrict__ pts, uint32_t n, double h, int knn, const uint32_t* __restrict__ sorted_idx,
    const CellEntry* __restrict__ table, uint32_t mask, const uint32_t* __restrict__ kept, uint32_t m,
    double* __restrict__ out_pts, double* __restrict__ out_covs, unsigned long long* __restrict__ out_idx,
    uint32_t* counters) {
  __shared__ double best_d[kMaxKnn][kKnnBlock];
  __shared__ uint32_t best_i[kMaxKnn][kKnnBlock];
  const uint32_t t = threadIdx.x;
  const uint32_t o = blockIdx.x * kKnnBlock + t;
  if (o >= m) return;
  const uint32_t qi = kept[o];
  const double qx = pts[3 * (size_t)qi], qy = pts[3 * (size_t)qi + 1], qz = pts[3 * (size_t)qi + 2];
  const int K = knn < (int)n ? knn : (int)n;
  int found = 0;

  auto offer = [&](uint32_t id) {
    const double dx = pts[3 * (size_t)id] - qx, dy = pts[3 * (size_t)id + 1] - qy, dz = pts[3 * (size_t)id + 2] - qz;
    const double d = dx * dx + dy * dy + dz * dz;
    if (found == K && !(d < best_d[K - 1][t] || (d == best_d[K - 1][t] && id < best_i[K - 1][t]))) return;
    int j = found < K ? found : K - 1;
    while (j > 0 && (best_d[j - 1][t] > d || (best_d[j - 1][t] == d && best_i[j - 1][t] > id))) {
      best_d[j][t] = best_d[j - 1][t];
      best_i[j][t] = best_i[j - 1][t];
      --j;
    }
    best_d[j][t] = d;
    best_i[j][t] = id;
    if (found < K) ++found;
  };

  const uint32_t cx = cell_coord(qx, h), cy = cell_coord(qy, h), cz = cell_coord(qz, h);
  bool done = false;
  for (int l = 0; l < kLevels && !done; ++l) {
    const int lx = (int)(cx >> l), ly = (int)(cy >> l), lz = (int)(cz >> l);
    const int top = 0x1FFFFF >> l;
    // can the 27-cell block hold K points at all?
    uint32_t population = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = lx + dx, y = ly + dy, z = lz + dz;
          if (x < 0 || y < 0 || z < 0 || x > top || y > top || z > top) continue;
          const CellEntry* e = find_cell(table, mask, cell_key(morton3((uint32_t)x, (uint32_t)y, (uint32_t)z), l));
          if (e) population += e->end - e->start;
        }
    if (population < (uint32_t)K && l + 1 < kLevels) continue;
    found = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = lx + dx, y = ly + dy, z = lz + dz;
          if (x < 0 || y < 0 || z < 0 || x > top || y > top || z > top) continue;
          const CellEntry* e = find_cell(table, mask, cell_key(morton3((uint32_t)x, (uint32_t)y, (uint32_t)z), l));
          if (!e) continue;
          for (uint32_t j = e->start; j < e->end; ++j) offer(sorted_idx[j]);
        }
    // every point outside the block is farther than the distance to the block's nearest face
    const double fx = qx / h - floor(qx / h), fy = qy / h - floor(qy / h), fz = qz / h - floor(qz / h);
    // position of the query inside its level-l cell, in units of h
    const double sub = (double)(1 << l);
    const double px = (double)(cx & ((1u << l) - 1u)) + fx, py = (double)(cy & ((1u << l) - 1u)) + fy,
                 pz = (double)(cz & ((1u << l) - 1u)) + fz;
    const double margin = fmin(fmin(fmin(px, sub - px), fmin(py, sub - py)), fmin(pz, sub - pz));
    const double safe = (sub + margin) * h * (1.0 - 1e-12);
    if (found == K && best_d[K - 1][t] <= safe * safe) done = true;
  }
  if (!done) {  // sparser than the coarsest level resolves (or fewer than K points near): look at everything
    found = 0;
    for (uint32_t id = 0; id < n; ++id) offer(id);
    atomicAdd(&counters[2], 1u);
  }

  double cov[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (found >= 3) {
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < found; ++k) {
      const uint32_t id = best_i[k][t];
      const double x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
      c[0] += x; c[1] += y; c[2] += z;
      c[3] += x * x; c[4] += x * y; c[5] += x * z;
      c[6] += y * y; c[7] += y * z; c[8] += z * z;
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
  double U[3][3];
  symmetric_eigen3(cov, U);
  out_pts[3 * (size_t)o] = qx; out_pts[3 * (size_t)o + 1] = qy; out_pts[3 * (size_t)o + 2] = qz;
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
#pragma unroll
    for (int r = 0; r < 3; ++r)
      out_covs[9 * (size_t)o + r + 3 * cc] = U[r][0] * U[cc][0] + U[r][1] * U[cc][1] + (U[r][2] * 1e-2) * U[cc][2];
  out_idx[o] = qi;
}

__global__ void cell_clear_kernel(CellEntry* table, uint64_t entries) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= entries) return;
  table[i].key = kEmptyCell;
  table[i].start = 0;
  table[i].end = 0;
}

inline uint32_t blocks_for(uint64_t work, uint32_t block) { return (uint32_t)((work + block - 1) / block); }
__host__ inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
__host__ inline uint64_t pow2_at_least(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

struct Layout {
  size_t codes_in, codes_out, idx_in, idx_out, kept_a, kept_b, cub, total;
  size_t cub_bytes;
};

__host__ inline Layout layout_for(uint32_t n) {
  Layout L;
  size_t sort_pairs = 0, sort_keys = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_pairs, (const unsigned long long*)nullptr,
                                           (unsigned long long*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)n);
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_keys, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
  L.cub_bytes = sort_pairs > sort_keys ? sort_pairs : sort_keys;
  size_t off = 0;
  L.codes_in = off; off += align256((size_t)n * 8);
  L.codes_out = off; off += align256((size_t)n * 8);
  L.idx_in = off; off += align256((size_t)n * 4);
  L.idx_out = off; off += align256((size_t)n * 4);
  L.kept_a = off; off += align256((size_t)n * 4);
  L.kept_b = off; off += align256((size_t)n * 4);
  L.cub = off; off += align256(L.cub_bytes);
  L.total = off + 256;
  return L;
}

}  // namespace

size_t preprocess_scratch_bytes(uint32_t n) { return layout_for(n ? n : 1).total; }
int preprocess_max_knn() { return kMaxKnn; }
uint64_t preprocess_cell_entries(uint32_t cells) { return pow2_at_least((uint64_t)cells * 2 + 64); }
size_t preprocess_cell_bytes(uint64_t entries) { return entries * sizeof(CellEntry); }

// Stage A: Morton codes, the sort, the kept list (counters[0] = kept points) and the number of cells
// over all levels (counters[1]).
hipError_t launch_preprocess_sort(hipStream_t s, const double* pts, uint32_t n, double h, void* scratch,
                                  uint32_t* counters) {
  const Layout L = layout_for(n);
  char* b = static_cast<char*>(scratch);
  auto* codes_in = reinterpret_cast<unsigned long long*>(b + L.codes_in);
  auto* codes_out = reinterpret_cast<unsigned long long*>(b + L.codes_out);
  auto* idx_in = reinterpret_cast<uint32_t*>(b + L.idx_in);
  auto* idx_out = reinterpret_cast<uint32_t*>(b + L.idx_out);
  auto* kept_a = reinterpret_cast<uint32_t*>(b + L.kept_a);
  hipLaunchKernelGGL(morton_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, pts, n, h, codes_in, idx_in);
  size_t cub_bytes = L.cub_bytes;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(b + L.cub, cub_bytes, codes_in, codes_out, idx_in, idx_out,
                                                    (int)n, 0, 63, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(run_count_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, codes_out, idx_out, n, kept_a,
                     counters);
  return hipGetLastError();
}

// Stage B: the cell table (entries = preprocess_cell_entries(counters[1])), the kept indices in
// ascending order, then the neighbour search + covariance for each of the m kept points.
hipError_t launch_preprocess_finish(hipStream_t s, const double* pts, uint32_t n, double h, int knn, uint32_t m,
                                    void* scratch, void* cell_table, uint64_t table_entries, double* out_pts,
                                    double* out_covs, unsigned long long* out_idx, uint32_t* counters) {
  if (m == 0) return hipSuccess;
  const Layout L = layout_for(n);
  char* b = static_cast<char*>(scratch);
  auto* codes_out = reinterpret_cast<unsigned long long*>(b + L.codes_out);
  auto* idx_out = reinterpret_cast<uint32_t*>(b + L.idx_out);
  auto* kept_a = reinterpret_cast<uint32_t*>(b + L.kept_a);
  auto* kept_b = reinterpret_cast<uint32_t*>(b + L.kept_b);
  auto* table = static_cast<CellEntry*>(cell_table);
  const uint32_t mask = (uint32_t)(table_entries - 1);
  hipLaunchKernelGGL(cell_clear_kernel, dim3(blocks_for(table_entries, 256)), dim3(256), 0, s, table,
                     table_entries);
  hipLaunchKernelGGL(cell_start_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, codes_out, n, table, mask);
  hipLaunchKernelGGL(cell_end_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, codes_out, n, table, mask);
  size_t cub_bytes = L.cub_bytes;
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(b + L.cub, cub_bytes, kept_a, kept_b, (int)m, 0, 32, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(knn_cov_kernel, dim3(blocks_for(m, kKnnBlock)), dim3(kKnnBlock), 0, s, pts, n, h, knn,
                     idx_out, table, mask, kept_b, m, out_pts, out_covs, out_idx, counters);
  return hipGetLastError();
}

}  // namespace vgicp
