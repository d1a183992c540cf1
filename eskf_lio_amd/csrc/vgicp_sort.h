// The preparation's one sort, hand-written for gfx950: (key, index) pairs, keys ascending, equal keys in ascending
// index order — the result of a stable sort when the indices come in ascending (the prologue writes idx[i] = i).
//
// A scan is 10^4 .. 10^6 pairs: small enough that a sort is paid for in LAUNCHES and in loads that wait for one another,
// not in bytes (60 000 pairs are 720 KB).  rocPRIM's merge sort, which this replaces, took one block sort (19 us: 15
// blocks of 4 096 on 256 CUs) and one merge launch per doubling of the run length (4 x 5.9 us): 42.5 us for 60 000
// pairs (profiles/r16_frame_kernel_stats.csv).  Here:
//   1. tile_sort_kernel: every WAVE sorts 256 pairs, four per lane, with a bitonic network over (key, index) — the index
//      breaks ties, which makes the order total and lets an unstable network give the stable result.  Exchanges at
//      distance 1 and 2 are inside a lane, the others are lane shuffles: no LDS, no barrier.  235 waves, 8 us.
//   2. rank_merge_kernel<G>: G sorted runs become one in ONE launch.  A pair's place in the merged run is its place in
//      its own run plus, for each of the other G - 1 runs, the number of pairs there that go before it — everything
//      `<=` its key in an earlier run (those hold lower indices), everything `<` its key in a later run: keys only, no
//      index is read.  A wave takes 64 consecutive pairs of one run.  Their counts in another run all lie between the
//      count of the wave's first key and that of its last, so the WAVE finds those two (64 lanes probe 64 places at a
//      time: the run's 64 splitters — every (run / 64)-th key, written aside by the launch that made the run — then the
//      64 keys between two of them: two coalesced loads in a row, whatever the run length up to 4 096), the windows
//      between the two counts (64 keys each on average) go to LDS one behind the other, and every lane finishes its
//      G - 1 searches there, side by side.  Windows that do not fit (few distinct keys) are searched where they are.
//      Every step is taken for all other runs at once, so a wave waits for four loads in a row whatever G is; what G
//      costs is instructions: measured 4 us + 0.85 .. 1.3 us per other run and level (60 000 pairs, one wave per SIMD:
//      nothing hides anything), i.e. 5.2 / 6.5 / 11.4 / 24 us per level for G = 2 / 4 / 8 / 16, and 60 000 pairs need
//      8 / 4 / 3 / 2 levels: G = 4 it is (tools/ab_sort.sh; -DVGICP_SORT_MAX_GROUP=8 or 16 builds the others).
// 60 000 pairs: 8 + 4 x 6.5 = 34 us in the frame chain against 42.5 us (same session, tools/ab_kernel.sh).
//
// The two buffers alternate; sort_pairs says where the result is wanted and starts on the side that ends there.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace vgicp {
namespace sortk {

#ifndef VGICP_SORT_MAX_GROUP
#define VGICP_SORT_MAX_GROUP 4
#endif
constexpr int kMaxGroup = VGICP_SORT_MAX_GROUP;   // runs merged by one launch, at most (2, 4, 8 or 16)
constexpr uint32_t kTile = 256;            // pairs a wave sorts
constexpr uint32_t kTileThreads = 256;     // four waves, four tiles per block
constexpr uint32_t kMergeThreads = 256;
constexpr uint32_t kFlat = 4096;           // keys of the other runs' windows a wave keeps in LDS (beyond: per-lane searches)

template <typename K>
struct Pair {
  K k;
  uint32_t v;
};
template <typename K>
__device__ __forceinline__ bool after(const Pair<K>& a, const Pair<K>& b) {   // a is placed after b
  return b.k < a.k || (!(a.k < b.k) && b.v < a.v);
}
__device__ __forceinline__ unsigned long long shfl_xor_key(unsigned long long k, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)k, mask), hi = (uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), mask);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shfl_xor_key(uint32_t k, int mask) { return (uint32_t)__shfl_xor((int)k, mask); }
// the value lane t holds, t wave-uniform: scalar reads, no LDS
__device__ __forceinline__ unsigned long long read_lane(unsigned long long k, uint32_t t) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, (int)t), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), (int)t);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t read_lane(uint32_t k, uint32_t t) { return (uint32_t)__builtin_amdgcn_readlane((int)k, (int)t); }

// splitters of a run of `run` pairs: the keys at the positions = S - 1 (mod S), S = run / 64, kept in an array of their
// own (position / S); a run's 64 splitters are 64 consecutive entries
template <typename K>
__global__ __launch_bounds__(kTileThreads) void tile_sort_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                                 K* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                                 K* __restrict__ split_out, uint32_t n) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t base = (blockIdx.x * (kTileThreads / 64u) + (threadIdx.x >> 6)) * kTile;   // wave-uniform
  if (base >= n) return;
  Pair<K> e[4];
  if (base + kTile <= n) {                        // (wave-uniform) a full tile: four consecutive pairs per lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r].k = keys_in[base + 4u * lane + (uint32_t)r];
      e[r].v = idx_in[base + 4u * lane + (uint32_t)r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {                 // unconditional loads from a clamped place, masked afterwards
      const uint32_t g = base + 4u * lane + (uint32_t)r;
      const bool there = g < n;
      const K k = keys_in[there ? g : 0u];
      const uint32_t v = idx_in[there ? g : 0u];
      e[r].k = there ? k : ~K(0);                  // beyond the end: after every pair of the scan
      e[r].v = there ? v : 0xFFFFFFFFu;
    }
  }
#pragma unroll
  for (uint32_t size = 2; size <= kTile; size <<= 1) {
#pragma unroll
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride < 4u) {                            // partner in this lane
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((uint32_t)r & stride) continue;
          const uint32_t p = 4u * lane + (uint32_t)r;
          const bool ascending = (p & size) == 0u;
          Pair<K>&a = e[r], &b = e[r | (int)stride];
          if (after(a, b) == ascending) { const Pair<K> t = a; a = b; b = t; }
        }
      } else {                                      // partner in lane ^ (stride / 4), same place there
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t p = 4u * lane + (uint32_t)r;
          const bool ascending = (p & size) == 0u, lower = (p & stride) == 0u;
          Pair<K> o;
          o.k = shfl_xor_key(e[r].k, (int)(stride >> 2));
          o.v = (uint32_t)__shfl_xor((int)e[r].v, (int)(stride >> 2));
          // the lower place keeps the earlier pair when ascending, the later one when descending
          const bool mine_after = after(e[r], o);
          if (mine_after == (lower == ascending)) e[r] = o;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t g = base + 4u * lane + (uint32_t)r;
    if (g < n) { keys_out[g] = e[r].k; idx_out[g] = e[r].v; }
  }
  if (split_out && base + 4u * lane + 3u < n) split_out[(base >> 2) + lane] = e[3].k;   // S = 256 / 64 = 4
}

// runs of `run` pairs (a power of two >= 256; the last one may be short), G of them side by side become one.
// Every step below is taken for all G - 1 other runs at once (unrolled: their loads are in flight together), so a wave
// waits for three loads in a row (splitters, the keys between two splitters, the windows) whatever G is.
template <typename K, int G>
__global__ __launch_bounds__(kMergeThreads) void rank_merge_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                                                                   const K* __restrict__ split_in, K* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ idx_out, K* __restrict__ split_out, uint32_t n,
                                                                   uint32_t run_log2) {
  constexpr int R = G - 1;
  const uint32_t run = 1u << run_log2;
  __shared__ K flat_sh[kMergeThreads / 64][kFlat + 8];
  K* flat = flat_sh[threadIdx.x >> 6];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_first = (blockIdx.x * kMergeThreads + threadIdx.x) & ~63u;     // the wave's first pair
  if (wave_first >= n) return;
  const uint32_t i = wave_first + lane;
  const bool valid = i < n;
  const uint32_t last_lane = (n - wave_first < 64u ? n - wave_first : 64u) - 1u;
  const K got_key = keys_in[valid ? i : 0u];                    // (unconditional, clamped: see below)
  const uint32_t my_idx = idx_in[valid ? i : 0u];               // asked for now, needed at the very end
  const K key = valid ? got_key : ~K(0);
  const K k_first = read_lane(key, 0u), k_last = read_lane(key, last_lane);
  const uint32_t r = wave_first >> run_log2;                    // the wave's run (64 divides run) ...
  const uint32_t first = r - r % (uint32_t)G;                   // ... and the first run of its group
  const unsigned long long group_start = (unsigned long long)first << run_log2;
  const uint32_t own = r - first;
  // the other runs in order: slot j is run first + j, + 1 from this wave's own run on.  A pair of slot j is merged before
  // a pair with key x of this run when its key is <= x and j < own (an earlier run: lower indices), < x otherwise.
  unsigned long long start[R];
#pragma unroll
  for (int j = 0; j < R; ++j) start[j] = group_start + ((unsigned long long)((uint32_t)j + ((uint32_t)j >= own ? 1u : 0u)) << run_log2);
#define VGICP_SORT_BEFORE(j, m, x) ((uint32_t)(j) < own ? !((x) < (m)) : (m) < (x))
  // --- where the wave's first and last key fall in every other run: pf[j] <= every lane's count <= pl[j] ---
  uint32_t pf[R], pl[R];
  uint32_t range = run >> 6;                                    // what is left to search after the splitters
  {
    K sv[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {                               // the run's 64 splitters: the keys that end its 64 parts
      // (every load of this kernel is unconditional, from a clamped place, and masked afterwards: a load under a
      //  condition becomes a branch, and fifteen branches in a row wait for one another's data)
      const unsigned long long at = start[j] + (unsigned long long)(lane + 1u) * range - 1u;
      const bool there = at < n;
      const K got = split_in[there ? (start[j] >> (run_log2 - 6u)) + lane : 0ull];
      sv[j] = there ? got : ~K(0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      pf[j] = (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, sv[j], k_first))) * range;
      pl[j] = (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, sv[j], k_last))) * range;
    }
  }
  while (range > 64u) {                                         // runs beyond 4 096: every (range / 64)-th key of the part
    const uint32_t st = range >> 6;
    K a[R], b[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t fa = pf[j] + (lane + 1u) * st - 1u, fb = pl[j] + (lane + 1u) * st - 1u;
      const bool ta = fa < run && start[j] + fa < n, tb = fb < run && start[j] + fb < n;
      const K ga = keys_in[ta ? start[j] + fa : 0ull], gb = keys_in[tb ? start[j] + fb : 0ull];
      a[j] = ta ? ga : ~K(0);
      b[j] = tb ? gb : ~K(0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      pf[j] += (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, a[j], k_first))) * st;
      pl[j] += (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, b[j], k_last))) * st;
    }
    range = st;
  }
  {
    K a[R], b[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {                               // the keys of the part itself
      const uint32_t fa = pf[j] + lane, fb = pl[j] + lane;
      const bool ta = lane < range && fa < run && start[j] + fa < n, tb = lane < range && fb < run && start[j] + fb < n;
      const K ga = keys_in[ta ? start[j] + fa : 0ull], gb = keys_in[tb ? start[j] + fb : 0ull];
      a[j] = ta ? ga : ~K(0);
      b[j] = tb ? gb : ~K(0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      pf[j] += (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, a[j], k_first)));
      pl[j] += (uint32_t)__builtin_popcountll(__ballot(VGICP_SORT_BEFORE(j, b[j], k_last)));
    }
  }
  // --- the windows [pf, pl) of all other runs, one behind the other, in LDS; every lane counts inside them ---
  uint32_t off[R + 1];
  off[0] = 0u;
  uint32_t widest = 0u;
  unsigned long long rank = i - ((unsigned long long)r << run_log2);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint32_t w = pl[j] - pf[j];
    off[j + 1] = off[j] + w;
    widest = w > widest ? w : widest;
    rank += pf[j];
  }
  const uint32_t total = off[R];
  const bool in_lds = total <= kFlat;
  if (in_lds) {
    for (uint32_t c = 0; c < total; c += 512u) {                // eight loads in flight
      K w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t f = c + 64u * (uint32_t)u + lane;
        unsigned long long from = start[0] + pf[0];            // the run f falls into: the last j with off[j] <= f
#pragma unroll
        for (int j = 1; j < R; ++j) from = f >= off[j] ? start[j] + pf[j] - off[j] : from;
        w[u] = keys_in[f < total ? from + f : 0ull];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t f = c + 64u * (uint32_t)u + lane;
        if (f < total) flat[f] = w[u];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // all searches side by side, one probe each per step: in LDS, or — windows too long to keep (few distinct keys, or
  // runs that do not overlap evenly) — in the runs themselves
  uint32_t p[R], rem[R];
#pragma unroll
  for (int j = 0; j < R; ++j) { p[j] = 0u; rem[j] = off[j + 1] - off[j]; }
  const uint32_t steps = widest ? 32u - (uint32_t)__builtin_clz(widest) : 0u;
  for (uint32_t s = 0; s < steps; ++s) {
    K m[R];
    if (in_lds) {
#pragma unroll
      for (int j = 0; j < R; ++j) m[j] = flat[off[j] + p[j] + (rem[j] >> 1)];            // at most flat[total]: there is room
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const unsigned long long at = start[j] + pf[j] + p[j] + (rem[j] >> 1);       // at most one past the window
        m[j] = keys_in[at < n ? at : 0ull];
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t half = rem[j] >> 1;
      const bool right = rem[j] > 0u && VGICP_SORT_BEFORE(j, m[j], key);
      p[j] = right ? p[j] + half + 1u : p[j];
      rem[j] = right ? rem[j] - half - 1u : half;
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) rank += p[j];
#undef VGICP_SORT_BEFORE
  if (!valid) return;
  const unsigned long long to = group_start + rank;
  keys_out[to] = key;
  idx_out[to] = my_idx;
  if (split_out) {                                               // the merged run's splitters: S = G run / 64
    constexpr uint32_t g_log2 = G == 2 ? 1u : G == 4 ? 2u : G == 8 ? 3u : 4u;
    const uint32_t s_log2 = run_log2 + g_log2 - 6u;
    if ((to & ((1ull << s_log2) - 1u)) == (1ull << s_log2) - 1u) split_out[to >> s_log2] = key;
  }
}

// how the runs of a scan of n pairs are merged: the group sizes of the levels (each 2, 4, 8 or 16), fewest levels first,
// then the smallest groups that still finish in that many
struct Plan {
  int levels;
  int group[8];
};
inline Plan plan_for(uint32_t n) {
  Plan p;
  p.levels = 0;
  uint64_t runs = ((uint64_t)n + kTile - 1) / kTile;
  int levels = 0;
  for (uint64_t reach = 1; reach < runs; reach *= kMaxGroup) ++levels;
  for (int l = 0; l < levels; ++l) {
    const int left = levels - l;                 // levels still to come, this one included
    int g = 2;
    for (; g < kMaxGroup; g <<= 1) {
      uint64_t reach = 1;
      for (int q = 0; q < left; ++q) reach *= (uint64_t)g;
      if (reach >= runs) break;
    }
    p.group[p.levels++] = g;
    runs = (runs + (uint64_t)g - 1) / (uint64_t)g;
  }
  return p;
}
inline uint32_t launches_for(uint32_t n) { return n ? 1u + (uint32_t)plan_for(n).levels : 0u; }
// room for the splitters: two arrays (they alternate like the pair buffers) of one key per four pairs
inline size_t split_bytes(uint32_t n, size_t key_bytes) { return 2 * ((((size_t)n / 4 + 64) * key_bytes + 255) & ~size_t(255)); }

// Sorts n pairs.  (keys_a, idx_a) holds the input and is overwritten; the result is left in (keys_b, idx_b).
// split: split_bytes(n, sizeof(K)) bytes of scratch.
template <typename K>
inline hipError_t sort_pairs(K* keys_a, uint32_t* idx_a, K* keys_b, uint32_t* idx_b, void* split, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const Plan p = plan_for(n);
  K* split_0 = static_cast<K*>(split);
  K* split_1 = reinterpret_cast<K*>(static_cast<char*>(split) + split_bytes(n, sizeof(K)) / 2);
  const uint32_t tiles = (n + kTile - 1) / kTile, tile_blocks = (tiles + kTileThreads / 64 - 1) / (kTileThreads / 64);
  const uint32_t blocks = (n + kMergeThreads - 1) / kMergeThreads;
  // an even number of merge levels: the tile sort writes to b; odd: in place on a, the first merge moves to b, ...
  K* from_k = (p.levels % 2 == 0) ? keys_b : keys_a;
  uint32_t* from_i = (p.levels % 2 == 0) ? idx_b : idx_a;
  K* from_s = split_0;
  hipLaunchKernelGGL((tile_sort_kernel<K>), dim3(tile_blocks), dim3(kTileThreads), 0, s, keys_a, idx_a, from_k, from_i,
                     p.levels ? from_s : static_cast<K*>(nullptr), n);
  uint32_t run_log2 = 8;   // kTile
  for (int l = 0; l < p.levels; ++l) {
    K* to_k = from_k == keys_a ? keys_b : keys_a;
    uint32_t* to_i = from_i == idx_a ? idx_b : idx_a;
    K* to_s = l + 1 < p.levels ? (from_s == split_0 ? split_1 : split_0) : static_cast<K*>(nullptr);
    switch (p.group[l]) {
      case 2: hipLaunchKernelGGL((rank_merge_kernel<K, 2>), dim3(blocks), dim3(kMergeThreads), 0, s, from_k, from_i, from_s, to_k, to_i, to_s, n, run_log2); break;
      case 4: hipLaunchKernelGGL((rank_merge_kernel<K, 4>), dim3(blocks), dim3(kMergeThreads), 0, s, from_k, from_i, from_s, to_k, to_i, to_s, n, run_log2); break;
      case 8: hipLaunchKernelGGL((rank_merge_kernel<K, 8>), dim3(blocks), dim3(kMergeThreads), 0, s, from_k, from_i, from_s, to_k, to_i, to_s, n, run_log2); break;
      default: hipLaunchKernelGGL((rank_merge_kernel<K, 16>), dim3(blocks), dim3(kMergeThreads), 0, s, from_k, from_i, from_s, to_k, to_i, to_s, n, run_log2); break;
    }
    run_log2 += p.group[l] == 2 ? 1u : p.group[l] == 4 ? 2u : p.group[l] == 8 ? 3u : 4u;
    from_k = to_k;
    from_i = to_i;
    from_s = to_s;
  }
  return hipGetLastError();
}

}  // namespace sortk
}  // namespace vgicp
