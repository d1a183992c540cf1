// GPU check of the preparation's sort (eskf_lio_amd/csrc/vgicp_sort.h) against std::stable_sort: sizes around every
// boundary of the plan (one tile, a tile's end, a group's end, a level more), keys with long runs of equal values (a
// scan's voxel codes), all equal, already sorted, reversed, and random 63-bit keys.  Built by the module's Makefile
// (`make sort_check`) into eskf_lio_amd/lib/, run by tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "../../eskf_lio_amd/csrc/vgicp_sort.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static unsigned long long rng_state = 0x2545F4914F6CDD1Dull;
static unsigned long long rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
  if (argc > 2) {   // sort_check time <n> [kind]: the launches of one sort, timed (for rocprofv3 --kernel-trace --stats)
    const uint32_t n = (uint32_t)std::atoi(argv[2]);
    const int kind = argc > 3 ? std::atoi(argv[3]) : 0;
    std::vector<unsigned long long> keys(n);
    // kind 0: random keys, ~6 pairs each; kind 1: a walk (consecutive pairs have nearby keys, as a sweep's points do)
    unsigned long long walk = 1ull << 40;
    for (uint32_t i = 0; i < n; ++i) {
      if (kind == 0) keys[i] = rng() % (n / 6 + 1);
      else { walk += (rng() % 2001) - 1000; if (i % 900 == 0) walk = (1ull << 40) + rng() % 3000000; keys[i] = walk >> 3; }
    }
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    unsigned long long *ka, *kb, *k0; uint32_t *ia, *ib, *i0; void* split;
    CK(hipMalloc(&ka, n * 8ull)); CK(hipMalloc(&kb, n * 8ull)); CK(hipMalloc(&k0, n * 8ull));
    CK(hipMalloc(&ia, n * 4ull)); CK(hipMalloc(&ib, n * 4ull)); CK(hipMalloc(&i0, n * 4ull));
    CK(hipMalloc(&split, vgicp::sortk::split_bytes(n, 8)));
    CK(hipMemcpy(k0, keys.data(), n * 8ull, hipMemcpyHostToDevice));
    CK(hipMemcpy(i0, idx.data(), n * 4ull, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float total = 0;
    for (int rep = 0; rep < 220; ++rep) {
      CK(hipMemcpyAsync(ka, k0, n * 8ull, hipMemcpyDeviceToDevice, s));
      CK(hipMemcpyAsync(ia, i0, n * 4ull, hipMemcpyDeviceToDevice, s));
      CK(hipEventRecord(e0, s));
      CK(vgicp::sortk::sort_pairs(ka, ia, kb, ib, split, n, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 20) total += ms;
    }
    std::printf("n = %u, kind %d: %.2f us per sort (%u launches)\n", n, kind, total / 200 * 1e3, vgicp::sortk::launches_for(n));
    return 0;
  }
  const uint32_t sizes[] = {1, 2, 3, 255, 1023, 1024, 1025, 2048, 2049, 4097, 8191, 8192, 8193, 10131, 16385, 60000, 65535, 65536,
                            65537, 100000, 131073, 262144, 262145, 300001, 1000003, 1048577};
  const uint32_t cap = 1048577 + 8;
  unsigned long long *ka, *kb;
  uint32_t *ia, *ib;
  CK(hipMalloc(&ka, cap * 8ull)); CK(hipMalloc(&kb, cap * 8ull)); CK(hipMalloc(&ia, cap * 4ull)); CK(hipMalloc(&ib, cap * 4ull));
  void* split;
  CK(hipMalloc(&split, vgicp::sortk::split_bytes(cap, 8)));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  int checked = 0;
  for (uint32_t n : sizes) {
    for (int kind = 0; kind < 6; ++kind) {
      if (n > 300001 && kind > 1) continue;
      std::vector<unsigned long long> keys(n);
      for (uint32_t i = 0; i < n; ++i) {
        switch (kind) {
          case 0: keys[i] = rng() % (n / 6 + 1); break;                     // ~6 pairs per key: voxel codes
          case 1: keys[i] = rng() >> 1; break;                                // 63 random bits
          case 2: keys[i] = 42; break;                                        // all equal: the order is the index order
          case 3: keys[i] = i / 3; break;                                     // sorted already
          case 4: keys[i] = (n - i) / 5; break;                               // reversed
          default: keys[i] = (rng() % 7) << 60 | (rng() % 3); break;          // few distinct keys, high and low bits
        }
      }
      std::vector<uint32_t> idx(n);
      std::iota(idx.begin(), idx.end(), 0u);
      CK(hipMemcpyAsync(ka, keys.data(), n * 8ull, hipMemcpyHostToDevice, s));
      CK(hipMemcpyAsync(ia, idx.data(), n * 4ull, hipMemcpyHostToDevice, s));
      CK(hipMemsetAsync(kb, 0xEE, cap * 8ull, s));
      CK(hipMemsetAsync(ib, 0xEE, cap * 4ull, s));
      CK(vgicp::sortk::sort_pairs(ka, ia, kb, ib, split, n, s));
      std::vector<unsigned long long> got_k(n + 1);
      std::vector<uint32_t> got_i(n + 1);
      CK(hipMemcpyAsync(got_k.data(), kb, (n + 1) * 8ull, hipMemcpyDeviceToHost, s));
      CK(hipMemcpyAsync(got_i.data(), ib, (n + 1) * 4ull, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
      for (uint32_t i = 0; i < n; ++i) {
        if (got_i[i] != idx[i] || got_k[i] != keys[idx[i]]) {
          std::printf("n = %u, kind %d: position %u holds (%llu, %u), expected (%llu, %u)\n", n, kind, i, got_k[i], got_i[i],
                      keys[idx[i]], idx[i]);
          return 1;
        }
      }
      if (got_k[n] != 0xEEEEEEEEEEEEEEEEull || got_i[n] != 0xEEEEEEEEu) { std::printf("n = %u, kind %d: wrote past the end\n", n, kind); return 1; }
      ++checked;
    }
  }
  std::printf("ok: %d sorts, %u launches for 60 000 pairs, %u for 1 000 000\n", checked, vgicp::sortk::launches_for(60000),
              vgicp::sortk::launches_for(1000000));
  return 0;
}
