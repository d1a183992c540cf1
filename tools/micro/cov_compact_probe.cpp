// Host-side micro-benchmark (no GPU): how fast can one / two threads move a scan's covariances from COLD pageable memory
// (every cloud read once, as vgicp_align's caller hands them over) into a staging buffer — whole (72 B per point, the
// streaming copy of rounds 5) or compacted to the six distinct entries of a symmetric matrix (48 B per point, round 6),
// in several instruction shapes.  Decides which shape eskf_lio_amd/csrc/vgicp_capi.hip uses.
//   g++ -O2 -std=c++17 -pthread -o cov_compact_probe cov_compact_probe.cpp && ./cov_compact_probe [threads]
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

__attribute__((target("avx2"))) static void plain_copy(char* dst, const char* src, size_t bytes) {
  size_t i = 0;
  for (; i + 128 <= bytes; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  _mm_sfence();
  if (i < bytes) std::memcpy(dst + i, src + i, bytes - i);
}
// V1: two points per turn, overlapping unaligned loads blended, scalar symmetry test, 3 x 32-byte streaming stores
__attribute__((target("avx2"))) static bool compact_v1(char* dst, const char* src, size_t cnt) {
  const double* s = (const double*)src; const uint64_t* w = (const uint64_t*)src; double* d = (double*)dst;
  uint64_t bad = 0; size_t i = 0;
  for (; i + 2 <= cnt; i += 2, s += 18, w += 18, d += 12) {
    const __m256d o0 = _mm256_blend_pd(_mm256_loadu_pd(s), _mm256_loadu_pd(s + 1), 0x8);
    const __m256d o1 = _mm256_blend_pd(_mm256_loadu_pd(s + 7), _mm256_loadu_pd(s + 5), 0x1);
    const __m256d o2 = _mm256_blend_pd(_mm256_blend_pd(_mm256_loadu_pd(s + 11), _mm256_loadu_pd(s + 12), 0x6), _mm256_loadu_pd(s + 14), 0x8);
    bad |= (w[1] ^ w[3]) | (w[2] ^ w[6]) | (w[5] ^ w[7]) | (w[10] ^ w[12]) | (w[11] ^ w[15]) | (w[14] ^ w[16]);
    _mm256_stream_pd(d, o0); _mm256_stream_pd(d + 4, o1); _mm256_stream_pd(d + 8, o2);
  }
  if (i < cnt) {
    bad |= (w[1] ^ w[3]) | (w[2] ^ w[6]) | (w[5] ^ w[7]);
    const uint64_t o[6] = {w[0], w[1], w[2], w[4], w[5], w[8]};
    std::memcpy(d, o, sizeof o);
  }
  _mm_sfence();
  return bad == 0;
}
// V3: eight points per turn in 512-bit registers: every output line (8 doubles) is one two-source permute of two
// unaligned loads, stored whole with one streaming store; the symmetry test compares shifted loads under constant masks
__attribute__((target("avx512f"))) static bool compact_v3(char* dst, const char* src, size_t cnt) {
  const double* s = (const double*)src; double* d = (double*)dst;
  // output vector k takes words first[k] .. : loads at la[k] and lb[k]; index < 8 -> from load a, >= 8 -> from load b
  static const int la[6] = {0, 11, 23, 36, 47, 59}, lb[6] = {8, 19, 31, 44, 55, 64};
  alignas(64) static long long idx[6][8];
  static bool init = false;
  if (!init) {
    static const int keep[6] = {0, 1, 2, 4, 5, 8};
    for (int k = 0; k < 6; ++k)
      for (int j = 0; j < 8; ++j) {
        const int o = 8 * k + j, word = 9 * (o / 6) + keep[o % 6];
        idx[k][j] = word >= lb[k] ? 8 + (word - lb[k]) : word - la[k];
      }
    init = true;
  }
  __m512i ix[6];
  for (int k = 0; k < 6; ++k) ix[k] = _mm512_load_si512(idx[k]);
  // symmetry: lanes of (load at 1 + 8v) vs (load at 3 + 8v) where word % 9 in {1, 5}; (load at 2 + 8v) vs (6 + 8v) where word % 9 == 2
  __mmask8 mA[9], mB[9];
  for (int v = 0; v < 9; ++v) {
    mA[v] = 0; mB[v] = 0;
    for (int j = 0; j < 8; ++j) {
      const int wa = 1 + 8 * v + j, wb = 2 + 8 * v + j;
      if (wa + 2 < 72 && (wa % 9 == 1 || wa % 9 == 5)) mA[v] |= (__mmask8)(1u << j);
      if (wb + 4 < 72 && wb % 9 == 2) mB[v] |= (__mmask8)(1u << j);
    }
  }
  unsigned bad = 0;
  size_t i = 0;
  for (; i + 8 <= cnt; i += 8, s += 72, d += 48) {
#pragma GCC unroll 6
    for (int k = 0; k < 6; ++k) {
      const __m512d a = _mm512_loadu_pd(s + la[k]), b = _mm512_loadu_pd(s + lb[k]);
      _mm512_stream_pd(d + 8 * k, _mm512_permutex2var_pd(a, ix[k], b));
    }
#pragma GCC unroll 9
    for (int v = 0; v < 9; ++v) {
      // (the last vectors' loads would run past the eight points: masked loads read nothing beyond the mask)
      const __m512i x1 = _mm512_maskz_loadu_epi64(mA[v], s + 1 + 8 * v), x3 = _mm512_maskz_loadu_epi64(mA[v], s + 3 + 8 * v);
      const __m512i x2 = _mm512_maskz_loadu_epi64(mB[v], s + 2 + 8 * v), x6 = _mm512_maskz_loadu_epi64(mB[v], s + 6 + 8 * v);
      bad |= _mm512_cmpneq_epi64_mask(x1, x3) | _mm512_cmpneq_epi64_mask(x2, x6);
    }
  }
  const uint64_t* w = (const uint64_t*)s;
  uint64_t bad2 = 0;
  for (; i < cnt; ++i, w += 9, d += 6) {
    bad2 |= (w[1] ^ w[3]) | (w[2] ^ w[6]) | (w[5] ^ w[7]);
    const uint64_t o[6] = {w[0], w[1], w[2], w[4], w[5], w[8]};
    std::memcpy(d, o, sizeof o);
  }
  _mm_sfence();
  return bad == 0 && bad2 == 0;
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 2;
  const size_t n = 100000, clouds = 160;   // 1.15 GB of source, every cloud read once per pass: cold
  std::vector<double*> src(clouds);
  for (auto& p : src) {
    p = (double*)std::malloc(n * 72);
    for (size_t i = 0; i < n; ++i) { double* c = p + 9 * i; for (int k = 0; k < 9; ++k) c[k] = (double)(i * 9 + k) + 0.25; c[3] = c[1]; c[6] = c[2]; c[7] = c[5]; }
  }
  char* dst = (char*)std::aligned_alloc(4096, n * 72);
  std::memset(dst, 0, n * 72);
  // correctness of the compacting shapes (also with one asymmetric entry, every kind, odd counts)
  for (auto fn : {compact_v1, compact_v3}) {
    if (fn == compact_v3 && !__builtin_cpu_supports("avx512f")) continue;
    for (size_t cnt : {size_t(1), size_t(7), size_t(8), size_t(9), size_t(2047), size_t(2048)}) {
      bool ok = fn(dst, (const char*)src[0], cnt);
      for (size_t i = 0; i < cnt && ok; ++i) { const double* c = src[0] + 9 * i; const double w[6] = {c[0], c[1], c[2], c[4], c[5], c[8]}; ok = std::memcmp(w, (double*)dst + 6 * i, 48) == 0; }
      if (!ok) { std::printf("WRONG result, cnt %zu\n", cnt); return 1; }
      const int pr[3][2] = {{1, 3}, {2, 6}, {5, 7}};
      for (int which = 0; which < 3; ++which) for (size_t at : {size_t(0), cnt / 2, cnt - 1}) {
        std::vector<double> s2(src[0], src[0] + cnt * 9);
        s2[at * 9 + pr[which][1]] += 1e-9;
        if (fn(dst, (const char*)s2.data(), cnt)) { std::printf("asymmetry missed, cnt %zu at %zu pair %d\n", cnt, at, which); return 1; }
      }
    }
  }
  auto run = [&](const char* name, auto fn) {
    double best = 1e9, sum = 0;
    for (size_t c = 0; c < clouds; ++c) {
      const auto t0 = std::chrono::steady_clock::now();
      std::atomic<size_t> next{0};
      auto work = [&] { for (;;) { const size_t u = next.fetch_add(2048); if (u >= n) return; fn(dst + u * 72, (const char*)(src[c] + u * 9), std::min<size_t>(2048, n - u)); } };
      std::vector<std::thread> th;
      for (int t = 1; t < threads; ++t) th.emplace_back(work);
      work();
      for (auto& x : th) x.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      best = std::min(best, ms); sum += ms;
    }
    std::printf("%-34s %d thread(s): mean %.3f ms  best %.3f ms per 100 000 covariances (7.2 MB in)\n", name, threads, sum / clouds, best);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("whole, streaming copy", [](char* d, const char* s, size_t k) { plain_copy(d, s, k * 72); return true; });
    run("compact v1 (avx2, 2 points/turn)", compact_v1);
    if (__builtin_cpu_supports("avx512f")) run("compact v3 (avx512, 8 points/turn)", compact_v3);
  }
  return 0;
}
