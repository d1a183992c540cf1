// CPU check of the drop-in's "is this host cloud still the resident scan" hash (include/eskf_lio_shim/LocalMap.hpp,
// shim::bufferHash / sampleHash): the AVX2 lanes and the plain ones give the same value, every single-bit edit of a
// buffer changes it, so do a swap of two words inside one lane and a shifted run; ResidentCheck::Sampled sees an edit of a
// sampled element and misses one of an unsampled element (the documented price of that mode), FullHash sees both.
// The same hash from pieces summed by helper threads (shim::HashCrew) equals the one-thread value for changing sizes and
// helper counts (also under ThreadSanitizer).
// Links nothing of the module (the C ABI is stubbed out: the header only needs the declarations).
#define ESKF_LIO_SHIM_FORCE_POD 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "eskf_lio_shim/LocalMap.hpp"

using namespace ESKF_LIO;

static uint64_t plain_lanes(const void* p, size_t bytes, uint64_t seed) {
  uint64_t s1[16], s2[16];
  for (int l = 0; l < 16; ++l) { s1[l] = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1); s2[l] = 0; }
  const uint64_t* w = static_cast<const uint64_t*>(p);
  for (size_t i = 0; i < bytes / 8; ++i) { const size_t l = i & 15u; s1[l] += w[i]; s2[l] += s1[l]; }
  uint64_t h = bytes * 0x100000001B3ull;
  for (int l = 0; l < 16; ++l) {
    h = (h ^ s1[l]) * 0x9FB21C651E98DF25ull; h ^= h >> 29;
    h = (h ^ s2[l]) * 0xC2B2AE3D27D4EB4Full; h ^= h >> 31;
  }
  return h;
}

int main(int argc, char** argv) {
  for (size_t words : {size_t(1), size_t(15), size_t(16), size_t(17), size_t(333), size_t(27000 * 12)}) {
    std::vector<uint64_t> buf(words);
    for (size_t i = 0; i < words; ++i) buf[i] = (i + 1) * 0x9E3779B97F4A7C15ull ^ (i << 7);
    const uint64_t h0 = shim::bufferHash(buf.data(), words * 8, 42);
    if (h0 != plain_lanes(buf.data(), words * 8, 42)) { std::printf("AVX2 and plain lanes differ at %zu words\n", words); return 1; }
    const size_t step = words > 4000 ? 997 : 1;
    for (size_t i = 0; i < words; i += step) {
      buf[i] ^= 1ull << (i % 64);
      if (shim::bufferHash(buf.data(), words * 8, 42) == h0) { std::printf("a single-bit edit of word %zu of %zu is not seen\n", i, words); return 1; }
      buf[i] ^= 1ull << (i % 64);
    }
    if (words > 40) {
      std::swap(buf[5], buf[21]);     // two words of one lane
      if (shim::bufferHash(buf.data(), words * 8, 42) == h0) { std::printf("a swap inside one lane is not seen\n"); return 1; }
      std::swap(buf[5], buf[21]);
      std::vector<uint64_t> shifted(buf.begin() + 1, buf.end());
      shifted.push_back(buf[0]);      // the same words, rotated by one
      if (shim::bufferHash(shifted.data(), words * 8, 42) == h0) { std::printf("a rotated buffer is not seen\n"); return 1; }
    }
  }
  // the same value from per-lane sums (what the fetch kernel posts): A = sum of the lane's words, B = sum of (m - k) x word
  for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(5), size_t(16), size_t(4999), size_t(10131)}) {
    PointCloud c;
    c.points_.resize(n);
    c.covariances_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      for (int a = 0; a < 3; ++a) c.points_[i].v[a] = -7.5 + 0.37 * (double)(3 * i + a);
      for (int k = 0; k < 9; ++k) c.covariances_[i].m[k] = 1e-3 * (double)((9 * i + k) % 101) - 0.02;
    }
    uint64_t sums[64] = {0};
    const uint64_t* arrays[2] = {reinterpret_cast<const uint64_t*>(c.points_.data()), reinterpret_cast<const uint64_t*>(c.covariances_.data())};
    const size_t words[2] = {3 * n, 9 * n};
    for (int a = 0; a < 2; ++a) {
      for (size_t i = 0; i < words[a]; ++i) {
        const size_t l = i & 15u, m = (words[a] - l + 15u) >> 4;
        sums[32 * a + l] += arrays[a][i];
        sums[32 * a + 16 + l] += (uint64_t)(m - (i >> 4)) * arrays[a][i];
      }
    }
    if (shim::fullHashFromSums(sums, n) != shim::sampleHash(c, false)) { std::printf("the hash from per-lane sums differs at %zu points\n", n); return 1; }
  }
  PointCloud cloud;
  cloud.points_.resize(5000);
  cloud.covariances_.resize(5000);
  for (size_t i = 0; i < 5000; ++i) {
    for (int a = 0; a < 3; ++a) cloud.points_[i].v[a] = 0.001 * (double)(3 * i + a);
    for (int k = 0; k < 9; ++k) cloud.covariances_[i].m[k] = 1.0 + 1e-6 * (double)(9 * i + k);
  }
  const uint64_t full = shim::sampleHash(cloud, false), sampled = shim::sampleHash(cloud, true);
  cloud.points_[0].v[0] += 1e-3;   // element 0 is sampled
  if (shim::sampleHash(cloud, false) == full || shim::sampleHash(cloud, true) == sampled) { std::printf("an edit of a sampled element is not seen\n"); return 1; }
  cloud.points_[0].v[0] -= 1e-3;
  cloud.covariances_[1].m[4] += 1e-9;   // element 1 of 5 000 is not
  if (shim::sampleHash(cloud, false) == full) { std::printf("FullHash misses an edit of an unsampled element\n"); return 1; }
  if (shim::sampleHash(cloud, true) != sampled) { std::printf("Sampled was expected to miss this edit (the test's premise)\n"); return 1; }
  // the same value from pieces summed on helper threads (shim::HashCrew), for changing sizes and helper counts, with the
  // caller taking part (sampleHash) and with the caller away (begin ... finish, as ICP::align runs it beside the device call)
  {
    shim::HashCrew & crew = shim::HashCrew::instance();
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3000;
    for (int k = 0; k < rounds; ++k) {
      const size_t n = 2500 + (size_t)(k * 37 % 9000);
      PointCloud c;
      c.points_.resize(n);
      c.covariances_.resize(n);
      for (size_t i = 0; i < n; i += 1 + n / 300) {
        c.points_[i].v[k % 3] = 0.25 * (double)(i + (size_t)k);
        c.covariances_[i].m[k % 9] = 1e-3 * (double)(i ^ (size_t)k);
      }
      crew.setHelpers(0);
      const uint64_t alone = shim::sampleHash(c, false);
      crew.setHelpers(1 + k % 3);
      if (shim::sampleHash(c, false) != alone) { std::printf("round %d: the crew's hash differs (caller taking part)\n", k); return 1; }
      shim::FullHashJob job;
      shim::planFullHash(c, job, crew.helpers());
      crew.begin(job.chunks, job.count);
      if (k % 5 == 0) { volatile uint64_t spin = 0; for (int q = 0; q < 20000; ++q) spin += (uint64_t)q; }   // "the device call"
      crew.finish();
      if (shim::foldFullHash(job) != alone) { std::printf("round %d: the crew's hash differs (caller away)\n", k); return 1; }
      c.covariances_[n / 2].m[4] += 1e-9;
      if (shim::sampleHash(c, false) == alone) { std::printf("round %d: an edit is not seen through the crew\n", k); return 1; }
    }
  }
  std::printf("ok\n");
  return 0;
}
