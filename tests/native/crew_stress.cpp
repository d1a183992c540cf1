// CPU stress of the copy crew (eskf_lio_amd/csrc/vgicp_context.h: CopyCrew) without a device: tens of thousands of tiny
// jobs, posted with and without waking the helpers, so that a helper regularly comes late to a job that is already
// over while the next one is open.  Every unit of every job must be copied exactly once and finish() must return.
// Built and run by tests/test_capi_cpu.py (g++, host only; the header's HIP types come from the ROCm headers).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "vgicp_context.h"

static void plain_copy(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }

int main(int argc, char** argv) {
  const int jobs = argc > 1 ? std::atoi(argv[1]) : 200000;
  const int helpers = argc > 2 ? std::atoi(argv[2]) : 2;
  constexpr uint32_t kUnit = 8, kUnits = 6, kN = kUnit * kUnits - 3;   // a ragged last unit
  std::vector<char> src_a(kN * 24), src_b(kN * 8), dst_a(kN * 24), dst_b(kN * 8);
  std::vector<uint32_t> flags(16 * kUnits, 0);
  CopyCrew crew;
  crew.start(helpers);
  std::atomic<int> progress{0};
  std::atomic<bool> done{false};
  std::thread watchdog([&] {   // a lost unit shows as a finish() that never returns
    int last = -1;
    for (;;) {
      for (int k = 0; k < 100 && !done.load(); ++k) std::this_thread::sleep_for(std::chrono::milliseconds(100));
      if (done.load()) return;
      const int now = progress.load();
      if (now == last) { std::printf("STUCK at job %d\n", now); std::fflush(stdout); std::_Exit(2); }
      last = now;
    }
  });
  uint32_t seq = 0;
  for (int j = 0; j < jobs; ++j) {
    if (++seq == 0) ++seq;
    for (size_t k = 0; k < src_a.size(); ++k) src_a[k] = (char)(j + k);
    for (size_t k = 0; k < src_b.size(); ++k) src_b[k] = (char)(3 * j + k);
    std::memset(dst_a.data(), 0, dst_a.size());
    std::memset(dst_b.data(), 0, dst_b.size());
    crew.pts = src_a.data(); crew.cov = src_b.data();
    crew.apts = dst_a.data(); crew.acov = dst_b.data();
    crew.flags = flags.data();
    crew.n = kN; crew.unit = kUnit; crew.units = kUnits; crew.seq = seq;
    crew.size_a = 24; crew.size_b = 8;
    crew.copy = plain_copy;
    const uint32_t job = crew.post(j % 3 != 0);   // every third job is not announced: helpers still awake may come late to it
    if (j % 5 == 0) std::this_thread::yield();    // ... or to the one before
    crew.work(job);
    crew.finish();
    for (uint32_t u = 0; u < kUnits; ++u)
      if (__atomic_load_n(&flags[16 * u], __ATOMIC_ACQUIRE) != seq) { std::printf("job %d: unit %u not published\n", j, u); return 1; }
    if (std::memcmp(src_a.data(), dst_a.data(), src_a.size()) || std::memcmp(src_b.data(), dst_b.data(), src_b.size())) {
      std::printf("job %d: bytes differ\n", j);
      return 1;
    }
    progress.store(j + 1);
  }
  done.store(true);
  watchdog.join();
  crew.stop();
  std::printf("ok %d jobs\n", jobs);
  return 0;
}
