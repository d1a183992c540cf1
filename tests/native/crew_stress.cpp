// CPU stress of the copy crew (eskf_lio_amd/csrc/vgicp_context.h: CopyCrew) without a device: tens of thousands of tiny
// jobs, posted with and without waking the helpers, so that a helper regularly comes late to a job that is already
// over while the next one is open.  The shape of the job CHANGES from job to job (units 2 ... 6, a ragged last unit,
// with and without the second array): a late helper that checked its old ticket against the NEXT job's larger unit
// count took a unit with the fields half written (round 5's advisor: "job 3641: unit 5 not published") — jobs are now
// closed by finish() before the next one's fields are written.  Every unit of every job must be copied exactly once,
// every flag published, and finish() must return.
// Second part (argv[3] != 0): a helper that NEVER returns from its copy — finish() must come back with `false` within
// its deadline, the crew must work on alone, and a completion that lands late must not be counted for a later job.
// Built and run by tests/test_capi_cpu.py (g++, host only, also under ThreadSanitizer; the header's HIP types come from
// the ROCm headers).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "vgicp_context.h"

static void plain_copy(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }

static std::atomic<int> g_block{0};       // 1: the next copy made by a HELPER thread hangs until g_release
static std::atomic<int> g_blocked{0};
static std::atomic<bool> g_release{false};
static thread_local bool t_is_caller = false;
static void blocking_copy(void* d, const void* s, size_t n) {
  int one = 1;
  if (!t_is_caller && g_block.compare_exchange_strong(one, 0)) {
    g_blocked.store(1);
    while (!g_release.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  std::memcpy(d, s, n);
}

int main(int argc, char** argv) {
  const int jobs = argc > 1 ? std::atoi(argv[1]) : 200000;
  const int helpers = argc > 2 ? std::atoi(argv[2]) : 2;
  const bool stuck_part = argc > 3 && std::atoi(argv[3]) != 0;
  t_is_caller = true;
  constexpr uint32_t kUnit = 8, kMaxUnits = 6, kMaxN = kUnit * kMaxUnits;
  struct Buffers {
    std::vector<char> src_a, src_b, dst_a, dst_b;
    std::vector<uint32_t> flags;
    Buffers() : src_a(kMaxN * 24), src_b(kMaxN * 8), dst_a(kMaxN * 24), dst_b(kMaxN * 8), flags(16 * kMaxUnits, 0) {}
  };
  // the jobs of the hanging part use buffers of their own: the helper that hangs writes into them when it is let go, long
  // after its job was given up (in the module: the context's staging memory, which is kept alive for exactly that reason)
  Buffers plain_set, hang_set;
  Buffers* use = &plain_set;
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
  auto one_job = [&](int j, void (*copy)(void*, const void*, size_t), double deadline, bool* finished_ok) -> int {
    // the job's shape: 2, 6, 3, 5, 4, 6, 2 ... units, the last one ragged every other job, the second array every third
    const uint32_t units = 2 + (uint32_t)((j * 7) % 5);
    const uint32_t n = units * kUnit - (j % 2 ? 3 : 0);
    const uint32_t size_b = j % 3 == 2 ? 0 : 8;
    if (++seq == 0) ++seq;
    std::vector<char>&src_a = use->src_a, &src_b = use->src_b, &dst_a = use->dst_a, &dst_b = use->dst_b;
    std::vector<uint32_t>& flags = use->flags;
    for (size_t k = 0; k < (size_t)n * 24; ++k) src_a[k] = (char)(j + k);
    for (size_t k = 0; k < (size_t)n * 8; ++k) src_b[k] = (char)(3 * j + k);
    std::memset(dst_a.data(), 0, dst_a.size());
    std::memset(dst_b.data(), 0, dst_b.size());
    crew.pts = src_a.data(); crew.cov = size_b ? src_b.data() : nullptr;
    crew.apts = dst_a.data(); crew.acov = size_b ? dst_b.data() : nullptr;
    crew.flags = flags.data();
    crew.n = n; crew.unit = kUnit; crew.units = units; crew.seq = seq;
    crew.size_a = 24; crew.size_b = size_b;
    crew.copy = copy;
    const uint32_t job = crew.post(j % 3 != 0);   // every third job is not announced: helpers still awake may come late to it
    if (j % 5 == 0) std::this_thread::yield();    // ... or to the one before
    crew.work(job);
    const bool ok = crew.finish(deadline);
    if (finished_ok) *finished_ok = ok;
    if (!ok) return 0;
    for (uint32_t u = 0; u < units; ++u)
      if (__atomic_load_n(&flags[16 * u], __ATOMIC_ACQUIRE) != seq) { std::printf("job %d: unit %u not published\n", j, u); return 1; }
    if (std::memcmp(src_a.data(), dst_a.data(), (size_t)n * 24) || (size_b && std::memcmp(src_b.data(), dst_b.data(), (size_t)n * 8))) {
      std::printf("job %d: bytes differ\n", j);
      return 1;
    }
    return 0;
  };
  for (int j = 0; j < jobs; ++j) {
    if (one_job(j, plain_copy, 10.0, nullptr)) return 1;
    progress.store(j + 1);
  }
  if (stuck_part && helpers > 0) {
    // a helper hangs inside its copy: jobs are posted (announced) until one of them catches it; that job's finish()
    // must return false within its deadline and mark the crew broken
    g_block.store(1);
    use = &hang_set;
    bool caught = false;
    int j = jobs;
    for (; j < jobs + 200000 && !caught; ++j) {
      bool ok = true;
      if (one_job(j, blocking_copy, 0.3, &ok)) return 1;
      progress.store(j + 1);
      if (!ok) caught = true;
    }
    if (!caught) {
      if (!g_blocked.load()) { std::printf("no helper ever took a unit: nothing to test\n"); g_block.store(0); }
      else { std::printf("a helper hangs but finish() never reported it\n"); return 1; }
    } else {
      if (!crew.broken) { std::printf("finish() gave up but the crew is not marked broken\n"); return 1; }
      // the crew works on alone (no helper is woken any more), every job complete
      use = &plain_set;
      for (int k = 0; k < 2000; ++k, ++j) {
        if (one_job(j, plain_copy, 10.0, nullptr)) return 1;
        progress.store(j + 1);
      }
      // the late helper comes back in the middle of later jobs: its completion must go nowhere
      g_release.store(true);
      for (int k = 0; k < 20000; ++k, ++j) {
        if (one_job(j, plain_copy, 10.0, nullptr)) return 1;
        progress.store(j + 1);
      }
      std::printf("stuck helper: reported, crew went on alone, late completion dropped\n");
    }
  }
  g_release.store(true);
  done.store(true);
  watchdog.join();
  crew.broken = false;   // every helper is back by now: join them
  crew.stop();
  std::printf("ok %d jobs\n", jobs);
  return 0;
}
