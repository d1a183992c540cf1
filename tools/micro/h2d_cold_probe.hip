// Micro-probe: 9.6 MB (2.4 MB + 7.2 MB) from PAGEABLE host buffers the HIP runtime has NEVER seen (a fresh cloud per
// frame, reference src/Registration.cpp:11) — how to get them onto the device fastest.  Every timed repetition uses
// its own freshly malloc'ed + written pair of buffers.
// build: hipcc -O3 --offload-arch=gfx950 -o h2d_cold_probe h2d_cold_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t a_bytes = 2400000, b_bytes = 7200000;

struct Pair { char* a; char* b; };
static Pair fresh(bool huge) {
  Pair p;
  if (huge) {
    p.a = static_cast<char*>(aligned_alloc(2 << 20, (a_bytes + (2 << 20) - 1) & ~size_t((2 << 20) - 1)));
    p.b = static_cast<char*>(aligned_alloc(2 << 20, (b_bytes + (2 << 20) - 1) & ~size_t((2 << 20) - 1)));
    madvise(p.a, a_bytes, MADV_HUGEPAGE);
    madvise(p.b, b_bytes, MADV_HUGEPAGE);
  } else {
    p.a = static_cast<char*>(malloc(a_bytes));
    p.b = static_cast<char*>(malloc(b_bytes));
  }
  memset(p.a, 1, a_bytes);
  memset(p.b, 2, b_bytes);
  return p;
}

int main() {
  char *da, *db, *pinned;
  hipMalloc(&da, a_bytes); hipMalloc(&db, b_bytes);
  hipHostMalloc(&pinned, 2 * (a_bytes + b_bytes), 0);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const double mb = (a_bytes + b_bytes) / 1e6;
  const int reps = 40;
  auto run = [&](const char* what, bool huge, const std::function<void(const Pair&)>& f) {
    std::vector<Pair> bufs;
    for (int i = 0; i < reps; ++i) bufs.push_back(fresh(huge));
    std::vector<double> t, t2;
    for (int i = 0; i < reps; ++i) { const double t0 = now(); f(bufs[i]); t.push_back((now() - t0) * 1e3); }
    for (int i = 0; i < reps; ++i) { const double t0 = now(); f(bufs[i]); t2.push_back((now() - t0) * 1e3); }  // seen now
    std::sort(t.begin(), t.end()); std::sort(t2.begin(), t2.end());
    printf("%-66s cold %.3f ms = %5.1f GB/s | seen %.3f ms = %5.1f GB/s\n", what, t[reps / 2], mb / t[reps / 2], t2[reps / 2], mb / t2[reps / 2]);
    for (auto& p : bufs) { free(p.a); free(p.b); }
  };
  auto chunked = [&](const Pair& p, size_t chunk) {
    for (size_t o = 0; o < a_bytes; o += chunk) hipMemcpyAsync(da + o, p.a + o, std::min(chunk, a_bytes - o), hipMemcpyHostToDevice, s1);
    for (size_t o = 0; o < b_bytes; o += chunk) hipMemcpyAsync(db + o, p.b + o, std::min(chunk, b_bytes - o), hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1);
  };
  run("two hipMemcpyAsync, one stream (current)", false, [&](const Pair& p) { chunked(p, 1 << 30); });
  run("chunks of 2 MB", false, [&](const Pair& p) { chunked(p, 2 << 20); });
  run("chunks of 1 MB", false, [&](const Pair& p) { chunked(p, 1 << 20); });
  run("chunks of 512 KB", false, [&](const Pair& p) { chunked(p, 512 << 10); });
  run("chunks of 256 KB", false, [&](const Pair& p) { chunked(p, 256 << 10); });
  run("chunks of 64 KB", false, [&](const Pair& p) { chunked(p, 64 << 10); });
  run("two threads, two streams", false, [&](const Pair& p) {
    std::thread t([&] { hipMemcpyAsync(da, p.a, a_bytes, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); });
    hipMemcpyAsync(db, p.b, b_bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); t.join(); });
  run("two threads, two streams, 4.8 MB each", false, [&](const Pair& p) {
    std::thread t([&] { hipMemcpyAsync(da, p.a, a_bytes, hipMemcpyHostToDevice, s2);
                        hipMemcpyAsync(db, p.b, a_bytes, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); });
    hipMemcpyAsync(db + a_bytes, p.b + a_bytes, b_bytes - a_bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); t.join(); });
  run("memcpy to pinned (1 thread) + 2 copies", false, [&](const Pair& p) {
    memcpy(pinned, p.a, a_bytes); memcpy(pinned + a_bytes, p.b, b_bytes);
    hipMemcpyAsync(da, pinned, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, pinned + a_bytes, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); });
  run("memcpy to pinned in 1 MB pieces, copy each piece as it is staged", false, [&](const Pair& p) {
    const size_t piece = 1 << 20; size_t off = 0;
    for (size_t o = 0; o < a_bytes; o += piece, off += piece) { const size_t n = std::min(piece, a_bytes - o);
      memcpy(pinned + off, p.a + o, n); hipMemcpyAsync(da + o, pinned + off, n, hipMemcpyHostToDevice, s1); }
    for (size_t o = 0; o < b_bytes; o += piece, off += piece) { const size_t n = std::min(piece, b_bytes - o);
      memcpy(pinned + off, p.b + o, n); hipMemcpyAsync(db + o, pinned + off, n, hipMemcpyHostToDevice, s1); }
    hipStreamSynchronize(s1); });
  run("the same with 4 staging threads", false, [&](const Pair& p) {
    const size_t piece = 1 << 20;
    struct Job { const char* src; char* dst; size_t off, n; };
    std::vector<Job> jobs; size_t off = 0;
    for (size_t o = 0; o < a_bytes; o += piece, off += piece) jobs.push_back({p.a + o, da + o, off, std::min(piece, a_bytes - o)});
    for (size_t o = 0; o < b_bytes; o += piece, off += piece) jobs.push_back({p.b + o, db + o, off, std::min(piece, b_bytes - o)});
    std::atomic<size_t> next{0};
    auto work = [&] { for (size_t j; (j = next.fetch_add(1)) < jobs.size();) { memcpy(pinned + jobs[j].off, jobs[j].src, jobs[j].n);
                        hipMemcpyAsync(jobs[j].dst, pinned + jobs[j].off, jobs[j].n, hipMemcpyHostToDevice, s1); } };
    std::thread t1(work), t2(work), t3(work); work(); t1.join(); t2.join(); t3.join();
    hipStreamSynchronize(s1); });
  run("hipHostRegister + copy + unregister", false, [&](const Pair& p) {
    hipHostRegister(p.a, a_bytes, 0); hipHostRegister(p.b, b_bytes, 0);
    hipMemcpyAsync(da, p.a, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, p.b, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); hipHostUnregister(p.a); hipHostUnregister(p.b); });
  {
    // where the time of the register path goes (cold buffers)
    std::vector<Pair> bufs;
    for (int i = 0; i < reps; ++i) bufs.push_back(fresh(false));
    double tr = 0, tc = 0, tu = 0;
    for (int i = 0; i < reps; ++i) {
      const Pair& p = bufs[i];
      double t0 = now();
      hipHostRegister(p.a, a_bytes, 0); hipHostRegister(p.b, b_bytes, 0);
      double t1 = now();
      hipMemcpyAsync(da, p.a, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, p.b, b_bytes, hipMemcpyHostToDevice, s1);
      hipStreamSynchronize(s1);
      double t2 = now();
      hipHostUnregister(p.a); hipHostUnregister(p.b);
      double t3 = now();
      tr += t1 - t0; tc += t2 - t1; tu += t3 - t2;
    }
    printf("register path, cold, mean: register %.3f ms, copies + sync %.3f ms, unregister %.3f ms\n", tr / reps * 1e3, tc / reps * 1e3, tu / reps * 1e3);
    for (auto& p : bufs) { free(p.a); free(p.b); }
  }
  run("register in 2 threads (one per array), copy, unregister in 2 threads", false, [&](const Pair& p) {
    std::thread t([&] { hipHostRegister(p.a, a_bytes, 0); hipMemcpyAsync(da, p.a, a_bytes, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); hipHostUnregister(p.a); });
    hipHostRegister(p.b, b_bytes, 0); hipMemcpyAsync(db, p.b, b_bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); hipHostUnregister(p.b);
    t.join(); });
  run("transparent huge pages (madvise), two hipMemcpyAsync", true, [&](const Pair& p) { chunked(p, 1 << 30); });
  run("transparent huge pages, chunks of 1 MB", true, [&](const Pair& p) { chunked(p, 1 << 20); });
  return 0;
}
