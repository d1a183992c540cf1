// Micro-probe: ways to get 9.6 MB (2.4 MB + 7.2 MB) of PAGEABLE host memory onto the device (MI355X box).
// build: hipcc -O3 --offload-arch=gfx950 -o h2d_probe h2d_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static double median_ms(F&& f, int reps = 30) {
  std::vector<double> t;
  for (int i = 0; i < 3; ++i) f();
  for (int i = 0; i < reps; ++i) { const double t0 = now(); f(); t.push_back((now() - t0) * 1e3); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const size_t a_bytes = 2400000, b_bytes = 7200000;
  char* ha = static_cast<char*>(malloc(a_bytes));
  char* hb = static_cast<char*>(malloc(b_bytes));
  memset(ha, 1, a_bytes); memset(hb, 2, b_bytes);
  char *da, *db, *pinned;
  hipMalloc(&da, a_bytes); hipMalloc(&db, b_bytes);
  hipHostMalloc(&pinned, a_bytes + b_bytes, 0);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const double mb = (a_bytes + b_bytes) / 1e6;
  auto report = [&](const char* what, double ms) { printf("%-62s %.3f ms = %.1f GB/s\n", what, ms, mb / ms); };
  report("two pageable hipMemcpyAsync on one stream + sync (current)", median_ms([&] {
    hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, hb, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); }));
  report("the same, large one first", median_ms([&] {
    hipMemcpyAsync(db, hb, b_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); }));
  report("two threads, two streams", median_ms([&] {
    std::thread t([&] { hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s2); hipStreamSynchronize(s2); });
    hipMemcpyAsync(db, hb, b_bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); t.join(); }));
  report("covariances in 4 chunks of 1.8 MB + points", median_ms([&] {
    for (int c = 0; c < 4; ++c) hipMemcpyAsync(db + c * (b_bytes / 4), hb + c * (b_bytes / 4), b_bytes / 4, hipMemcpyHostToDevice, s1);
    hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }));
  report("hipHostRegister both + copy + unregister", median_ms([&] {
    hipHostRegister(ha, a_bytes, 0); hipHostRegister(hb, b_bytes, 0);
    hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, hb, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); hipHostUnregister(ha); hipHostUnregister(hb); }));
  hipHostRegister(ha, a_bytes, 0); hipHostRegister(hb, b_bytes, 0);
  report("already registered (pinned) sources", median_ms([&] {
    hipMemcpyAsync(da, ha, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, hb, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); }));
  hipHostUnregister(ha); hipHostUnregister(hb);
  report("memcpy into one pinned buffer (1 thread) + one copy", median_ms([&] {
    memcpy(pinned, ha, a_bytes); memcpy(pinned + a_bytes, hb, b_bytes);
    hipMemcpyAsync(da, pinned, a_bytes, hipMemcpyHostToDevice, s1); hipMemcpyAsync(db, pinned + a_bytes, b_bytes, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1); }));
  return 0;
}
