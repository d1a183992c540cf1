// Micro-probe (round 5): how fast can a 9.6 MB scan (n x 24 B points + n x 72 B covariances, pageable, never seen by
// the runtime, FREED by the caller right after the call) reach the device as 12 SoA planes WITHOUT the runtime ever
// registering the caller's pages?  A crew of plain-memcpy threads (no HIP calls) fills a page-locked arena in units of
// U points; the device reads the ARENA directly (no blit): either (a) per-group pack launches enqueued by the caller as
// groups complete, or (b) ONE pack launch whose workgroups poll per-unit flags in the arena.
// build: hipcc -O3 --offload-arch=gfx950 -o stage_crew_probe stage_crew_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
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

__attribute__((target("avx2"))) static void stream_copy(char* dst, const char* src, size_t bytes) {
  size_t i = 0;
  for (; i + 128 <= bytes; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i));
    const __m256i b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64));
    const __m256i d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a);
    _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c);
    _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  if (i < bytes) memcpy(dst + i, src + i, bytes - i);
  _mm_sfence();
}

// ---- the crew: T-1 helper threads + optionally the caller; jobs = units of U points -----------------------------
struct Crew {
  std::vector<std::thread> th;
  std::atomic<uint32_t> gen{0}, next{0}, finished{0};
  std::atomic<bool> quit{false};
  // the job
  const char *pts = nullptr, *cov = nullptr;
  char *apts = nullptr, *acov = nullptr;
  volatile uint32_t* flags = nullptr;
  uint32_t n = 0, U = 0, units = 0, seq = 0;
  bool streaming = true;
  void work() {
    for (uint32_t u; (u = next.fetch_add(1, std::memory_order_relaxed)) < units;) {
      const size_t p0 = (size_t)u * U, cnt = std::min<size_t>(U, n - p0);
      if (streaming) { stream_copy(apts + p0 * 24, pts + p0 * 24, cnt * 24); stream_copy(acov + p0 * 72, cov + p0 * 72, cnt * 72); }
      else { memcpy(apts + p0 * 24, pts + p0 * 24, cnt * 24); memcpy(acov + p0 * 72, cov + p0 * 72, cnt * 72); std::atomic_thread_fence(std::memory_order_release); }
      __atomic_store_n(const_cast<uint32_t*>(flags + 16 * u), seq, __ATOMIC_RELEASE);
      finished.fetch_add(1, std::memory_order_release);
    }
  }
  void run(int id) {
    uint32_t seen = gen.load(std::memory_order_acquire);   // a thread that starts late must not take a stale job
    while (!quit.load(std::memory_order_relaxed)) {
      const uint32_t g = gen.load(std::memory_order_acquire);
      if (g == seen) { _mm_pause(); continue; }
      seen = g;
      work();
    }
  }
  void start(int helpers) { for (int i = 0; i < helpers; ++i) th.emplace_back([this, i] { run(i); }); }
  void stop() { quit = true; for (auto& t : th) t.join(); th.clear(); quit = false; }
  void post() { next.store(0); finished.store(0); gen.fetch_add(1, std::memory_order_release); }
};

// ---- device: read AoS from the arena, write 12 planes ------------------------------------------------------------
// one workgroup = BLOCK points; the 96*BLOCK bytes come in as coalesced 16-byte loads into LDS, go out plane by plane
template <int BLOCK>
__device__ __forceinline__ void pack_block(const char* apts, const char* acov, uint32_t p0, uint32_t cnt, double* soa, size_t stride,
                                           double* lds) {
  const uint32_t t = threadIdx.x;
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i* sp = reinterpret_cast<const v4i*>(apts + (size_t)p0 * 24);
  const v4i* sc = reinterpret_cast<const v4i*>(acov + (size_t)p0 * 72);
  v4i* lp = reinterpret_cast<v4i*>(lds);
  v4i* lc = reinterpret_cast<v4i*>(lds + 3 * BLOCK);
  const uint32_t np = (cnt * 24 + 15) / 16, nc = (cnt * 72 + 15) / 16;
  for (uint32_t k = t; k < np; k += BLOCK) lp[k] = __builtin_nontemporal_load(sp + k);
  for (uint32_t k = t; k < nc; k += BLOCK) lc[k] = __builtin_nontemporal_load(sc + k);
  __syncthreads();
  if (t < cnt) {
    const size_t i = p0 + t;
#pragma unroll
    for (int k = 0; k < 3; ++k) soa[k * stride + i] = lds[3 * t + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) soa[(3 + k) * stride + i] = lds[3 * BLOCK + 9 * t + k];
  }
  __syncthreads();
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pack_range_kernel(const char* apts, const char* acov, uint32_t first, uint32_t last,
                                                           double* soa, size_t stride) {
  __shared__ double lds[12 * BLOCK];
  const uint32_t p0 = first + blockIdx.x * BLOCK;
  if (p0 >= last) return;
  pack_block<BLOCK>(apts, acov, p0, min((uint32_t)BLOCK, last - p0), soa, stride, lds);
}

// (b) one launch: workgroup g takes blocks g, g + G, ...; a block waits for its unit's flag (host memory)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void pack_polling_kernel(const char* apts, const char* acov, uint32_t n, uint32_t U,
                                                             const uint32_t* flags, uint32_t seq, double* soa, size_t stride,
                                                             uint32_t spin_limit, uint32_t* gave_up) {
  __shared__ double lds[12 * BLOCK];
  __shared__ uint32_t ok;
  const uint32_t blocks = (n + BLOCK - 1) / BLOCK;
  uint32_t have_unit = 0xFFFFFFFFu;
  for (uint32_t b = blockIdx.x; b < blocks; b += gridDim.x) {
    const uint32_t p0 = b * BLOCK, cnt = min((uint32_t)BLOCK, n - p0);
    const uint32_t u_last = (p0 + cnt - 1) / U;   // a block may straddle two units: wait for the later one (units finish nearly in order) AND the earlier
    const uint32_t u_first = p0 / U;
    if (have_unit != u_last) {
      if (threadIdx.x == 0) {
        uint32_t spins = 0, good = 1;
        for (uint32_t u = u_first; u <= u_last; ++u) {
          while (__hip_atomic_load(flags + 16 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins > spin_limit) { good = 0; break; }
            __builtin_amdgcn_s_sleep(20);
          }
          if (!good) break;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        ok = good;
      }
      __syncthreads();
      if (!ok) { if (threadIdx.x == 0) *gave_up = seq; return; }
      have_unit = u_last;
    }
    pack_block<BLOCK>(apts, acov, p0, cnt, soa, stride, lds);
  }
}

__global__ void pack_simple_kernel(const double* pts, const double* covs, uint32_t n, double* soa, size_t stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 3; ++k) soa[k * stride + i] = pts[3 * (size_t)i + k];
  for (int k = 0; k < 9; ++k) soa[(3 + k) * stride + i] = covs[9 * (size_t)i + k];
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 100000;
  const int reps = argc > 2 ? atoi(argv[2]) : 60;
  const size_t a_bytes = (size_t)n * 24, b_bytes = (size_t)n * 72, stride = (n + 63) & ~63u;
  char *arena; double *soa, *daos; uint32_t* d_gave;
  const size_t arena_bytes = ((a_bytes + 255) & ~size_t(255)) + ((b_bytes + 255) & ~size_t(255)) + (1 << 16);
  hipHostMalloc((void**)&arena, arena_bytes, 0);
  memset(arena, 0, arena_bytes);
  hipMalloc((void**)&soa, stride * 12 * 8);
  hipMalloc((void**)&daos, a_bytes + b_bytes);
  hipMalloc((void**)&d_gave, 4); hipMemset(d_gave, 0, 4);
  char* apts = arena; char* acov = arena + ((a_bytes + 255) & ~size_t(255));
  uint32_t* flags = reinterpret_cast<uint32_t*>(acov + ((b_bytes + 255) & ~size_t(255)));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const double mb = (a_bytes + b_bytes) / 1e6;

  struct Pair { char* a; char* b; };
  auto fresh = [&] { Pair p{(char*)malloc(a_bytes), (char*)malloc(b_bytes)};
    for (size_t i = 0; i < a_bytes / 8; ++i) ((double*)p.a)[i] = (double)i * 0.5;
    for (size_t i = 0; i < b_bytes / 8; ++i) ((double*)p.b)[i] = (double)i * 0.25 + 1.0;
    return p; };
  std::vector<double> h_soa(stride * 12);
  auto check = [&](const char* what) {
    hipMemcpy(h_soa.data(), soa, stride * 12 * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (uint32_t i = 0; i < n; i += 97) {
      for (int k = 0; k < 3; ++k) bad += h_soa[k * stride + i] != (double)(3 * (size_t)i + k) * 0.5;
      for (int k = 0; k < 9; ++k) bad += h_soa[(3 + k) * stride + i] != (double)(9 * (size_t)i + k) * 0.25 + 1.0;
    }
    if (bad) printf("   !! %s: %zu wrong values\n", what, bad);
    hipMemset(soa, 0xFF, stride * 12 * 8);
    hipDeviceSynchronize();
  };
  // f(pair) does the whole upload + pack and returns when the planes are complete; the pair is FREED right after
  auto run = [&](const char* what, const std::function<void(const Pair&)>& f, bool free_in_loop = true) {
    std::vector<Pair> bufs;
    for (int i = 0; i < reps; ++i) bufs.push_back(fresh());
    // evict the CPU caches' idea of the last buffers
    std::vector<double> t, tf;
    f(bufs[0]); check(what);
    for (int i = 1; i < reps; ++i) {
      const double t0 = now(); f(bufs[i]); const double t1 = now();
      if (free_in_loop) { free(bufs[i].a); free(bufs[i].b); bufs[i].a = bufs[i].b = nullptr; }
      const double t2 = now();
      t.push_back((t1 - t0) * 1e3); tf.push_back((t2 - t1) * 1e3);
    }
    check(what);
    std::vector<double> ts = t; std::sort(ts.begin(), ts.end());
    std::vector<double> fs = tf; std::sort(fs.begin(), fs.end());
    int slow_free = 0, worst = 0, slow_call = 0;
    for (size_t i = 0; i < tf.size(); ++i) { slow_free += tf[i] > 1.0; if (tf[i] > tf[worst]) worst = (int)i; slow_call += t[i] > 1.0; }
    printf("%-78s median %.3f ms = %5.1f GB/s, p90 %.3f, max %.3f (%d > 1 ms) | free(): median %.3f max %.3f ms (#%d), %d > 1 ms\n", what, ts[ts.size() / 2],
           mb / ts[ts.size() / 2], ts[ts.size() * 9 / 10], ts.back(), slow_call, fs[fs.size() / 2], fs.back(), worst, slow_free);
    fflush(stdout);
    for (auto& p : bufs) { free(p.a); free(p.b); }
  };

  // --- references -------------------------------------------------------------------------------------------------
  run("staged, 1 thread, 384 KB pieces, hipMemcpyAsync each + simple pack (round 4)", [&](const Pair& p) {
    const size_t piece = 384 << 10;
    for (size_t o = 0; o < a_bytes; o += piece) { const size_t l = std::min(piece, a_bytes - o); stream_copy(apts + o, p.a + o, l);
      hipMemcpyAsync((char*)daos + o, apts + o, l, hipMemcpyHostToDevice, s); }
    for (size_t o = 0; o < b_bytes; o += piece) { const size_t l = std::min(piece, b_bytes - o); stream_copy(acov + o, p.b + o, l);
      hipMemcpyAsync((char*)daos + a_bytes + o, acov + o, l, hipMemcpyHostToDevice, s); }
    pack_simple_kernel<<<(n + 255) / 256, 256, 0, s>>>(daos, (double*)((char*)daos + a_bytes), n, soa, stride);
    hipStreamSynchronize(s); });

  // --- the device reading the arena directly, data already staged: what PCIe gives a kernel -----------------------
  {
    Pair p = fresh();
    stream_copy(apts, p.a, a_bytes); stream_copy(acov, p.b, b_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int block : {256, 512, 1024}) {
      float best = 1e9f;
      for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0, s);
        if (block == 256) pack_range_kernel<256><<<(n + 255) / 256, 256, 0, s>>>(apts, acov, 0, n, soa, stride);
        if (block == 512) pack_range_kernel<512><<<(n + 511) / 512, 512, 0, s>>>(apts, acov, 0, n, soa, stride);
        if (block == 1024) pack_range_kernel<1024><<<(n + 1023) / 1024, 1024, 0, s>>>(apts, acov, 0, n, soa, stride);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
      }
      check("pack_range");
      printf("pack_range_kernel<%d> reading the whole arena over PCIe: %.1f us = %.1f GB/s\n", block, best * 1e3, mb / best);
    }
    for (int grid : {16, 32, 64, 128, 256}) {
      float best = 1e9f;
      for (uint32_t k = 0; k < 64; ++k) flags[16 * k] = 7;
      for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0, s);
        pack_polling_kernel<1024><<<grid, 1024, 0, s>>>(apts, acov, n, 4096, flags, 7, soa, stride, 1000000, d_gave);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
      }
      check("pack_polling (flags preset)");
      printf("pack_polling_kernel<1024> grid %3d, flags already set: %.1f us = %.1f GB/s\n", grid, best * 1e3, mb / best);
    }
    {
      hipMemcpy(daos, p.a, a_bytes, hipMemcpyHostToDevice); hipMemcpy((char*)daos + a_bytes, p.b, b_bytes, hipMemcpyHostToDevice);
      float best = 1e9f;
      for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0, s);
        pack_range_kernel<256><<<(n + 255) / 256, 256, 0, s>>>((char*)daos, (char*)daos + a_bytes, 0, n, soa, stride);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
      }
      printf("pack_range_kernel<256> from DEVICE memory (for scale): %.1f us\n", best * 1e3);
    }
    free(p.a); free(p.b);
  }

  // --- CPU side alone: how fast does a crew fill the arena from cold buffers ----------------------------------------
  Crew crew;
  crew.apts = apts; crew.acov = acov; crew.flags = flags; crew.n = n;
  uint32_t seq = 100;
  for (int T : {1, 2, 3, 4, 8}) {
    crew.start(T - 1);
    for (uint32_t U : {2048u, 8192u}) {
      crew.U = U; crew.units = (n + U - 1) / U;
      char label[160];
      snprintf(label, sizeof label, "CPU only: %2d threads (caller included), units of %u points, streaming stores", T, U);
      for (int streaming = 1; streaming >= 0; --streaming) {
        crew.streaming = streaming != 0;
        if (!streaming) snprintf(label, sizeof label, "CPU only: %2d threads (caller included), units of %u points, libc memcpy", T, U);
        run(label, [&](const Pair& p) { crew.pts = p.a; crew.cov = p.b; crew.seq = ++seq; crew.post(); crew.work();
          while (crew.finished.load(std::memory_order_acquire) < crew.units) _mm_pause(); });
      }
      crew.streaming = true;
    }
    crew.stop();
  }

  // --- the pipelines ----------------------------------------------------------------------------------------------------
  for (int T : {1, 2, 3, 4, 8}) {
    crew.start(T);   // T helpers; the caller orchestrates only
    for (uint32_t U : {2048u, 4096u}) {
      crew.U = U; crew.units = (n + U - 1) / U;
      crew.streaming = true;
      for (uint32_t group : {4u, 8u}) {
        char label[160];
        snprintf(label, sizeof label, "(a) %2d helpers, U=%u, caller launches pack_range per %u units as they complete", T, U, group);
        run(label, [&](const Pair& p) { crew.pts = p.a; crew.cov = p.b; crew.seq = ++seq; crew.post();
          for (uint32_t u0 = 0; u0 < crew.units; u0 += group) {
            const uint32_t u1 = std::min(crew.units, u0 + group);
            for (uint32_t u = u0; u < u1; ++u) while (__atomic_load_n(const_cast<uint32_t*>(flags + 16 * u), __ATOMIC_ACQUIRE) != crew.seq) _mm_pause();
            const uint32_t first = u0 * U, last = std::min(n, u1 * U);
            pack_range_kernel<256><<<(last - first + 255) / 256, 256, 0, s>>>(apts, acov, first, last, soa, stride);
          }
          hipStreamSynchronize(s); });
      }
      for (int grid : {32, 64}) {
        char label[160];
        snprintf(label, sizeof label, "(b) %2d helpers, U=%u, ONE polling launch of %d x 1024 threads", T, U, grid);
        run(label, [&](const Pair& p) { crew.pts = p.a; crew.cov = p.b; crew.seq = ++seq; crew.post();
          pack_polling_kernel<1024><<<grid, 1024, 0, s>>>(apts, acov, n, U, flags, crew.seq, soa, stride, 200000, d_gave);
          hipStreamSynchronize(s); });
        snprintf(label, sizeof label, "(b') %2d helpers + the caller copying, U=%u, ONE polling launch of %d x 1024 threads", T, U, grid);
        run(label, [&](const Pair& p) { crew.pts = p.a; crew.cov = p.b; crew.seq = ++seq; crew.post();
          pack_polling_kernel<1024><<<grid, 1024, 0, s>>>(apts, acov, n, U, flags, crew.seq, soa, stride, 200000, d_gave);
          crew.work();
          hipStreamSynchronize(s); });
      }
    }
    crew.stop();
  }
  run("in place: two hipMemcpyAsync from pageable + simple pack (buffers kept)", [&](const Pair& p) {
    hipMemcpyAsync(daos, p.a, a_bytes, hipMemcpyHostToDevice, s); hipMemcpyAsync((char*)daos + a_bytes, p.b, b_bytes, hipMemcpyHostToDevice, s);
    pack_simple_kernel<<<(n + 255) / 256, 256, 0, s>>>(daos, (double*)((char*)daos + a_bytes), n, soa, stride);
    hipStreamSynchronize(s); }, false);
  run("in place, buffers FREED after each call (the registered-range stall)", [&](const Pair& p) {
    hipMemcpyAsync(daos, p.a, a_bytes, hipMemcpyHostToDevice, s); hipMemcpyAsync((char*)daos + a_bytes, p.b, b_bytes, hipMemcpyHostToDevice, s);
    pack_simple_kernel<<<(n + 255) / 256, 256, 0, s>>>(daos, (double*)((char*)daos + a_bytes), n, soa, stride);
    hipStreamSynchronize(s); });
  uint32_t gave = 0; hipMemcpy(&gave, d_gave, 4, hipMemcpyDeviceToHost);
  printf("polling kernel gave up: %s\n", gave ? "YES" : "never");
  return 0;
}
