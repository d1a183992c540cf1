// LocalMap.hpp — drop-in for the reference's include/ESKF_LIO/LocalMap.hpp + src/LocalMap.cpp.
//
// Same class name, namespace and public methods as the reference (include/ESKF_LIO/LocalMap.hpp:
// 28-61 constructors, :91-98 methods), so src/Odometry.cpp:61,86 and src/ErrorStateKF.cpp:115-130
// compile against it unchanged.  What differs is where the data lives:
//   * the host std::unordered_map stays authoritative (save() and the running-mean insertion rule of
//     Voxel::addPoint, LocalMap.hpp:79-87, need it) and is updated exactly as
//     LocalMap::updateLocalMap does (src/LocalMap.cpp:10-76);
//   * every voxel touched by an update is forwarded to the device mirror as ONE vgicp_map_upsert
//     batch, every evicted voxel as ONE vgicp_map_erase batch, so the HIP registration path reads a
//     table that is current before the next ICP::align;
//   * correspondenceMatching() (src/LocalMap.cpp:78-112) runs on the device through vgicp_match and
//     returns the reference's tuple (srcPoints, srcCovs, mapPoints, mapCovs) in ascending point order
//     (the reference's order is thread-arrival order, i.e. unspecified).
// Deviations, both documented in DESIGN.md: prevTransform_ is initialised (the reference reads it
// uninitialised on the first frame, LocalMap.hpp:113 / LocalMap.cpp:39,134) so the first update
// always inserts; Open3D GUI calls are dropped (visualizeLocalMap() returns true without a window).
// There is no CPU fallback: every method that needs the device throws std::runtime_error with the
// library's message when the HIP module reports an error.
#ifndef ESKF_LIO_SHIM_LOCAL_MAP_HPP_
#define ESKF_LIO_SHIM_LOCAL_MAP_HPP_

#include <chrono>
#include <cmath>
#if defined(__x86_64__) && (defined(__GNUC__) || defined(__clang__))
#include <immintrin.h>
#endif
#if defined(__linux__)
#include <sys/mman.h>
#endif
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../vgicp_hip.h"
#include "ShimTypes.hpp"

#if defined(ESKF_LIO_SHIM_NATIVE_TYPES) && __has_include(<yaml-cpp/yaml.h>)
#include <yaml-cpp/yaml.h>
#define ESKF_LIO_SHIM_HAVE_YAML 1
#endif

namespace ESKF_LIO
{

// The keys LocalMap's YAML constructor reads (config/hilti_config.yaml:36-45).
struct LocalMapConfig
{
  double voxelSize = 0.3;
  size_t maxNumPointsPerVoxel = 1000;
  double translationSquaredThreshold = 1.0e-2;
  double cosineThreshold = 0.985;
  bool removeDistantPoints = true;
  double distanceThreshold = 100.0;
  double removePeriod = 10.0;
  // false: the host std::unordered_map is authoritative and the device mirror is fed batches (keeps
  //        every raw point for save(), as the reference does).
  // true:  the voxel grid the registration reads lives on the device — insertion and eviction run there
  //        (vgicp_map_insert_resident_async / vgicp_map_evict, same arithmetic, same results, nothing waited for).
  //        With keepRawPoints the raw points every voxel holds (what save() writes, src/LocalMap.cpp:156-167) are kept
  //        in a host-side SHADOW of the grid that a worker thread of this object maintains from the prepared clouds
  //        updateLocalMap is handed (the reference's own insertion and eviction loops, in the reference's order, off
  //        the caller's thread): save() then writes exactly what the reference writes.  The shadow needs the prepared
  //        scan on the host (CloudPreprocessorConfig::HostCopy::Eager, the default); with a deferred host copy, or
  //        with keepRawPoints = false, save() writes one point per voxel (its mean).
  // Default since round 5: true / true — the classes as a maintainer gets them by swapping the headers are the fast
  // ones (0.5 - 0.6 ms per 60 000-point frame instead of 3.6 - 4.5), and save() still writes the reference's content.
  bool deviceResident = true;
  bool keepRawPoints = true;
};

namespace shim
{
inline void check(vgicp_ctx * ctx, int rc, const char * what)
{
  if (rc != VGICP_OK) {
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " +
            vgicp_last_error(ctx));
  }
}

// Developer aid: where a frame's host time goes inside the classes (tools/probe_eager.py through libvgicp_host.so).
// Off unless shim::trace().on is set; a disabled scope costs one predictable branch.
struct Trace
{
  enum Slot {ProcessEnqueue, ProcessWait, ProcessResize, ProcessDownload, ProcessStamp, AlignVerify, AlignCall,
    UpdateVerify, UpdateRest, UpdateInsert, UpdateShadow, Slots};   // (the last two are parts of UpdateRest)
  bool on = false;
  double seconds[Slots] = {0};
  uint64_t calls[Slots] = {0};
};
inline Trace & trace()
{
  static Trace t;
  return t;
}
struct TraceScope
{
  int slot;
  std::chrono::steady_clock::time_point t0;
  explicit TraceScope(int s)
  : slot(trace().on ? s : -1)
  {
    if (slot >= 0) {t0 = std::chrono::steady_clock::now();}
  }
  ~TraceScope()
  {
    if (slot >= 0) {
      trace().seconds[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      ++trace().calls[slot];
    }
  }
};

// Storage of clouds that died inside these classes (a cloud moved into updateLocalMap, src/Odometry.cpp:86, ends in the
// map's hands) is kept for the clouds to come: process() needs 72 bytes per kept point for the covariances of every
// frame, and a FRESH allocation of that size is what the C library maps anew and the kernel faults in page by page
// (measured: 0.44 ms of a 1.1 ms frame were `covariances_.resize()`); storage that was used before costs nothing.
// At most four vectors of each kind are kept; everything else is freed as before.
struct StoragePool
{
  std::mutex mutex;
  std::vector<std::vector<Vector3d>> points;
  std::vector<std::vector<Matrix3d>> covariances;
};
inline StoragePool & storagePool()
{
  static StoragePool pool;
  return pool;
}
// the cloud is about to be destroyed by its last owner: keep its buffers (four of a kind at most: a fifth replaces the
// smallest one kept when it is larger)
template<typename T>
inline void keepStorage(std::vector<T> & v, std::vector<std::vector<T>> & kept)
{
  if (!v.capacity()) {return;}
  v.clear();
  if (kept.size() < 4) {
    kept.emplace_back(std::move(v));
    return;
  }
  size_t smallest = 0;
  for (size_t i = 1; i < kept.size(); ++i) {
    if (kept[i].capacity() < kept[smallest].capacity()) {smallest = i;}
  }
  if (kept[smallest].capacity() < v.capacity()) {kept[smallest].swap(v);}
}
inline void recycleStorage(PointCloud & cloud)
{
  StoragePool & pool = storagePool();
  std::lock_guard<std::mutex> lk(pool.mutex);
  keepStorage(cloud.covariances_, pool.covariances);
  keepStorage(cloud.points_, pool.points);
}
// make room for n elements in v, out of the pool when v has none of its own (v's contents are not kept): the smallest
// kept buffer that is large enough
template<typename T>
inline void adoptStorage(std::vector<T> & v, std::vector<std::vector<T>> & kept, size_t n)
{
  if (v.capacity() >= n) {return;}
  {
    StoragePool & pool = storagePool();
    std::lock_guard<std::mutex> lk(pool.mutex);
    size_t best = kept.size();
    for (size_t i = 0; i < kept.size(); ++i) {
      if (kept[i].capacity() >= n && (best == kept.size() || kept[i].capacity() < kept[best].capacity())) {best = i;}
    }
    if (best != kept.size()) {
      v.swap(kept[best]);
      v.clear();
      kept.erase(kept.begin() + static_cast<std::ptrdiff_t>(best));
      return;
    }
  }
  // nothing to reuse (the clouds of the last frames are still with the shadow grid's worker, or were smaller): a fresh
  // allocation with room for the next frames' sizes (a scan's kept count moves by a few per cent from frame to frame:
  // an exact fit would send every other frame here), its pages brought in by ONE call instead of one fault each
  // (Linux >= 5.14; ignored where it is not known)
  v.reserve(n + n / 4 + 64);
#if defined(__linux__)
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(v.data()) + 4095u) & ~uintptr_t(4095u);
  const uintptr_t hi = reinterpret_cast<uintptr_t>(v.data() + v.capacity()) & ~uintptr_t(4095u);
  if (hi > lo + (256u << 10)) {(void)madvise(reinterpret_cast<void *>(lo), hi - lo, 23 /* MADV_POPULATE_WRITE */);}
#endif
}

// One context per process, created on first use: device $VGICP_DEVICE (default 0), or — VGICP_DEVICES=0,1,2,3 — ONE
// context that drives several devices from this thread (vgicp_create_multi: replicated map, point-sharded align;
// an ordinal may repeat, "0,0", to split one device).  The reference's single caller thread (src/main.cpp:68-70)
// reaches the multi-GPU path through the unchanged ICP::align that way.
inline vgicp_ctx * defaultContext()
{
  static vgicp_ctx * ctx = [] {
      vgicp_ctx * c = nullptr;
      int rc;
      if (const char * list = std::getenv("VGICP_DEVICES")) {
        std::vector<int> ids;
        for (const char * p = list; *p; ) {
          char * end = nullptr;
          const long v = std::strtol(p, &end, 10);
          if (end == p) {break;}
          ids.push_back(static_cast<int>(v));
          p = (*end == ',') ? end + 1 : end;
        }
        if (ids.empty()) {throw std::runtime_error("VGICP_DEVICES names no device");}
        rc = vgicp_create_multi(ids.data(), static_cast<int>(ids.size()), &c);
      } else {
        int dev = 0;
        if (const char * env = std::getenv("VGICP_DEVICE")) {dev = std::atoi(env);}
        rc = vgicp_create(dev, &c);
      }
      if (rc != VGICP_OK) {
        throw std::runtime_error(std::string("vgicp_create failed: ") + vgicp_last_error(nullptr));
      }
      return c;
    }();
  return ctx;
}

// ---- "this host cloud IS the scan that is resident on the device" --------------------------------------------
// CloudPreprocessor::process leaves the prepared scan on the device and stamps the host cloud; ICP::align and
// LocalMap::updateLocalMap (src/Odometry.cpp:74,79,86 hand the SAME cloud from one to the next) recognise the stamp
// and work on the resident scan instead of uploading the cloud again.  The stamp is the cloud's address, its
// buffers' addresses and sizes, a hash over the buffers' contents and the library's
// scan generation (VGICP_COUNTER_SCAN_GENERATION: anything else that replaced the resident scan voids it).  A
// cloud that was resized, reallocated or edited in place — ANY byte of either buffer: by default every byte is hashed
// (ResidentCheck::FullHash, ~35 GB/s out of the caches: 0.07 ms per check of a 27 000-point prepared cloud) —
// falls back to the upload path, which is what the reference does with every cloud (src/Registration.cpp:11,
// src/LocalMap.cpp:45-58 always read the host cloud).  ResidentCheck::Sampled hashes 64 evenly spaced elements of each
// buffer instead (first and last included): ~1 us, but an edit of an UNSAMPLED element in place is not seen — only for
// callers that never edit a prepared cloud in place, or call shim::forget(cloud) when they do.  Chosen through the
// configuration (CloudPreprocessorConfig::residentCheck — the class that makes the stamp —, YAML key
// cloud_preprocessor.resident_check: sampled), never through the environment; a stamp remembers how it was made, and
// ICP::align / LocalMap::updateLocalMap check it the way it was made.
enum class ResidentCheck {FullHash, Sampled};
struct ResidentStamp
{
  vgicp_ctx * ctx = nullptr;
  const void * cloud = nullptr;
  const void * pointData = nullptr;
  const void * covData = nullptr;
  size_t pointCount = 0, covCount = 0;
  uint64_t hash = 0, generation = 0;
  bool sampled = false;      // the hash covers 64 elements of each buffer only
  bool wantSampled = false;  // what the configuration asked for (ResidentCheck::Sampled); `sampled` is also set for a deferred host copy
  size_t kept = 0;           // points of the resident scan, when known (0 while the preparation has not reported)
  bool hostIsCurrent = false;  // the host buffers hold the prepared scan (false: the raw sweep, the scan is on the device only)
};
inline std::vector<ResidentStamp> & residentStamps()
{
  static std::vector<ResidentStamp> stamps;
  return stamps;
}
// Every byte of a buffer in one pass at the speed the caches deliver it: 16 interleaved lanes of 64-bit words, each a
// pair of running sums (s1 += w; s2 += s1 — position-dependent, so a changed word, a swapped pair or a shifted run all
// show), folded with odd multipliers at the end.  Not cryptographic: a change detector for buffers nobody attacks.
// AVX2 when the CPU has it (four 4-lane vectors), the same lanes in plain C++ otherwise — the same value either way.
#if defined(__x86_64__) && (defined(__GNUC__) || defined(__clang__))
#define ESKF_LIO_SHIM_HASH_AVX2 1
__attribute__((target("avx2"))) inline void lanesAvx2(const uint64_t * w, size_t blocks, uint64_t (&s1)[16], uint64_t (&s2)[16])
{
  __m256i a[4], b[4];
  for (int v = 0; v < 4; ++v) {
    a[v] = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s1 + 4 * v));
    b[v] = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s2 + 4 * v));
  }
  for (size_t k = 0; k < blocks; ++k, w += 16) {
    for (int v = 0; v < 4; ++v) {
      a[v] = _mm256_add_epi64(a[v], _mm256_loadu_si256(reinterpret_cast<const __m256i *>(w + 4 * v)));
      b[v] = _mm256_add_epi64(b[v], a[v]);
    }
  }
  for (int v = 0; v < 4; ++v) {
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(s1 + 4 * v), a[v]);
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(s2 + 4 * v), b[v]);
  }
}
#endif
// The sums of a run of full 16-word blocks, from zero: A[l] = the lane's words added up, B[l] = its running sums added
// up (= sum of (blocks - k) x word k).  Runs that follow one another combine (blocks' weights shift by what comes after),
// so a buffer can be summed in pieces, on several cores (HashCrew below) or by the device (vgicp_scan_fetch_sums).
struct HashChunk
{
  const uint64_t * w = nullptr;
  size_t blocks = 0;
  uint64_t A[16], B[16];
};
inline void laneSums(HashChunk & c)
{
  for (int l = 0; l < 16; ++l) {c.A[l] = 0; c.B[l] = 0;}
  size_t done = 0;
#ifdef ESKF_LIO_SHIM_HASH_AVX2
  static const bool wide = __builtin_cpu_supports("avx2");
  if (wide) {lanesAvx2(c.w, c.blocks, c.A, c.B); done = c.blocks;}
#endif
  for (size_t k = done; k < c.blocks; ++k) {
    for (size_t l = 0; l < 16; ++l) {
      c.A[l] += c.w[16 * k + l];
      c.B[l] += c.A[l];
    }
  }
}
// consecutive chunks of one buffer -> the sums of all their blocks
inline void combineChunks(const HashChunk * chunks, int count, uint64_t (&A)[16], uint64_t (&B)[16])
{
  for (int l = 0; l < 16; ++l) {A[l] = 0; B[l] = 0;}
  size_t after = 0;
  for (int i = count - 1; i >= 0; --i) {
    for (int l = 0; l < 16; ++l) {
      A[l] += chunks[i].A[l];
      B[l] += chunks[i].B[l] + static_cast<uint64_t>(after) * chunks[i].A[l];
    }
    after += chunks[i].blocks;
  }
}
// the hash of a buffer whose full blocks were summed (A, B): the lanes' start values, the last partial block, the fold
inline uint64_t finishBufferHash(const void * p, size_t bytes, uint64_t seed, const uint64_t (&A)[16], const uint64_t (&B)[16])
{
  const uint64_t * w = static_cast<const uint64_t *>(p);
  const size_t words = bytes / 8, blocks = words / 16;
  uint64_t s1[16], s2[16];
  for (int l = 0; l < 16; ++l) {
    const uint64_t c = seed + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(l + 1);
    s1[l] = c + A[l];
    s2[l] = static_cast<uint64_t>(blocks) * c + B[l];
  }
  for (size_t i = blocks * 16; i < words; ++i) {
    const size_t l = i & 15u;
    s1[l] += w[i];
    s2[l] += s1[l];
  }
  uint64_t h = bytes * 0x100000001B3ull;
  for (int l = 0; l < 16; ++l) {
    h = (h ^ s1[l]) * 0x9FB21C651E98DF25ull;
    h ^= h >> 29;
    h = (h ^ s2[l]) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 31;
  }
  return h;
}
inline uint64_t bufferHash(const void * p, size_t bytes, uint64_t seed)
{
  HashChunk c;
  c.w = static_cast<const uint64_t *>(p);
  c.blocks = bytes / 8 / 16;
  laneSums(c);
  return finishBufferHash(p, bytes, seed, c.A, c.B);
}
// A few helper threads that sum chunks beside the caller (or instead of it, while the caller waits for the device): the
// full-hash check reads 96 bytes per point twice a frame, ~70 us each on one core for a 35 000-point scan.  Chunks are
// taken from one atomic word that carries the job's number (a helper that comes late for a job finds the word closed
// or the next job's number and takes nothing); finish() takes what is left itself, so a job ends even when no helper
// ever runs.  Helpers spin for ~100 us after a job (a frame's second check follows its first closely), then sleep.
class HashCrew
{
public:
  static HashCrew & instance()
  {
    static HashCrew crew;
    return crew;
  }
  // how many helper threads jobs may use (0: the caller alone); threads are started when first needed
  void setHelpers(int n) {wanted_.store(n < 0 ? 0 : (n > 3 ? 3 : n), std::memory_order_relaxed);}
  int helpers() const {return wanted_.load(std::memory_order_relaxed);}
  // the chunks' laneSums start on the helpers; finish() must follow (same thread), the chunks stay where they are until then
  void begin(HashChunk * chunks, int count)
  {
    const int want = helpers();
    if (want > static_cast<int>(threads_.size())) {
      std::lock_guard<std::mutex> lk(mutex_);
      while (static_cast<int>(threads_.size()) < want) {threads_.emplace_back([this] {loop();});}
    }
    chunks_.store(chunks, std::memory_order_relaxed);   // (a helper late for the last job may look: its exchange then fails)
    count_.store(count, std::memory_order_relaxed);
    done_.store(0, std::memory_order_relaxed);
    job_ = (job_ + 1u) & 0x7FFFFFFFu;
    ticket_.store(static_cast<uint64_t>(job_) << 32, std::memory_order_seq_cst);     // open, chunk 0 next
    if (want > 0 && sleepers_.load(std::memory_order_seq_cst) > 0) {
      {std::lock_guard<std::mutex> lk(mutex_);}
      wake_.notify_all();
    }
  }
  void finish()
  {
    while (takeOne()) {}
    ticket_.store((static_cast<uint64_t>(job_) << 32) | kClosed, std::memory_order_seq_cst);
    // chunks a helper has taken are being summed (no lock, no device in there): they arrive
    for (uint32_t spins = 0; done_.load(std::memory_order_acquire) != count_.load(std::memory_order_relaxed); ++spins) {
      if (spins < 4096u) {pauseCpu();} else {std::this_thread::yield();}
    }
  }
  ~HashCrew()
  {
    {
      std::lock_guard<std::mutex> lk(mutex_);
      quit_.store(true, std::memory_order_seq_cst);
    }
    wake_.notify_all();
    for (auto & t : threads_) {t.join();}
  }

private:
  static constexpr uint64_t kClosed = 0xFFFFFFFFull;
  static void pauseCpu()
  {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  // one chunk of the open job, if there is one left: summed here
  bool takeOne()
  {
    for (;;) {
      uint64_t t = ticket_.load(std::memory_order_acquire);
      const uint64_t index = t & 0xFFFFFFFFull;
      if (index == kClosed) {return false;}
      HashChunk * chunks = chunks_.load(std::memory_order_relaxed);   // this job's, if the exchange below succeeds (the word
      const int count = count_.load(std::memory_order_relaxed);       //  only changes job when closed)
      if (index >= static_cast<uint64_t>(count)) {return false;}
      if (!ticket_.compare_exchange_weak(t, t + 1u, std::memory_order_acq_rel, std::memory_order_acquire)) {continue;}
      laneSums(chunks[index]);
      done_.fetch_add(1, std::memory_order_release);
      return true;
    }
  }
  void loop()
  {
    for (;;) {
      if (takeOne()) {continue;}
      // nothing to take: watch the word for a while, then sleep until begin() says so
      const uint64_t seen = ticket_.load(std::memory_order_acquire);
      const auto t0 = std::chrono::steady_clock::now();
      bool changed = false;
      for (uint32_t spins = 0; !changed; ++spins) {
        pauseCpu();
        changed = ticket_.load(std::memory_order_acquire) != seen || quit_.load(std::memory_order_relaxed);
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) {break;}
      }
      if (quit_.load(std::memory_order_seq_cst)) {return;}
      if (changed) {continue;}
      std::unique_lock<std::mutex> lk(mutex_);
      sleepers_.fetch_add(1, std::memory_order_seq_cst);
      wake_.wait(lk, [&] {return quit_.load(std::memory_order_seq_cst) || ticket_.load(std::memory_order_seq_cst) != seen;});
      sleepers_.fetch_sub(1, std::memory_order_seq_cst);
      if (quit_.load(std::memory_order_seq_cst)) {return;}
    }
  }
  std::mutex mutex_;
  std::condition_variable wake_;
  std::vector<std::thread> threads_;
  std::atomic<uint64_t> ticket_{kClosed};
  std::atomic<int> done_{0}, sleepers_{0}, wanted_{2};
  std::atomic<bool> quit_{false};
  std::atomic<HashChunk *> chunks_{nullptr};
  std::atomic<int> count_{0};
  uint32_t job_ = 0;
};
// sampleHash(cloud, false) in pieces: the chunks of both buffers (points first), summed anywhere, then folded
struct FullHashJob
{
  static constexpr int kMaxChunks = 16;
  HashChunk chunks[kMaxChunks];
  int count = 0, pointChunks = 0;
  const PointCloud * cloud = nullptr;
};
inline void planFullHash(const PointCloud & cloud, FullHashJob & job, int workers)
{
  job.cloud = &cloud;
  job.count = 0;
  const size_t bytesP = cloud.points_.size() * sizeof(Vector3d), bytesC = cloud.covariances_.size() * sizeof(Matrix3d);
  const size_t blocksP = bytesP / 128, blocksC = bytesC / 128;
  // pieces of at least 64 KB, about two per worker so that whoever is faster takes more
  const size_t total = blocksP + blocksC;
  size_t pieces = static_cast<size_t>(workers < 1 ? 1 : workers) * 2;
  const size_t most = total / 512 ? total / 512 : 1;
  pieces = pieces > most ? most : pieces;
  pieces = pieces > static_cast<size_t>(FullHashJob::kMaxChunks - 2) ? static_cast<size_t>(FullHashJob::kMaxChunks - 2) : pieces;
  const size_t per = (total + pieces - 1) / pieces;
  auto cut = [&](const void * p, size_t blocks) {
      const uint64_t * w = static_cast<const uint64_t *>(p);
      size_t at = 0;
      do {
        const size_t take = blocks - at < per + per / 4 ? blocks - at : per;   // no sliver at the end
        HashChunk & c = job.chunks[job.count++];
        c.w = w + 16 * at;
        c.blocks = take;
        at += take;
      } while (at < blocks && job.count < FullHashJob::kMaxChunks - 1);
      if (at < blocks) {job.chunks[job.count - 1].blocks += blocks - at;}
    };
  if (bytesP) {cut(cloud.points_.data(), blocksP);}
  job.pointChunks = job.count;
  if (bytesC) {cut(cloud.covariances_.data(), blocksC);}
}
inline uint64_t foldFullHash(const FullHashJob & job)
{
  const PointCloud & cloud = *job.cloud;
  const size_t n = cloud.points_.size(), m = cloud.covariances_.size();
  uint64_t h = n * 0x100000001B3ull ^ m;
  uint64_t A[16], B[16];
  if (n) {
    combineChunks(job.chunks, job.pointChunks, A, B);
    h = finishBufferHash(cloud.points_.data(), n * sizeof(Vector3d), h, A, B);
  }
  if (m) {
    combineChunks(job.chunks + job.pointChunks, job.count - job.pointChunks, A, B);
    h = finishBufferHash(cloud.covariances_.data(), m * sizeof(Matrix3d), h, A, B);
  }
  return h;
}
// The same value from sums the DEVICE made while it wrote the buffer (vgicp_scan_fetch_sums): a lane that starts at c
// and adds m words ends with s1 = c + A and s2 = m c + B, A the sum of its words and B the sum of (m - k) x word.
inline uint64_t bufferHashFromSums(const uint64_t * A, const uint64_t * B, size_t bytes, uint64_t seed)
{
  const size_t words = bytes / 8;
  uint64_t h = bytes * 0x100000001B3ull;
  for (size_t l = 0; l < 16; ++l) {
    const uint64_t c = seed + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(l + 1);
    const uint64_t m = (words - l + 15u) >> 4;
    h = (h ^ (c + A[l])) * 0x9FB21C651E98DF25ull;
    h ^= h >> 29;
    h = (h ^ (m * c + B[l])) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 31;
  }
  return h;
}
// sampleHash(cloud, false) of a cloud of n points and n covariances whose bytes the fetch kernel summed
inline uint64_t fullHashFromSums(const uint64_t (&sums)[64], size_t n)
{
  uint64_t h = n * 0x100000001B3ull ^ n;
  if (n) {
    h = bufferHashFromSums(sums, sums + 16, n * sizeof(Vector3d), h);
    h = bufferHashFromSums(sums + 32, sums + 48, n * sizeof(Matrix3d), h);
  }
  return h;
}
inline uint64_t sampleHash(const PointCloud & cloud, bool sampled)
{
  const size_t n = cloud.points_.size(), m = cloud.covariances_.size();
  if (!sampled) {
    HashCrew & crew = HashCrew::instance();
    FullHashJob job;
    const bool shared = crew.helpers() > 0 && (n * sizeof(Vector3d) + m * sizeof(Matrix3d)) >= (256u << 10);
    planFullHash(cloud, job, shared ? crew.helpers() + 1 : 1);
    if (shared) {
      crew.begin(job.chunks, job.count);
      crew.finish();
    } else {
      for (int i = 0; i < job.count; ++i) {laneSums(job.chunks[i]);}
    }
    return foldFullHash(job);
  }
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](const void * p, size_t bytes) {
      const unsigned char * b = static_cast<const unsigned char *>(p);
      for (size_t i = 0; i < bytes; ++i) {h = (h ^ b[i]) * 1099511628211ull;}
    };
  for (size_t k = 0; k < 64 && n; ++k) {mix(&cloud.points_[k * (n - 1) / 63], sizeof(Vector3d));}
  for (size_t k = 0; k < 64 && m; ++k) {mix(&cloud.covariances_[k * (m - 1) / 63], sizeof(Matrix3d));}
  return h;
}
inline uint64_t scanGeneration(vgicp_ctx * ctx)
{
  uint64_t g = 0;
  check(ctx, vgicp_get_counter(ctx, VGICP_COUNTER_SCAN_GENERATION, &g), "vgicp_get_counter");
  return g;
}
inline ResidentStamp * findStamp(vgicp_ctx * ctx)
{
  for (auto & st : residentStamps()) {
    if (st.ctx == ctx) {return &st;}
  }
  return nullptr;
}
inline void stampResident(
  vgicp_ctx * ctx, const PointCloud & cloud, size_t kept, bool hostIsCurrent,
  ResidentCheck how = ResidentCheck::FullHash, const uint64_t * knownFullHash = nullptr)
{
  ResidentStamp * st = findStamp(ctx);
  if (!st) {
    residentStamps().push_back(ResidentStamp{});
    st = &residentStamps().back();
  }
  st->ctx = ctx;
  st->cloud = &cloud;
  st->pointData = cloud.points_.data();
  st->covData = cloud.covariances_.data();
  st->pointCount = cloud.points_.size();
  st->covCount = cloud.covariances_.size();
  // a host cloud that does NOT hold the prepared scan (HostCopy::Deferred: it still holds the raw sweep, and its contract
  // says "materialize before you touch it") has no content the device's copy could be compared with: its stamp guards
  // the object's identity (address, buffers, sizes, 64 samples), whatever the configuration asks for
  st->wantSampled = how == ResidentCheck::Sampled;
  st->sampled = st->wantSampled || !hostIsCurrent;
  // knownFullHash: the full hash of exactly these bytes, made elsewhere (by the device as it delivered them)
  st->hash = knownFullHash && !st->sampled ? *knownFullHash : sampleHash(cloud, st->sampled);
  st->generation = scanGeneration(ctx);
  st->kept = kept;
  st->hostIsCurrent = hostIsCurrent;
}
// The stamp of `cloud` if object, buffers, sizes and the context's scan generation still are what was stamped (the
// cheap part of the check: the contents are NOT looked at), else nullptr.
inline ResidentStamp * residentIdentityOf(vgicp_ctx * ctx, const PointCloud & cloud)
{
  ResidentStamp * st = findStamp(ctx);
  if (!st || st->cloud != &cloud || st->pointData != cloud.points_.data() ||
    st->covData != cloud.covariances_.data() || st->pointCount != cloud.points_.size() ||
    st->covCount != cloud.covariances_.size())
  {
    return nullptr;
  }
  return st->generation == scanGeneration(ctx) ? st : nullptr;
}
// The stamp of `cloud` if it still is the resident scan of ctx, else nullptr.
inline ResidentStamp * residentStampOf(vgicp_ctx * ctx, const PointCloud & cloud)
{
  ResidentStamp * st = residentIdentityOf(ctx, cloud);
  return st && st->hash == sampleHash(cloud, st->sampled) ? st : nullptr;
}
inline void forget(vgicp_ctx * ctx)
{
  if (ResidentStamp * st = findStamp(ctx)) {st->cloud = nullptr;}
}
inline void forget(const PointCloud & cloud)
{
  for (auto & st : residentStamps()) {
    if (st.cloud == &cloud) {st.cloud = nullptr;}
  }
}
// The host buffers of a cloud whose prepared scan lives on the device only (CloudPreprocessorConfig::HostCopy::
// Deferred) are filled now: one synchronisation and one download.  No-op for any other cloud.
inline void materialize(vgicp_ctx * ctx, PointCloud & cloud)
{
  ResidentStamp * st = residentStampOf(ctx, cloud);
  if (!st || st->hostIsCurrent) {return;}
  size_t n = 0;
  check(ctx, vgicp_scan_download(ctx, 0, nullptr, nullptr, &n), "vgicp_scan_download");
  cloud.points_.resize(n);
  cloud.covariances_.resize(n);
  if (n) {
    check(
      ctx, vgicp_scan_download(
        ctx, n, reinterpret_cast<double *>(cloud.points_.data()),
        reinterpret_cast<double *>(cloud.covariances_.data()), &n), "vgicp_scan_download");
  }
  stampResident(ctx, cloud, n, true, st->wantSampled ? ResidentCheck::Sampled : ResidentCheck::FullHash);
}
}  // namespace shim

class LocalMap
{
public:
  struct Voxel;

  using PointVector = typename std::vector<Vector3d>;
  using CovarianceVector = typename std::vector<Matrix3d>;
  using Correspondence = typename std::tuple<PointVector, CovarianceVector, PointVector,
      CovarianceVector>;

  struct Key
  {
    int32_t i, j, k;
    bool operator==(const Key & o) const {return i == o.i && j == o.j && k == o.k;}
  };
  // open3d::utility::hash_eigen<Eigen::Vector3i> (boost-style combine); only iteration order
  // depends on it.
  struct VoxelHash
  {
    size_t operator()(const Key & key) const
    {
      size_t seed = 0;
      const int32_t e[3] = {key.i, key.j, key.k};
      for (int n = 0; n < 3; ++n) {
        seed ^= std::hash<int>()(e[n]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
      }
      return seed;
    }
  };
  using VoxelGrid = typename std::unordered_map<Key, Voxel, VoxelHash>;

  struct Voxel
  {
    size_t maxNumPoints;
    size_t numPoints;
    PointVector points;
    Vector3d mean;
    Matrix3d covariance;
    bool dirty = false;  // touched since the last device sync

    Voxel(size_t maxNumPoints_, const Vector3d & point, const Matrix3d & covariance_)
    : maxNumPoints(maxNumPoints_), numPoints(1), mean(point), covariance(covariance_)
    {
      points.reserve(maxNumPoints);
      points.push_back(point);
    }

    // returns true when the voxel changed
    bool addPoint(const Vector3d & point, const Matrix3d & covariance_)
    {
      if (numPoints >= maxNumPoints) {return false;}
      points.push_back(point);
      const double n = static_cast<double>(numPoints), n1 = static_cast<double>(numPoints + 1);
      for (int a = 0; a < 3; ++a) {mean(a) = (n * mean(a) + point(a)) / n1;}
      for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) {
          covariance(r, c) = (n * covariance(r, c) + covariance_(r, c)) / n1;
        }
      }
      ++numPoints;
      return true;
    }
  };

  explicit LocalMap(const LocalMapConfig & config, bool visualize = false, vgicp_ctx * ctx = nullptr)
  : voxelSize_(config.voxelSize)
    , maxNumPointsPerVoxel_(config.maxNumPointsPerVoxel)
    , translationSquaredThreshold_(config.translationSquaredThreshold)
    , cosineThreshold_(config.cosineThreshold)
    , removeDistantPoints_(config.removeDistantPoints)
    , distanceThreshold_(config.distanceThreshold)
    , removePeriod_(config.removePeriod)
    , deviceResident_(config.deviceResident)
    , keepRawPoints_(config.keepRawPoints)
    , visualize_(visualize)
    , ctx_(ctx ? ctx : shim::defaultContext())
  {
    shim::check(ctx_, vgicp_map_reset(ctx_, voxelSize_, 0), "vgicp_map_reset");
  }

  ~LocalMap() {shadowStop();}
  LocalMap(const LocalMap &) = delete;
  LocalMap & operator=(const LocalMap &) = delete;

  // reference: LocalMap(double voxelSize, size_t maxNumPointsPerVoxel, bool visualize = false).
  // The reference leaves the update thresholds uninitialised here (LocalMap.hpp:54-61); this one
  // disables the motion gate and the eviction instead, i.e. every update inserts.
  LocalMap(double voxelSize, size_t maxNumPointsPerVoxel, bool visualize = false,
    vgicp_ctx * ctx = nullptr)
  : voxelSize_(voxelSize)
    , maxNumPointsPerVoxel_(maxNumPointsPerVoxel)
    , translationSquaredThreshold_(-1.0)
    , cosineThreshold_(2.0)
    , removeDistantPoints_(false)
    , distanceThreshold_(std::numeric_limits<double>::infinity())
    , removePeriod_(std::numeric_limits<double>::infinity())
    , visualize_(visualize)
    , ctx_(ctx ? ctx : shim::defaultContext())
  {
    shim::check(ctx_, vgicp_map_reset(ctx_, voxelSize_, 0), "vgicp_map_reset");
  }

#if defined(ESKF_LIO_SHIM_HAVE_YAML)
  // reference: LocalMap(const YAML::Node &, const PinholeCameraParameters &, bool visualize = true)
  LocalMap(
    const YAML::Node & config, const open3d::camera::PinholeCameraParameters &,
    bool visualize = true)
  : LocalMap(fromYaml(config), visualize) {}

  static LocalMapConfig fromYaml(const YAML::Node & config)
  {
    LocalMapConfig c;
    const auto & m = config["local_map"];
    c.voxelSize = m["voxel_size"].as<double>();
    c.maxNumPointsPerVoxel = m["max_num_points_per_voxel"].as<size_t>();
    c.translationSquaredThreshold = m["update"]["translation_sq_threshold"].as<double>();
    c.cosineThreshold = m["update"]["cosine_threshold"].as<double>();
    c.removeDistantPoints = m["remove_distant_points"]["enabled"].as<bool>();
    c.distanceThreshold = m["remove_distant_points"]["distance_threshold"].as<double>();
    c.removePeriod = m["remove_distant_points"]["removing_period"].as<double>();
    return c;
  }
#endif

  // reference: src/LocalMap.cpp:10-76. The cloud is moved into the world frame in place, as there.
  void updateLocalMap(PointCloudPtr cloud, const Isometry3d & transform, bool initialize = false)
  {
    // a cloud that ends its life here (moved in by its only owner, src/Odometry.cpp:86, and not passed on to the shadow
    // grid's worker) leaves its buffers to the frames to come
    struct Recycler
    {
      PointCloudPtr & c;
      ~Recycler() {if (c && c.use_count() == 1) {shim::recycleStorage(*c);}}
    } recycler{cloud};
    // The cloud CloudPreprocessor::process prepared and ICP::align registered is still resident on the device
    // (src/Odometry.cpp:74,79,86 pass the same cloud along): with the grid on the device the insertion runs there
    // on that resident scan, enqueued only — no upload, nothing waited for.  The host cloud is moved into the world
    // frame as the reference does only when it holds the prepared scan (eager host copy).
    shim::ResidentStamp * resident = nullptr;
    {
      shim::TraceScope ts(shim::Trace::UpdateVerify);
      resident = deviceResident_ ? shim::residentStampOf(ctx_, *cloud) : nullptr;
    }
    shim::TraceScope tsRest(shim::Trace::UpdateRest);
    if (resident) {
      const bool hostIsCurrent = resident->hostIsCurrent;
      trajectory_.push_back(transform);
      const bool insert = initialize || !hasPrevTransform_ || needsMapUpdate(transform);
      bool evicted = false;
      if (insert) {
        {
          shim::TraceScope tsInsert(shim::Trace::UpdateInsert);
          shim::check(
            ctx_, vgicp_map_insert_resident_async(ctx_, shim::poseData(transform), maxNumPointsPerVoxel_),
            "vgicp_map_insert_resident_async");
        }
        if (removeDistantPoints_ && now() - currentRemoveTime_ > removePeriod_) {
          evicted = true;
          const Vector3d position = transform.translation();
          const double pos[3] = {position(0), position(1), position(2)};
          size_t numRemovedVoxels = 0;
          shim::check(ctx_, vgicp_map_evict(ctx_, pos, distanceThreshold_, &numRemovedVoxels), "vgicp_map_evict");
          currentRemoveTime_ = now();
          std::cout << "removed " << numRemovedVoxels << " voxels\n";
        }
        hasPrevTransform_ = true;
      }
      prevTransform_ = transform;
      // The host side of the same update, for save(): the cloud moved into the world frame in place (src/LocalMap.cpp:15)
      // and, when it was inserted, the reference's insertion loop on the shadow grid -- on the worker thread when nobody
      // else can see the cloud (the caller moved its pointer in, src/Odometry.cpp:86), else the transform at least here.
      if (hostIsCurrent && keepRawPoints_ && shadowComplete_) {
        const bool mine = cloud.use_count() == 1;
        ShadowOp op;
        if (mine) {
          op.cloud = std::move(cloud);
        } else {                       // somebody else still holds the cloud: it is moved here and now, the worker gets a copy
          cloud->Transform(transform.matrix());
          op.cloud = std::make_shared<PointCloud>(*cloud);
        }
        op.transform = transform;
        op.transformFirst = mine;
        op.insert = insert;
        op.evict = evicted;
        op.position = transform.translation();
        shim::TraceScope tsShadow(shim::Trace::UpdateShadow);
        shadowPush(std::move(op));
      } else {
        if (hostIsCurrent) {cloud->Transform(transform.matrix());}   // in place, as src/LocalMap.cpp:15 (the stamp is void now)
        if (insert) {shadowComplete_ = false;}   // this frame's points never reached the host: save() falls back to the means
      }
      shim::forget(ctx_);
      return;
    }
    shim::materialize(ctx_, *cloud);   // a cloud whose prepared scan is on the device only: the host map needs the data
    cloud->Transform(transform.matrix());
    trajectory_.push_back(transform);

    if (initialize == false && hasPrevTransform_ && needsMapUpdate(transform) == false) {
      prevTransform_ = transform;
      return;
    }

    const auto & points = cloud->points_;
    const auto & covariances = cloud->covariances_;
    if (deviceResident_) {
      // the cloud is already in the world frame (transformed in place above, as the reference does)
      const Isometry3d identity = Isometry3d::Identity();
      if (!points.empty()) {
        shim::check(
          ctx_, vgicp_map_insert_scan(
            ctx_, points.size(), points.data()->data(), covariances.data()->data(),
            shim::poseData(identity), maxNumPointsPerVoxel_, nullptr), "vgicp_map_insert_scan");
      }
      bool evicted = false;
      if (removeDistantPoints_ && now() - currentRemoveTime_ > removePeriod_) {
        const Vector3d position = transform.translation();
        const double pos[3] = {position(0), position(1), position(2)};
        size_t numRemovedVoxels = 0;
        shim::check(ctx_, vgicp_map_evict(ctx_, pos, distanceThreshold_, &numRemovedVoxels), "vgicp_map_evict");
        currentRemoveTime_ = now();
        evicted = true;
        std::cout << "removed " << numRemovedVoxels << " voxels\n";
      }
      if (keepRawPoints_ && shadowComplete_) {
        ShadowOp op;
        // the worker thread reads the cloud later: it gets the caller's object only when nobody else can reach it
        // (src/Odometry.cpp:86 moves its pointer in), else a copy — a caller that keeps its pointer may edit or resize
        // the cloud as soon as this call returns
        if (cloud.use_count() == 1) {op.cloud = std::move(cloud);} else {op.cloud = std::make_shared<PointCloud>(*cloud);}
        op.transformFirst = false;
        op.insert = true;
        op.evict = evicted;
        op.position = transform.translation();
        shadowPush(std::move(op));
      }
      prevTransform_ = transform;
      hasPrevTransform_ = true;
      return;
    }
    std::vector<Voxel *> touched;
    std::vector<Key> touchedKeys;
    for (size_t i = 0; i < points.size(); ++i) {
      const Key key = toKey(getVoxelIndex(points[i]));
      auto found = voxelGrid_.find(key);
      Voxel * voxel = nullptr;
      bool changed = true;
      if (found == voxelGrid_.end()) {
        voxel = &voxelGrid_.emplace(key, Voxel(maxNumPointsPerVoxel_, points[i], covariances[i]))
          .first->second;
      } else {
        voxel = &found->second;
        changed = voxel->addPoint(points[i], covariances[i]);
      }
      if (changed && !voxel->dirty) {
        voxel->dirty = true;
        touched.push_back(voxel);
        touchedKeys.push_back(key);
      }
    }

    std::vector<int32_t> erased;
    if (removeDistantPoints_ && now() - currentRemoveTime_ > removePeriod_) {
      size_t numRemovedVoxels = 0;
      const Vector3d position = transform.translation();
      for (auto it = voxelGrid_.begin(); it != voxelGrid_.end(); ) {
        if (needsPointRemoval(it->first, position)) {
          erased.push_back(it->first.i);
          erased.push_back(it->first.j);
          erased.push_back(it->first.k);
          it->second.dirty = false;
          it = voxelGrid_.erase(it);
          ++numRemovedVoxels;
        } else {
          ++it;
        }
      }
      currentRemoveTime_ = now();
      std::cout << "removed " << numRemovedVoxels << " voxels\n";
    }

    syncDevice(touched, touchedKeys, erased);
    prevTransform_ = transform;
    hasPrevTransform_ = true;
  }

  // reference: src/LocalMap.cpp:78-112 (device lookup; ascending point order).
  Correspondence correspondenceMatching(
    const PointVector & points, const CovarianceVector & covariances) const
  {
    Correspondence correspondence;
    auto & [srcPoints, srcCovs, mapPoints, mapCovs] = correspondence;
    const size_t n = points.size();
    srcPoints.resize(n);
    srcCovs.resize(n);
    mapPoints.resize(n);
    mapCovs.resize(n);
    size_t matched = 0;
    if (n > 0) {
      shim::check(
        ctx_, vgicp_match(
          ctx_, n, points.data()->data(), covariances.data()->data(),
          srcPoints.data()->data(), srcCovs.data()->data(), mapPoints.data()->data(),
          mapCovs.data()->data(), nullptr, &matched), "vgicp_match");
    }
    srcPoints.resize(matched);
    srcCovs.resize(matched);
    mapPoints.resize(matched);
    mapCovs.resize(matched);
    return correspondence;
  }

  // reference: src/LocalMap.cpp:120-130 polls an Open3D window; there is none here.
  bool visualizeLocalMap() const {return true;}

  // reference: src/LocalMap.cpp:156-167 writes a .pcd through Open3D and a PinholeCameraTrajectory
  // JSON. Written here without Open3D: ASCII PCD v0.7 — of every stored point when the host map is
  // authoritative (as the reference writes), of ONE point per voxel (its mean, read back with
  // vgicp_map_export) in deviceResident mode, where the raw points are not kept — and the 4x4 poses as a
  // JSON array of column-major "extrinsic" arrays (the field Open3D's trajectory reader uses).
  void save(const std::string & cloud_path, const std::string & trajectory_path) const
  {
    shadowDrain();
    std::vector<double> deviceMeans;
    if (deviceResident_ && !(keepRawPoints_ && shadowComplete_)) {
      const size_t n = size();
      std::vector<int32_t> keys(3 * n);
      std::vector<double> covs(9 * n);
      std::vector<uint64_t> counts(n);
      deviceMeans.resize(3 * n);
      size_t written = 0;
      if (n) {
        shim::check(
          ctx_, vgicp_map_export(
            ctx_, n, keys.data(), deviceMeans.data(), covs.data(), counts.data(),
            &written), "vgicp_map_export");
      }
      deviceMeans.resize(3 * written);
    }
    size_t total = deviceMeans.size() / 3;
    for (const auto & kv : voxelGrid_) {total += kv.second.points.size();}
    std::ofstream pcd(cloud_path);
    pcd << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 8 8 8\n"
        << "TYPE F F F\nCOUNT 1 1 1\nWIDTH " << total << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
        << "POINTS " << total << "\nDATA ascii\n";
    pcd.precision(17);
    for (const auto & kv : voxelGrid_) {
      for (const auto & p : kv.second.points) {pcd << p(0) << ' ' << p(1) << ' ' << p(2) << '\n';}
    }
    for (size_t v = 0; v + 2 < deviceMeans.size(); v += 3) {
      pcd << deviceMeans[v] << ' ' << deviceMeans[v + 1] << ' ' << deviceMeans[v + 2] << '\n';
    }
    std::ofstream traj(trajectory_path);
    traj.precision(17);
    traj << "{\n\"class_name\" : \"PinholeCameraTrajectory\",\n\"parameters\" : [\n";
    for (size_t t = 0; t < trajectory_.size(); ++t) {
      const double * m = shim::poseData(trajectory_[t]);
      traj << "{ \"extrinsic\" : [";
      for (int e = 0; e < 16; ++e) {traj << (e ? ", " : " ") << m[e];}
      traj << " ] }" << (t + 1 < trajectory_.size() ? ",\n" : "\n");
    }
    traj << "],\n\"version_major\" : 1,\n\"version_minor\" : 0\n}\n";
  }

  // ---- additions (not in the reference) ----
  size_t size() const
  {
    if (!deviceResident_) {return voxelGrid_.size();}
    size_t voxels = 0;
    shim::check(ctx_, vgicp_map_size(ctx_, &voxels, nullptr), "vgicp_map_size");
    return voxels;
  }
  bool deviceResident() const {return deviceResident_;}
  double voxelSize() const {return voxelSize_;}
  vgicp_ctx * context() const {return ctx_;}
  // the host's voxel grid: authoritative without deviceResident, else the shadow kept for save() (brought up to date first)
  const VoxelGrid & grid() const
  {
    shadowDrain();
    return voxelGrid_;
  }
  // true while save() will write every stored raw point, as the reference does (false once a frame was inserted on the
  // device whose prepared scan never reached the host)
  bool savesRawPoints() const {return !deviceResident_ || (keepRawPoints_ && shadowComplete_);}

private:
  static Key toKey(const Vector3i & v) {return Key{v(0), v(1), v(2)};}
  static double now()
  {
    return std::chrono::duration<double>(
      std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  // reference: src/LocalMap.cpp:114-118
  Vector3i getVoxelIndex(const Vector3d & point) const
  {
    Vector3i idx;
    for (int a = 0; a < 3; ++a) {idx(a) = static_cast<int>(std::floor(point(a) / voxelSize_));}
    return idx;
  }

  // reference: src/LocalMap.cpp:132-147
  bool needsMapUpdate(const Isometry3d & transform) const
  {
    const Isometry3d moved = prevTransform_.inverse() * transform;
    const auto R = moved.linear();
    const double cosine = 0.5 * (R(0, 0) + R(1, 1) + R(2, 2) - 1.0);
    if (cosine < cosineThreshold_) {return true;}
    const auto t = moved.translation();
    const double translationSq = t(0) * t(0) + t(1) * t(1) + t(2) * t(2);
    if (translationSq > translationSquaredThreshold_) {return true;}
    return false;
  }

  // reference: src/LocalMap.cpp:149-154
  bool needsPointRemoval(const Key & key, const Vector3d & currentPos) const
  {
    const double c[3] = {(key.i + 0.5) * voxelSize_, (key.j + 0.5) * voxelSize_,
      (key.k + 0.5) * voxelSize_};
    const double d0 = c[0] - currentPos(0), d1 = c[1] - currentPos(1), d2 = c[2] - currentPos(2);
    return std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) > distanceThreshold_;
  }

  // ---- the shadow grid's worker (deviceResident + keepRawPoints) --------------------------------------------------
  struct ShadowOp
  {
    PointCloudPtr cloud;
    Isometry3d transform = Isometry3d::Identity();
    bool transformFirst = false;   // the cloud is still in the scan frame (nobody else holds it): move it here
    bool insert = false;
    bool evict = false;
    Vector3d position;
  };
  void shadowApply(ShadowOp & op)
  {
    if (op.cloud && op.transformFirst) {op.cloud->Transform(op.transform.matrix());}
    if (op.cloud && op.insert) {
      const auto & points = op.cloud->points_;
      const auto & covariances = op.cloud->covariances_;
      for (size_t i = 0; i < points.size(); ++i) {          // src/LocalMap.cpp:47-58
        const Key key = toKey(getVoxelIndex(points[i]));
        auto found = voxelGrid_.find(key);
        if (found == voxelGrid_.end()) {
          voxelGrid_.emplace(key, Voxel(maxNumPointsPerVoxel_, points[i], covariances[i]));
        } else {
          found->second.addPoint(points[i], covariances[i]);
        }
      }
    }
    if (op.evict) {                                          // src/LocalMap.cpp:60-72
      for (auto it = voxelGrid_.begin(); it != voxelGrid_.end(); ) {
        if (needsPointRemoval(it->first, op.position)) {it = voxelGrid_.erase(it);} else {++it;}
      }
    }
    if (op.cloud && op.cloud.use_count() == 1) {shim::recycleStorage(*op.cloud);}   // its last owner: the buffers stay
    op.cloud.reset();
  }
  void shadowLoop()
  {
    std::unique_lock<std::mutex> lk(shadowMutex_);
    for (;;) {
      shadowCv_.wait(lk, [&] {return shadowQuit_ || !shadowQueue_.empty();});
      if (shadowQueue_.empty()) {return;}                    // quit, and nothing left to apply
      ShadowOp op = std::move(shadowQueue_.front());
      shadowQueue_.pop_front();
      shadowBusy_ = true;
      lk.unlock();
      shadowApply(op);
      lk.lock();
      shadowBusy_ = false;
      shadowIdle_.notify_all();
    }
  }
  void shadowPush(ShadowOp && op)
  {
    std::unique_lock<std::mutex> lk(shadowMutex_);
    if (!shadowThread_.joinable()) {shadowThread_ = std::thread([this] {shadowLoop();});}
    // at the sensor's rate the worker is idle most of the time; a caller that runs frames back to back faster than the
    // host can file their points waits here rather than let the backlog grow without bound
    shadowIdle_.wait(lk, [&] {return shadowQueue_.size() < 256;});
    shadowQueue_.push_back(std::move(op));
    lk.unlock();
    shadowCv_.notify_one();
  }
  void shadowDrain() const
  {
    std::unique_lock<std::mutex> lk(shadowMutex_);
    shadowIdle_.wait(lk, [&] {return shadowQueue_.empty() && !shadowBusy_;});
  }
  void shadowStop()
  {
    {
      std::lock_guard<std::mutex> lk(shadowMutex_);
      shadowQuit_ = true;
    }
    shadowCv_.notify_all();
    if (shadowThread_.joinable()) {shadowThread_.join();}
  }

  void syncDevice(
    const std::vector<Voxel *> & touched, const std::vector<Key> & keys,
    const std::vector<int32_t> & erased)
  {
    // a voxel both touched and evicted in this update is gone: its Voxel* is dangling, skip by key
    std::vector<int32_t> k;
    std::vector<double> means, covs;
    k.reserve(3 * touched.size());
    means.reserve(3 * touched.size());
    covs.reserve(9 * touched.size());
    for (size_t n = 0; n < touched.size(); ++n) {
      if (!erased.empty() && voxelGrid_.find(keys[n]) == voxelGrid_.end()) {continue;}
      Voxel * v = touched[n];
      v->dirty = false;
      k.push_back(keys[n].i);
      k.push_back(keys[n].j);
      k.push_back(keys[n].k);
      for (int a = 0; a < 3; ++a) {means.push_back(v->mean(a));}
      for (int e = 0; e < 9; ++e) {covs.push_back(v->covariance.data()[e]);}
    }
    if (!erased.empty()) {
      shim::check(ctx_, vgicp_map_erase(ctx_, erased.size() / 3, erased.data()), "vgicp_map_erase");
    }
    if (!k.empty()) {
      shim::check(
        ctx_, vgicp_map_upsert(ctx_, k.size() / 3, k.data(), means.data(), covs.data()),
        "vgicp_map_upsert");
    }
  }

  double voxelSize_;
  size_t maxNumPointsPerVoxel_;
  double translationSquaredThreshold_;
  double cosineThreshold_;
  bool removeDistantPoints_;
  double distanceThreshold_;
  double removePeriod_;
  bool deviceResident_ = false;
  bool keepRawPoints_ = false;
  bool shadowComplete_ = true;       // every frame inserted on the device so far has also reached the shadow grid
  std::thread shadowThread_;
  mutable std::mutex shadowMutex_;
  mutable std::condition_variable shadowCv_, shadowIdle_;
  std::deque<ShadowOp> shadowQueue_;
  bool shadowBusy_ = false, shadowQuit_ = false;
  double currentRemoveTime_ = std::numeric_limits<double>::lowest();
  Isometry3d prevTransform_ = Isometry3d::Identity();
  bool hasPrevTransform_ = false;

  VoxelGrid voxelGrid_;

  bool visualize_;
  std::vector<Isometry3d> trajectory_;
  vgicp_ctx * ctx_;
};
}  // namespace ESKF_LIO

#endif  // ESKF_LIO_SHIM_LOCAL_MAP_HPP_
