// vgicp_context.h — the host-side context behind the C ABI (include/vgicp_hip.h), shared by vgicp_capi.hip (one
// device) and vgicp_multi.hip (one caller thread driving several devices).  Not part of the ABI: callers only ever
// see the opaque vgicp_ctx*.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vgicp_hip.h"
#include "vgicp_device.h"

using namespace vgicp;

// what a frame costs the host besides kernels: copies / memsets enqueued and synchronisations (vgicp_get_frame_stats)
namespace vgicp { extern thread_local uint64_t g_copy_ops, g_sync_ops; }
using vgicp::g_copy_ops;
using vgicp::g_sync_ops;
#define hipMemcpyAsync(...) (++g_copy_ops, hipMemcpyAsync(__VA_ARGS__))
#define hipMemsetAsync(...) (++g_copy_ops, hipMemsetAsync(__VA_ARGS__))
#define hipStreamSynchronize(...) (++g_sync_ops, hipStreamSynchronize(__VA_ARGS__))
#define hipEventSynchronize(...) (++g_sync_ops, hipEventSynchronize(__VA_ARGS__))

namespace vgicp {

// ---- the few RCCL entry points used, bound at run time so the library loads without RCCL ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[VGICP_UNIQUE_ID_BYTES]; } ncclUniqueId;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclDouble = 8;  // ncclFloat64, rccl.h
constexpr int kNcclSum = 0;
constexpr int kNcclChar = 0;   // ncclInt8

extern thread_local std::string g_create_error;
// what vgicp_sweep_stage* report: those may run on another thread than the context's owner (who writes ctx->err), so
// their text lives with the calling thread; vgicp_last_error(ctx) returns it to that thread until it fails elsewhere
extern thread_local std::string g_stage_error;
// ... identified by the context's creation number, not by its address: a context destroyed on one thread and another
// allocated at the same address must not inherit a stale text that a third thread still keeps
extern thread_local uint64_t g_stage_error_ctx;
extern std::atomic<uint64_t> g_context_ids;

inline double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

constexpr uint64_t kMinSlots = 1024;
constexpr int kDefaultChunk = 4;
constexpr int kMaxChunksInFlight = 2;
constexpr int kPersistentCooldownAligns = 8;  // aligns on the per-launch loop after the single launch gave up

}  // namespace vgicp
using namespace vgicp;

// The host side of the scan upload (vgicp_align, vgicp_scan_upload): a crew of plain-memcpy threads.
// The caller's scan lives in pageable memory that the caller frees right after the call (the reference deep-copies a
// fresh cloud per frame, src/Registration.cpp:11).  Handing such a buffer to the runtime registers its pages with the
// driver, and the free() then takes every queue of the process off the device for ~20 ms (profiles/r10_sync_stall.txt);
// staging it through page-locked memory with copy COMMANDS cost 0.45 - 0.68 ms for 9.6 MB (25 commands of ~19 us).
// So: the crew — the caller's own thread plus `helpers` threads that make no HIP call — copies the scan into page-locked
// staging memory in units of pack_arena_unit() points and publishes every unit with a flag, and ONE kernel launch
// (pack_arena_kernel) reads the staging memory over PCIe behind them.  Measured (tools/micro/stage_crew_probe.hip,
// 2 x EPYC 9575F): one thread copies 9.6 MB from never-seen pages in 0.22 ms (43 GB/s), two in 0.13 - 0.15 ms; the kernel
// gets 54 GB/s out of the link; copy + transfer + packing overlapped: 0.19 - 0.20 ms with ONE helper, the link's rate.
// A helper spins for a short while after a job (the next frame's upload follows soon in a loop) and then sleeps.
struct CopyCrew {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<uint32_t> gen{0};        // the job the helpers were last woken for
  // job number << 32 | next unit to take.  A job is OPEN from post() to finish(): finish() closes it (unit word ~0)
  // BEFORE the caller may write a field of the next job, so a thread that comes late to a job that is over can never
  // take a unit with the next job's fields half written (round 5's advisor reproduced exactly that with a unit count
  // that varies from job to job: its bounds check read the NEXT job's larger count, its compare-and-swap on the OLD
  // job's exhausted ticket succeeded).  A unit is taken by compare-and-swap, so a closed ticket is left alone.
  std::atomic<uint64_t> next{0};
  // job number << 32 | units of that job completed: a completion that lands after its job was given up (deadline) finds
  // another job's number here and is dropped instead of being counted for the job that follows
  std::atomic<uint64_t> finished{0};
  std::atomic<uint32_t> taken{0};      // units whose job fields a thread has read so far (see work())
  uint32_t job = 0;                    // jobs posted so far (the caller's thread only)
  bool quit = false;
  bool broken = false;                 // a helper missed finish()'s deadline: no helper is woken again (caller's thread only)
  // the job (written while no job is open, read only after a unit of THIS job was taken)
  const char* pts = nullptr;
  const char* cov = nullptr;
  char* apts = nullptr;
  char* acov = nullptr;
  uint32_t* flags = nullptr;
  uint32_t n = 0, unit = 0, seq = 0;
  std::atomic<uint32_t> units{0};      // read BEFORE a unit is taken (by a thread that may have come late to an older job): atomic
  uint32_t size_a = 24, size_b = 72;   // bytes per point of the two arrays (size_b = 0: one array only)
  void (*copy)(void*, const void*, size_t) = nullptr;
  // optional: the second array's units are written by this instead (dst = the unit's place, cnt points of size_b bytes
  // in src); what it returns is stored into the unit's flag line (word 1) before the flag: the form the unit was
  // staged in (the covariances of a scan upload: compact when all of a unit's are bitwise symmetric)
  uint32_t (*copy_b_form)(void*, const void*, size_t) = nullptr;
  static constexpr uint32_t kClosed = 0xFFFFFFFFu;

  void work(uint32_t my_job) {
    for (;;) {
      uint64_t v = next.load(std::memory_order_acquire);
      for (;;) {
        // closed (~0) or exhausted or another job's ticket: nothing to take.  The count may belong to a LATER job only
        // when this ticket is closed already, and then the compare-and-swap below cannot succeed.
        if ((uint32_t)(v >> 32) != my_job || (uint32_t)v >= units.load(std::memory_order_relaxed)) return;
        if (next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
      }
      const uint32_t u = (uint32_t)v;
      // the job was open when the unit was taken and stays open until the unit is delivered — unless finish() gives
      // up on this thread: everything it touches from here on is what the job was THEN, never a later job's fields
      const char *j_pts = pts, *j_cov = cov;
      char *j_apts = apts, *j_acov = acov;
      uint32_t* j_flags = flags;
      const uint32_t j_seq = seq, j_size_a = size_a, j_size_b = size_b;
      const size_t p0 = (size_t)u * unit, cnt = std::min<size_t>(unit, n - p0);
      void (*j_copy)(void*, const void*, size_t) = copy;
      uint32_t (*j_copy_b_form)(void*, const void*, size_t) = copy_b_form;
      // "I have read the job": a finish() that gives up on this thread writes the next job's fields afterwards, and
      // this is the edge that orders those writes behind the reads above (a completed unit orders them through
      // `finished`; a thread that never completes has nothing else to show)
      taken.fetch_add(1, std::memory_order_release);
      j_copy(j_apts + p0 * j_size_a, j_pts + p0 * j_size_a, cnt * j_size_a);
      if (j_size_b && j_copy_b_form)
        __atomic_store_n(j_flags + 16 * (size_t)u + 1, j_copy_b_form(j_acov + p0 * j_size_b, j_cov + p0 * j_size_b, cnt), __ATOMIC_RELAXED);
      else if (j_size_b) j_copy(j_acov + p0 * j_size_b, j_cov + p0 * j_size_b, cnt * j_size_b);
      // the unit's bytes (streaming stores, fenced by `copy`) are globally visible before its flag
      __atomic_store_n(j_flags + 16 * (size_t)u, j_seq, __ATOMIC_RELEASE);
      uint64_t f = finished.load(std::memory_order_relaxed);
      while ((uint32_t)(f >> 32) == my_job &&
             !finished.compare_exchange_weak(f, f + 1, std::memory_order_release, std::memory_order_relaxed)) {}
    }
  }
  void run() {
    uint32_t seen = gen.load(std::memory_order_acquire);
    for (;;) {
      bool have = false;
      for (int spins = 0; spins < 40000 && !have; ++spins) {   // ~1 ms
        have = gen.load(std::memory_order_acquire) != seen;
        if (!have) __builtin_ia32_pause();
        // should the scheduler have put this thread on the CPU of one that has work to do (a wake-up lands on the
        // waker's CPU): let it run.  Costs a fraction of a microsecond when nobody else wants this CPU.
        if (!have && (spins & 1023) == 1023) sched_yield();
      }
      if (!have) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return quit || gen.load(std::memory_order_acquire) != seen; });
        if (quit) return;
      }
      seen = gen.load(std::memory_order_acquire);
      work(seen);
    }
  }
  void start(int helpers) {
    for (int i = 0; i < helpers; ++i) th.emplace_back([this] { run(); });
  }
  // everything of the job is in place: open it (to the helpers too if asked); the caller then works on it itself
  // (work(job)) and waits for the units others took (finish())
  uint32_t post(bool wake_helpers) {
    if (++job == 0) ++job;
    finished.store((uint64_t)job << 32, std::memory_order_relaxed);
    next.store((uint64_t)job << 32, std::memory_order_release);
    if (wake_helpers && !broken && !th.empty()) {
      {
        std::lock_guard<std::mutex> lk(m);
        gen.store(job, std::memory_order_release);
      }
      cv.notify_all();
    }
    return job;
  }
  // Wait for the units other threads took, then CLOSE the job.  A helper that was woken for this job may sit on THIS
  // thread's CPU (a wake-up lands on the waker's CPU) with a unit half copied: spinning here would keep it off the CPU
  // for a whole scheduler slice (3 ms steps were measured); after a short spin the CPU is offered to whoever else wants
  // it.  Bounded: a helper that has not delivered its unit `deadline_seconds` after the caller ran out of work (it died,
  // or the machine is not scheduling it) makes this return false — the job is closed, the crew never wakes a helper
  // again (`broken`), and the caller reports the failure; the late helper's completion is dropped by its job number.
  // (What cannot be taken back is the helper's pointer into the caller's buffer: the entry point says so in its error.)
  bool finish(double deadline_seconds = 10.0) {
    const uint32_t want = units.load(std::memory_order_relaxed);
    bool ok = true;
    double t0 = 0.0;
    for (uint32_t spins = 0; (uint32_t)finished.load(std::memory_order_acquire) < want; ++spins) {
      if (spins < 512) { __builtin_ia32_pause(); continue; }
      sched_yield();
      if ((spins & 255u) == 0) {
        const double t = now_seconds();
        if (t0 == 0.0) t0 = t;
        else if (t - t0 > deadline_seconds) { ok = false; broken = true; (void)taken.load(std::memory_order_acquire); break; }
      }
    }
    next.store(((uint64_t)job << 32) | kClosed, std::memory_order_release);
    if (!ok) finished.store(0, std::memory_order_relaxed);   // job 0 is never posted: late completions go nowhere
    return ok;
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv.notify_all();
    for (auto& t : th)
      if (t.joinable()) {
        if (broken) t.detach();   // one of them never came back: nobody may wait for it (the owner leaks this object)
        else t.join();
      }
    th.clear();
  }
};

struct vgicp_multi;  // vgicp_multi.hip: the sub-contexts of an in-process multi-device context

struct vgicp_ctx {
  uint64_t id = 0;                // creation number (never 0, never reused)
  int device = -1;
  vgicp_multi* multi = nullptr;   // this handle IS a multi-device context: every entry point forwards to vgicp_multi.hip
  vgicp_multi* owner = nullptr;   // this context is one of a multi-device context's sub-contexts (rank = peer_rank)
  hipStream_t stream = nullptr;
  mutable std::string err;
  std::string peer_status;   // "" while the device-initiated exchange is wired (or nothing to exchange); else why it is not (vgicp_peer_status)
  int cu_count = 0;
  uint64_t hbm_bytes = 0;
  std::string arch;

  // voxel table
  double voxel_size = 0.0;
  VoxelRecord* table = nullptr;
  uint64_t slots = 0;
  uint64_t voxels = 0;      // FULL records
  uint64_t tombstones = 0;
  uint32_t* d_counters = nullptr;  // 8 words + 64 of developer histograms (VGICP_DEBUG_PREP=2)
  uint32_t* h_counters = nullptr;  // pinned

  // dense copy of the FULL records for tables far beyond the caches' reach (PersistArgs::dense): rebuilt lazily before
  // an align when the map has changed since (map_version counts every mutation)
  VoxelRecord* d_dense = nullptr;
  uint64_t dense_capacity = 0;      // records
  uint32_t* d_dense_counts = nullptr;
  uint32_t dense_counts_capacity = 0;
  uint64_t map_version = 1, dense_version = 0;
  uint64_t dense_slots_threshold = 1ull << 24;   // tables of this many slots (2 GiB) and more; VGICP_DENSE_SLOTS at creation, 0 = never
  // batch staging (upsert / erase / hooks)
  void* d_stage = nullptr;
  size_t stage_bytes = 0;
  void* d_cells = nullptr;  // cell table of the scan preparation (vgicp_preprocess)
  size_t cells_bytes = 0;

  // resident scan
  double* d_scan_aos = nullptr;  // points (3n) then covs (9n)
  double* d_scan = nullptr;      // SoA planes
  void* d_memo = nullptr;        // per point {key, slot}: the launch-per-round loop's memory between launches (IterArgs::memo)
  size_t scan_capacity = 0;      // points
  uint32_t n = 0;
  uint64_t stride = 0;
  bool scan_ready = false;

  // align state
  AlignState* d_state = nullptr;  // two, ping-pong: launch j reads [j&1], writes [(j+1)&1]
  AlignState* h_state = nullptr;  // pinned, kMaxChunksInFlight + 1 slots
  double* d_rows[2] = {nullptr, nullptr};  // partial rows, ping-pong like the state
  double* d_sums = nullptr;       // one row: the all-reduce message (multi-GPU)
  // persistent single-launch align (single GPU)
  uint32_t persist_round0 = 0;       // rounds the persistent launches of this context have executed, mod 3
  uint32_t persist_seq = 0;
  uint32_t persist_lds_budget = 0;   // dynamic LDS a persistent workgroup may plan with (0 = the whole CU); sub-contexts that share a device take less
  uint32_t persist_grid = 0;         // workgroups of every persistent launch: min(CUs, kExchangeRows), all resident
  double* d_rows_persist = nullptr;  // [3][kExchangeRows][kSlots] (vgicp_device.h, PersistArgs)
  double* d_parts_persist = nullptr; // [3][kFolders][kSlots]
  void* h_exchange_image = nullptr;  // pinned: what the two buffers hold between launches
  bool persistent_enabled = true;    // cleared by VGICP_PERSISTENT=0 or when a workgroup does not fit a CU
  double prefetch_margin = 0.015;    // see PersistArgs::prefetch_margin; VGICP_PREFETCH_MARGIN overrides (0 = off).  Round 6: 0.03 -> 0.015
                                     // once the workgroups that are no folders stopped polling early (C2: 0 7.18, 0.01 6.41, 0.015 6.34, 0.02 6.35, 0.03 6.52, 0.04 6.66 us per round)
  uint32_t persist_spin_limit = 50000;  // polls (>= ~1 us each) before an in-kernel wait gives up
  int persistent_cooldown = 0;       // aligns left on the per-launch loop after an in-kernel wait timed out
  uint64_t persistent_launches = 0;  // diagnostics (vgicp_get_counter)
  uint64_t persistent_fallbacks = 0;
  uint64_t upload_bytes = 0;
  double upload_seconds = 0.0;
  uint64_t prep_indefinite = 0;      // kept points of the last scan preparation with an indefinite covariance
  // developer / test switches of the environment, read ONCE when the context is created (never inside a call)
  struct DevSwitches {
    bool no_sym = false, no_stash = false, no_memo = false, verbose = false, insert_sort = false;
    int debug_prep = 0;
    uint32_t pack_spin_limit = 0;     // 0 = the module's default
    long debug_upload_delay_us = 0;
  } dev;
  CopyCrew* crew = nullptr;          // the upload's copy threads, created with the first upload that wants a helper
  int upload_threads = 3;            // threads that copy a scan into the staging memory, the caller's included (VGICP_UPLOAD_THREADS).
                                     // Round 6: 2 -> 3 — with symmetric covariances crossing the link as six doubles the HOST copy out of
                                     // never-seen pages became the limit at two threads (C2 from fresh clouds: 2 threads 0.366 ms per align
                                     // with or without the compaction, 3 threads 0.321, 4 threads 0.316 - 0.321)
  char* h_upload = nullptr;          // page-locked staging memory of the scan upload: [unit flags][points][covariances]
  size_t upload_cap = 0;             // bytes behind the flags
  size_t upload_flag_bytes = 0;      // one 64-byte line per unit the capacity can hold; never holds anything but flags
  hipEvent_t ev_upload = nullptr;    // behind the kernel that read h_upload last (the next upload overwrites it)
  bool upload_in_flight = false;
  uint64_t upload_slow = 0;          // uploads whose copy threads took so long that the packing was repeated behind them
  // scan preparation without host round trips
  void* d_tiles = nullptr;           // tile slots of the two device-wide scans
  uint32_t* h_prep = nullptr;        // pinned: the counter block as a preparation left it (kCounterWords)
  uint32_t prep_epoch = 0;
  bool scan_pending = false;         // a prepared scan is resident but the host has not read its size / verdict yet
  uint32_t n_upper = 0;              // raw points of the pending scan (>= its kept points)
  uint64_t scan_generation = 0;      // replacements of the resident scan so far (VGICP_COUNTER_SCAN_GENERATION)
  uint32_t scan_seq = 0;             // uploads so far; pack_scan_kernel marks an asymmetric covariance with it
  bool scan_sym_known = false;       // the resident scan went through pack_scan_kernel (not a scan prepared on the device)
  int64_t prep_deskewed = 0;
  bool reference_order = false;      // VGICP_OPTION_REFERENCE_ORDER: prepared scans come in the reference's unordered_map order
  bool prep_with_deskew = false;
  double prep_voxel = 0.0;           // > 0: the resident scan was down-sampled on the device to one point per voxel of this size
  // the deskew's state table on its way to the device: pinned, two slots in turn (an enqueue-only preparation returns
  // before the copy has run, so the table cannot live on the caller's stack)
  // ... and the raw sweep (points, then capture times): mid-sized copies out of the caller's pageable memory are staged
  // through these instead of handed to the runtime, which would pin the caller's pages on the fly — measured on the
  // round's boxes: 1.4 MB of points 0.05 ms staged, but 12-22 ms (first copy of every frame) through the pinning path
  // once the caller allocates and frees its clouds per frame, as the reference does (examples/frame_chain)
  // ... and for every OTHER copy between the caller's pageable memory and the device (map batches, hooks, scan_download,
  // the scan of a vgicp_align up to 4 MB): one page-locked arena.  A copy of more than 512 KB goes through it (h2d: CPU
  // copy in, then DMA; d2h: DMA, then CPU copy out after the call's synchronisation) -- never through the runtime's
  // pin-on-the-fly path, whose registrations stall the whole process when the caller frees the buffer
  char* h_arena = nullptr;
  size_t arena_used = 0;
  size_t upload_stage_limit = 512u << 20;   // scans up to this many bytes are staged by the copy crew; larger ones (and
                                            // all of them with the limit 0) are handed to the runtime in place
                                            // (VGICP_UPLOAD_STAGE_LIMIT, VGICP_OPTION_UPLOAD_STAGE_KB)
  size_t upload_whole_hint = 0;             // a sub-context's shard: the size of the caller's WHOLE scan decides, not the shard's
  struct PendingOut { void* dst; const char* src; size_t bytes; };
  std::vector<PendingOut> pending_out;
  // vgicp_scan_fetch_*: the prepared scan written into page-locked memory by a kernel, piece by piece
  unsigned long long* h_fetch_hdr = nullptr;      // pinned: word 0 epoch << 32 | kept (run_scan_kernel), word 8 seq << 32 | refused (fetch_kernel), words 16 .. 79 the checksums
  unsigned long long* h_fetch_hdr_dev = nullptr;  // the same as the device addresses it
  char* h_fetch = nullptr;           // pinned: [one 64-byte flag line per piece][points, padded to 256 bytes][covariances]
  char* h_fetch_dev = nullptr;
  size_t fetch_cap_points = 0;       // points the staging area can hold
  size_t fetch_flag_bytes = 0;
  uint32_t fetch_seq = 0;
  unsigned long long* d_fetch_sums = nullptr;   // device, 65 words, zero between launches (fetch_kernel's checksums + ticket)
  bool fetch_sums_valid = false;     // the page-locked copy (h_fetch_hdr + 16 .. + 80) holds the sums of the last completed fetch
  bool fetch_open = false;           // a fetch kernel is enqueued behind the pending preparation
  uint32_t fetch_kept = 0;
  // sweeps staged AHEAD of their preparation (vgicp_sweep_stage: the lidar callback's thread copies a sweep into
  // page-locked memory when it arrives; vgicp_scan_prepare_staged_async consumes it by ticket).  Guarded by
  // ahead_mutex: the one part of a context that another thread may enter while the owner thread is inside a call.
  struct AheadSlot {
    char* mem = nullptr;
    size_t cap = 0, n = 0;
    uint64_t ticket = 0;
    bool has_times = false;
    uint32_t step = 0, off[3] = {0, 0, 0};   // != 0: sensor records (vgicp_sweep_stage_cloud2), float32 x y z at these offsets
    size_t times_at = 0;           // byte offset of the capture times (n doubles) inside mem
    int state = 0;                 // 0 free, 1 staged, 2 handed to the device (`done` recorded behind its readers), 3 being filled
    hipEvent_t done = nullptr;
  };
  AheadSlot ahead[3];
  std::mutex ahead_mutex;
  uint64_t ahead_tickets = 0;
  char* h_raw_stage[2] = {nullptr, nullptr};
  size_t raw_stage_cap[2] = {0, 0};
  double* h_state_table[2] = {nullptr, nullptr};
  size_t state_table_cap[2] = {0, 0};
  hipEvent_t ev_state_table[2] = {nullptr, nullptr};
  uint32_t state_table_next = 0;
  // deferred map insertion (vgicp_map_insert_resident_async): running totals on the device, read at the next sync
  uint32_t* d_ins_counters = nullptr;
  uint32_t* h_ins_counters = nullptr;  // pinned
  uint32_t ins_seen[2] = {0, 0};
  bool insert_pending = false;
  bool ins_copy_enqueued = false;    // some device-to-host copy behind the pending insertion carries its totals ...
  bool ins_from_prep = false;        // ... in the tail of h_prep (the next preparation's counter copy) rather than h_ins_counters
  uint64_t insert_pending_upper = 0;
  // frame statistics
  uint64_t stat_launches0 = 0, stat_copies0 = 0, stat_syncs0 = 0;
  bool stage_events = false;
  hipEvent_t ev_stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [6]: behind the prologue
  bool ev_stage_set[7] = {false, false, false, false, false, false, false};
  int iter_block = 512;           // threads per workgroup of the iteration kernel (measured best at C2)
  double* d_log = nullptr;
  double* h_log = nullptr;  // pinned
  double* h_log_dev = nullptr;  // the same memory as the device addresses it (the persistent launch writes state + log there)
  int log_capacity = 0;     // iterations
  uint64_t* d_stamps = nullptr;  // only with VGICP_DEBUG_STAMPS=1
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_chunk[kMaxChunksInFlight] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_prof;

  // device-initiated exchange between GPUs: peer-mapped mailboxes (vgicp_peer_*)
  double* d_mail = nullptr;            // this rank's mailbox, fine-grained device memory, [3][kMaxRanks][kSlots]
  double* peer_mail[kMaxRanks] = {nullptr};  // every rank's mailbox as mapped here ([peer_rank] = d_mail)
  double** d_mail_table = nullptr;     // device copy of peer_mail
  int peer_world = 1, peer_rank = 0;
  bool peers_connected = false;
  bool peer_mail_is_ipc = true;        // false: plain pointers of the same process (sub-contexts), nothing to close
  bool peer_enabled = true;            // cleared for good when a launch gave up waiting for a peer
  uint32_t mail_round0 = 0;            // rounds executed through the mailboxes so far (same on every rank)
  uint32_t mail_seq = 0;               // aligns attempted through the mailboxes so far (same on every rank)

  // RCCL
  RcclApi rccl;
  ncclComm_t comm = nullptr;
  int world_size = 1;
  int rank = 0;
};


// ---- helpers shared by the two translation units ----
namespace vgicp {
inline int fail(const vgicp_ctx* ctx, int code, const std::string& text) {
  if (ctx) {
    ctx->err = text;
    if (g_stage_error_ctx == ctx->id) g_stage_error_ctx = 0;
  } else {
    g_create_error = text;
  }
  return code;
}
inline int fail_stage(const vgicp_ctx* ctx, int code, const std::string& text) {
  g_stage_error = text;
  g_stage_error_ctx = ctx ? ctx->id : 0;
  return code;
}
inline int fail_hip(const vgicp_ctx* ctx, hipError_t e, const char* what) {
  return fail(ctx, VGICP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace vgicp

#define VG_HIP(ctx, call)                                              \
  do {                                                                 \
    hipError_t e__ = (call);                                           \
    if (e__ != hipSuccess) return fail_hip((ctx), e__, #call);         \
  } while (0)

// ---- what vgicp_multi.hip needs from vgicp_capi.hip besides the public entry points ----
namespace vgicp_internal {
// A sub-context gives up an align it cannot finish alone with this status (never seen by a caller of the ABI): the
// in-kernel exchange between the sub-contexts timed out, or the align asks for the launch-per-round loop — the
// multi-device context then runs that loop itself, adding the sub-contexts' rows on the host.
constexpr int kNeedGroupLoop = 1000;
// vgicp_create with a cap on the persistent launch's workgroups (sub-contexts that share one device split its CUs).
int create_context(int device_id, uint32_t max_persist_grid, vgicp_ctx** out);
// Mailboxes of the device-initiated exchange wired by plain pointers (one process: no IPC handles); (re-)initialises
// every mailbox and the running round / align numbers.  The streams of all sub-contexts must be idle.
int wire_mailboxes(vgicp_ctx* const* subs, int n);
// ICP::align over sub-contexts that each hold a shard, one launch per round on every device, the rank rows added on
// the host in rank order (the same pairwise tree the mailbox path uses): the fallback of the in-kernel exchange and
// the path of VGICP_FLAG_PROFILE / VGICP_FLAG_NO_PERSISTENT.  The caller's thread drives every device.
int align_host_summed(vgicp_ctx* const* subs, int n, const double guess[16], const vgicp_params* params,
                      double out_pose[16], vgicp_stats* stats);
// The resident scan of `ctx` becomes n points that are already on a device as AoS (points n x 3, covs n x 9; peer
// memory is fine): copied into the context's own storage and packed into the planes the registration reads.
// Enqueued on the context's stream (which first waits for `ready`, if given); not synchronised.
int adopt_device_scan(vgicp_ctx* ctx, int src_device, const double* d_points, const double* d_covs, size_t n,
                      double prep_voxel, hipEvent_t ready);
// LocalMap::updateLocalMap's insertion for n points that are already on THIS context's device as AoS.
int map_insert_device(vgicp_ctx* ctx, const double* d_points, const double* d_covs, size_t n, const double transform[16],
                      size_t max_points_per_voxel, bool short_lists, bool deferred, size_t* new_voxels);
int settle_context(vgicp_ctx* ctx);
bool align_needs_allocation(const vgicp_ctx* ctx, size_t n, int max_it);
int reserve_for_align(vgicp_ctx* ctx, size_t n, int max_it);
bool insertion_lists_stay_short_for(const vgicp_ctx* ctx, double prep_voxel);
}  // namespace vgicp_internal

// ---- vgicp_multi.hip: the entry points of a multi-device context (ctx->multi != nullptr) ----
namespace vgicp_multi_api {
int destroy(vgicp_ctx* ctx);
int device_info(const vgicp_ctx* ctx, char* name, size_t name_len, int32_t* cu_count, uint64_t* hbm_bytes);
int get_counter(const vgicp_ctx* ctx, int which, uint64_t* value);
int map_reset(vgicp_ctx* ctx, double voxel_size, size_t capacity_hint);
int map_upsert(vgicp_ctx* ctx, size_t n, const int32_t* keys, const double* means, const double* covs);
int map_erase(vgicp_ctx* ctx, size_t n, const int32_t* keys);
int map_size(const vgicp_ctx* ctx, size_t* voxels, size_t* table_slots);
int map_insert_scan(vgicp_ctx* ctx, size_t n, const double* points, const double* covs, const double transform[16],
                    size_t max_points_per_voxel, size_t* new_voxels);
int map_insert_resident(vgicp_ctx* ctx, const double transform[16], size_t max_points_per_voxel, size_t* new_voxels,
                        bool deferred);
int map_evict(vgicp_ctx* ctx, const double position[3], double distance_threshold, size_t* removed);
int map_export(vgicp_ctx* ctx, size_t capacity, int32_t* keys, double* means, double* covs, uint64_t* counts,
               size_t* written);
int align(vgicp_ctx* ctx, size_t n, const double* points, const double* covs, const double guess[16],
          const vgicp_params* params, double out_pose[16], vgicp_stats* stats);
int scan_upload(vgicp_ctx* ctx, size_t n, const double* points, const double* covs);
int align_resident(vgicp_ctx* ctx, const double guess[16], const vgicp_params* params, double out_pose[16],
                   vgicp_stats* stats);
int scan_prepare(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time, size_t num_states,
                 const double* states, const double extrinsic[16], double voxel_size, int knn, size_t* kept,
                 int64_t* deskewed, bool deferred, uint64_t ticket = 0);
int scan_info(vgicp_ctx* ctx, size_t* kept, int64_t* deskewed, uint64_t* indefinite);
int scan_download(vgicp_ctx* ctx, size_t capacity, double* points, double* covs, size_t* n);
int get_frame_stats(vgicp_ctx* ctx, vgicp_frame_stats* out, int reset);
int set_option(vgicp_ctx* ctx, int option, int value);
vgicp_ctx* first(const vgicp_ctx* ctx);  // sub-context 0: the hooks that work on one device
const char* peer_status(const vgicp_ctx* ctx);
void scan_replaced(vgicp_ctx* ctx);
}  // namespace vgicp_multi_api
