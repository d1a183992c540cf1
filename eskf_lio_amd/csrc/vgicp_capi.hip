// vgicp_capi.hip — the C ABI declared in include/vgicp_hip.h (host side of the HIP module).
//
// Owns: the device context, the voxel-table mirror of LocalMap (growth / rehash policy), the resident
// scan, the iteration launch schedule of ICP::align (reference src/Registration.cpp:15-28) and the
// optional RCCL communicator that merges the normal equations across GPUs (the cross-device form of
// the thread merge at reference src/Registration.cpp:71-75).
// No PyTorch, no oracle, no CPU fallback: without a gfx950 device every compute entry point fails
// with VGICP_ERR_NO_DEVICE / VGICP_ERR_HIP.
#include <immintrin.h>

#include <array>
#include <unordered_map>
#include "vgicp_context.h"

namespace vgicp {
thread_local uint64_t g_copy_ops = 0, g_sync_ops = 0;
thread_local std::string g_create_error;
thread_local std::string g_stage_error;
thread_local uint64_t g_stage_error_ctx = 0;
std::atomic<uint64_t> g_context_ids{0};
}  // namespace vgicp

namespace {

int settle(vgicp_ctx* ctx);         // defined with the scan preparation below
int fetch_insert_totals(vgicp_ctx* ctx);
int settle_scan(vgicp_ctx* ctx);
int settle_insert(vgicp_ctx* ctx);

// ---- copies between the CALLER'S pageable memory and the device -------------------------------------------------
// hipMemcpyAsync registers a pageable range of more than 1 MB with the driver and lets the DMA engine read it in
// place.  That is the fastest way to move a buffer once -- and a trap for a caller that allocates and frees its buffers
// per frame, as the reference does: when such a range is unmapped (free() of anything above glibc's mmap threshold),
// the driver takes ALL queues of the process off the device until the registration is torn down: 20 - 24 ms in which
// nothing runs (profiles/r10_sync_stall.txt: 23 of 30 ten-frame runs saw it; none with a malloc that keeps its memory).
// So copies of 512 KB - 16 MB go through a page-locked arena of the context instead (smaller ones the runtime stages
// itself; larger ones -- a 10 M-voxel map, a 100 k-point scan -- go up directly, once).  VGICP_STAGE_LIMIT=0: never.
constexpr size_t kArenaBytes = 16u << 20, kArenaMin = 512u << 10;

// The CPU copy into page-locked staging memory sets the pace of a frame's first phase (the device idles until the sweep
// has arrived).  The destination is read next by the DMA engine, never by this CPU: streaming stores write it without
// first fetching the lines (no read-for-ownership) and without evicting the caller's data from the caches.
// VGICP_STAGE_COPY=memcpy keeps libc's copy.
__attribute__((target("avx2"))) void stage_copy_avx2(char* dst, const char* src, size_t bytes) {
  size_t i = 0;
  // dst is 64-byte aligned at every call site (page-locked buffers, offsets in multiples of 256 bytes)
  for (; i + 128 <= bytes; i += 128) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
    const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
  }
  _mm_sfence();
  if (i < bytes) std::memcpy(dst + i, src + i, bytes - i);
}
// earliest / latest capture time and "is any NaN" of a sweep, as the plain loops  e = t < e ? t : e;  l = t > l ? t : l
// give them (a NaN never replaces anything; a NaN in t[0] stays): vminpd / vmaxpd return their SECOND operand when the
// comparison fails, which is exactly that.  One dependent chain of 60 000 vminsd is 27 us per sweep; eight lanes: 4 us.
__attribute__((target("avx2"))) void time_range_avx2(const double* t, size_t n, double* earliest, double* latest, bool* any_nan) {
  __m256d mn0 = _mm256_set1_pd(t[0]), mn1 = mn0, mx0 = mn0, mx1 = mn0;
  __m256d un = _mm256_cmp_pd(mn0, mn0, _CMP_UNORD_Q);
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256d a = _mm256_loadu_pd(t + i), b = _mm256_loadu_pd(t + i + 4);
    mn0 = _mm256_min_pd(a, mn0);
    mn1 = _mm256_min_pd(b, mn1);
    mx0 = _mm256_max_pd(a, mx0);
    mx1 = _mm256_max_pd(b, mx1);
    un = _mm256_or_pd(un, _mm256_or_pd(_mm256_cmp_pd(a, a, _CMP_UNORD_Q), _mm256_cmp_pd(b, b, _CMP_UNORD_Q)));
  }
  double lo[8], hi[8];
  _mm256_storeu_pd(lo, mn0); _mm256_storeu_pd(lo + 4, mn1);
  _mm256_storeu_pd(hi, mx0); _mm256_storeu_pd(hi + 4, mx1);
  double e = lo[0], l = hi[0];
  for (int k = 1; k < 8; ++k) { e = lo[k] < e ? lo[k] : e; l = hi[k] > l ? hi[k] : l; }
  bool nan = _mm256_movemask_pd(un) != 0;
  for (; i < n; ++i) { e = t[i] < e ? t[i] : e; l = t[i] > l ? t[i] : l; nan |= !(t[i] == t[i]); }
  *earliest = e; *latest = l; *any_nan = nan;
}
void time_range(const double* t, size_t n, double* earliest, double* latest, bool* any_nan) {
  static const bool wide = __builtin_cpu_supports("avx2");
  if (wide && n >= 16) { time_range_avx2(t, n, earliest, latest, any_nan); return; }
  double e = t[0], l = t[0];
  bool nan = !(t[0] == t[0]);
  for (size_t i = 1; i < n; ++i) { e = t[i] < e ? t[i] : e; l = t[i] > l ? t[i] : l; nan |= !(t[i] == t[i]); }
  *earliest = e; *latest = l; *any_nan = nan;
}
void stage_copy(void* dst, const void* src, size_t bytes) {
  static const bool streaming = __builtin_cpu_supports("avx2") &&
                                !(std::getenv("VGICP_STAGE_COPY") && std::strcmp(std::getenv("VGICP_STAGE_COPY"), "memcpy") == 0);
  if (streaming && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0) stage_copy_avx2(static_cast<char*>(dst), static_cast<const char*>(src), bytes);
  else std::memcpy(dst, src, bytes);
}
// A unit of covariances (cnt x 9 doubles, column-major) into the staging memory of the scan upload: when every one of
// them is bitwise symmetric — c10 == c01, c20 == c02, c21 == c12; every covariance the reference makes is — only the six
// entries c00 c10 c20 c11 c21 c22 are written (48 instead of 72 bytes per point cross the link; pack_arena_kernel
// mirrors them), else the unit is copied whole.  Returns the form for the unit's flag line.
__attribute__((target("avx2"))) bool cov_unit_compact_avx2(char* dst, const char* src, size_t cnt) {
  const double* s = reinterpret_cast<const double*>(src);
  const uint64_t* w = reinterpret_cast<const uint64_t*>(src);
  double* d = reinterpret_cast<double*>(dst);
  uint64_t bad = 0;
  size_t i = 0;
  // two points per turn: 18 doubles in, 12 out = three aligned 32-byte streaming stores (dst is 64-byte aligned and a
  // pair's 96 bytes keep it 32-byte aligned).  No shuffles: every output vector is two or three overlapping unaligned
  // loads blended (the load ports have room; cross-lane permutes were the bottleneck of a first version), and the
  // symmetry test is scalar on the same cache lines.
  // in: A0 .. A8 at s[0..8], B0 .. B8 at s[9..17]; out: A0 A1 A2 A4 | A5 A8 B0 B1 | B2 B4 B5 B8
  for (; i + 2 <= cnt; i += 2, s += 18, w += 18, d += 12) {
    const __m256d o0 = _mm256_blend_pd(_mm256_loadu_pd(s), _mm256_loadu_pd(s + 1), 0x8);
    const __m256d o1 = _mm256_blend_pd(_mm256_loadu_pd(s + 7), _mm256_loadu_pd(s + 5), 0x1);
    const __m256d o2 = _mm256_blend_pd(_mm256_blend_pd(_mm256_loadu_pd(s + 11), _mm256_loadu_pd(s + 12), 0x6), _mm256_loadu_pd(s + 14), 0x8);
    bad |= (w[1] ^ w[3]) | (w[2] ^ w[6]) | (w[5] ^ w[7]) | (w[10] ^ w[12]) | (w[11] ^ w[15]) | (w[14] ^ w[16]);
    _mm256_stream_pd(d, o0);
    _mm256_stream_pd(d + 4, o1);
    _mm256_stream_pd(d + 8, o2);
  }
  if (i < cnt) {   // an odd count: the unit's (the scan's) last point
    bad |= (w[1] ^ w[3]) | (w[2] ^ w[6]) | (w[5] ^ w[7]);
    const uint64_t o[6] = {w[0], w[1], w[2], w[4], w[5], w[8]};
    std::memcpy(d, o, sizeof o);
  }
  _mm_sfence();
  return bad == 0;
}
uint32_t stage_cov_unit(void* dst, const void* src, size_t cnt) {
  static const bool wide = __builtin_cpu_supports("avx2");
  static const bool off = std::getenv("VGICP_UPLOAD_COMPACT") && std::getenv("VGICP_UPLOAD_COMPACT")[0] == '0';   // A/B aid
  if (wide && !off && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0 &&
      cov_unit_compact_avx2(static_cast<char*>(dst), static_cast<const char*>(src), cnt))
    return kArenaCompact;
  stage_copy(dst, src, cnt * 9 * sizeof(double));   // one asymmetric covariance (or no AVX2): the unit as it is
  return kArenaFull;
}
void arena_reset(vgicp_ctx* ctx) {
  ctx->arena_used = 0;
  ctx->pending_out.clear();
}
char* arena_take(vgicp_ctx* ctx, size_t bytes) {
  static const bool off = std::getenv("VGICP_STAGE_LIMIT") && std::atoll(std::getenv("VGICP_STAGE_LIMIT")) == 0;
  if (off || bytes <= kArenaMin || bytes > kArenaBytes - ctx->arena_used) return nullptr;
  if (!ctx->h_arena && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_arena), kArenaBytes, 0) != hipSuccess) {
    ctx->h_arena = nullptr;
    return nullptr;
  }
  char* p = ctx->h_arena + ctx->arena_used;
  ctx->arena_used += (bytes + 255) & ~size_t(255);
  return p;
}
// page-locked memory (hipHostMalloc / vgicp_host_register): the DMA engine reads it in place, nothing to stage
bool is_pagelocked(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
int user_h2d(vgicp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return VGICP_OK;
  char* p = bytes > kArenaMin && is_pagelocked(src) ? nullptr : arena_take(ctx, bytes);
  if (!p) {
    VG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return VGICP_OK;
  }
  const size_t piece = 384u << 10;   // each piece travels while the CPU copies the next
  for (size_t off = 0; off < bytes; off += piece) {
    const size_t len = std::min(piece, bytes - off);
    stage_copy(p + off, static_cast<const char*>(src) + off, len);
    VG_HIP(ctx, hipMemcpyAsync(static_cast<char*>(dst) + off, p + off, len, hipMemcpyHostToDevice, ctx->stream));
  }
  return VGICP_OK;
}
// device -> the caller's memory; complete only after the stream has been synchronised AND user_copies_finish ran
int user_d2h(vgicp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return VGICP_OK;
  char* p = bytes > kArenaMin && is_pagelocked(dst) ? nullptr : arena_take(ctx, bytes);
  if (!p) {
    VG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return VGICP_OK;
  }
  VG_HIP(ctx, hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ctx->pending_out.push_back({dst, p, bytes});
  return VGICP_OK;
}
void user_copies_finish(vgicp_ctx* ctx) {
  for (const auto& o : ctx->pending_out) std::memcpy(o.dst, o.src, o.bytes);
  ctx->pending_out.clear();
}
#define VG_RC(call) do { const int rc__ = (call); if (rc__ != VGICP_OK) return rc__; } while (0)

int ensure_stage(vgicp_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return VGICP_OK;
  if (ctx->d_stage) VG_HIP(ctx, hipFree(ctx->d_stage));
  ctx->d_stage = nullptr;
  ctx->stage_bytes = 0;
  const size_t want = bytes + bytes / 2;
  VG_HIP(ctx, hipMalloc(&ctx->d_stage, want));
  ctx->stage_bytes = want;
  return VGICP_OK;
}

int alloc_table(vgicp_ctx* ctx, uint64_t slots, VoxelRecord** out) {
  if (slots > (1ull << 32)) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table would exceed 2^32 slots");
  VoxelRecord* t = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&t), slots * sizeof(VoxelRecord));
  if (e != hipSuccess)
    return fail(ctx, VGICP_ERR_TABLE_FULL, std::string("hipMalloc(voxel table): ") + hipGetErrorString(e));
  VG_HIP(ctx, launch_table_clear(ctx->stream, t, slots));
  *out = t;
  return VGICP_OK;
}

int reserve_dense(vgicp_ctx* ctx);
// Keep load (FULL + TOMB + incoming) <= 1/2 at all times; size new tables for load <= 1/4.
int ensure_table(vgicp_ctx* ctx, uint64_t incoming) {
  const uint64_t used = ctx->voxels + ctx->tombstones + incoming + ctx->insert_pending_upper;
  if (ctx->table && used * 2 <= ctx->slots) return VGICP_OK;
  const uint64_t slots = next_pow2(std::max<uint64_t>(kMinSlots, (ctx->voxels + incoming) * 4));
  VoxelRecord* fresh = nullptr;
  int rc = alloc_table(ctx, slots, &fresh);
  if (rc != VGICP_OK) return rc;
  if (ctx->table) {
    if (ctx->voxels > 0) {
      // one scratch word per OLD slot between the claim and the write launch (its own allocation: the staging area may
      // hold the batch that made the table grow)
      uint32_t* claimed = nullptr;
      VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&claimed), ctx->slots * sizeof(uint32_t)));
      VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
      VG_HIP(ctx, launch_rehash(ctx->stream, ctx->table, ctx->slots, fresh, (uint32_t)(slots - 1),
                                ctx->d_counters, claimed));
      VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
      VG_HIP(ctx, hipFree(claimed));
    }
    VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VG_HIP(ctx, hipFree(ctx->table));
  }
  ctx->table = fresh;
  ctx->slots = slots;
  ctx->tombstones = 0;
  ++ctx->map_version;
  return reserve_dense(ctx);   // the dense copy's storage follows the table's size here, never inside an align
}

int ensure_scan(vgicp_ctx* ctx, size_t n) {
  if (n <= ctx->scan_capacity && ctx->d_scan) return VGICP_OK;
  if (ctx->d_scan) VG_HIP(ctx, hipFree(ctx->d_scan));
  if (ctx->d_scan_aos) VG_HIP(ctx, hipFree(ctx->d_scan_aos));
  if (ctx->d_memo) VG_HIP(ctx, hipFree(ctx->d_memo));
  ctx->d_scan = ctx->d_scan_aos = nullptr;
  ctx->d_memo = nullptr;
  ctx->scan_capacity = 0;
  size_t cap = std::max<size_t>(n + n / 4, 1024);
  cap = (cap + 63) & ~size_t(63);  // planes stay 512-byte aligned
  VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_scan), cap * kScanPlanes * sizeof(double)));
  VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_scan_aos), cap * kScanPlanes * sizeof(double)));
  VG_HIP(ctx, hipMalloc(&ctx->d_memo, cap * 16));
  ctx->scan_capacity = cap;
  return VGICP_OK;
}

int ensure_log(vgicp_ctx* ctx, int iterations) {
  if (iterations <= ctx->log_capacity) return VGICP_OK;
  if (ctx->d_log) VG_HIP(ctx, hipFree(ctx->d_log - kSlots));
  if (ctx->h_log) VG_HIP(ctx, hipHostFree(ctx->h_log - kSlots));
  ctx->d_log = ctx->h_log = nullptr;
  ctx->log_capacity = 0;
  // one header row in front of the log: the persistent launch leaves its final AlignState there, so a
  // single device-to-host copy brings state and log back
  const int cap = std::max(iterations, 128);
  double* d = nullptr;
  double* h = nullptr;
  VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d), (size_t)(cap + 1) * kSlots * sizeof(double)));
  VG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&h), (size_t)(cap + 1) * kSlots * sizeof(double), 0));
  VG_HIP(ctx, hipMemset(d, 0, kSlots * sizeof(double)));
  void* hd = nullptr;
  VG_HIP(ctx, hipHostGetDevicePointer(&hd, h, 0));
  ctx->d_log = d + kSlots;
  ctx->h_log = h + kSlots;
  ctx->h_log_dev = static_cast<double*>(hd) + kSlots;
  ctx->log_capacity = cap;
  return VGICP_OK;
}

uint32_t iterate_grid(const vgicp_ctx* ctx) {
  const uint32_t block = (uint32_t)ctx->iter_block - 64;  // wave 0 of a workgroup solves, the rest own points
  const uint32_t want = (ctx->n + block - 1) / block;
  return std::min<uint32_t>(std::max<uint32_t>(want, 1), kMaxIterBlocks);
}

IterArgs base_args(const vgicp_ctx* ctx) {
  IterArgs a;
  std::memset(&a, 0, sizeof a);
  a.scan = ctx->d_scan;
  a.stride = ctx->stride;
  a.n = ctx->n;
  a.mask = (uint32_t)(ctx->slots - 1);
  a.table = ctx->table;
  a.voxel_size = ctx->voxel_size;
  a.log = ctx->d_log;
  a.stamps = ctx->d_stamps;
  a.memo = static_cast<int4*>(ctx->d_memo);
  a.memo_valid = 0;   // the caller knows which launch of the align this is
  a.scan_seq = ctx->scan_seq;
  a.asym_dev = (ctx->scan_sym_known && !ctx->dev.no_sym) ? ctx->d_ins_counters + 2 : nullptr;
  // the dense record copy (tables far beyond the caches' reach): used where it is current — the aligns that reach the
  // loop after a persistent launch has rebuilt it, or run_align's own ensure_dense
  a.dense = (ctx->d_dense && ctx->dense_version == ctx->map_version && ctx->slots >= ctx->dense_slots_threshold &&
             ctx->dense_slots_threshold != 0) ? ctx->d_dense : nullptr;
  return a;
}

int load_rccl(vgicp_ctx* ctx) {
  if (ctx->rccl.lib) return VGICP_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(ctx, VGICP_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
  RcclApi api;
  api.lib = lib;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(lib, "ncclAllGather"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce)
    return fail(ctx, VGICP_ERR_RCCL, "librccl lacks a required symbol");
  ctx->rccl = api;
  return VGICP_OK;
}

int fail_rccl(const vgicp_ctx* ctx, int code, const char* what) {
  const char* txt = ctx->rccl.GetErrorString ? ctx->rccl.GetErrorString(code) : "?";
  return fail(ctx, VGICP_ERR_RCCL, std::string(what) + ": " + txt);
}

bool finite16(const double* m) {
  for (int i = 0; i < 16; ++i)
    if (!std::isfinite(m[i])) return false;
  return true;
}

// Enqueue launch j of an align on the context's stream.  Launch j's prologue closes round j-1 (fold
// its rows, solve, advance the pose) and its body accumulates round j; the launch after the last
// round is prologue-only and runs as a single workgroup (`closing`).  With a communicator each body
// launch is followed by this rank's row fold and the 256-byte all-reduce the next prologue reads.
int enqueue_launch(vgicp_ctx* ctx, const IterArgs& base, int j, uint32_t body_grid, bool closing,
                   bool use_comm) {
  IterArgs a = base;
  a.state_in = ctx->d_state + (j & 1);
  a.state_out = ctx->d_state + ((j + 1) & 1);
  a.rows = ctx->d_rows[j & 1];
  a.memo_valid = j > 0 ? 1u : 0u;   // launch 0 of an align writes every point's memo
  if (use_comm) {
    a.prev = ctx->d_sums;
    a.prev_rows = j > 0 ? 1u : 0u;
  } else {
    a.prev = ctx->d_rows[(j + 1) & 1];
    a.prev_rows = j > 0 ? body_grid : 0u;
  }
  if (closing) VG_HIP(ctx, launch_close(ctx->stream, a, ctx->iter_block));
  else VG_HIP(ctx, launch_iterate(ctx->stream, a, body_grid, ctx->iter_block));
  if (use_comm && !closing) {
    VG_HIP(ctx, launch_fold_rows(ctx->stream, a.rows, body_grid, a.state_out, ctx->d_sums));
    const int rc = ctx->rccl.AllReduce(ctx->d_sums, ctx->d_sums, kSlots, kNcclDouble, kNcclSum,
                                       ctx->comm, ctx->stream);
    if (rc != 0) return fail_rccl(ctx, rc, "ncclAllReduce");
  }
  return VGICP_OK;
}

int check_params(const vgicp_ctx* ctx, const vgicp_params* p) {
  if (!p) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "params is NULL");
  if (p->max_iteration < 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "max_iteration < 0");
  return VGICP_OK;
}

void pose_to_state(const double* m16, double* pose12) {
  Pose T;
  pose_from_mat4(m16, T);
  for (int k = 0; k < 9; ++k) pose12[k] = T.R[k];
  for (int k = 0; k < 3; ++k) pose12[9 + k] = T.t[k];
}
void state_to_pose(const double* pose12, double* m16) {
  Pose T;
  for (int k = 0; k < 9; ++k) T.R[k] = pose12[k];
  for (int k = 0; k < 3; ++k) T.t[k] = pose12[9 + k];
  pose_to_mat4(T, m16);
}

// Put the exchange buffers of the persistent launch into their initial state (everything unset, round 0):
// at context creation and after a launch that gave up.
int reset_persistent_exchange(vgicp_ctx* ctx) {
  const size_t rw = persistent_rows_words(), pw = persistent_parts_words();
  unsigned long long* img = static_cast<unsigned long long*>(ctx->h_exchange_image);
  persistent_exchange_image(ctx->persist_grid, img, img + rw);
  VG_HIP(ctx, hipMemcpyAsync(ctx->d_rows_persist, img, rw * 8, hipMemcpyHostToDevice, ctx->stream));
  VG_HIP(ctx, hipMemcpyAsync(ctx->d_parts_persist, img + rw, pw * 8, hipMemcpyHostToDevice, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->persist_round0 = 0;
  return VGICP_OK;
}

// A table that is far larger than what caches and TLBs reach (2^24 slots = 2 GiB and more: BASELINE config C5 has
// 8.6 GB) gets a dense copy of its FULL records for the several-points-per-thread launch: tools/micro/gather_pieces
// measured 6.7 ns per random 128-byte line and CU out of a 5-10 GB table against 5.4 ns out of 2.5 GB, and a cliff for
// more lines in flight above 4 GB.  Smaller tables (C2: 512 MB) never use it.  VGICP_DENSE_SLOTS (read when the context is created) overrides the threshold, 0 = never.
bool wants_dense(const vgicp_ctx* ctx, uint32_t n_upper) {
  return ctx->table && ctx->dense_slots_threshold != 0 && ctx->slots >= ctx->dense_slots_threshold && ctx->voxels > 0 &&
         (uint64_t)n_upper > (uint64_t)ctx->persist_grid * 448u;
}
// Storage of the dense copy: sized when the TABLE is (re)allocated (vgicp_map_reset, a growing upsert / insertion) —
// never inside an align.  The table keeps FULL + tombstones + incoming <= slots / 2, so slots / 2 records always suffice.
int reserve_dense(vgicp_ctx* ctx) {
  if (!ctx->table || ctx->dense_slots_threshold == 0 || ctx->slots < ctx->dense_slots_threshold) return VGICP_OK;
  const uint64_t cap = ctx->slots / 2;
  if (cap > ctx->dense_capacity) {
    if (ctx->d_dense) { VG_HIP(ctx, hipStreamSynchronize(ctx->stream)); VG_HIP(ctx, hipFree(ctx->d_dense)); }
    ctx->d_dense = nullptr;
    ctx->dense_capacity = 0;
    ctx->dense_version = 0;
    VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_dense), cap * sizeof(VoxelRecord)));
    ctx->dense_capacity = cap;
  }
  const uint32_t nb = table_dense_blocks(ctx->slots);
  if (nb + 1 > ctx->dense_counts_capacity) {
    if (ctx->d_dense_counts) { VG_HIP(ctx, hipStreamSynchronize(ctx->stream)); VG_HIP(ctx, hipFree(ctx->d_dense_counts)); }
    ctx->d_dense_counts = nullptr;
    ctx->dense_counts_capacity = 0;
    VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_dense_counts), (size_t)(nb + 1) * sizeof(uint32_t)));
    ctx->dense_counts_capacity = nb + 1;
  }
  return VGICP_OK;
}
// The align's part: rebuild the copy (three launches, no allocation) when the map changed since the last align.
// *usable = false when there is no storage for it (the threshold was lowered after the table was made): the launch then
// simply reads the table.
int ensure_dense(vgicp_ctx* ctx, bool* usable) {
  *usable = false;
  const uint32_t nb = table_dense_blocks(ctx->slots);
  if (!ctx->d_dense || ctx->dense_capacity < ctx->slots / 2 || nb + 1 > ctx->dense_counts_capacity) return VGICP_OK;
  *usable = true;
  if (ctx->dense_version == ctx->map_version) return VGICP_OK;
  VG_HIP(ctx, launch_table_dense(ctx->stream, ctx->table, ctx->slots, ctx->d_dense, ctx->dense_capacity, ctx->d_dense_counts));
  ctx->dense_version = ctx->map_version;
  return VGICP_OK;
}

// The whole align in one launch (single GPU). Returns VGICP_OK and *ran = true when the kernel
// completed; *ran = false when it gave up (the caller then uses launches).
int run_align_persistent(vgicp_ctx* ctx, const double* guess, const vgicp_params* params,
                         AlignState* result, bool* ran, float* device_ms) {
  *ran = false;
  static_assert(sizeof(AlignState) <= kSlots * sizeof(double), "the state must fit the log's header row");
  const uint32_t grid = ctx->persist_grid;  // always the same, all resident: the exchange buffers rely on it
  const int max_it = params->max_iteration;
  PersistArgs a;
  std::memset(&a, 0, sizeof a);
  a.scan = ctx->d_scan;
  a.stride = ctx->stride;
  a.n = ctx->n;                                        // a pending scan: the raw count, an upper bound ...
  a.n_dev = ctx->scan_pending ? ctx->d_counters : nullptr;  // ... and the kept count is read from the device
  a.asym_dev = ctx->scan_sym_known ? ctx->d_ins_counters + 2 : nullptr;  // word 2 of that block: the symmetry verdict
  a.scan_seq = ctx->scan_seq;
  if (ctx->dev.no_sym) a.asym_dev = nullptr;  // developer A/B: always read all twelve planes
  a.mask = (uint32_t)(ctx->slots - 1);
  a.table = ctx->table;
  if (wants_dense(ctx, ctx->n)) {
    bool usable = false;
    const int rc_dense = ensure_dense(ctx, &usable);   // a no-op unless the map changed since the last align; never allocates
    if (rc_dense != VGICP_OK) return rc_dense;
    if (usable) a.dense = ctx->d_dense;
  }
  a.voxel_size = ctx->voxel_size;
  a.rows = ctx->d_rows_persist;
  a.parts = ctx->d_parts_persist;
  a.round0 = ctx->persist_round0;
  // final state and per-round log go straight into pinned host memory (posted PCIe writes, 5.4 KB per align):
  // no copy-back to enqueue after the launch
  a.state = reinterpret_cast<AlignState*>(ctx->h_log_dev - kSlots);
  a.log = ctx->h_log_dev;
  // between GPUs the ranks' host threads reach the launch at slightly different times: a rank waits much longer
  // for a peer (~1 s) than for a workgroup of its own device (~50 ms) before it gives up
  a.spin_limit = (ctx->peers_connected && ctx->peer_world > 1) ? ctx->persist_spin_limit * 20u : ctx->persist_spin_limit;
  a.seq = ++ctx->persist_seq == 0 ? ++ctx->persist_seq : ctx->persist_seq;  // never 0
  pose_to_state(guess, a.pose0);
  a.cosine_threshold = params->cosine_threshold;
  a.translation_sq_threshold = params->translation_sq_threshold;
  a.max_iteration = max_it;
  persistent_lds_plan(ctx->n, grid, &a.memo_points, &a.stash_points, &a.stash_bytes, ctx->persist_lds_budget);
  if (ctx->dev.no_stash) a.stash_points = a.stash_bytes = 0;
  if (ctx->dev.no_memo) a.memo_points = 0;
  a.prefetch_margin = (a.memo_points == 0 && a.stash_points == 0 && ctx->n <= grid * 448u) ? ctx->prefetch_margin : 0.0;
  a.stamps = ctx->d_stamps;
  const bool multi = ctx->peers_connected && ctx->peer_world > 1;
  a.world = multi ? (uint32_t)ctx->peer_world : 1u;
  a.rank = multi ? (uint32_t)ctx->peer_rank : 0u;
  a.mail = ctx->d_mail_table;
  a.mail_round0 = ctx->mail_round0;
  a.mail_seq = multi ? ++ctx->mail_seq : 0u;
  // the launch reports into the header row of the pinned log: who gave up (any workgroup) and workgroup 0's verdict
  AlignState* header = reinterpret_cast<AlignState*>(ctx->h_log - kSlots);
  header->abort_seq = 0;
  header->outcome = kOutcomeNone;
  { const int rc_copy = fetch_insert_totals(ctx); if (rc_copy != VGICP_OK) return rc_copy; }   // normally carried by the preparation's copy
  // one launch, one synchronisation
  static const bool trace_align = std::getenv("VGICP_TRACE_ALIGN") != nullptr;   // developer aid: where the host time of an align goes
  const double ta0 = trace_align ? now_seconds() : 0.0;
  VG_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
  VG_HIP(ctx, launch_persistent(ctx->stream, a, grid));
  VG_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
  if (ctx->stage_events) VG_HIP(ctx, hipEventRecord(ctx->ev_stage[3], ctx->stream));
  const double ta1 = trace_align ? now_seconds() : 0.0;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double ta2 = trace_align ? now_seconds() : 0.0;
  VG_HIP(ctx, hipEventElapsedTime(device_ms, ctx->ev_begin, ctx->ev_end));
  if (trace_align && ta2 - ta0 > 2e-3)
    std::fprintf(stderr, "[vgicp trace] align: enqueue %.3f ms, hipStreamSynchronize %.3f ms, the launch itself %.3f ms (events)\n",
                 (ta1 - ta0) * 1e3, (ta2 - ta1) * 1e3, (double)*device_ms);
  std::memcpy(result, header, sizeof(AlignState));
  ++ctx->persistent_launches;
  {
    // the frame's ONE synchronisation has happened: what was deferred is known now (a pending scan's size and
    // verdict, the counts of the previous frame's map insertion)
    const int rc_scan = settle_scan(ctx);
    const int rc_ins = settle_insert(ctx);
    if (rc_scan != VGICP_OK || rc_ins != VGICP_OK) {
      // the launch itself may well have completed: keep the exchange buffers' rotation in step before reporting
      if (result->seq == a.seq && result->outcome == kOutcomeCommitted && result->abort_seq != a.seq) {
        ctx->persist_round0 = (ctx->persist_round0 + (uint32_t)result->iteration) % 3u;
        if (multi) ctx->mail_round0 += (uint32_t)result->iteration;
      } else {
        (void)reset_persistent_exchange(ctx);
      }
      return rc_scan != VGICP_OK ? rc_scan : rc_ins;
    }
  }
  const bool committed = result->seq == a.seq && result->outcome == kOutcomeCommitted;
  const bool someone_gave_up = result->abort_seq == a.seq;
  if (!committed || someone_gave_up) {
    // An in-kernel wait timed out (a workgroup was not resident: something else holds CUs of this device; or a
    // peer GPU did not deliver).  `someone_gave_up` with `committed`: workgroup 0 arrived late, found every row in
    // place and finished while another workgroup had already stopped waiting — its rows of the later rounds are
    // missing, the result must not be used.  Put the exchange back into its initial state, use the per-launch
    // loop for this align and the next few, then try the single launch again.
    ++ctx->persistent_fallbacks;
    if (ctx->owner && multi) {
      if (ctx->dev.verbose)
        std::fprintf(stderr, "[vgicp] rank %d of %d: persistent launch did not commit (echo %s, outcome %u, a workgroup gave up: %s, "
                     "rounds reported %d, %u points)\n", ctx->peer_rank, ctx->peer_world, result->seq == a.seq ? "yes" : "no",
                     result->outcome, someone_gave_up ? "yes" : "no", result->iteration, ctx->n);
      // a sub-context of an in-process multi-device context: every sub-context's launch has ended when its thread
      // returns, so the group itself re-arms all mailboxes and runs this align with the rows added on the host
      const int rc_reset = reset_persistent_exchange(ctx);
      return rc_reset != VGICP_OK ? rc_reset : vgicp_internal::kNeedGroupLoop;
    }
    ctx->persistent_cooldown = kPersistentCooldownAligns;
    if (ctx->persistent_fallbacks == 1 || ctx->dev.verbose)
      std::fprintf(stderr, "[vgicp] persistent align launch gave up waiting for a workgroup%s (fallback #%llu): using one "
                   "launch per iteration for the next %d aligns\n", multi ? " or a peer GPU" : "",
                   (unsigned long long)ctx->persistent_fallbacks, kPersistentCooldownAligns);
    int rc = reset_persistent_exchange(ctx);
    if (multi) {
      // Between GPUs the outcome is collective (the verdict words at the end of the launch): every rank leaves the
      // mailboxes for good in the SAME align and re-runs it through the host collective, so the all-reduces pair up.
      // A peer's kernel may still be writing into a mailbox, so they are not touched again.
      ctx->peer_enabled = false;
      const bool agreed = result->outcome == kOutcomeAgreedAbort || (result->outcome == kOutcomeNone && !committed);
      std::fprintf(stderr, "[vgicp] rank %d: the in-kernel exchange between GPUs gave up (%s); this communicator "
                   "continues with one launch + one RCCL all-reduce per iteration\n", ctx->peer_rank,
                   result->outcome == kOutcomeAgreedAbort ? "a peer reported it" :
                   result->outcome == kOutcomeNoAgreement ? "a peer's verdict never arrived" :
                   committed ? "a workgroup of this rank, after the verdict was sent" : "this rank timed out");
      if (rc == VGICP_OK && !agreed)
        return fail(ctx, VGICP_ERR_RCCL, "the ranks could not agree on the outcome of this align (a peer's verdict is missing "
                    "or this rank's verdict was sent before one of its workgroups gave up): not re-running it alone");
    }
    return rc;
  }
  ctx->persist_round0 = (ctx->persist_round0 + (uint32_t)result->iteration) % 3u;
  if (multi) ctx->mail_round0 += (uint32_t)result->iteration;
  *ran = true;
  return VGICP_OK;
}

int run_align(vgicp_ctx* ctx, const double* guess, const vgicp_params* params, double* out_pose,
              vgicp_stats* stats) {
  const double t0 = now_seconds();
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident: call vgicp_scan_upload first");
  int rc = check_params(ctx, params);
  if (rc != VGICP_OK) return rc;
  const int max_it = params->max_iteration;
  rc = ensure_log(ctx, max_it);
  if (rc != VGICP_OK) return rc;
  const bool profile = (params->flags & VGICP_FLAG_PROFILE) != 0;
  int chunk = params->chunk_iterations > 0 ? params->chunk_iterations : kDefaultChunk;
  if (profile) chunk = 1;

  const bool peer_path = ctx->peers_connected && ctx->peer_enabled && ctx->peer_world > 1;
  const bool alone = ctx->world_size == 1;  // also a communicator of one rank: nothing to exchange
  const bool single_launch = !(ctx->persistent_cooldown > 0 && !peer_path) && ctx->persistent_enabled &&
                             (alone || peer_path) && !profile && max_it > 0 &&
                             (params->flags & VGICP_FLAG_NO_PERSISTENT) == 0;
  if (ctx->owner && ctx->peer_world > 1 && !single_launch) return vgicp_internal::kNeedGroupLoop;  // the group's host-summed loop
  if (!single_launch) {
    // the launch-per-round loop sizes its grid from the scan: a pending scan has to be settled first
    rc = settle(ctx);
    if (rc != VGICP_OK) return rc;
    if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident");
  }
  if (ctx->persistent_cooldown > 0 && !peer_path) --ctx->persistent_cooldown;
  else if (single_launch) {
    bool ran = false;
    float ms = 0.f;
    AlignState* hf = &ctx->h_state[0];
    if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[2], ctx->stream)); ctx->ev_stage_set[2] = true; }
    rc = run_align_persistent(ctx, guess, params, &ctx->h_state[1], &ran, &ms);
    if (rc != VGICP_OK) return rc;
    if (ran) {
      if (ctx->stage_events) ctx->ev_stage_set[3] = true;
      *hf = ctx->h_state[1];
      state_to_pose(hf->pose, out_pose);
      if (stats) {
        stats->iterations = hf->iteration;
        stats->converged = hf->converged;
        stats->world_size = peer_path ? ctx->peer_world : 1;
        stats->launches = 1;
        stats->device_seconds = ms * 1e-3;
        for (int it = 0; it < hf->iteration; ++it) {
          const double* row = ctx->h_log + (size_t)it * kSlots;
          if (stats->corr_count) stats->corr_count[it] = (uint64_t)row[kCountSlot];
          if (stats->normal_eq) std::memcpy(stats->normal_eq + (size_t)it * kNormalEq, row, kNormalEq * sizeof(double));
        }
        stats->seconds = now_seconds() - t0;
      }
      if (!finite16(out_pose)) return fail(ctx, VGICP_ERR_DEGENERATE, "solved pose is not finite (singular normal equations)");
      return VGICP_OK;
    }
  }

  if (ctx->peers_connected && ctx->peer_world > 1 && ctx->comm == nullptr)
    return fail(ctx, VGICP_ERR_RCCL, "the in-kernel exchange between GPUs is not available for this align (gave up earlier, "
                "profiling or VGICP_FLAG_NO_PERSISTENT) and there is no RCCL communicator to fall back to");
  AlignState* h0 = &ctx->h_state[0];
  std::memset(h0, 0, sizeof(AlignState));
  pose_to_state(guess, h0->pose);
  h0->cosine_threshold = params->cosine_threshold;
  h0->translation_sq_threshold = params->translation_sq_threshold;
  h0->max_iteration = max_it;
  h0->done = (max_it == 0) ? 1 : 0;
  VG_HIP(ctx, hipMemcpyAsync(ctx->d_state, h0, sizeof(AlignState), hipMemcpyHostToDevice, ctx->stream));

  // a table far beyond the caches' reach: the loop reads remembered records from the dense copy too (rebuilt here
  // when the map changed since; storage was made with the table, nothing is allocated)
  if (ctx->table && ctx->dense_slots_threshold != 0 && ctx->slots >= ctx->dense_slots_threshold && ctx->voxels > 0) {
    bool usable = false;
    rc = ensure_dense(ctx, &usable);
    if (rc != VGICP_OK) return rc;
  }
  const IterArgs base = base_args(ctx);
  const uint32_t grid = iterate_grid(ctx);
  const bool use_comm = ctx->comm != nullptr;
  const int total_launches = max_it > 0 ? max_it + 1 : 0;  // max_it bodies + the closing prologue
  if (profile && (int)ctx->ev_prof.size() < 2 * total_launches) {
    const size_t old = ctx->ev_prof.size();
    ctx->ev_prof.resize(2 * (size_t)total_launches, nullptr);
    for (size_t k = old; k < ctx->ev_prof.size(); ++k) VG_HIP(ctx, hipEventCreate(&ctx->ev_prof[k]));
  }

  VG_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
  int launched = 0;
  int chunks_enqueued = 0, chunks_checked = 0;
  bool finished = total_launches == 0;
  // Keep up to two chunks in flight: enqueue chunk k+1 before looking at chunk k's status, so the
  // device never idles behind the host; launches enqueued past convergence exit at their first load.
  while (!finished) {
    while (launched < total_launches && chunks_enqueued - chunks_checked < kMaxChunksInFlight) {
      // the first chunk carries one extra launch: launch j closes round j-1
      const int todo = std::min(chunk + (launched == 0 ? 1 : 0), total_launches - launched);
      for (int k = 0; k < todo; ++k) {
        const int j = launched + k;
        if (profile) VG_HIP(ctx, hipEventRecord(ctx->ev_prof[2 * j], ctx->stream));
        rc = enqueue_launch(ctx, base, j, grid, /*closing=*/j == max_it, use_comm);
        if (rc != VGICP_OK) return rc;
        if (profile) VG_HIP(ctx, hipEventRecord(ctx->ev_prof[2 * j + 1], ctx->stream));
      }
      launched += todo;
      const int slot = chunks_enqueued % kMaxChunksInFlight;
      VG_HIP(ctx, hipMemcpyAsync(&ctx->h_state[1 + slot], ctx->d_state + (launched & 1),
                                 sizeof(AlignState), hipMemcpyDeviceToHost, ctx->stream));
      VG_HIP(ctx, hipEventRecord(ctx->ev_chunk[slot], ctx->stream));
      ++chunks_enqueued;
    }
    const int slot = chunks_checked % kMaxChunksInFlight;
    VG_HIP(ctx, hipEventSynchronize(ctx->ev_chunk[slot]));
    ++chunks_checked;
    if (ctx->h_state[1 + slot].done || (launched >= total_launches && chunks_checked == chunks_enqueued))
      finished = true;
  }
  VG_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
  AlignState* hf = &ctx->h_state[0];
  VG_HIP(ctx, hipMemcpyAsync(hf, ctx->d_state + (launched & 1), sizeof(AlignState),
                             hipMemcpyDeviceToHost, ctx->stream));
  const bool want_log = stats && (stats->corr_count || stats->normal_eq);
  if (want_log && max_it > 0)
    VG_HIP(ctx, hipMemcpyAsync(ctx->h_log, ctx->d_log, (size_t)max_it * kSlots * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));

  state_to_pose(hf->pose, out_pose);
  if (stats) {
    stats->iterations = hf->iteration;
    stats->converged = hf->converged;
    stats->world_size = ctx->world_size;
    stats->launches = launched;
    float ms = 0.f;
    VG_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    stats->device_seconds = ms * 1e-3;
    for (int it = 0; it < hf->iteration; ++it) {
      const double* row = ctx->h_log + (size_t)it * kSlots;
      if (stats->corr_count) stats->corr_count[it] = (uint64_t)row[kCountSlot];
      if (stats->normal_eq) std::memcpy(stats->normal_eq + (size_t)it * kNormalEq, row, kNormalEq * sizeof(double));
    }
    if (profile && stats->kernel_ms) {
      // one entry per body launch (the closing single-workgroup launch is not a round)
      for (int it = 0; it < std::min(launched, max_it); ++it) {
        float k = 0.f;
        VG_HIP(ctx, hipEventElapsedTime(&k, ctx->ev_prof[2 * it], ctx->ev_prof[2 * it + 1]));
        stats->kernel_ms[it] = k;
      }
    }
    stats->seconds = now_seconds() - t0;
  }
  if (!finite16(out_pose)) return fail(ctx, VGICP_ERR_DEGENERATE, "solved pose is not finite (singular normal equations)");
  return VGICP_OK;
}

}  // namespace

namespace {
int ensure_mailbox(vgicp_ctx* ctx) {
  if (ctx->d_mail) return VGICP_OK;
  // fine-grained: stores of another GPU's kernel become visible to this GPU's running kernel
  VG_HIP(ctx, hipExtMallocWithFlags(reinterpret_cast<void**>(&ctx->d_mail), kMailWords * 8, hipDeviceMallocFinegrained));
  VG_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_mail_table), kMaxRanks * sizeof(double*)));
  return VGICP_OK;
}

void close_peers(vgicp_ctx* ctx) {
  for (int r = 0; r < kMaxRanks; ++r) {
    if (ctx->peer_mail_is_ipc && ctx->peer_mail[r] && ctx->peer_mail[r] != ctx->d_mail) (void)hipIpcCloseMemHandle(ctx->peer_mail[r]);
    ctx->peer_mail[r] = nullptr;
  }
  ctx->peers_connected = false;
  ctx->peer_world = 1;
  ctx->peer_rank = 0;
}
}  // namespace

extern "C" {

int vgicp_abi_version(void) { return VGICP_ABI_VERSION; }

int vgicp_create(int device_id, vgicp_ctx** out) { return vgicp_internal::create_context(device_id, 0, out); }

}  // extern "C"

int vgicp_internal::create_context(int device_id, uint32_t max_persist_grid, vgicp_ctx** out) {
  if (!out) return fail(nullptr, VGICP_ERR_BAD_ARGUMENT, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(nullptr, VGICP_ERR_NO_DEVICE,
                std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  if (device_id < 0 || device_id >= count)
    return fail(nullptr, VGICP_ERR_BAD_ARGUMENT, "device_id out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceProperties");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, VGICP_ERR_NO_DEVICE,
                std::string("device is ") + prop.gcnArchName + ", this module is built for gfx950 only");
  vgicp_ctx* ctx = new vgicp_ctx;
  ctx->id = ++g_context_ids;
  ctx->device = device_id;
  ctx->cu_count = prop.multiProcessorCount;
  ctx->hbm_bytes = prop.totalGlobalMem;
  ctx->arch = prop.gcnArchName;
  auto bail = [&](hipError_t err, const char* what) {
    int rc = fail_hip(nullptr, err, what);
    delete ctx;
    return rc;
  };
#define VG_CREATE(call)                                  \
  do {                                                   \
    hipError_t e__ = (call);                             \
    if (e__ != hipSuccess) return bail(e__, #call);      \
  } while (0)
  VG_CREATE(hipSetDevice(device_id));
  VG_CREATE(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  // the insertion's running totals (4 words) sit right behind the counter block: ONE copy after a preparation brings
  // both back, so a deferred insertion needs no copy of its own in the frame chain
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), (kCounterWords + 4) * sizeof(uint32_t)));
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_counters), kCounterWords * sizeof(uint32_t), 0));
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_prep), (kCounterWords + 4) * sizeof(uint32_t), 0));
  VG_CREATE(hipMalloc(&ctx->d_tiles, preprocess_tile_bytes()));
  VG_CREATE(hipMemset(ctx->d_tiles, 0, preprocess_tile_bytes()));
  VG_CREATE(hipMemset(ctx->d_counters, 0, (kCounterWords + 4) * sizeof(uint32_t)));
  ctx->d_ins_counters = ctx->d_counters + kCounterWords;
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_fetch_hdr), 128, 0));
  std::memset(ctx->h_fetch_hdr, 0, 128);
  { void* dev = nullptr; VG_CREATE(hipHostGetDevicePointer(&dev, ctx->h_fetch_hdr, 0)); ctx->h_fetch_hdr_dev = static_cast<unsigned long long*>(dev); }
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_ins_counters), 4 * sizeof(uint32_t), 0));
  ctx->h_ins_counters[0] = ctx->h_ins_counters[1] = 0;
  if (const char* se = std::getenv("VGICP_STAGE_EVENTS"); se && se[0] == '1') {
    for (auto& e : ctx->ev_stage) VG_CREATE(hipEventCreate(&e));
    ctx->stage_events = true;
  }
  ctx->stat_launches0 = g_kernel_launches;
  ctx->stat_copies0 = g_copy_ops;
  ctx->stat_syncs0 = g_sync_ops;
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_state), 2 * sizeof(AlignState)));
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_state),
                          (1 + kMaxChunksInFlight) * sizeof(AlignState), 0));
  for (int k = 0; k < 2; ++k)
    VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_rows[k]),
                        (size_t)kMaxIterBlocks * kSlots * sizeof(double)));
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_rows_persist), persistent_rows_words() * 8));
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_parts_persist), persistent_parts_words() * 8));
  VG_CREATE(hipHostMalloc(&ctx->h_exchange_image, (persistent_rows_words() + persistent_parts_words()) * 8, 0));
  ctx->persist_grid = (uint32_t)std::min<int>(ctx->cu_count, kExchangeRows);
  if (max_persist_grid >= 1 && max_persist_grid < ctx->persist_grid) ctx->persist_grid = max_persist_grid;
  if (const char* pg = std::getenv("VGICP_PERSIST_GRID")) {  // fewer workgroups: several contexts sharing one device
    const long v = std::atol(pg);
    if (v >= 1 && v <= (long)ctx->persist_grid) ctx->persist_grid = (uint32_t)v;
  }
  if (const char* pe = std::getenv("VGICP_PERSISTENT")) ctx->persistent_enabled = pe[0] != '0';
  if (const char* ds = std::getenv("VGICP_DENSE_SLOTS")) ctx->dense_slots_threshold = std::strtoull(ds, nullptr, 10);
  if (const char* pm = std::getenv("VGICP_PREFETCH_MARGIN")) ctx->prefetch_margin = std::atof(pm);
  if (const char* sl = std::getenv("VGICP_SPIN_LIMIT")) ctx->persist_spin_limit = (uint32_t)std::strtoul(sl, nullptr, 10);
  if (const char* ut = std::getenv("VGICP_UPLOAD_THREADS")) ctx->upload_threads = std::max(1, std::min(16, std::atoi(ut)));
  if (const char* ul = std::getenv("VGICP_UPLOAD_STAGE_LIMIT")) ctx->upload_stage_limit = (size_t)std::max(0ll, std::atoll(ul));
  // developer / test switches: read here, once; no entry point looks at the environment again
  ctx->dev.no_sym = std::getenv("VGICP_NO_SYM") != nullptr;       // A/B: always read all twelve planes
  ctx->dev.no_stash = std::getenv("VGICP_NO_STASH") != nullptr;
  ctx->dev.no_memo = std::getenv("VGICP_NO_MEMO") != nullptr;
  ctx->dev.verbose = std::getenv("VGICP_VERBOSE") != nullptr;
  ctx->dev.insert_sort = std::getenv("VGICP_INSERT_SORT") != nullptr;
  if (const char* dp = std::getenv("VGICP_DEBUG_PREP")) ctx->dev.debug_prep = std::atoi(dp);
  if (const char* ps = std::getenv("VGICP_PACK_SPIN_LIMIT")) ctx->dev.pack_spin_limit = (uint32_t)std::strtoul(ps, nullptr, 10);
  if (const char* dd = std::getenv("VGICP_DEBUG_UPLOAD_DELAY_US")) ctx->dev.debug_upload_delay_us = std::atol(dd);
  {
    // the in-kernel exchange needs every workgroup resident: one 512-thread workgroup with the LARGEST dynamic LDS
    // a launch plan asks for (memo + parked points of a scan bigger than the grid: 150 KB) must fit a CU — checked
    // once here instead of found out by a timeout on every align
    uint32_t resident = 0;
    VG_CREATE(persistent_prepare_device());
    VG_CREATE(persistent_max_resident(persistent_max_dyn_lds_bytes(), ctx->cu_count, &resident));
    if (resident < ctx->persist_grid) {
      ctx->persistent_enabled = false;
      std::fprintf(stderr, "[vgicp] a persistent workgroup with %u bytes of LDS does not fit a compute unit of this device: "
                   "aligns use one launch per iteration\n", persistent_max_dyn_lds_bytes());
    }
  }
  if (const char* blk = std::getenv("VGICP_ITER_BLOCK")) {
    const int b = std::atoi(blk);
    if (b == 256 || b == 512 || b == 1024) ctx->iter_block = b;
  }
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_sums), kSlots * sizeof(double)));
  VG_CREATE(hipMemset(ctx->d_state, 0, 2 * sizeof(AlignState)));
  if (const char* dbg = std::getenv("VGICP_DEBUG_STAMPS"); dbg && (dbg[0] == '1' || dbg[0] == '2')) {
    VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_stamps), (32 + kExchangeRows) * sizeof(uint64_t)));
    VG_CREATE(hipMemset(ctx->d_stamps, 0, (32 + kExchangeRows) * sizeof(uint64_t)));
  }
  VG_CREATE(hipEventCreate(&ctx->ev_begin));
  VG_CREATE(hipEventCreate(&ctx->ev_end));
  for (int k = 0; k < kMaxChunksInFlight; ++k)
    VG_CREATE(hipEventCreateWithFlags(&ctx->ev_chunk[k], hipEventDisableTiming));
#undef VG_CREATE
  if (reset_persistent_exchange(ctx) != VGICP_OK) {
    g_create_error = ctx->err;
    vgicp_destroy(ctx);
    return VGICP_ERR_HIP;
  }
  *out = ctx;
  return VGICP_OK;
}

extern "C" {

int vgicp_destroy(vgicp_ctx* ctx) {
  if (!ctx) return VGICP_OK;
  if (ctx->multi) return vgicp_multi_api::destroy(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  bool crew_lost = false;   // a copy thread that never came back may still write the staging memory: both are leaked then
  if (ctx->crew) {
    crew_lost = ctx->crew->broken;
    ctx->crew->stop();
    if (!crew_lost) delete ctx->crew;
    ctx->crew = nullptr;
  }
  if (ctx->h_upload && !crew_lost) (void)hipHostFree(ctx->h_upload);
  if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
  for (auto& s : ctx->ahead) {
    if (s.mem) (void)hipHostFree(s.mem);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  close_peers(ctx);
  if (ctx->d_mail) (void)hipFree(ctx->d_mail);
  if (ctx->d_mail_table) (void)hipFree(ctx->d_mail_table);
  if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
  if (ctx->d_stamps) {
    uint64_t h[32] = {0};
    if (hipMemcpy(h, ctx->d_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[4] > 0) {
      const double k = 0.01 / (double)h[4];  // 100 MHz ticks -> us per launch
      std::fprintf(stderr, "[vgicp stamps] body launches %llu | workgroup 0, first worker lane: loads+fold+barrier %.2f us, "
                   "speculative probe || solve, to 2nd barrier %.2f us, verify+accumulate loop %.2f us, "
                   "butterfly+row store %.2f us | solver wave: solve+publish %.2f us\n", (unsigned long long)h[4], h[0] * k,
                   h[5] * k, h[1] * k, h[2] * k, h[6] * k);
    }
    if (hipMemcpy(h, ctx->d_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[13] > 0) {
      for (int o = 8; o <= 16; o += 8) {
        const double k = 0.01 / (double)h[o + 5];  // 100 MHz ticks -> us per round
        std::fprintf(stderr, "[vgicp stamps] persistent, workgroup 0 %s, %llu rounds: accumulate+butterfly (to the barrier) "
                     "%.2f us, publish + level-1 fold %.2f us, level-2 poll %.2f us, solve+broadcast %.2f us\n",
                     o == 8 ? "solver wave" : "first worker lane", (unsigned long long)h[o + 5], h[o] * k, h[o + 1] * k,
                     h[o + 2] * k, h[o + 3] * k);
      }
      const double kf = 0.01 / (double)h[13];
      std::fprintf(stderr, "[vgicp stamps] inside solve+broadcast (solver wave of workgroup 0): re-arm + totals through LDS to registers "
                   "%.3f us, LDL^T %.3f us, exponential + compose + test %.3f us, pose to LDS (+ state, workgroup 0) %.3f us, the rest "
                   "(barrier, pose read by every wave) %.3f us\n", h[24] * kf, h[25] * kf, h[26] * kf, h[27] * kf,
                   (h[11] - h[24] - h[25] - h[26] - h[27]) * kf);
    }
    uint64_t wg[kExchangeRows];
    if (h[13] > 0 && hipMemcpy(wg, ctx->d_stamps + 32, sizeof wg, hipMemcpyDeviceToHost) == hipSuccess) {
      const double k = 0.01 / (double)h[13];
      double lo = 1e30, hi = 0.0, sum = 0.0;
      int hi_at = 0;
      const int g = (int)ctx->persist_grid;
      for (int b = 0; b < g; ++b) {
        const double v = wg[b] * k;
        sum += v;
        if (v < lo) lo = v;
        if (v > hi) { hi = v; hi_at = b; }
      }
      std::fprintf(stderr, "[vgicp stamps] persistent, time to the first barrier per workgroup (mean over rounds): min %.2f us, "
                   "mean %.2f us, max %.2f us (workgroup %d)\n", lo, sum / g, hi, hi_at);
      if (const char* all = std::getenv("VGICP_DEBUG_STAMPS"); all && all[0] == '2') {  // every workgroup's figure
        for (int b = 0; b < g; ++b) std::fprintf(stderr, "%s%.2f", b % 16 ? " " : "\n[vgicp stamps wg] ", wg[b] * k);
        std::fprintf(stderr, "\n");
      }
    }
    (void)hipFree(ctx->d_stamps);
  }
  (void)hipFree(ctx->table);
  (void)hipFree(ctx->d_dense);
  (void)hipFree(ctx->d_dense_counts);
  (void)hipFree(ctx->d_counters);
  (void)hipHostFree(ctx->h_counters);
  (void)hipHostFree(ctx->h_prep);
  (void)hipFree(ctx->d_tiles);
  (void)hipHostFree(ctx->h_ins_counters);
  for (int k = 0; k < 2; ++k) {
    if (ctx->h_state_table[k]) (void)hipHostFree(ctx->h_state_table[k]);
    if (ctx->h_raw_stage[k] && !crew_lost) (void)hipHostFree(ctx->h_raw_stage[k]);
    if (k == 0 && ctx->h_arena) (void)hipHostFree(ctx->h_arena);
    if (ctx->ev_state_table[k]) (void)hipEventDestroy(ctx->ev_state_table[k]);
  }
  for (auto& e : ctx->ev_stage) if (e) (void)hipEventDestroy(e);
  if (ctx->h_fetch_hdr) (void)hipHostFree(ctx->h_fetch_hdr);
  if (ctx->h_fetch) (void)hipHostFree(ctx->h_fetch);
  (void)hipFree(ctx->d_stage);
  (void)hipFree(ctx->d_cells);
  (void)hipFree(ctx->d_scan);
  (void)hipFree(ctx->d_scan_aos);
  (void)hipFree(ctx->d_memo);
  (void)hipFree(ctx->d_state);
  (void)hipHostFree(ctx->h_state);
  (void)hipFree(ctx->d_rows_persist);
  (void)hipFree(ctx->d_parts_persist);
  (void)hipHostFree(ctx->h_exchange_image);
  (void)hipFree(ctx->d_rows[0]);
  (void)hipFree(ctx->d_rows[1]);
  (void)hipFree(ctx->d_sums);
  if (ctx->d_log) (void)hipFree(ctx->d_log - kSlots);
  if (ctx->h_log) (void)hipHostFree(ctx->h_log - kSlots);
  if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
  if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
  for (auto& e : ctx->ev_chunk) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_prof) if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (g_stage_error_ctx == ctx->id) g_stage_error_ctx = 0;
  delete ctx;
  return VGICP_OK;
}

const char* vgicp_last_error(const vgicp_ctx* ctx) {
  if (ctx && g_stage_error_ctx == ctx->id) return g_stage_error.c_str();   // this thread's last failure was a vgicp_sweep_stage*
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int vgicp_device_info(const vgicp_ctx* ctx, char* name, size_t name_len, int32_t* cu_count,
                      uint64_t* hbm_bytes) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::device_info(ctx, name, name_len, cu_count, hbm_bytes);
  if (name && name_len) {
    std::strncpy(name, ctx->arch.c_str(), name_len - 1);
    name[name_len - 1] = '\0';
  }
  if (cu_count) *cu_count = ctx->cu_count;
  if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
  return VGICP_OK;
}

int vgicp_get_counter(const vgicp_ctx* ctx, int which, uint64_t* value) {
  if (!ctx || !value) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::get_counter(ctx, which, value);
  switch (which) {
    case VGICP_COUNTER_PERSISTENT_LAUNCHES: *value = ctx->persistent_launches; break;
    case VGICP_COUNTER_PERSISTENT_FALLBACKS: *value = ctx->persistent_fallbacks; break;
    case VGICP_COUNTER_UPLOAD_BYTES: *value = ctx->upload_bytes; break;
    case VGICP_COUNTER_UPLOAD_NANOSECONDS: *value = (uint64_t)(ctx->upload_seconds * 1e9); break;
    case VGICP_COUNTER_PREP_INDEFINITE: {
      // a preparation that was only enqueued has not reported yet: bring it up to date like every other reader
      const int rc_settle = settle(const_cast<vgicp_ctx*>(ctx));
      if (rc_settle != VGICP_OK) return rc_settle;
      *value = ctx->prep_indefinite;
      break;
    }
    case VGICP_COUNTER_SCAN_GENERATION: *value = ctx->scan_generation; break;
    case VGICP_COUNTER_UPLOAD_SLOW: *value = ctx->upload_slow; break;
    default: return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "unknown counter");
  }
  return VGICP_OK;
}

int vgicp_map_reset(vgicp_ctx* ctx, double voxel_size, size_t capacity_hint) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_reset(ctx, voxel_size, capacity_hint);
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!(voxel_size > 0.0) || !std::isfinite(voxel_size))
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "voxel_size must be positive and finite");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->table) VG_HIP(ctx, hipFree(ctx->table));
  ctx->table = nullptr;
  ctx->slots = ctx->voxels = ctx->tombstones = 0;
  ++ctx->map_version;
  ctx->voxel_size = voxel_size;
  const uint64_t slots = next_pow2(std::max<uint64_t>(kMinSlots, (uint64_t)capacity_hint * 4));
  int rc = alloc_table(ctx, slots, &ctx->table);
  if (rc != VGICP_OK) return rc;
  ctx->slots = slots;
  rc = reserve_dense(ctx);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VGICP_OK;
}

int vgicp_map_upsert(vgicp_ctx* ctx, size_t n, const int32_t* keys, const double* means,
                     const double* covs) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_upsert(ctx, n, keys, means, covs);
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (n == 0) return VGICP_OK;
  if (!keys || !means || !covs) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL batch pointer");
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "batch too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_table(ctx, n);
  if (rc != VGICP_OK) return rc;
  const size_t kb = n * 3 * sizeof(int32_t), mb = n * 3 * sizeof(double), cb = n * 9 * sizeof(double);
  const size_t koff = 0, moff = (kb + 255) & ~size_t(255), coff = moff + mb, qoff = (coff + cb + 255) & ~size_t(255);
  rc = ensure_stage(ctx, qoff + n * sizeof(uint32_t));
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, base + koff, keys, kb));
  VG_RC(user_h2d(ctx, base + moff, means, mb));
  VG_RC(user_h2d(ctx, base + coff, covs, cb));
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_upsert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), (uint32_t)n,
                            reinterpret_cast<const int32_t*>(base + koff),
                            reinterpret_cast<const double*>(base + moff),
                            reinterpret_cast<const double*>(base + coff), ctx->d_counters,
                            reinterpret_cast<uint32_t*>(base + qoff)));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t),
                             hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels += ctx->h_counters[0];
  if (ctx->h_counters[1] != 0) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table probe sequence exhausted");
  return VGICP_OK;
}

int vgicp_map_erase(vgicp_ctx* ctx, size_t n, const int32_t* keys) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_erase(ctx, n, keys);
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (n == 0) return VGICP_OK;
  if (!keys) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL batch pointer");
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "batch too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t kb = n * 3 * sizeof(int32_t);
  int rc = ensure_stage(ctx, kb);
  if (rc != VGICP_OK) return rc;
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, ctx->d_stage, keys, kb));
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_erase(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), (uint32_t)n,
                           static_cast<const int32_t*>(ctx->d_stage), ctx->d_counters));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t),
                             hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels -= ctx->h_counters[0];
  ctx->tombstones += ctx->h_counters[0];
  return VGICP_OK;
}

int vgicp_map_size(const vgicp_ctx* ctx, size_t* voxels, size_t* table_slots) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_size(ctx, voxels, table_slots);
  { const int rc_settle = settle(const_cast<vgicp_ctx*>(ctx)); if (rc_settle != VGICP_OK) return rc_settle; }  // a deferred insertion
  if (voxels) *voxels = ctx->voxels;
  if (table_slots) *table_slots = ctx->slots;
  return VGICP_OK;
}

int vgicp_map_insert_scan(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                          const double transform[16], size_t max_points_per_voxel, size_t* new_voxels) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_insert_scan(ctx, n, points, covs, transform, max_points_per_voxel, new_voxels);
  if (new_voxels) *new_voxels = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (n == 0) return VGICP_OK;
  if (!points || !covs || !transform) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (max_points_per_voxel == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "max_points_per_voxel must be >= 1");
  if (n > 0x7FFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_table(ctx, n);  // every point may open a voxel
  if (rc != VGICP_OK) return rc;
  const size_t pb = ((n * 3 * sizeof(double)) + 255) & ~size_t(255);
  const size_t cb = ((n * 9 * sizeof(double)) + 255) & ~size_t(255);
  const size_t sb = map_insert_scratch_bytes((uint32_t)n);
  rc = ensure_stage(ctx, pb + cb + sb);
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  double pose12[12];
  pose_to_state(transform, pose12);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, base, points, n * 3 * sizeof(double)));
  VG_RC(user_h2d(ctx, base + pb, covs, n * 9 * sizeof(double)));
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_map_insert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), ctx->voxel_size,
                                reinterpret_cast<const double*>(base), reinterpret_cast<const double*>(base + pb),
                                (uint32_t)n, pose12, (uint64_t)max_points_per_voxel, base + pb + cb, sb,
                                ctx->d_counters));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels += ctx->h_counters[0];
  if (new_voxels) *new_voxels = ctx->h_counters[0];
  if (ctx->h_counters[1] != 0) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table probe sequence exhausted");
  return VGICP_OK;
}

namespace {
// A scan the device down-sampled itself holds one point per voxel of ITS grid: a voxel of the map then receives at
// most (map voxel / scan voxel + 1)^3 of them, and when that is a handful the insertion goes without its sort
// (launch_map_insert, short_lists).  Any other resident scan (uploaded as it came) keeps the sort.
bool insertion_lists_stay_short(const vgicp_ctx* ctx) {
  if (!(ctx->prep_voxel > 0.0) || ctx->dev.insert_sort) return false;
  const double per_axis = std::ceil(ctx->voxel_size / ctx->prep_voxel) + 1.0;
  return per_axis * per_axis * per_axis <= 64.0;
}
}  // namespace

int vgicp_map_insert_resident(vgicp_ctx* ctx, const double transform[16], size_t max_points_per_voxel,
                              size_t* new_voxels) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_insert_resident(ctx, transform, max_points_per_voxel, new_voxels, false);
  if (new_voxels) *new_voxels = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident: call vgicp_scan_upload first");
  if (!transform) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (max_points_per_voxel == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "max_points_per_voxel must be >= 1");
  if ((ctx->comm || ctx->peers_connected) && !ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "resident scan is a shard: use vgicp_map_insert_scan with the whole scan");
  const size_t n = ctx->n;
  if (n == 0) return VGICP_OK;
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_table(ctx, n);
  if (rc != VGICP_OK) return rc;
  const size_t sb = map_insert_scratch_bytes((uint32_t)n);
  rc = ensure_stage(ctx, sb);
  if (rc != VGICP_OK) return rc;
  double pose12[12];
  pose_to_state(transform, pose12);
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_map_insert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), ctx->voxel_size,
                                ctx->d_scan_aos, ctx->d_scan_aos + 3 * ctx->scan_capacity, (uint32_t)n, pose12,
                                (uint64_t)max_points_per_voxel, ctx->d_stage, sb, ctx->d_counters,
                                insertion_lists_stay_short(ctx)));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels += ctx->h_counters[0];
  if (new_voxels) *new_voxels = ctx->h_counters[0];
  if (ctx->h_counters[1] != 0) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table probe sequence exhausted");
  return VGICP_OK;
}

int vgicp_map_insert_resident_async(vgicp_ctx* ctx, const double transform[16], size_t max_points_per_voxel) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_insert_resident(ctx, transform, max_points_per_voxel, nullptr, true);
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident: call vgicp_scan_upload first");
  if (!transform) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (max_points_per_voxel == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "max_points_per_voxel must be >= 1");
  if ((ctx->comm || ctx->peers_connected) && !ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "resident scan is a shard: use vgicp_map_insert_scan with the whole scan");
  // the scan's size has to be known (the align that registered it has settled it); an insertion still pending
  // from an earlier frame is settled by the same synchronisation
  int rc = (ctx->scan_pending || ctx->insert_pending) ? settle(ctx) : VGICP_OK;
  if (rc != VGICP_OK) return rc;
  const size_t n = ctx->n;
  if (n == 0) return VGICP_OK;
  VG_HIP(ctx, hipSetDevice(ctx->device));
  rc = ensure_table(ctx, n);  // every point may open a voxel (grows / rehashes with a synchronisation when it has to)
  if (rc != VGICP_OK) return rc;
  const size_t sb = map_insert_scratch_bytes((uint32_t)n);
  rc = ensure_stage(ctx, sb);
  if (rc != VGICP_OK) return rc;
  double pose12[12];
  pose_to_state(transform, pose12);
  if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[4], ctx->stream)); ctx->ev_stage_set[4] = true; }
  ++ctx->map_version;
  VG_HIP(ctx, launch_map_insert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), ctx->voxel_size,
                                ctx->d_scan_aos, ctx->d_scan_aos + 3 * ctx->scan_capacity, (uint32_t)n, pose12,
                                (uint64_t)max_points_per_voxel, ctx->d_stage, sb, ctx->d_ins_counters,
                                insertion_lists_stay_short(ctx)));
  if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[5], ctx->stream)); ctx->ev_stage_set[5] = true; }
  ctx->insert_pending = true;
  ctx->ins_copy_enqueued = false;   // the next preparation's counter copy carries the totals (or settle() fetches them)
  ctx->insert_pending_upper = n;
  return VGICP_OK;
}

int vgicp_get_frame_stats(vgicp_ctx* ctx, vgicp_frame_stats* out, int reset) {
  if (!ctx || !out) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::get_frame_stats(ctx, out, reset);
  std::memset(out, 0, sizeof *out);
  out->kernel_launches = g_kernel_launches - ctx->stat_launches0;
  out->copies = g_copy_ops - ctx->stat_copies0;
  out->host_syncs = g_sync_ops - ctx->stat_syncs0;
  out->prepare_us = out->align_us = out->insert_us = out->prepare_head_us = -1.0;
  if (ctx->stage_events) {
    VG_HIP(ctx, hipSetDevice(ctx->device));
    float ms = 0.f;
    if (ctx->ev_stage_set[0] && ctx->ev_stage_set[1] && hipEventElapsedTime(&ms, ctx->ev_stage[0], ctx->ev_stage[1]) == hipSuccess)
      out->prepare_us = ms * 1e3;
    if (ctx->ev_stage_set[0] && ctx->ev_stage_set[6] && hipEventElapsedTime(&ms, ctx->ev_stage[0], ctx->ev_stage[6]) == hipSuccess)
      out->prepare_head_us = ms * 1e3;
    if (ctx->ev_stage_set[2] && ctx->ev_stage_set[3] && hipEventElapsedTime(&ms, ctx->ev_stage[2], ctx->ev_stage[3]) == hipSuccess)
      out->align_us = ms * 1e3;
    if (ctx->ev_stage_set[4] && ctx->ev_stage_set[5] && hipEventElapsedTime(&ms, ctx->ev_stage[4], ctx->ev_stage[5]) == hipSuccess)
      out->insert_us = ms * 1e3;
  }
  if (reset) {
    ctx->stat_launches0 = g_kernel_launches;
    ctx->stat_copies0 = g_copy_ops;
    ctx->stat_syncs0 = g_sync_ops;
  }
  return VGICP_OK;
}

int vgicp_set_option(vgicp_ctx* ctx, int option, int value) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::set_option(ctx, option, value);
  switch (option) {
    case VGICP_OPTION_STAGE_EVENTS:
      VG_HIP(ctx, hipSetDevice(ctx->device));
      if (value && !ctx->ev_stage[0])
        for (auto& e : ctx->ev_stage) VG_HIP(ctx, hipEventCreate(&e));
      ctx->stage_events = value != 0;
      for (bool& b : ctx->ev_stage_set) b = false;
      return VGICP_OK;
    case VGICP_OPTION_UPLOAD_STAGE_KB:
      if (value < 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "negative size");
      ctx->upload_stage_limit = (size_t)value << 10;
      return VGICP_OK;
    case VGICP_OPTION_REFERENCE_ORDER:
      ctx->reference_order = value != 0;
      return VGICP_OK;
    default:
      return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "unknown option");
  }
}

int vgicp_map_evict(vgicp_ctx* ctx, const double position[3], double distance_threshold, size_t* removed) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_evict(ctx, position, distance_threshold, removed);
  if (removed) *removed = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (!position) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_map_evict(ctx->stream, ctx->table, ctx->slots, ctx->voxel_size, position,
                               distance_threshold, ctx->d_counters));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels -= ctx->h_counters[0];
  ctx->tombstones += ctx->h_counters[0];
  if (removed) *removed = ctx->h_counters[0];
  return VGICP_OK;
}

int vgicp_map_export(vgicp_ctx* ctx, size_t capacity, int32_t* keys, double* means, double* covs,
                     uint64_t* counts, size_t* written) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::map_export(ctx, capacity, keys, means, covs, counts, written);
  if (!written) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "written is NULL");
  *written = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (capacity == 0 || ctx->voxels == 0) return VGICP_OK;
  if (!keys || !means || !covs || !counts) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL array pointer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t cap = std::min<size_t>(capacity, ctx->voxels);
  const size_t kb = (cap * 3 * sizeof(int32_t) + 255) & ~size_t(255);
  const size_t mb = (cap * 3 * sizeof(double) + 255) & ~size_t(255);
  const size_t cb = (cap * 9 * sizeof(double) + 255) & ~size_t(255);
  const size_t nb = cap * sizeof(uint64_t);
  int rc = ensure_stage(ctx, kb + mb + cb + nb);
  if (rc != VGICP_OK) return rc;
  char* b = static_cast<char*>(ctx->d_stage);
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  VG_HIP(ctx, launch_map_export(ctx->stream, ctx->table, ctx->slots, (uint32_t)cap,
                                reinterpret_cast<int32_t*>(b), reinterpret_cast<double*>(b + kb),
                                reinterpret_cast<double*>(b + kb + mb), reinterpret_cast<uint64_t*>(b + kb + mb + cb),
                                ctx->d_counters));
  arena_reset(ctx);
  VG_RC(user_d2h(ctx, keys, b, cap * 3 * sizeof(int32_t)));
  VG_RC(user_d2h(ctx, means, b + kb, cap * 3 * sizeof(double)));
  VG_RC(user_d2h(ctx, covs, b + kb + mb, cap * 9 * sizeof(double)));
  VG_RC(user_d2h(ctx, counts, b + kb + mb + cb, cap * sizeof(uint64_t)));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  user_copies_finish(ctx);
  *written = cap;
  return VGICP_OK;
}

namespace {
// Copy the scan to the device and pack it into the SoA planes (reference: the deep copy of the cloud at
// src/Registration.cpp:11, which here is the copy to the device).  The caller's buffers are ordinary pageable memory
// (std::vector storage) that the caller may free on return, as the reference frees its cloud every frame
// (src/Odometry.cpp:84-87) — so the runtime must never get to register them (a freed registered range takes every
// queue of the process off the device for ~20 ms).  The copy crew (vgicp_context.h) moves the scan into page-locked
// staging memory of the context, this thread and `upload_threads - 1` helpers, unit by unit, while ONE kernel launch
// reads the staged units over PCIe behind them and packs them: 9.6 MB in 0.19 - 0.20 ms, the link's rate, where the
// runtime's in-place path took 0.27 - 0.6 ms and staging with copy commands 0.45 - 0.68 ms.  The pack kernel (and whatever
// the caller enqueues next) runs in stream order; nothing on the DEVICE is waited for here, but the copy threads are:
// the caller's buffers are free again on return.
// Page-locked buffers (vgicp_host_register, hipHostMalloc) are read by the copy engine in place.  In place as well,
// through the runtime's pin-on-the-fly path: scans larger than the stage limit (default 512 MB) and every scan with
// the limit 0 (VGICP_OPTION_UPLOAD_STAGE_KB / VGICP_UPLOAD_STAGE_LIMIT / VGICP_STAGE_LIMIT=0) — for callers that keep
// their buffers.
constexpr uint32_t kPackSpinLimit = 400000;   // polls of a staged unit's flag (>= 1 us each) before the pack kernel gives up
constexpr double kCrewSlowSeconds = 0.1;      // copy threads slower than this: the packing is repeated behind the launch

// CopyCrew::finish() ran into its deadline: a helper thread took a unit of the upload and never delivered it.  The kernel
// that waits for that unit's flag gives up by itself (kPackSpinLimit); the scan is not resident; the context copies
// alone from now on.  The one thing that cannot be taken back is that helper's pointer into the caller's buffer.
int crew_gave_up(vgicp_ctx* ctx) {
  ctx->scan_ready = false;
  ctx->upload_threads = 1;
  (void)hipStreamSynchronize(ctx->stream);
  // that thread may still write the staging memory it was copying into: later uploads get memory of their own
  ctx->h_upload = nullptr;
  ctx->upload_cap = ctx->upload_flag_bytes = 0;
  for (int k = 0; k < 2; ++k) { ctx->h_raw_stage[k] = nullptr; ctx->raw_stage_cap[k] = 0; }
  return fail(ctx, VGICP_ERR_TIMEOUT,
              "a copy thread of the scan upload did not deliver its unit within 10 s (dead or never scheduled): the scan is "
              "not resident, this context stages alone from now on; that thread may still read the caller's buffer");
}

int ensure_upload_stage(vgicp_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->upload_cap) return VGICP_OK;
  if (ctx->upload_in_flight) { VG_HIP(ctx, hipEventSynchronize(ctx->ev_upload)); ctx->upload_in_flight = false; }
  if (ctx->h_upload) VG_HIP(ctx, hipHostFree(ctx->h_upload));
  ctx->h_upload = nullptr;
  ctx->upload_cap = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 4, 4u << 20);
  const size_t flag_bytes = (want / ((size_t)pack_arena_unit() * kScanPlanes * sizeof(double)) + 2) * 64;
  VG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_upload), flag_bytes + want, 0));
  std::memset(ctx->h_upload, 0, flag_bytes);   // "no upload yet" (a sequence number is never 0)
  ctx->upload_cap = want;
  ctx->upload_flag_bytes = flag_bytes;
  return VGICP_OK;
}

int scan_upload_enqueue(vgicp_ctx* ctx, size_t n, const double* points, const double* covs) {
  if (n > 0 && (!points || !covs)) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL scan pointer");
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_scan(ctx, n);
  if (rc != VGICP_OK) return rc;
  ++ctx->scan_generation;
  ctx->scan_ready = false;
  ctx->prep_voxel = 0.0;
  ctx->prep_with_deskew = false;   // what vgicp_scan_info reports belongs to a PREPARED scan, not to this one
  ctx->prep_deskewed = 0;
  ctx->prep_indefinite = 0;
  ctx->n = (uint32_t)n;
  ctx->stride = ctx->scan_capacity;
  if (n == 0) return VGICP_OK;
  const double t0 = now_seconds();
  double* aos_pts = ctx->d_scan_aos;
  double* aos_cov = ctx->d_scan_aos + 3 * ctx->scan_capacity;
  if (++ctx->scan_seq == 0) ++ctx->scan_seq;
  ctx->scan_sym_known = true;
  const size_t bytes = n * kScanPlanes * sizeof(double);
  static const bool stage_off = std::getenv("VGICP_STAGE_LIMIT") && std::atoll(std::getenv("VGICP_STAGE_LIMIT")) == 0;
  const size_t whole_bytes = ctx->upload_whole_hint ? ctx->upload_whole_hint : bytes;
  const bool staged = !stage_off && whole_bytes <= ctx->upload_stage_limit && bytes > (256u << 10) &&
                      !(is_pagelocked(points) && is_pagelocked(covs));
  if (staged) {
    const uint32_t unit = pack_arena_unit(), units = (uint32_t)((n + unit - 1) / unit);
    const size_t pb = (n * 3 * sizeof(double) + 255 + 16) & ~size_t(255), cb = (n * 9 * sizeof(double) + 255 + 16) & ~size_t(255);
    rc = ensure_upload_stage(ctx, pb + cb);
    if (rc != VGICP_OK) return rc;
    if (!ctx->ev_upload) VG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    // the kernel that read the staging memory last has long finished (every align ends in a synchronisation); make sure
    if (ctx->upload_in_flight && hipEventQuery(ctx->ev_upload) != hipSuccess) VG_HIP(ctx, hipEventSynchronize(ctx->ev_upload));
    ctx->upload_in_flight = false;
    const bool want_helpers = ctx->upload_threads > 1 && bytes >= (2u << 20);
    if (!ctx->crew) ctx->crew = new CopyCrew;
    CopyCrew* crew = ctx->crew;
    if (want_helpers && crew->th.empty()) crew->start(ctx->upload_threads - 1);
    crew->pts = reinterpret_cast<const char*>(points);
    crew->cov = reinterpret_cast<const char*>(covs);
    crew->flags = reinterpret_cast<uint32_t*>(ctx->h_upload);
    crew->apts = ctx->h_upload + ctx->upload_flag_bytes;
    crew->acov = crew->apts + pb;
    crew->n = (uint32_t)n;
    crew->unit = unit;
    crew->units = units;
    crew->seq = ctx->scan_seq;
    crew->size_a = 3 * sizeof(double);
    crew->size_b = 9 * sizeof(double);
    crew->copy = stage_copy;
    crew->copy_b_form = stage_cov_unit;
    const double t_post = now_seconds();
    const uint32_t job = crew->post(want_helpers);
    // the launch first (it starts reading as soon as unit 0 is published), then this thread copies too
    // test aids: a pack kernel with little patience and a copy thread that is held up (the repeat below is then what counts)
    const uint32_t spin_limit = ctx->dev.pack_spin_limit ? ctx->dev.pack_spin_limit : kPackSpinLimit;
    const long debug_delay_us = ctx->dev.debug_upload_delay_us;
    const hipError_t e_launch = launch_pack_arena(ctx->stream, crew->apts, crew->acov, (uint32_t)n, crew->flags, true, ctx->scan_seq,
                                                  spin_limit, aos_pts, aos_cov, ctx->d_scan, ctx->stride,
                                                  ctx->d_ins_counters + 2);
    if (debug_delay_us > 0 && !want_helpers) std::this_thread::sleep_for(std::chrono::microseconds(debug_delay_us));
    crew->work(job);
    const bool crew_done = crew->finish();   // always: the caller's buffers must not be in use on return
    if (e_launch != hipSuccess) return fail_hip(ctx, e_launch, "launch_pack_arena");
    if (!crew_done) return crew_gave_up(ctx);
    if (now_seconds() - t_post > kCrewSlowSeconds) {
      // the copy threads were held up for so long that a workgroup of the launch may have stopped waiting: everything
      // is staged now, pack it again behind the launch (no flags to wait for)
      ++ctx->upload_slow;
      VG_HIP(ctx, launch_pack_arena(ctx->stream, crew->apts, crew->acov, (uint32_t)n, crew->flags, false, ctx->scan_seq, 0, aos_pts,
                                    aos_cov, ctx->d_scan, ctx->stride, ctx->d_ins_counters + 2));
    }
    VG_HIP(ctx, hipEventRecord(ctx->ev_upload, ctx->stream));
    ctx->upload_in_flight = true;
  } else {
    VG_HIP(ctx, hipMemcpyAsync(aos_pts, points, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VG_HIP(ctx, hipMemcpyAsync(aos_cov, covs, n * 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VG_HIP(ctx, launch_pack_scan(ctx->stream, aos_pts, aos_cov, (uint32_t)n, ctx->d_scan, ctx->stride,
                                 ctx->d_ins_counters + 2, ctx->scan_seq));
  }
  ctx->upload_bytes += bytes;
  ctx->upload_seconds += now_seconds() - t0;  // host side: the staging copy (or the copy calls) + the enqueue of the pack kernel
  return VGICP_OK;
}
}  // namespace

int vgicp_scan_upload(vgicp_ctx* ctx, size_t n, const double* points, const double* covs) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::scan_upload(ctx, n, points, covs);
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  const double t0 = now_seconds();
  int rc = scan_upload_enqueue(ctx, n, points, covs);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)t0;
  ctx->scan_ready = true;
  return VGICP_OK;
}

int vgicp_host_register(vgicp_ctx* ctx, const void* buffer, size_t bytes) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {  // page-locked once, for every device (portable)
    if (!buffer || bytes == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL / empty buffer");
    VG_HIP(ctx, hipHostRegister(const_cast<void*>(buffer), bytes, hipHostRegisterPortable));
    return VGICP_OK;
  }
  if (!buffer || bytes == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL / empty buffer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  VG_HIP(ctx, hipHostRegister(const_cast<void*>(buffer), bytes, hipHostRegisterDefault));
  return VGICP_OK;
}

int vgicp_host_unregister(vgicp_ctx* ctx, const void* buffer) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    if (!buffer) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL buffer");
    size_t unused = 0;
    const int rc_sync = vgicp_multi_api::map_size(ctx, &unused, nullptr);  // settles every sub-context: no copy in flight
    if (rc_sync != VGICP_OK) return rc_sync;
    VG_HIP(ctx, hipHostUnregister(const_cast<void*>(buffer)));
    return VGICP_OK;
  }
  if (!buffer) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL buffer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // no copy out of the buffer may still be in flight
  VG_HIP(ctx, hipHostUnregister(const_cast<void*>(buffer)));
  return VGICP_OK;
}

int vgicp_align_resident(vgicp_ctx* ctx, const double guess[16], const vgicp_params* params,
                         double out_pose[16], vgicp_stats* stats) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (!guess || !out_pose) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pose pointer");
  if (ctx->multi) return vgicp_multi_api::align_resident(ctx, guess, params, out_pose, stats);
  VG_HIP(ctx, hipSetDevice(ctx->device));
  return run_align(ctx, guess, params, out_pose, stats);
}

int vgicp_align(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                const double guess[16], const vgicp_params* params, double out_pose[16],
                vgicp_stats* stats) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    if (!guess || !out_pose) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pose pointer");
    return vgicp_multi_api::align(ctx, n, points, covs, guess, params, out_pose, stats);
  }
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  const double t0 = now_seconds();
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  // the upload is only enqueued: the pack kernel and the align's first launch follow it in stream order
  int rc = scan_upload_enqueue(ctx, n, points, covs);
  if (rc != VGICP_OK) return rc;
  ctx->scan_ready = true;
  rc = vgicp_align_resident(ctx, guess, params, out_pose, stats);
  if (stats) stats->seconds = now_seconds() - t0;
  return rc;
}

int vgicp_accumulate(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                     const double pose[16], double JTJ[36], double JTr[6], uint64_t* count) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {  // a hook that works on one device ("local rank only"): the whole scan on sub-context 0
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    vgicp_multi_api::scan_replaced(ctx);
    const int rc = vgicp_accumulate(first, n, points, covs, pose, JTJ, JTr, count);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  if (!pose || !JTJ || !JTr) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL output pointer");
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  int rc = vgicp_scan_upload(ctx, n, points, covs);
  if (rc != VGICP_OK) return rc;
  rc = ensure_log(ctx, 1);
  if (rc != VGICP_OK) return rc;
  AlignState* h0 = &ctx->h_state[0];
  std::memset(h0, 0, sizeof(AlignState));
  pose_to_state(pose, h0->pose);
  h0->cosine_threshold = 2.0;
  h0->max_iteration = 1;
  VG_HIP(ctx, hipMemcpyAsync(ctx->d_state, h0, sizeof(AlignState), hipMemcpyHostToDevice, ctx->stream));
  // local rank only (never the communicator): one body launch, then the closing prologue
  const IterArgs base = base_args(ctx);
  const uint32_t grid = iterate_grid(ctx);
  rc = enqueue_launch(ctx, base, 0, grid, false, false);
  if (rc != VGICP_OK) return rc;
  rc = enqueue_launch(ctx, base, 1, grid, true, false);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_log, ctx->d_log, kSlots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double* row = ctx->h_log;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c <= r; ++c) {
      JTJ[r + 6 * c] = row[tri6(r, c)];
      JTJ[c + 6 * r] = row[tri6(r, c)];
    }
  for (int k = 0; k < 6; ++k) JTr[k] = row[21 + k];
  if (count) *count = (uint64_t)row[kCountSlot];
  return VGICP_OK;
}

int vgicp_solve_step(vgicp_ctx* ctx, const double JTJ[36], const double JTr[6], double cosine_threshold,
                     double translation_sq_threshold, uint32_t flags, double se3[6], double step[16],
                     int32_t* used_pivoted, int32_t* converged) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    const int rc = vgicp_solve_step(first, JTJ, JTr, cosine_threshold, translation_sq_threshold, flags, se3, step, used_pivoted, converged);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  if (!JTJ || !JTr || !se3 || !step) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage(ctx, 64 * sizeof(double));
  if (rc != VGICP_OK) return rc;
  double packed[32] = {0.0};
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c <= r; ++c) packed[tri6(r, c)] = JTJ[r + 6 * c];  // the lower triangle, as Eigen's LDLT reads it
  for (int k = 0; k < 6; ++k) packed[21 + k] = JTr[k];
  double* d_in = static_cast<double*>(ctx->d_stage);
  double* d_out = d_in + 32;
  double out[20];
  VG_HIP(ctx, hipMemcpyAsync(d_in, packed, sizeof packed, hipMemcpyHostToDevice, ctx->stream));
  VG_HIP(ctx, launch_solve_step(ctx->stream, d_in, cosine_threshold, translation_sq_threshold,
                                (flags & VGICP_SOLVE_FORCE_PIVOTED) ? 1 : 0, d_out));
  VG_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 6; ++k) se3[k] = out[k];
  Pose T;
  for (int k = 0; k < 9; ++k) T.R[k] = out[6 + k];
  for (int k = 0; k < 3; ++k) T.t[k] = out[15 + k];
  pose_to_mat4(T, step);
  if (used_pivoted) *used_pivoted = out[18] != 0.0;
  if (converged) *converged = out[19] != 0.0;
  return VGICP_OK;
}

int vgicp_match(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                double* src_points, double* src_covs, double* map_points, double* map_covs,
                uint64_t* src_index, size_t* matched) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {  // the map is replicated: any replica answers
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    const int rc = vgicp_match(first, n, points, covs, src_points, src_covs, map_points, map_covs, src_index, matched);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  if (!matched) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "matched is NULL");
  *matched = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (n == 0) return VGICP_OK;
  if (!points || !covs || !src_points || !src_covs || !map_points || !map_covs)
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL array pointer");
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nb = match_blocks((uint32_t)n);
  // staging layout: in_pts | in_cov | out src_pts | src_cov | map_pts | map_cov | index | counts
  const size_t pb = n * 3 * sizeof(double), cb = n * 9 * sizeof(double), ib = n * sizeof(uint64_t);
  const size_t nbb = ((size_t)nb * sizeof(uint32_t) + 255) & ~size_t(255);
  const size_t total = 3 * pb + 3 * cb + ib + nbb + 256;
  int rc = ensure_stage(ctx, total);
  if (rc != VGICP_OK) return rc;
  char* b = static_cast<char*>(ctx->d_stage);
  double* in_pts = reinterpret_cast<double*>(b);
  double* in_cov = reinterpret_cast<double*>(b + pb);
  double* o_sp = reinterpret_cast<double*>(b + pb + cb);
  double* o_sc = reinterpret_cast<double*>(b + 2 * pb + cb);
  double* o_mp = reinterpret_cast<double*>(b + 2 * pb + 2 * cb);
  double* o_mc = reinterpret_cast<double*>(b + 3 * pb + 2 * cb);
  uint64_t* o_ix = reinterpret_cast<uint64_t*>(b + 3 * pb + 3 * cb);
  uint32_t* counts = reinterpret_cast<uint32_t*>(b + 3 * pb + 3 * cb + ib);
  uint32_t* d_total = reinterpret_cast<uint32_t*>(b + 3 * pb + 3 * cb + ib + nbb);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, in_pts, points, pb));
  VG_RC(user_h2d(ctx, in_cov, covs, cb));
  VG_HIP(ctx, launch_match(ctx->stream, in_pts, in_cov, (uint32_t)n, ctx->table,
                           (uint32_t)(ctx->slots - 1), ctx->voxel_size, counts, d_total, o_sp, o_sc,
                           o_mp, o_mc, o_ix));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t m = ctx->h_counters[0];
  if (m > 0) {
    arena_reset(ctx);   // the inputs have been consumed (synchronised above)
    VG_RC(user_d2h(ctx, src_points, o_sp, m * 3 * sizeof(double)));
    VG_RC(user_d2h(ctx, src_covs, o_sc, m * 9 * sizeof(double)));
    VG_RC(user_d2h(ctx, map_points, o_mp, m * 3 * sizeof(double)));
    VG_RC(user_d2h(ctx, map_covs, o_mc, m * 9 * sizeof(double)));
    if (src_index) VG_RC(user_d2h(ctx, src_index, o_ix, m * sizeof(uint64_t)));
    VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    user_copies_finish(ctx);
  }
  *matched = m;
  return VGICP_OK;
}

int vgicp_voxel_index(vgicp_ctx* ctx, size_t n, const double* points, int32_t* keys) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    const int rc = vgicp_voxel_index(first, n, points, keys);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (n == 0) return VGICP_OK;
  if (!points || !keys) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL array pointer");
  if (!(ctx->voxel_size > 0.0)) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel size: call vgicp_map_reset first");
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pb = n * 3 * sizeof(double), kb = n * 3 * sizeof(int32_t);
  int rc = ensure_stage(ctx, pb + kb);
  if (rc != VGICP_OK) return rc;
  char* b = static_cast<char*>(ctx->d_stage);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, b, points, pb));
  VG_HIP(ctx, launch_voxel_index(ctx->stream, reinterpret_cast<const double*>(b), (uint32_t)n,
                                 ctx->voxel_size, reinterpret_cast<int32_t*>(b + pb)));
  VG_RC(user_d2h(ctx, keys, b + pb, kb));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  user_copies_finish(ctx);
  return VGICP_OK;
}

namespace {
int ensure_cells(vgicp_ctx* ctx, size_t need) {
  if (need <= ctx->cells_bytes) return VGICP_OK;
  if (ctx->d_cells) VG_HIP(ctx, hipFree(ctx->d_cells));
  ctx->d_cells = nullptr;
  ctx->cells_bytes = 0;
  VG_HIP(ctx, hipMalloc(&ctx->d_cells, need));
  ctx->cells_bytes = need;
  return VGICP_OK;
}

// The scan preparation on points that are already on the device, ENQUEUED as one sequence without a host round
// trip: tables and grids are sized from n, the kept count stays on the device (counter word 0) and a copy of the
// counter block travels to pinned host memory behind the last kernel. resolve_prepare() reads it after a
// synchronisation of the stream.
struct DeskewOnDevice {
  const double* point_time = nullptr;
  const double* state_time = nullptr;
  const double* poses = nullptr;
  uint32_t states = 0;
  bool ordered = false;
  uint32_t max_hits = 0;   // deskew_table: the largest hit count of any point (ordered queues)
  uint32_t* ends = nullptr;
};
// The raw points of a preparation that are still to be copied into page-locked staging memory (scan_prepare_enqueue):
// enqueue_prepare launches the kernels that read them, copies (this thread and the crew's helpers), and launches the rest.
struct StagedPoints {
  const double* points = nullptr;   // the caller's, n x 3; nullptr: the staging memory holds them already (vgicp_sweep_stage)
  char* stage = nullptr;            // page-locked, n x 24 bytes (+ padding)
  uint32_t* flags = nullptr;        // one 64-byte line per unit
  hipEvent_t done = nullptr;        // recorded behind the last kernel that reads the staging memory
  uint32_t step = 0, off[3] = {0, 0, 0};   // staged ahead as sensor records (vgicp_sweep_stage_cloud2): PrepareArgs::src_step
  uint32_t job = 0, seq = 0;        // the copy crew's job (posted by scan_prepare_enqueue: a helper is copying already)
  bool helpers = false;
  double t_post = 0.0;
  CopyCrew* open_with = nullptr;    // the job is open: whoever leaves early has to finish it (the caller's buffer is read)
  ~StagedPoints() {
    if (open_with) { open_with->work(job); (void)open_with->finish(); }
  }
};
// The copy of a sweep's points into `stage`, opened to the crew: the helpers (if any are awake or worth waking) start at
// once, the caller joins through crew->work(job) when it has launched the kernels that read the staging memory.
void post_sweep_copy(vgicp_ctx* ctx, size_t n, StagedPoints* sp, const double* times = nullptr, double* times_stage = nullptr) {
  if (++ctx->scan_seq == 0) ++ctx->scan_seq;
  sp->seq = ctx->scan_seq;
  const uint32_t unit = pack_arena_unit();
  sp->helpers = ctx->upload_threads > 1 && n * 3 * sizeof(double) >= (1u << 20);
  if (!ctx->crew) ctx->crew = new CopyCrew;
  CopyCrew* crew = ctx->crew;
  if (sp->helpers && crew->th.empty()) crew->start(ctx->upload_threads - 1);
  crew->pts = reinterpret_cast<const char*>(sp->points);
  crew->cov = reinterpret_cast<const char*>(times);           // a unit's capture times travel with its points (or nullptr)
  crew->apts = sp->stage;
  crew->acov = reinterpret_cast<char*>(times_stage);
  crew->flags = sp->flags;
  crew->n = (uint32_t)n;
  crew->unit = unit;
  crew->units = (uint32_t)((n + unit - 1) / unit);
  crew->seq = sp->seq;
  crew->size_a = 24;
  crew->size_b = times ? 8 : 0;
  crew->copy = stage_copy;
  crew->copy_b_form = nullptr;
  sp->t_post = now_seconds();
  sp->job = crew->post(sp->helpers);
  sp->open_with = crew;
}
int enqueue_prepare(vgicp_ctx* ctx, double* d_pts, size_t n, double voxel_size, int knn, const double* extrinsic16,
                    const DeskewOnDevice& dk, void* scratch, double* d_out_pts, double* d_out_covs,
                    unsigned long long* d_out_idx, double* soa, uint64_t soa_stride, StagedPoints* staged = nullptr) {
  const uint64_t entries = preprocess_cell_entries_for((uint32_t)n);
  int rc = ensure_cells(ctx, preprocess_cell_bytes(entries));
  if (rc != VGICP_OK) return rc;
  const int debug = ctx->dev.debug_prep;
  if (debug) VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 72 * sizeof(uint32_t), ctx->stream));
  if (++ctx->prep_epoch == 0) ++ctx->prep_epoch;
  PrepareArgs a;
  std::memset(&a, 0, sizeof a);
  a.pts = d_pts;
  a.n = (uint32_t)n;
  a.voxel_size = voxel_size;
  a.knn = knn;
  a.extrinsic16 = extrinsic16;
  a.point_time = dk.point_time;
  a.state_time = dk.state_time;
  a.poses = dk.poses;
  a.states = dk.states;
  a.ordered_states = dk.ordered;
  a.max_hits_known = dk.ordered;
  a.max_hits = dk.max_hits;
  a.ends = dk.ends;
  a.scratch = scratch;
  a.cell_table = ctx->d_cells;
  a.table_entries = entries;
  a.out_pts = d_out_pts;
  a.out_covs = d_out_covs;
  a.out_idx = d_out_idx;
  a.soa = soa;
  a.soa_stride = soa_stride;
  a.counters = ctx->d_counters;
  a.host_kept = ctx->h_fetch_hdr_dev;
  a.tiles = ctx->d_tiles;
  a.epoch = ctx->prep_epoch;
  a.debug = debug;
  a.ev_after_prologue = ctx->stage_events ? ctx->ev_stage[6] : nullptr;
  if (ctx->stage_events) ctx->ev_stage_set[6] = true;
  if (!staged) {
    VG_HIP(ctx, launch_prepare(ctx->stream, a));
  } else if (!staged->points) {
    // staged ahead of time: the prologue reads the page-locked copy where it lies, nothing to wait for
    a.src_points = staged->stage;
    a.src_flags = nullptr;
    a.src_step = staged->step;
    for (int k = 0; k < 3; ++k) a.src_off[k] = staged->off[k];
    VG_HIP(ctx, launch_prepare_head(ctx->stream, a));
    if (staged->done) VG_HIP(ctx, hipEventRecord(staged->done, ctx->stream));
    VG_HIP(ctx, launch_prepare_tail(ctx->stream, a));
  } else {
    // the sweep's points go up without a copy command: the copy threads fill the staging memory unit by unit, the
    // prologue (launched FIRST) reads the units over PCIe as they are published (see scan_upload_enqueue)
    const uint32_t spin_limit = ctx->dev.pack_spin_limit ? ctx->dev.pack_spin_limit : kPackSpinLimit;
    const long debug_delay_us = ctx->dev.debug_upload_delay_us;
    CopyCrew* crew = ctx->crew;
    const bool want_helpers = staged->helpers;
    const uint32_t job = staged->job;
    const double t_post = staged->t_post;
    a.src_points = staged->stage;
    a.src_flags = staged->flags;
    a.src_seq = staged->seq;
    a.src_unit = pack_arena_unit();
    a.src_spin = spin_limit;
    const hipError_t e_head = launch_prepare_head(ctx->stream, a);
    if (debug_delay_us > 0 && !want_helpers) std::this_thread::sleep_for(std::chrono::microseconds(debug_delay_us));
    crew->work(job);
    const bool crew_done = crew->finish();   // always: the caller's buffer is free again on return
    staged->open_with = nullptr;
    if (e_head != hipSuccess) return fail_hip(ctx, e_head, "launch_prepare_head");
    if (!crew_done) return crew_gave_up(ctx);
    if (now_seconds() - t_post > kCrewSlowSeconds) {
      // the copy threads were held up so long that a workgroup of the prologue may have stopped waiting (and said so in
      // the counter block under this epoch): everything is staged now — the head once more, nothing to wait for
      ++ctx->upload_slow;
      if (++ctx->prep_epoch == 0) ++ctx->prep_epoch;
      a.epoch = ctx->prep_epoch;
      a.src_flags = nullptr;
      VG_HIP(ctx, launch_prepare_head(ctx->stream, a));
    }
    if (staged->done) VG_HIP(ctx, hipEventRecord(staged->done, ctx->stream));
    VG_HIP(ctx, launch_prepare_tail(ctx->stream, a));
  }
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_prep, ctx->d_counters, (kCounterWords + 4) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->insert_pending && !ctx->ins_copy_enqueued) {   // the deferred insertion's totals travel with this copy
    ctx->ins_copy_enqueued = true;
    ctx->ins_from_prep = true;
  }
  return VGICP_OK;
}

// After the stream has been synchronised: what the preparation found. *kept is set even on refusal.
int resolve_prepare(vgicp_ctx* ctx, uint32_t* kept) {
  const uint32_t* h = ctx->h_prep;
  *kept = 0;
  if (h[kScanTimeout] == ctx->prep_epoch)
    return fail(ctx, VGICP_ERR_HIP, "a device-wide scan of the scan preparation gave up waiting for a tile");
  if (h[kBeyondGrid] == ctx->prep_epoch)
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a point lies beyond the search grid (more than 2^17 voxel sizes from the origin)");
  *kept = h[0];
  ctx->prep_indefinite = h[kIndefiniteCounter];
  ctx->prep_deskewed = (int64_t)h[kDeskewedCounter];
  if (ctx->dev.debug_prep) {
    const int debug = ctx->dev.debug_prep;
    if (debug >= 2) {
      std::fprintf(stderr, "[vgicp prep] queries by cells taken (buckets of 8):");
      for (int i = 0; i < 32; ++i) std::fprintf(stderr, " %u", h[8 + i]);
      std::fprintf(stderr, "\n[vgicp prep] queries by time in the search (buckets of 8 us):");
      for (int i = 0; i < 32; ++i) std::fprintf(stderr, " %u", h[40 + i]);
      std::fprintf(stderr, "\n");
    }
    const uint32_t m = h[0];
    if (debug) std::fprintf(stderr, "[vgicp prep] kept %u cells %u queries that spilled %u | point batches total %u (%.1f/query) max %u | cells taken total %u (%.1f/query) max %u | queries starting above the voxel level: %u\n",
                            m, h[1], h[2], h[3], h[3] / (double)(m ? m : 1), h[4], h[5], h[5] / (double)(m ? m : 1), h[6], h[7]);
  }
  return VGICP_OK;
}

// A deferred insertion whose totals no copy has picked up yet (no preparation followed it): a copy of its own, now.
int fetch_insert_totals(vgicp_ctx* ctx) {
  if (!ctx->insert_pending || ctx->ins_copy_enqueued) return VGICP_OK;
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_ins_counters, ctx->d_ins_counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  ctx->ins_copy_enqueued = true;
  ctx->ins_from_prep = false;
  return VGICP_OK;
}

// The deferred map insertion's counts (running totals), once the stream has been synchronised.
int settle_insert(vgicp_ctx* ctx) {
  if (!ctx->insert_pending) return VGICP_OK;
  ctx->insert_pending = false;
  ctx->insert_pending_upper = 0;
  const uint32_t* totals = ctx->ins_from_prep ? ctx->h_prep + kCounterWords : ctx->h_ins_counters;
  const uint32_t created = totals[0] - ctx->ins_seen[0];
  const uint32_t failed = totals[1] - ctx->ins_seen[1];
  ctx->ins_seen[0] = totals[0];
  ctx->ins_seen[1] = totals[1];
  ctx->voxels += created;
  if (failed) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table probe sequence exhausted (deferred map insertion)");
  return VGICP_OK;
}

// A pending scan's size and verdict, once the stream has been synchronised.
int settle_scan(vgicp_ctx* ctx) {
  if (!ctx->scan_pending) return VGICP_OK;
  ctx->scan_pending = false;
  uint32_t m = 0;
  const int rc = resolve_prepare(ctx, &m);
  if (rc != VGICP_OK) {
    ctx->scan_ready = false;
    ctx->n = 0;
    return rc;
  }
  ctx->n = m;
  return VGICP_OK;
}

// Everything deferred is brought up to date (one synchronisation if anything is pending): entry points that read
// or change what a pending operation still owns call this first.
int settle(vgicp_ctx* ctx) {
  if (!ctx->scan_pending && !ctx->insert_pending) return VGICP_OK;
  VG_HIP(ctx, hipSetDevice(ctx->device));
  { const int rc_copy = fetch_insert_totals(ctx); if (rc_copy != VGICP_OK) return rc_copy; }
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int rc_scan = settle_scan(ctx);
  const int rc_ins = settle_insert(ctx);
  return rc_scan != VGICP_OK ? rc_scan : rc_ins;
}

// VGICP_OPTION_REFERENCE_ORDER — the kept points in the sequence the reference emits them (src/CloudPreprocessor.cpp:85-99):
// the iteration order of a std::unordered_map<Eigen::Vector3i, int, open3d::utility::hash_eigen<Eigen::Vector3i>> that was
// filled in scan order.  Only the first point of a voxel creates a node, so the container sees the kept points' voxel
// keys in ascending input index — the order the device emits — and its node order is libstdc++'s for that hash and that
// insertion sequence (the reference's platform, Ubuntu 22.04 / GCC 11, ships the libstdc++ this module is built
// against; the C++ standard leaves the order open).  The container itself is what is asked here: the same type with the
// same hash, filled the same way, on the host (27 000 insertions: ~1.5 ms — a parity mode, not the fast path).
struct VoxelKeyHostHash {   // open3d::utility::hash_eigen<Eigen::Vector3i>: boost-style combine of std::hash<int>
  size_t operator()(const std::array<int32_t, 3>& k) const {
    size_t seed = 0;
    for (int n = 0; n < 3; ++n) seed ^= std::hash<int>()(k[n]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};
// perm[o] = the ascending-order slot of the point the reference emits o-th.  pts: m x 3, the kept points in ascending input index.
void reference_order_of(const double* pts, size_t m, double voxel_size, std::vector<uint32_t>* perm) {
  std::unordered_map<std::array<int32_t, 3>, uint32_t, VoxelKeyHostHash> grid;
  for (size_t i = 0; i < m; ++i) {
    std::array<int32_t, 3> key;   // the preprocessor's getVoxelIndex: floor(p / voxel) as int (src/CloudPreprocessor.cpp:129-133)
    for (int a = 0; a < 3; ++a) key[a] = static_cast<int32_t>(std::floor(pts[3 * i + a] / voxel_size));
    if (grid.find(key) == grid.end()) grid[key] = (uint32_t)i;   // as the reference writes it (:88-91)
  }
  perm->clear();
  perm->reserve(grid.size());
  for (const auto& kv : grid) perm->push_back(kv.second);
}
// The resident scan (just prepared, stream synchronised, ctx->n kept points) is put into the reference's order in place.
int reorder_resident_scan(vgicp_ctx* ctx, double voxel_size) {
  const size_t m = ctx->n;
  if (m < 2) return VGICP_OK;
  std::vector<double> pts(3 * m);
  VG_HIP(ctx, hipMemcpyAsync(pts.data(), ctx->d_scan_aos, m * 24, hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint32_t> perm;
  reference_order_of(pts.data(), m, voxel_size, &perm);
  if (perm.size() != m) return fail(ctx, VGICP_ERR_HIP, "reference order: the kept points do not lie in distinct voxels");
  const size_t pb = (m * 24 + 255) & ~size_t(255), cb = (m * 72 + 255) & ~size_t(255), ib = (m * 4 + 255) & ~size_t(255);
  const int rc = ensure_stage(ctx, pb + cb + ib);
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  double* t_pts = reinterpret_cast<double*>(base);
  double* t_cov = reinterpret_cast<double*>(base + pb);
  uint32_t* d_perm = reinterpret_cast<uint32_t*>(base + pb + cb);
  double* aos_cov = ctx->d_scan_aos + 3 * ctx->scan_capacity;
  VG_HIP(ctx, hipMemcpyAsync(d_perm, perm.data(), m * 4, hipMemcpyHostToDevice, ctx->stream));
  VG_HIP(ctx, launch_gather_scan(ctx->stream, d_perm, (uint32_t)m, ctx->d_scan_aos, aos_cov, nullptr, t_pts, t_cov, nullptr));
  VG_HIP(ctx, hipMemcpyAsync(ctx->d_scan_aos, t_pts, m * 24, hipMemcpyDeviceToDevice, ctx->stream));
  VG_HIP(ctx, hipMemcpyAsync(aos_cov, t_cov, m * 72, hipMemcpyDeviceToDevice, ctx->stream));
  // the planes the registration reads, from the reordered AoS copy (the symmetry word is not consulted for a prepared scan)
  VG_HIP(ctx, launch_pack_scan(ctx->stream, ctx->d_scan_aos, aos_cov, (uint32_t)m, ctx->d_scan, ctx->stride, ctx->d_ins_counters + 3, 0));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `perm` and `pts` die with this frame
  return VGICP_OK;
}

int check_preprocess_args(vgicp_ctx* ctx, size_t n, double voxel_size, int knn) {
  if (!(voxel_size > 0.0)) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "voxel_size must be positive");
  if (knn < 1 || knn > preprocess_max_knn())
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "knn must be in [1, " + std::to_string(preprocess_max_knn()) + "]");
  if (n > (size_t)kMaxScanTiles * 2048u) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large (more than 8 M points)");
  return VGICP_OK;
}
}  // namespace

int vgicp_preprocess(vgicp_ctx* ctx, size_t n, const double* points, double voxel_size, int knn,
                     size_t capacity, double* out_points, double* out_covs, uint64_t* out_index,
                     size_t* kept) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {  // scan preparation with the result returned to the host: one device's work
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    const int rc = vgicp_preprocess(first, n, points, voxel_size, knn, capacity, out_points, out_covs, out_index, kept);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  if (!kept) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "kept is NULL");
  *kept = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  int rc = check_preprocess_args(ctx, n, voxel_size, knn);
  if (rc != VGICP_OK) return rc;
  if (n == 0) return VGICP_OK;
  if (!points) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL scan pointer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  // stage: [points 3n][out points 3n][out covs 9n][out index n][scratch]
  const size_t pb = (n * 3 * sizeof(double) + 255) & ~size_t(255);
  const size_t cb = (n * 9 * sizeof(double) + 255) & ~size_t(255);
  const size_t ib = (n * sizeof(uint64_t) + 255) & ~size_t(255);
  const size_t sb = preprocess_scratch_bytes((uint32_t)n);
  rc = ensure_stage(ctx, pb + pb + cb + ib + sb);
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  double* d_out_pts = reinterpret_cast<double*>(base + pb);
  double* d_out_covs = reinterpret_cast<double*>(base + 2 * pb);
  unsigned long long* d_out_idx = reinterpret_cast<unsigned long long*>(base + 2 * pb + cb);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, base, points, n * 3 * sizeof(double)));
  rc = enqueue_prepare(ctx, reinterpret_cast<double*>(base), n, voxel_size, knn, nullptr, DeskewOnDevice(),
                       base + 2 * pb + cb + ib, d_out_pts, d_out_covs, d_out_idx, nullptr, 0);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  uint32_t m = 0;
  rc = resolve_prepare(ctx, &m);
  *kept = m;
  if (rc != VGICP_OK) return rc;
  if (m > (out_points && out_covs ? capacity : 0))
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "output capacity smaller than the number of occupied voxels");
  arena_reset(ctx);   // the input has been consumed (synchronised above)
  VG_RC(user_d2h(ctx, out_points, d_out_pts, (size_t)m * 3 * sizeof(double)));
  VG_RC(user_d2h(ctx, out_covs, d_out_covs, (size_t)m * 9 * sizeof(double)));
  if (out_index) VG_RC(user_d2h(ctx, out_index, d_out_idx, (size_t)m * sizeof(uint64_t)));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  user_copies_finish(ctx);
  if (ctx->reference_order && m > 1) {   // VGICP_OPTION_REFERENCE_ORDER: the host arrays, through copies
    std::vector<uint32_t> perm;
    reference_order_of(out_points, m, voxel_size, &perm);
    if (perm.size() != m) return fail(ctx, VGICP_ERR_HIP, "reference order: the kept points do not lie in distinct voxels");
    std::vector<double> p(out_points, out_points + 3 * (size_t)m), c(out_covs, out_covs + 9 * (size_t)m);
    std::vector<uint64_t> ix;
    if (out_index) ix.assign(out_index, out_index + m);
    for (size_t o = 0; o < m; ++o) {
      std::memcpy(out_points + 3 * o, p.data() + 3 * (size_t)perm[o], 24);
      std::memcpy(out_covs + 9 * o, c.data() + 9 * (size_t)perm[o], 72);
      if (out_index) out_index[o] = ix[perm[o]];
    }
  }
  return VGICP_OK;
}

namespace {
// Host side of the deskew: one pose per IMU state, (pose at the end of the sweep)^-1 * state pose. Same
// formulas and evaluation order as Eigen's Quaterniond::toRotationMatrix / slerp and Isometry3d products.
struct Pose12 { double R[9]; double t[3]; };  // R column-major
void quat_matrix(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[3] = txy - twz; R[6] = txz + twy;
  R[1] = txy + twz; R[4] = 1.0 - (txx + tzz); R[7] = tyz - twx;
  R[2] = txz - twy; R[5] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
void rotate(const double R[9], const double v[3], double out[3]) {
  for (int r = 0; r < 3; ++r) out[r] = R[r] * v[0] + R[r + 3] * v[1] + R[r + 6] * v[2];
}
Pose12 pose_product(const Pose12& A, const Pose12& B) {
  Pose12 C;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      double s = A.R[r] * B.R[3 * c];
      s += A.R[r + 3] * B.R[3 * c + 1];
      s += A.R[r + 6] * B.R[3 * c + 2];
      C.R[r + 3 * c] = s;
    }
  double rt[3];
  rotate(A.R, B.t, rt);
  for (int k = 0; k < 3; ++k) C.t[k] = rt[k] + A.t[k];
  return C;
}
Pose12 pose_inverted(const Pose12& A) {
  Pose12 C;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) C.R[r + 3 * c] = A.R[c + 3 * r];
  double rt[3];
  rotate(C.R, A.t, rt);
  for (int k = 0; k < 3; ++k) C.t[k] = -rt[k];
  return C;
}
void quat_slerp(const double a[4], const double b[4], double t, double out[4]) {
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = std::fabs(d);
  double scale0, scale1;
  if (absD >= one) {
    scale0 = 1.0 - t;
    scale1 = t;
  } else {
    const double theta = std::acos(absD);
    const double sinTheta = std::sin(theta);
    scale0 = std::sin((1.0 - t) * theta) / sinTheta;
    scale1 = std::sin(t * theta) / sinTheta;
  }
  if (d < 0.0) scale1 = -scale1;
  for (int k = 0; k < 4; ++k) out[k] = scale0 * a[k] + scale1 * b[k];
}
}  // namespace

namespace {
// [state times | 12 doubles per state] for the states that can own points; false where the reference would
// step off its deque (no state at or before the last point's time, or none after it).
bool deskew_table(size_t n, const double* point_time, size_t num_states, const double* states,
                  std::vector<double>& host, size_t& used, bool& ordered, uint32_t* max_hits = nullptr) {
  const double t_end = point_time[n - 1];
  long before = (long)num_states - 1;
  while (before >= 0 && states[8 * before] > t_end) --before;
  if (before < 0 || (size_t)before + 1 >= num_states) return false;
  const double* s1 = states + 8 * before;
  const double* s2 = s1 + 8;
  const double factor = (t_end - s1[0]) / (s2[0] - s1[0] + 1e-6);
  double q[4];
  quat_slerp(s1 + 4, s2 + 4, factor, q);
  Pose12 end_pose;
  quat_matrix(q, end_pose.R);
  for (int k = 0; k < 3; ++k) end_pose.t[k] = s1[1 + k] + factor * (s2[1 + k] - s1[1 + k]);
  const Pose12 end_inv = pose_inverted(end_pose);
  // The reference walks ALL states (its deque is never trimmed, so it grows by 400 entries per second). A
  // state whose timestamp is not above the smallest capture time can never take a point (the walk's
  // test "pointTime < timestamp" fails for whichever point it looks at), so leading ones are skipped here.
  double earliest, latest;
  bool any_nan;
  time_range(point_time, n, &earliest, &latest, &any_nan);
  size_t first = 0;
  while (first + 1 < (size_t)before + 2 && states[8 * first] <= earliest) ++first;
  states += 8 * first;
  used = (size_t)before + 2 - first;  // up to the first state after the end of the sweep
  host.assign(used * 13, 0.0);
  for (size_t s = 0; s < used; ++s) {
    Pose12 T;
    quat_matrix(states + 8 * s + 4, T.R);
    for (int k = 0; k < 3; ++k) T.t[k] = states[8 * s + 1 + k];
    T = pose_product(end_inv, T);
    host[s] = states[8 * s];
    std::memcpy(&host[used + 12 * s], T.R, 9 * sizeof(double));
    std::memcpy(&host[used + 12 * s + 9], T.t, 3 * sizeof(double));
  }
  ordered = true;  // finite, non-decreasing state times: the device finds the segment bounds in parallel
  for (size_t s = 0; s < used; ++s)
    if (!(host[s] - host[s] == 0.0) || (s && host[s] < host[s - 1])) ordered = false;
  if (max_hits) {
    // the largest number of states that any point is a "hit" for (!(t < timestamp), nested for ordered timestamps):
    // the count of the latest capture time -- every state when a time is NaN (a hit for all of them).  The states from
    // this number on own no point (the walk finds no hit for them and keeps its bound): what the prologue needs to know
    // about the WHOLE sweep, so that nothing on the device has to wait for all of it.
    size_t hits = 0;
    if (any_nan) hits = used;
    else while (hits < used && !(latest < host[hits])) ++hits;
    *max_hits = (uint32_t)hits;
  }
  return true;
}
}  // namespace

int vgicp_deskew(vgicp_ctx* ctx, size_t n, double* points, const double* point_time, size_t num_states,
                 const double* states, int64_t* transformed) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    vgicp_ctx* first = vgicp_multi_api::first(ctx);
    const int rc = vgicp_deskew(first, n, points, point_time, num_states, states, transformed);
    if (rc != VGICP_OK) ctx->err = first->err;
    return rc;
  }
  if (!transformed) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "transformed is NULL");
  *transformed = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (n == 0 || num_states == 0) return VGICP_OK;
  if (!points || !point_time || !states) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (n > 0x7FFFFFFFull || num_states > 0x7FFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan or state queue too large");
  std::vector<double> host;
  size_t used = 0;
  bool ordered = false;
  if (!deskew_table(n, point_time, num_states, states, host, used, ordered)) {
    *transformed = -1;
    return VGICP_OK;
  }
  VG_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pb = (n * 3 * sizeof(double) + 255) & ~size_t(255);
  const size_t tb = (n * sizeof(double) + 255) & ~size_t(255);
  const size_t sb = (used * 13 * sizeof(double) + 255) & ~size_t(255);
  const size_t eb = (deskew_scratch_words((uint32_t)used) * sizeof(uint32_t) + 255) & ~size_t(255);
  int rc = ensure_stage(ctx, pb + tb + sb + eb);
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  double* d_pts = reinterpret_cast<double*>(base);
  double* d_time = reinterpret_cast<double*>(base + pb);
  double* d_states = reinterpret_cast<double*>(base + pb + tb);
  uint32_t* d_ends = reinterpret_cast<uint32_t*>(base + pb + tb + sb);
  arena_reset(ctx);
  VG_RC(user_h2d(ctx, d_pts, points, n * 3 * sizeof(double)));
  VG_RC(user_h2d(ctx, d_time, point_time, n * sizeof(double)));
  VG_HIP(ctx, hipMemcpyAsync(d_states, host.data(), used * 13 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  VG_HIP(ctx, launch_deskew(ctx->stream, d_pts, (uint32_t)n, d_time, d_states, (uint32_t)used, d_states + used, d_ends, ordered));
  VG_RC(user_d2h(ctx, points, d_pts, n * 3 * sizeof(double)));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, d_ends + (used - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  user_copies_finish(ctx);
  *transformed = (int64_t)ctx->h_counters[0];
  return VGICP_OK;
}

namespace {
// CloudPreprocessor::process enqueued on the context's stream with the prepared scan left resident: upload of the
// raw sweep, then launch_prepare writing the AoS scan AND the SoA planes the registration reads. Nothing is waited
// for: the scan is `pending` (its size is on the device, ctx->n_upper bounds it).
// ahead: the sweep was staged by vgicp_sweep_stage (points / point_time then point INTO that page-locked slot and nothing
// is copied here; its `done` event is recorded behind the kernels that read it).
int scan_prepare_enqueue(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time, size_t num_states,
                         const double* states, const double extrinsic[16], double voxel_size, int knn,
                         vgicp_ctx::AheadSlot* ahead = nullptr) {
  int rc = check_preprocess_args(ctx, n, voxel_size, knn);
  if (rc != VGICP_OK) return rc;
  if ((ctx->comm || ctx->peers_connected) && !ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "the prepared scan is whole: not available on a communicator (shards)");
  if (n > 0 && !points) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL scan pointer");
  const bool with_deskew = n > 0 && num_states > 0;
  if (with_deskew && (!point_time || !states)) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (num_states > 0x7FFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "state queue too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  rc = ensure_scan(ctx, n);
  if (rc != VGICP_OK) return rc;
  ++ctx->scan_generation;
  ctx->scan_ready = false;
  ctx->scan_pending = false;
  ctx->n = 0;
  ctx->n_upper = 0;
  ctx->stride = ctx->scan_capacity;
  ctx->prep_with_deskew = with_deskew;
  ctx->prep_voxel = voxel_size;
  ctx->prep_deskewed = 0;
  ctx->prep_indefinite = 0;
  if (n == 0) {
    ctx->scan_ready = true;
    return VGICP_OK;
  }
  std::vector<double> host;
  size_t used = 0;
  bool ordered = false;
  uint32_t max_hits = 0;
  static const bool trace_table = std::getenv("VGICP_TRACE_PREPARE") != nullptr;
  const double tt0 = trace_table ? now_seconds() : 0.0;
  const bool table_ok = !with_deskew || deskew_table(n, point_time, num_states, states, host, used, ordered, &max_hits);
  if (trace_table) std::fprintf(stderr, "[vgicp trace] deskew_table %.3f ms\n", (now_seconds() - tt0) * 1e3);
  if (!table_ok) {
    ctx->prep_deskewed = -1;
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "the IMU states do not bracket the end of the sweep");
  }
  // the fused prologue keeps the segment ends of the states that can own points in LDS (4 bytes each, 64 KB by default)
  if (used > kPrepareMaxStates)
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "more than 16000 IMU states inside one sweep: use vgicp_deskew + vgicp_preprocess");
  // stage: [points 3n][times n][state table][segment ends + first hits][kept index n][scratch]
  const size_t pb = (n * 3 * sizeof(double) + 255) & ~size_t(255);
  const size_t tb = (n * sizeof(double) + 255) & ~size_t(255);
  const size_t sb = (used * 13 * sizeof(double) + 255) & ~size_t(255);
  const size_t eb = (deskew_scratch_words((uint32_t)used) * sizeof(uint32_t) + 255) & ~size_t(255);
  const size_t ib = (n * sizeof(uint64_t) + 255) & ~size_t(255);
  rc = ensure_stage(ctx, pb + tb + sb + eb + ib + preprocess_scratch_bytes((uint32_t)n));
  if (rc != VGICP_OK) return rc;
  char* base = static_cast<char*>(ctx->d_stage);
  double* d_pts = reinterpret_cast<double*>(base);
  double* d_time = reinterpret_cast<double*>(base + pb);
  double* d_states = reinterpret_cast<double*>(base + pb + tb);
  unsigned long long* d_idx = reinterpret_cast<unsigned long long*>(base + pb + tb + sb + eb);
  void* scratch = base + pb + tb + sb + eb + ib;
  if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[0], ctx->stream)); ctx->ev_stage_set[0] = true; }
  static const bool trace = std::getenv("VGICP_TRACE_PREPARE") != nullptr;   // developer aid: where the host time of the enqueue goes
  const double tr0 = trace ? now_seconds() : 0.0;
  // two pinned slots in turn, guarded by one event each (recorded behind the last copy out of the slot)
  const uint32_t slot = ctx->state_table_next++ & 1u;
  if (ctx->ev_state_table[slot]) {
    // the copies out of this slot two preparations ago: long complete, normally (no host wait then)
    if (hipEventQuery(ctx->ev_state_table[slot]) != hipSuccess) VG_HIP(ctx, hipEventSynchronize(ctx->ev_state_table[slot]));
  } else {
    VG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_state_table[slot], hipEventDisableTiming));
  }
  const size_t raw_bytes = n * 3 * sizeof(double) + (with_deskew ? n * sizeof(double) : 0);
  static const size_t stage_limit = std::getenv("VGICP_STAGE_LIMIT") ? (size_t)std::atoll(std::getenv("VGICP_STAGE_LIMIT")) : (16u << 20);
  const bool staged = ahead != nullptr || raw_bytes <= stage_limit;   // larger sweeps go up straight from the caller's memory
  const bool walk = with_deskew && !(ordered && used <= kDeskewMaxStates);   // the serial bounds walk reads the times many times over: on the device
  StagedPoints sp;
  const double* time_src = nullptr;   // where the deskew's first kernel reads the capture times
  if (ahead) {
    sp.points = nullptr;
    sp.stage = const_cast<char*>(reinterpret_cast<const char*>(points));
    sp.done = ahead->done;
    sp.step = ahead->step;
    for (int k = 0; k < 3; ++k) sp.off[k] = ahead->off[k];
    if (with_deskew) {
      time_src = point_time;
      if (walk) VG_HIP(ctx, hipMemcpyAsync(d_time, point_time, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
  } else if (staged) {
    // slot layout: [unit flags][points, padded][capture times]
    const size_t pts_bytes = n * 3 * sizeof(double);
    const size_t flag_bytes = ((n + pack_arena_unit() - 1) / pack_arena_unit() + 1) * 64;
    const size_t pts_room = (pts_bytes + 16 + 255) & ~size_t(255);
    if (ctx->raw_stage_cap[slot] < flag_bytes + pts_room + n * sizeof(double)) {
      if (ctx->h_raw_stage[slot]) VG_HIP(ctx, hipHostFree(ctx->h_raw_stage[slot]));
      ctx->h_raw_stage[slot] = nullptr;
      ctx->raw_stage_cap[slot] = 0;
      const size_t cap = (flag_bytes + pts_room + n * sizeof(double)) * 5 / 4 + 4096;   // a quarter more
      VG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_raw_stage[slot]), cap, 0));
      std::memset(ctx->h_raw_stage[slot], 0, cap);   // flags of "no sweep yet" (a sequence number is never 0)
      ctx->raw_stage_cap[slot] = cap;
    }
    char* stage = ctx->h_raw_stage[slot];
    // a flag only ever means "this unit of THIS sweep": the flag area moves with the sweep's size, so it is wiped
    std::memset(stage, 0, flag_bytes);
    sp.points = points;
    sp.flags = reinterpret_cast<uint32_t*>(stage);
    sp.stage = stage + flag_bytes;
    sp.done = ctx->ev_state_table[slot];
    double* times_stage = reinterpret_cast<double*>(stage + flag_bytes + pts_room);
    // the prologue finds the deskew's segments itself and reads a workgroup's capture times behind the wait for its
    // unit: they are staged unit by unit with the points, by whoever copies the unit
    const bool times_by_unit = with_deskew && !walk && prepare_bounds_fused((uint32_t)n, (uint32_t)used, ordered);
    post_sweep_copy(ctx, n, &sp, times_by_unit ? point_time : nullptr, times_by_unit ? times_stage : nullptr);   // a helper that is awake starts now
    if (times_by_unit) {
      time_src = times_stage;
    } else if (with_deskew) {
      // this thread: the capture times first (a sixth of the bytes): the deskew's bounds need nothing else, and its
      // kernel reads them where they are staged
      stage_copy(times_stage, point_time, n * sizeof(double));
      time_src = times_stage;
      if (walk) VG_HIP(ctx, hipMemcpyAsync(d_time, times_stage, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
  } else {
    VG_HIP(ctx, hipMemcpyAsync(d_pts, points, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (with_deskew) VG_HIP(ctx, hipMemcpyAsync(d_time, point_time, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  const double tr1 = trace ? now_seconds() : 0.0;
  DeskewOnDevice dk;
  if (with_deskew) {
    if (ctx->state_table_cap[slot] < used * 13) {
      if (ctx->h_state_table[slot]) VG_HIP(ctx, hipHostFree(ctx->h_state_table[slot]));
      ctx->h_state_table[slot] = nullptr;
      ctx->state_table_cap[slot] = 0;
      const size_t cap = std::max<size_t>(used * 13 * 2, 13 * 256);
      VG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_state_table[slot]), cap * sizeof(double), 0));
      ctx->state_table_cap[slot] = cap;
    }
    std::memcpy(ctx->h_state_table[slot], host.data(), used * 13 * sizeof(double));
    VG_HIP(ctx, hipMemcpyAsync(d_states, ctx->h_state_table[slot], used * 13 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dk.point_time = (staged && !walk) ? time_src : d_time;
    dk.state_time = d_states;
    dk.poses = d_states + used;
    dk.states = (uint32_t)used;
    dk.ordered = ordered;
    dk.max_hits = max_hits;
    dk.ends = reinterpret_cast<uint32_t*>(base + pb + tb + sb);
  }
  if (!staged) {
    VG_HIP(ctx, hipEventRecord(ctx->ev_state_table[slot], ctx->stream));   // behind the last copy out of this slot's pinned buffers
    // a sweep too large to stage was handed to the runtime in place: its copies have to be over before the call returns,
    // because the caller's buffers are free again on return whatever the size (the drop-in releases the capture times at once)
    VG_HIP(ctx, hipEventSynchronize(ctx->ev_state_table[slot]));
  }
  const double tr2 = trace ? now_seconds() : 0.0;
  // (staged: the slot's event is recorded inside, behind the kernels that read the staging memory)
  rc = enqueue_prepare(ctx, d_pts, n, voxel_size, knn, extrinsic, dk, scratch, ctx->d_scan_aos,
                       ctx->d_scan_aos + 3 * ctx->scan_capacity, d_idx, ctx->d_scan, ctx->stride, staged ? &sp : nullptr);
  if (rc != VGICP_OK) return rc;
  if (ahead) VG_HIP(ctx, hipEventRecord(ctx->ev_state_table[slot], ctx->stream));   // (the state table's pinned slot)
  if (trace)
    std::fprintf(stderr, "[vgicp trace] prepare enqueue: staging + points copy %.3f ms, times + states copies %.3f ms, kernels %.3f ms\n",
                 (tr1 - tr0) * 1e3, (tr2 - tr1) * 1e3, (now_seconds() - tr2) * 1e3);
  if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[1], ctx->stream)); ctx->ev_stage_set[1] = true; }
  ctx->n_upper = (uint32_t)n;
  ctx->scan_sym_known = false;   // covariances made on the device: all twelve planes are read
  ctx->n = (uint32_t)n;          // an upper bound until the pending scan is settled
  ctx->scan_pending = true;
  ctx->scan_ready = true;
  if (ctx->reference_order) {
    // the parity mode: this preparation is waited for, and its result put into the reference's sequence, before anything
    // else sees it (an "async" preparation is synchronous under this option)
    rc = settle(ctx);
    if (rc != VGICP_OK) return rc;
    return reorder_resident_scan(ctx, voxel_size);
  }
  return VGICP_OK;
}
}  // namespace

int vgicp_scan_prepare_async(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time,
                             size_t num_states, const double* states, const double extrinsic[16],
                             double voxel_size, int knn) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::scan_prepare(ctx, n, points, point_time, num_states, states, extrinsic, voxel_size, knn, nullptr, nullptr, true);
  // nothing is settled here: a map insertion still pending from the previous frame has counters of its own and is
  // read at this frame's one synchronisation (the align); a scan that was prepared but never used is simply replaced
  return scan_prepare_enqueue(ctx, n, points, point_time, num_states, states, extrinsic, voxel_size, knn);
}

// A sweep copied into page-locked memory of the context WHEN IT ARRIVES (the lidar callback's thread, reference
// include/ESKF_LIO/Subscriber.hpp:80-103; src/Odometry.cpp:43-48 pops the sweep long before :74 prepares it), so that
// the preparation later starts from bytes the device can read at once.  Only plain CPU copies here, under a mutex
// of its own: the one entry point another thread may call while the context's owner is inside a call.
namespace {
// data: n x 3 doubles (step == 0) or n sensor records of `step` bytes (float32 x y z at off[], a float64 capture time at
// off_time, or none: SIZE_MAX).  times: n doubles for the first form (or nullptr).
int stage_sweep_ahead(vgicp_ctx* ctx, size_t n, const void* data, const double* times, uint32_t step, const uint32_t off[3],
                      size_t off_time, uint64_t* ticket) {
  vgicp_ctx::AheadSlot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
    for (auto& s : ctx->ahead)
      if (s.state == 0) { slot = &s; slot->state = 3; break; }
  }
  for (int k = 0; k < 3 && !slot; ++k) {
    // handed to the device two preparations ago: its readers have long finished.  The event is asked OUTSIDE the mutex
    // (the slot is reserved meanwhile), so the owner thread never waits for a runtime call made by this one.
    vgicp_ctx::AheadSlot* cand = nullptr;
    {
      std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
      if (ctx->ahead[k].state == 2) { cand = &ctx->ahead[k]; cand->state = 3; }
    }
    if (!cand) continue;
    if (!cand->done || hipEventQuery(cand->done) == hipSuccess) { slot = cand; break; }
    std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
    cand->state = 2;
  }
  if (!slot) return fail_stage(ctx, VGICP_ERR_NOT_READY, "three sweeps are staged ahead already: prepare one (or vgicp_sweep_unstage it) first");
  const size_t rec = step ? step : 3 * sizeof(double);
  const size_t pts_room = (n * rec + 16 + 255) & ~size_t(255);
  const bool has_times = step ? off_time != SIZE_MAX : times != nullptr;
  const size_t need = pts_room + n * sizeof(double);
  if (slot->cap < need) {
    if (hipSetDevice(ctx->device) != hipSuccess) { slot->state = 0; return fail_stage(ctx, VGICP_ERR_HIP, "hipSetDevice"); }
    if (slot->mem) (void)hipHostFree(slot->mem);
    slot->mem = nullptr;
    slot->cap = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&slot->mem), need * 5 / 4 + 4096, 0) != hipSuccess) {
      slot->state = 0;
      return fail_stage(ctx, VGICP_ERR_HIP, "hipHostMalloc(sweep staging)");
    }
    slot->cap = need * 5 / 4 + 4096;
  }
  stage_copy(slot->mem, data, n * rec);   // the records as they are: the device picks the floats out and widens them
  double* t_dst = reinterpret_cast<double*>(slot->mem + pts_room);
  if (step && has_times) {
    // the capture times out of the records into an array of their own (the deskew's first kernel reads them contiguously)
    const char* src = static_cast<const char*>(data) + off_time;
    for (size_t i = 0; i < n; ++i) std::memcpy(t_dst + i, src + i * step, sizeof(double));
  } else if (has_times) {
    stage_copy(t_dst, times, n * sizeof(double));
  }
  std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
  slot->n = n;
  slot->has_times = has_times;
  slot->step = step;
  for (int k = 0; k < 3; ++k) slot->off[k] = step ? off[k] : 0u;
  slot->times_at = pts_room;
  slot->ticket = ++ctx->ahead_tickets;
  slot->state = 1;
  *ticket = slot->ticket;
  return VGICP_OK;
}
}  // namespace

int vgicp_sweep_stage(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time, uint64_t* ticket) {
  if (!ctx || !ticket) return VGICP_ERR_BAD_ARGUMENT;
  *ticket = 0;
  if (ctx->multi) {
    const int rc = vgicp_sweep_stage(vgicp_multi_api::first(ctx), n, points, point_time, ticket);
    if (rc != VGICP_OK) g_stage_error_ctx = ctx->id;
    return rc;
  }
  if (n == 0 || !points) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "empty sweep");
  if (n > 0xFFFFFFFFull) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "sweep too large");
  const uint32_t none[3] = {0, 0, 0};
  return stage_sweep_ahead(ctx, n, points, point_time, 0, none, SIZE_MAX, ticket);
}

int vgicp_sweep_stage_cloud2(vgicp_ctx* ctx, size_t n, const void* data, size_t point_step, size_t off_x, size_t off_y,
                             size_t off_z, size_t off_time, uint64_t* ticket) {
  if (!ctx || !ticket) return VGICP_ERR_BAD_ARGUMENT;
  *ticket = 0;
  if (ctx->multi) {
    const int rc = vgicp_sweep_stage_cloud2(vgicp_multi_api::first(ctx), n, data, point_step, off_x, off_y, off_z, off_time, ticket);
    if (rc != VGICP_OK) g_stage_error_ctx = ctx->id;
    return rc;
  }
  if (n == 0 || !data) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "empty sweep");
  if (n > 0xFFFFFFFFull) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "sweep too large");
  if (point_step < 12 || point_step > 64 || point_step % 4 != 0)
    return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "point_step must be a multiple of 4 between 12 and 64 bytes");
  for (size_t o : {off_x, off_y, off_z})
    if (o % 4 != 0 || o + 4 > point_step) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "x / y / z must be float32 fields inside the record, 4-byte aligned");
  if (off_time != SIZE_MAX && off_time + 8 > point_step) return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "the float64 capture time must lie inside the record");
  const uint32_t off[3] = {(uint32_t)off_x, (uint32_t)off_y, (uint32_t)off_z};
  return stage_sweep_ahead(ctx, n, data, nullptr, (uint32_t)point_step, off, off_time, ticket);
}

int vgicp_sweep_unstage(vgicp_ctx* ctx, uint64_t ticket) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    const int rc = vgicp_sweep_unstage(vgicp_multi_api::first(ctx), ticket);
    if (rc != VGICP_OK) g_stage_error_ctx = ctx->id;
    return rc;
  }
  std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
  for (auto& s : ctx->ahead)
    if (ticket != 0 && s.state == 1 && s.ticket == ticket) {
      s.state = 0;   // staged, never handed to the device: nothing reads it
      s.ticket = 0;
      return VGICP_OK;
    }
  return fail_stage(ctx, VGICP_ERR_BAD_ARGUMENT, "no sweep staged under this ticket (used, dropped already, or never given out)");
}

int vgicp_scan_prepare_staged_async(vgicp_ctx* ctx, uint64_t ticket, size_t num_states, const double* states,
                                    const double extrinsic[16], double voxel_size, int knn) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::scan_prepare(ctx, 0, nullptr, nullptr, num_states, states, extrinsic, voxel_size, knn, nullptr, nullptr, true, ticket);
  vgicp_ctx::AheadSlot* slot = nullptr;
  {
    std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
    for (auto& s : ctx->ahead)
      if (s.state == 1 && s.ticket == ticket) { slot = &s; break; }
  }
  if (!slot || ticket == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "no sweep staged under this ticket (staged by vgicp_sweep_stage, used once)");
  if (num_states > 0 && !slot->has_times) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "the sweep was staged without capture times");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  if (!slot->done) VG_HIP(ctx, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
  const int rc = scan_prepare_enqueue(ctx, slot->n, reinterpret_cast<const double*>(slot->mem),
                                      reinterpret_cast<const double*>(slot->mem + slot->times_at), num_states, states, extrinsic,
                                      voxel_size, knn, slot);
  std::lock_guard<std::mutex> lk(ctx->ahead_mutex);
  // whatever the outcome the ticket is used up; the slot is free again once the kernels that read it are through
  // (an enqueue that failed before it launched anything left `done` as it was: an old, completed event)
  slot->state = 2;
  return rc;
}

int vgicp_scan_info(vgicp_ctx* ctx, size_t* kept, int64_t* deskewed, uint64_t* indefinite) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::scan_info(ctx, kept, deskewed, indefinite);
  if (kept) *kept = 0;
  if (deskewed) *deskewed = 0;
  if (indefinite) *indefinite = 0;
  int rc = settle(ctx);
  if (rc != VGICP_OK) return rc;
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident");
  if (kept) *kept = ctx->n;
  if (deskewed) *deskewed = ctx->prep_with_deskew ? ctx->prep_deskewed : 0;
  if (indefinite) *indefinite = ctx->prep_indefinite;
  return VGICP_OK;
}

int vgicp_scan_prepare(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time,
                       size_t num_states, const double* states, const double extrinsic[16],
                       double voxel_size, int knn, size_t* kept, int64_t* deskewed) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) {
    if (!kept) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "kept is NULL");
    return vgicp_multi_api::scan_prepare(ctx, n, points, point_time, num_states, states, extrinsic, voxel_size, knn, kept, deskewed, false);
  }
  if (!kept) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "kept is NULL");
  *kept = 0;
  if (deskewed) *deskewed = 0;
  int rc = settle(ctx);
  if (rc != VGICP_OK) return rc;
  rc = scan_prepare_enqueue(ctx, n, points, point_time, num_states, states, extrinsic, voxel_size, knn);
  if (rc != VGICP_OK) {
    if (deskewed && ctx->prep_deskewed < 0) *deskewed = -1;
    return rc;
  }
  rc = settle(ctx);  // one synchronisation
  if (rc != VGICP_OK) return rc;
  *kept = ctx->n;
  if (deskewed && ctx->prep_with_deskew) *deskewed = ctx->prep_deskewed;
  return VGICP_OK;
}

// CloudPreprocessor::process with the host copy the reference leaves behind (src/CloudPreprocessor.cpp:8-23 ends with the
// prepared scan IN the caller's cloud), for a preparation that was only enqueued (vgicp_scan_prepare_async):
//   vgicp_scan_fetch_begin   enqueues ONE kernel behind the preparation that will write the prepared scan into page-locked
//                            memory piece by piece, and returns as soon as the down-sampling has told the host how many
//                            points it keeps (a posted write of run_scan_kernel, ~0.1 ms before the neighbour search and
//                            the covariances are through): the caller sizes its vectors in that time;
//   vgicp_scan_fetch_end     copies every piece out the moment its flag arrives, then brings the context up to date
//                            (what vgicp_scan_info does) — no copy command, one synchronisation at the very end.
// Against vgicp_scan_info + vgicp_scan_download (a synchronisation, two copy commands, a second synchronisation and a
// 2.6 MB memcpy in a row: 0.25 - 0.33 ms for a 27 000-point scan) this is the transfer itself.
namespace {
constexpr uint32_t kFetchPiece = 64u << 10;
constexpr double kFetchPatienceSeconds = 5.0;
int ensure_fetch_stage(vgicp_ctx* ctx, size_t points) {
  if (points <= ctx->fetch_cap_points && ctx->h_fetch) return VGICP_OK;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_fetch) VG_HIP(ctx, hipHostFree(ctx->h_fetch));
  ctx->h_fetch = nullptr;
  ctx->fetch_cap_points = 0;
  const size_t cap = std::max<size_t>(points + points / 4, 4096);
  const size_t data = (((cap * 24u) + 255u) & ~size_t(255)) + cap * 72u + 256u;
  const size_t flag_bytes = (data / kFetchPiece + 2) * 64;
  VG_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_fetch), flag_bytes + data, 0));
  std::memset(ctx->h_fetch, 0, flag_bytes);
  void* dev = nullptr;
  VG_HIP(ctx, hipHostGetDevicePointer(&dev, ctx->h_fetch, 0));
  ctx->h_fetch_dev = static_cast<char*>(dev);
  ctx->fetch_cap_points = cap;
  ctx->fetch_flag_bytes = flag_bytes;
  return VGICP_OK;
}
}  // namespace

int vgicp_scan_fetch_begin(vgicp_ctx* ctx, size_t* kept) {
  if (!ctx || !kept) return VGICP_ERR_BAD_ARGUMENT;
  *kept = 0;
  if (ctx->multi || !ctx->scan_pending) {
    // nothing pending (or a multi-device context, whose prepared scan is dealt out first): the two-step path
    ctx->fetch_open = false;
    return vgicp_scan_info(ctx, kept, nullptr, nullptr);
  }
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_fetch_stage(ctx, ctx->n_upper);
  if (rc != VGICP_OK) return rc;
  if (++ctx->fetch_seq == 0) ++ctx->fetch_seq;
  VG_HIP(ctx, launch_fetch(ctx->stream, ctx->d_scan_aos, ctx->d_scan_aos + 3 * ctx->scan_capacity, ctx->d_counters, ctx->prep_epoch,
                           (uint32_t)std::min<size_t>(ctx->fetch_cap_points, ctx->n_upper),
                           ctx->h_fetch_dev + ctx->fetch_flag_bytes, reinterpret_cast<uint32_t*>(ctx->h_fetch_dev),
                           ctx->h_fetch_hdr_dev + 8, ctx->fetch_seq, kFetchPiece));
  ctx->fetch_open = true;
  // how many points the down-sampling kept (or the fetch kernel's first word: the preparation is through, refused or not)
  const double t0 = now_seconds();
  for (uint32_t spins = 0;; ++spins) {
    const unsigned long long k = __atomic_load_n(ctx->h_fetch_hdr, __ATOMIC_ACQUIRE);
    if ((uint32_t)(k >> 32) == ctx->prep_epoch) { ctx->fetch_kept = (uint32_t)k; break; }
    const unsigned long long d = __atomic_load_n(ctx->h_fetch_hdr + 8, __ATOMIC_ACQUIRE);
    if ((uint32_t)(d >> 32) == ctx->fetch_seq) {   // the fetch kernel has started: the preparation ended without a count
      ctx->fetch_open = false;
      return vgicp_scan_info(ctx, kept, nullptr, nullptr);   // reports why (a refused scan), or the count after all
    }
    __builtin_ia32_pause();
    if ((spins & 4095u) == 4095u && now_seconds() - t0 > kFetchPatienceSeconds) {
      ctx->fetch_open = false;
      const int rc_info = vgicp_scan_info(ctx, kept, nullptr, nullptr);
      return rc_info != VGICP_OK ? rc_info : fail(ctx, VGICP_ERR_TIMEOUT, "the preparation did not report its size within 5 s");
    }
  }
  *kept = ctx->fetch_kept;
  return VGICP_OK;
}

int vgicp_scan_fetch_end(vgicp_ctx* ctx, size_t capacity, double* points, double* covs, size_t* n) {
  if (!ctx || !n) return VGICP_ERR_BAD_ARGUMENT;
  *n = 0;
  if (!ctx->fetch_open) return vgicp_scan_download(ctx, capacity, points, covs, n);
  ctx->fetch_open = false;
  const size_t kept = ctx->fetch_kept;
  int rc_copy = VGICP_OK;
  if (kept > 0 && (capacity < kept || !points || !covs)) {
    rc_copy = fail(ctx, VGICP_ERR_BAD_ARGUMENT, "capacity smaller than the prepared scan (or a NULL output pointer)");
  } else if (kept > 0) {
    const size_t pb = kept * 24u, pb_pad = (pb + 255u) & ~size_t(255), total = pb_pad + kept * 72u;
    const uint32_t pieces = (uint32_t)((total + kFetchPiece - 1) / kFetchPiece);
    const uint32_t* flags = reinterpret_cast<const uint32_t*>(ctx->h_fetch);
    const char* stage = ctx->h_fetch + ctx->fetch_flag_bytes;
    const double t0 = now_seconds();
    for (uint32_t piece = 0; piece < pieces && rc_copy == VGICP_OK; ++piece) {
      for (uint32_t spins = 0; __atomic_load_n(flags + 16u * piece, __ATOMIC_ACQUIRE) != ctx->fetch_seq; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 4095u) == 4095u && now_seconds() - t0 > kFetchPatienceSeconds) {
          rc_copy = fail(ctx, VGICP_ERR_TIMEOUT, "the prepared scan did not arrive within 5 s");
          break;
        }
      }
      if (rc_copy != VGICP_OK) break;
      const size_t off = (size_t)piece * kFetchPiece, len = std::min<size_t>(kFetchPiece, total - off);
      // a piece may hold the end of the points, the padding and the beginning of the covariances
      if (off < pb) std::memcpy(reinterpret_cast<char*>(points) + off, stage + off, std::min(len, pb - off));
      if (off + len > pb_pad) {
        const size_t from = std::max(off, pb_pad);
        std::memcpy(reinterpret_cast<char*>(covs) + (from - pb_pad), stage + from, off + len - from);
      }
    }
  }
  // the preparation's own verdict and counters, the pending map insertion's totals: as every synchronising entry point
  const int rc = settle(ctx);
  if (rc != VGICP_OK) return rc;
  if (rc_copy != VGICP_OK) return rc_copy;
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident");
  if (ctx->n != kept) return fail(ctx, VGICP_ERR_HIP, "the preparation reported two different sizes");
  *n = kept;
  return VGICP_OK;
}

int vgicp_scan_download(vgicp_ctx* ctx, size_t capacity, double* points, double* covs, size_t* n) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi) return vgicp_multi_api::scan_download(ctx, capacity, points, covs, n);
  if (!n) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "n is NULL");
  *n = 0;
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident: call vgicp_scan_upload or vgicp_scan_prepare first");
  *n = ctx->n;
  if (ctx->n == 0 || (!points && !covs)) return VGICP_OK;  // both NULL: size query
  if (capacity < ctx->n) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "capacity smaller than the resident scan");
  if (!points || !covs) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL output pointer");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  arena_reset(ctx);
  const size_t pb = (size_t)ctx->n * 3 * sizeof(double), cb = (size_t)ctx->n * 9 * sizeof(double);
  VG_RC(user_d2h(ctx, points, ctx->d_scan_aos, pb));
  VG_RC(user_d2h(ctx, covs, ctx->d_scan_aos + 3 * ctx->scan_capacity, cb));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  user_copies_finish(ctx);
  return VGICP_OK;
}

int vgicp_peer_export(vgicp_ctx* ctx, void* handle64) {
  if (!ctx || !handle64) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  static_assert(sizeof(hipIpcMemHandle_t) == VGICP_PEER_HANDLE_BYTES, "handle size");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_mailbox(ctx);
  if (rc != VGICP_OK) return rc;
  hipIpcMemHandle_t h;
  VG_HIP(ctx, hipIpcGetMemHandle(&h, ctx->d_mail));
  std::memcpy(handle64, &h, sizeof h);
  return VGICP_OK;
}

int vgicp_peer_connect(vgicp_ctx* ctx, int world_size, int rank, const void* handles) {
  if (!ctx || !handles) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (world_size < 1 || world_size > kMaxRanks || rank < 0 || rank >= world_size)
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "bad world_size / rank (at most 16 ranks)");
  if (ctx->peers_connected) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "peers already connected");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_mailbox(ctx);
  if (rc != VGICP_OK) return rc;
  // this rank's mailbox: the rows the ranks write are unset, the others +0.0 for good
  std::vector<unsigned long long> img(kMailWords, 0ull);
  for (int buf = 0; buf < 3; ++buf)
    for (int r = 0; r < world_size; ++r)
      for (int sl = 0; sl <= kCountSlot; ++sl) img[((size_t)buf * kMaxRanks + r) * kSlots + sl] = kRowUnset;
  VG_HIP(ctx, hipMemcpy(ctx->d_mail, img.data(), kMailWords * 8, hipMemcpyHostToDevice));
  for (int r = 0; r < world_size; ++r) {
    if (r == rank) {
      ctx->peer_mail[r] = ctx->d_mail;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * VGICP_PEER_HANDLE_BYTES, sizeof h);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      close_peers(ctx);
      return fail_hip(ctx, e, "hipIpcOpenMemHandle(peer mailbox)");
    }
    ctx->peer_mail[r] = static_cast<double*>(p);
  }
  VG_HIP(ctx, hipMemcpy(ctx->d_mail_table, ctx->peer_mail, kMaxRanks * sizeof(double*), hipMemcpyHostToDevice));
  ctx->peer_world = world_size;
  ctx->peer_rank = rank;
  ctx->world_size = world_size;
  ctx->rank = rank;
  ctx->mail_round0 = 0;
  ctx->mail_seq = 0;
  ctx->peer_enabled = true;
  ctx->peers_connected = true;
  return VGICP_OK;
}

int vgicp_peer_disconnect(vgicp_ctx* ctx) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  close_peers(ctx);
  if (!ctx->comm) {
    ctx->world_size = 1;
    ctx->rank = 0;
  }
  return VGICP_OK;
}

int vgicp_comm_unique_id(vgicp_ctx* ctx, void* id128) {
  if (!ctx || !id128) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  int rc = load_rccl(ctx);
  if (rc != VGICP_OK) return rc;
  ncclUniqueId id;
  const int e = ctx->rccl.GetUniqueId(&id);
  if (e != 0) return fail_rccl(ctx, e, "ncclGetUniqueId");
  std::memcpy(id128, id.internal, VGICP_UNIQUE_ID_BYTES);
  return VGICP_OK;
}

int vgicp_comm_init(vgicp_ctx* ctx, int world_size, int rank, const void* id128) {
  if (!ctx || !id128) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  { const int rc_settle = settle(ctx); if (rc_settle != VGICP_OK) return rc_settle; }
  if (world_size < 1 || rank < 0 || rank >= world_size)
    return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "bad world_size / rank");
  if (ctx->comm) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "communicator already initialised");
  int rc = load_rccl(ctx);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(id.internal, id128, VGICP_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const int e = ctx->rccl.CommInitRank(&comm, world_size, id, rank);
  if (e != 0) return fail_rccl(ctx, e, "ncclCommInitRank");
  ctx->comm = comm;
  ctx->world_size = world_size;
  ctx->rank = rank;
  // Device-initiated exchange on top: every rank's mailbox handle travels through ONE RCCL all-gather, peers
  // are mapped, and one all-reduce makes sure every mailbox is initialised before any kernel writes into one.
  // Any failure leaves the communicator on the host-enqueued all-reduce (VGICP_PEER_EXCHANGE=0 asks for that).
  const char* want = std::getenv("VGICP_PEER_EXCHANGE");
  if (world_size > 1 && world_size <= kMaxRanks && !(want && want[0] == '0') && ctx->rccl.AllGather &&
      !ctx->peers_connected) {
    std::string why;
    char mine[VGICP_PEER_HANDLE_BYTES];
    char* d_all = nullptr;
    std::vector<char> all((size_t)world_size * VGICP_PEER_HANDLE_BYTES);
    bool ok = vgicp_peer_export(ctx, mine) == VGICP_OK;
    if (!ok) why = ctx->err;
    // every rank must take part in the collectives whatever happened locally: a failed export sends zeros
    if (!ok) std::memset(mine, 0, sizeof mine);
    // (a rank that could not even allocate these few bytes cannot take part in the collectives below and fails
    // the whole call; its peers would wait for it inside RCCL as they would for any rank that died)
    if (hipMalloc(reinterpret_cast<void**>(&d_all), all.size() + VGICP_PEER_HANDLE_BYTES) != hipSuccess)
      return fail(ctx, VGICP_ERR_HIP, "hipMalloc(handle exchange) failed");
    {
      char* d_mine = d_all + all.size();
      bool coll = hipMemcpyAsync(d_mine, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                  ctx->rccl.AllGather(d_mine, d_all, VGICP_PEER_HANDLE_BYTES, kNcclChar, ctx->comm, ctx->stream) == 0 &&
                  hipMemcpyAsync(all.data(), d_all, all.size(), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                  hipStreamSynchronize(ctx->stream) == hipSuccess;
      if (!coll) { ok = false; why = "handle all-gather failed"; }
      bool any_zero = false;
      for (int r = 0; r < world_size && coll; ++r) {
        bool zero = true;
        for (int k = 0; k < VGICP_PEER_HANDLE_BYTES; ++k) zero = zero && all[(size_t)r * VGICP_PEER_HANDLE_BYTES + k] == 0;
        any_zero = any_zero || zero;
      }
      if (any_zero) { ok = false; why = "a rank could not export its mailbox"; }
      if (ok && vgicp_peer_connect(ctx, world_size, rank, all.data()) != VGICP_OK) { ok = false; why = ctx->err; }
      // agreement + barrier: the sum of the ranks' verdicts; the peer path is used only if all of them connected
      double verdict = ok ? 1.0 : 0.0;
      double* d_v = reinterpret_cast<double*>(d_all);
      if (coll && hipMemcpyAsync(d_v, &verdict, sizeof verdict, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
          ctx->rccl.AllReduce(d_v, d_v, 1, kNcclDouble, kNcclSum, ctx->comm, ctx->stream) == 0 &&
          hipMemcpyAsync(&verdict, d_v, sizeof verdict, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
          hipStreamSynchronize(ctx->stream) == hipSuccess) {
        if (verdict != (double)world_size) {
          if (ctx->peers_connected) close_peers(ctx);
          ctx->world_size = world_size;
          ctx->rank = rank;
          if (why.empty()) why = "another rank could not connect";
        }
      } else if (ctx->peers_connected) {
        close_peers(ctx);
        ctx->world_size = world_size;
        ctx->rank = rank;
      }
      (void)hipFree(d_all);
    }
    if (!ctx->peers_connected && ctx->dev.verbose)
      std::fprintf(stderr, "[vgicp] rank %d: no device-initiated exchange (%s); using RCCL all-reduce per iteration\n", rank,
                   why.c_str());
    ctx->peer_status = ctx->peers_connected ? std::string() : ("mailboxes not wired: " + (why.empty() ? std::string("unknown reason") : why));
    ctx->err.clear();
  } else if (world_size > 1 && !ctx->peers_connected) {
    ctx->peer_status = (want && want[0] == '0') ? "mailboxes not wired: VGICP_PEER_EXCHANGE=0" :
                       world_size > kMaxRanks ? "mailboxes not wired: more than 16 ranks" : "mailboxes not wired: librccl has no ncclAllGather";
  }
  return VGICP_OK;
}

const char* vgicp_peer_status(const vgicp_ctx* ctx) {
  if (!ctx) return "no context";
  if (ctx->multi) return vgicp_multi_api::peer_status(ctx);
  if (ctx->peers_connected && !ctx->peer_enabled) return "mailboxes wired, but a launch gave up waiting for a peer: one launch + one RCCL all-reduce per iteration since";
  return ctx->peer_status.c_str();
}

int vgicp_comm_destroy(vgicp_ctx* ctx) {
  if (!ctx) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || ctx->owner) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "a multi-device context has its exchange built in: communicators and hand-wired peers are for one-process-per-GPU hosts");
  if (ctx->peers_connected) {
    VG_HIP(ctx, hipSetDevice(ctx->device));
    VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    close_peers(ctx);
  }
  if (ctx->comm) {
    VG_HIP(ctx, hipSetDevice(ctx->device));
    VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->world_size = 1;
  ctx->rank = 0;
  return VGICP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// What the in-process multi-device context (vgicp_multi.hip) needs from this file besides the entry points above.
// ---------------------------------------------------------------------------------------------------------------
namespace vgicp_internal {

int settle_context(vgicp_ctx* ctx) { return settle(ctx); }

bool insertion_lists_stay_short_for(const vgicp_ctx* ctx, double prep_voxel) {
  if (!(prep_voxel > 0.0) || ctx->dev.insert_sort) return false;
  const double per_axis = std::ceil(ctx->voxel_size / prep_voxel) + 1.0;
  return per_axis * per_axis * per_axis <= 64.0;
}

// Whether an align of n points / max_it rounds would have to (re)allocate on this context, and the allocation itself.
// hipFree waits for the whole DEVICE: sub-contexts that share a device must not meet one between their launches (a
// neighbour's persistent kernel is already running and waiting for this sub-context's), so the multi-device context
// grows every sub-context's buffers in a phase of its own before anybody launches.
bool align_needs_allocation(const vgicp_ctx* ctx, size_t n, int max_it) {
  // (the dense copy's storage is made when the table is: reserve_dense; an align only rebuilds its contents)
  return !ctx->d_scan || n > ctx->scan_capacity || max_it > ctx->log_capacity || ctx->log_capacity == 0;
}
int reserve_for_align(vgicp_ctx* ctx, size_t n, int max_it) {
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = settle(ctx);
  if (rc != VGICP_OK) return rc;
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!ctx->d_scan || n > ctx->scan_capacity) {
    rc = ensure_scan(ctx, n);
    if (rc != VGICP_OK) return rc;
    ctx->scan_ready = false;   // whatever was resident went with the old buffers
    ctx->n = 0;
  }
  return ensure_log(ctx, std::max(max_it, 1));
}

int wire_mailboxes(vgicp_ctx* const* subs, int n) {
  if (n < 1 || n > kMaxRanks) return fail(subs[0], VGICP_ERR_BAD_ARGUMENT, "at most 16 devices");
  // every device must be able to store into every other device's mailbox (xGMI / PCIe peer access)
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      if (subs[a]->device == subs[b]->device) continue;
      int can = 0;
      VG_HIP(subs[a], hipDeviceCanAccessPeer(&can, subs[a]->device, subs[b]->device));
      if (!can) return fail(subs[a], VGICP_ERR_HIP, "device " + std::to_string(subs[a]->device) + " cannot access device " +
                            std::to_string(subs[b]->device) + " as a peer");
      VG_HIP(subs[a], hipSetDevice(subs[a]->device));
      const hipError_t e = hipDeviceEnablePeerAccess(subs[b]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail_hip(subs[a], e, "hipDeviceEnablePeerAccess");
      (void)hipGetLastError();
    }
  for (int r = 0; r < n; ++r) {
    vgicp_ctx* ctx = subs[r];
    VG_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = ensure_mailbox(ctx);
    if (rc != VGICP_OK) return rc;
    // this rank's mailbox: the rows the ranks write are unset, the others +0.0 for good; verdict words 0
    std::vector<unsigned long long> img(kMailWords, 0ull);
    for (int buf = 0; buf < 3; ++buf)
      for (int q = 0; q < n; ++q)
        for (int sl = 0; sl <= kCountSlot; ++sl) img[((size_t)buf * kMaxRanks + q) * kSlots + sl] = kRowUnset;
    VG_HIP(ctx, hipMemcpy(ctx->d_mail, img.data(), kMailWords * 8, hipMemcpyHostToDevice));
  }
  for (int r = 0; r < n; ++r) {
    vgicp_ctx* ctx = subs[r];
    VG_HIP(ctx, hipSetDevice(ctx->device));
    for (int q = 0; q < kMaxRanks; ++q) ctx->peer_mail[q] = q < n ? subs[q]->d_mail : nullptr;
    VG_HIP(ctx, hipMemcpy(ctx->d_mail_table, ctx->peer_mail, kMaxRanks * sizeof(double*), hipMemcpyHostToDevice));
    ctx->peer_mail_is_ipc = false;
    ctx->peer_world = n;
    ctx->peer_rank = r;
    ctx->world_size = n;
    ctx->rank = r;
    ctx->mail_round0 = 0;
    ctx->mail_seq = 0;
    ctx->peer_enabled = true;
    ctx->peers_connected = true;
  }
  return VGICP_OK;
}

namespace {
// tree_sum<16> of vgicp_kernels.hip on the host: the order in which poll_and_sum<true> adds the ranks' rows
double tree_sum_host(const double* v, int count) {
  if (count == 1) return v[0];
  return tree_sum_host(v, count / 2) + tree_sum_host(v + count / 2, count - count / 2);
}
}  // namespace

int align_host_summed(vgicp_ctx* const* subs, int n, const double guess[16], const vgicp_params* params,
                      double out_pose[16], vgicp_stats* stats) {
  const double t0 = now_seconds();
  vgicp_ctx* lead = subs[0];
  int rc = check_params(lead, params);
  if (rc != VGICP_OK) return rc;
  const int max_it = params->max_iteration;
  const bool profile = (params->flags & VGICP_FLAG_PROFILE) != 0;
  std::vector<uint32_t> grid((size_t)n);
  for (int r = 0; r < n; ++r) {
    vgicp_ctx* ctx = subs[r];
    VG_HIP(ctx, hipSetDevice(ctx->device));
    rc = settle(ctx);
    if (rc != VGICP_OK) return rc;
    if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
    if (!ctx->scan_ready) return fail(ctx, VGICP_ERR_NOT_READY, "no scan resident: call vgicp_scan_upload first");
    rc = ensure_log(ctx, max_it);
    if (rc != VGICP_OK) return rc;
    AlignState* h0 = &ctx->h_state[0];
    std::memset(h0, 0, sizeof(AlignState));
    pose_to_state(guess, h0->pose);
    h0->cosine_threshold = params->cosine_threshold;
    h0->translation_sq_threshold = params->translation_sq_threshold;
    h0->max_iteration = max_it;
    h0->done = (max_it == 0) ? 1 : 0;
    VG_HIP(ctx, hipMemcpyAsync(ctx->d_state, h0, sizeof(AlignState), hipMemcpyHostToDevice, ctx->stream));
    grid[(size_t)r] = iterate_grid(ctx);
  }
  const int total_launches = max_it > 0 ? max_it + 1 : 0;  // max_it bodies + the closing prologue
  VG_HIP(lead, hipSetDevice(lead->device));
  if (profile && (int)lead->ev_prof.size() < 2 * total_launches) {
    const size_t old = lead->ev_prof.size();
    lead->ev_prof.resize(2 * (size_t)total_launches, nullptr);
    for (size_t k = old; k < lead->ev_prof.size(); ++k) VG_HIP(lead, hipEventCreate(&lead->ev_prof[k]));
  }
  VG_HIP(lead, hipEventRecord(lead->ev_begin, lead->stream));
  int launched = 0;
  for (int j = 0; j < total_launches; ++j) {
    const bool closing = j == max_it;
    for (int r = 0; r < n; ++r) {
      vgicp_ctx* ctx = subs[r];
      VG_HIP(ctx, hipSetDevice(ctx->device));
      IterArgs a = base_args(ctx);
      a.state_in = ctx->d_state + (j & 1);
      a.state_out = ctx->d_state + ((j + 1) & 1);
      a.rows = ctx->d_rows[j & 1];
      a.prev = ctx->d_sums;           // the row the host summed over the ranks
      a.prev_rows = j > 0 ? 1u : 0u;
      a.memo_valid = j > 0 ? 1u : 0u;
      if (profile && r == 0) VG_HIP(ctx, hipEventRecord(ctx->ev_prof[2 * j], ctx->stream));
      if (closing) VG_HIP(ctx, launch_close(ctx->stream, a, ctx->iter_block));
      else {
        VG_HIP(ctx, launch_iterate(ctx->stream, a, grid[(size_t)r], ctx->iter_block));
        VG_HIP(ctx, launch_fold_rows(ctx->stream, a.rows, grid[(size_t)r], a.state_out, ctx->d_sums));
        // the rank's row goes to pinned memory (the header row of the pinned log: unused outside a persistent launch)
        VG_HIP(ctx, hipMemcpyAsync(ctx->h_log - kSlots, ctx->d_sums, kSlots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      }
      if (profile && r == 0) VG_HIP(ctx, hipEventRecord(ctx->ev_prof[2 * j + 1], ctx->stream));
      VG_HIP(ctx, hipMemcpyAsync(&ctx->h_state[1], a.state_out, sizeof(AlignState), hipMemcpyDeviceToHost, ctx->stream));
    }
    ++launched;
    for (int r = 0; r < n; ++r) {
      VG_HIP(subs[r], hipSetDevice(subs[r]->device));
      VG_HIP(subs[r], hipStreamSynchronize(subs[r]->stream));
    }
    if (closing || lead->h_state[1].done) break;
    // the ranks' rows, added in the order the mailbox path adds them (identical bits on every device)
    double total[kSlots];
    for (int sl = 0; sl < kSlots; ++sl) {
      double x[kMaxRanks];
      for (int q = 0; q < kMaxRanks; ++q) x[q] = (q < n && sl <= kCountSlot) ? (subs[q]->h_log - kSlots)[sl] : 0.0;
      total[sl] = tree_sum_host(x, kMaxRanks);
    }
    for (int r = 0; r < n; ++r) {
      vgicp_ctx* ctx = subs[r];
      VG_HIP(ctx, hipSetDevice(ctx->device));
      std::memcpy(ctx->h_log - kSlots, total, sizeof total);
      VG_HIP(ctx, hipMemcpyAsync(ctx->d_sums, ctx->h_log - kSlots, kSlots * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  VG_HIP(lead, hipSetDevice(lead->device));
  VG_HIP(lead, hipEventRecord(lead->ev_end, lead->stream));
  AlignState* hf = &lead->h_state[0];
  VG_HIP(lead, hipMemcpyAsync(hf, lead->d_state + (launched & 1), sizeof(AlignState), hipMemcpyDeviceToHost, lead->stream));
  const bool want_log = stats && (stats->corr_count || stats->normal_eq);
  if (want_log && max_it > 0)
    VG_HIP(lead, hipMemcpyAsync(lead->h_log, lead->d_log, (size_t)max_it * kSlots * sizeof(double), hipMemcpyDeviceToHost, lead->stream));
  VG_HIP(lead, hipStreamSynchronize(lead->stream));
  state_to_pose(hf->pose, out_pose);
  if (stats) {
    stats->iterations = hf->iteration;
    stats->converged = hf->converged;
    stats->world_size = n;
    stats->launches = launched;
    float ms = 0.f;
    VG_HIP(lead, hipEventElapsedTime(&ms, lead->ev_begin, lead->ev_end));
    stats->device_seconds = ms * 1e-3;
    for (int it = 0; it < hf->iteration; ++it) {
      const double* row = lead->h_log + (size_t)it * kSlots;
      if (stats->corr_count) stats->corr_count[it] = (uint64_t)row[kCountSlot];
      if (stats->normal_eq) std::memcpy(stats->normal_eq + (size_t)it * kNormalEq, row, kNormalEq * sizeof(double));
    }
    if (profile && stats->kernel_ms) {
      for (int it = 0; it < std::min(launched, max_it); ++it) {
        float k = 0.f;
        VG_HIP(lead, hipEventElapsedTime(&k, lead->ev_prof[2 * it], lead->ev_prof[2 * it + 1]));
        stats->kernel_ms[it] = k;
      }
    }
    stats->seconds = now_seconds() - t0;
  }
  if (!finite16(out_pose)) return fail(lead, VGICP_ERR_DEGENERATE, "solved pose is not finite (singular normal equations)");
  return VGICP_OK;
}

int adopt_device_scan(vgicp_ctx* ctx, int src_device, const double* d_points, const double* d_covs, size_t n,
                      double prep_voxel, hipEvent_t ready) {
  if (n > 0xFFFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = settle(ctx);
  if (rc != VGICP_OK) return rc;
  rc = ensure_scan(ctx, n);
  if (rc != VGICP_OK) return rc;
  ++ctx->scan_generation;
  ctx->scan_ready = false;
  ctx->prep_voxel = prep_voxel;
  ctx->prep_with_deskew = false;
  ctx->prep_deskewed = 0;
  ctx->prep_indefinite = 0;
  ctx->n = (uint32_t)n;
  ctx->stride = ctx->scan_capacity;
  if (ready) VG_HIP(ctx, hipStreamWaitEvent(ctx->stream, ready, 0));
  if (n > 0) {
    double* aos_pts = ctx->d_scan_aos;
    double* aos_cov = ctx->d_scan_aos + 3 * ctx->scan_capacity;
    if (src_device != ctx->device) {
      VG_HIP(ctx, hipMemcpyPeerAsync(aos_pts, ctx->device, d_points, src_device, n * 3 * sizeof(double), ctx->stream));
      VG_HIP(ctx, hipMemcpyPeerAsync(aos_cov, ctx->device, d_covs, src_device, n * 9 * sizeof(double), ctx->stream));
      g_copy_ops += 2;
    } else {
      VG_HIP(ctx, hipMemcpyAsync(aos_pts, d_points, n * 3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
      VG_HIP(ctx, hipMemcpyAsync(aos_cov, d_covs, n * 9 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (++ctx->scan_seq == 0) ++ctx->scan_seq;
    ctx->scan_sym_known = true;
    VG_HIP(ctx, launch_pack_scan(ctx->stream, aos_pts, aos_cov, (uint32_t)n, ctx->d_scan, ctx->stride,
                                 ctx->d_ins_counters + 2, ctx->scan_seq));
  }
  ctx->scan_ready = true;
  return VGICP_OK;
}

int map_insert_device(vgicp_ctx* ctx, const double* d_points, const double* d_covs, size_t n, const double transform[16],
                      size_t max_points_per_voxel, bool short_lists, bool deferred, size_t* new_voxels) {
  if (new_voxels) *new_voxels = 0;
  if (!ctx->table) return fail(ctx, VGICP_ERR_NOT_READY, "no voxel map: call vgicp_map_reset first");
  if (!transform) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "NULL pointer");
  if (max_points_per_voxel == 0) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "max_points_per_voxel must be >= 1");
  if (n > 0x7FFFFFFFull) return fail(ctx, VGICP_ERR_BAD_ARGUMENT, "scan too large");
  VG_HIP(ctx, hipSetDevice(ctx->device));
  int rc = (ctx->scan_pending || ctx->insert_pending) ? settle(ctx) : VGICP_OK;
  if (rc != VGICP_OK) return rc;
  if (n == 0) return VGICP_OK;
  rc = ensure_table(ctx, n);
  if (rc != VGICP_OK) return rc;
  const size_t sb = map_insert_scratch_bytes((uint32_t)n);
  rc = ensure_stage(ctx, sb);
  if (rc != VGICP_OK) return rc;
  double pose12[12];
  pose_to_state(transform, pose12);
  if (deferred) {
    if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[4], ctx->stream)); ctx->ev_stage_set[4] = true; }
    ++ctx->map_version;
  VG_HIP(ctx, launch_map_insert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), ctx->voxel_size, d_points, d_covs,
                                  (uint32_t)n, pose12, (uint64_t)max_points_per_voxel, ctx->d_stage, sb, ctx->d_ins_counters,
                                  short_lists));
    if (ctx->stage_events) { VG_HIP(ctx, hipEventRecord(ctx->ev_stage[5], ctx->stream)); ctx->ev_stage_set[5] = true; }
    ctx->insert_pending = true;
    ctx->ins_copy_enqueued = false;
    ctx->insert_pending_upper = n;
    return VGICP_OK;
  }
  VG_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
  ++ctx->map_version;
  VG_HIP(ctx, launch_map_insert(ctx->stream, ctx->table, (uint32_t)(ctx->slots - 1), ctx->voxel_size, d_points, d_covs,
                                (uint32_t)n, pose12, (uint64_t)max_points_per_voxel, ctx->d_stage, sb, ctx->d_counters, short_lists));
  VG_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  VG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->voxels += ctx->h_counters[0];
  if (new_voxels) *new_voxels = ctx->h_counters[0];
  if (ctx->h_counters[1] != 0) return fail(ctx, VGICP_ERR_TABLE_FULL, "voxel table probe sequence exhausted");
  return VGICP_OK;
}

}  // namespace vgicp_internal
