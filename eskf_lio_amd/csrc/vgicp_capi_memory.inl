// vgicp_capi_memory.inl — part of vgicp_capi.hip (one translation unit, cut by concern; see that file).
// Copies between the caller's pageable memory and the device (the page-locked arena, the streaming CPU copies, the
// symmetric-covariance compaction of the scan upload), and the context's device buffers: staging area, voxel table
// (growth / rehash policy), scan, log.
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
}  // namespace
