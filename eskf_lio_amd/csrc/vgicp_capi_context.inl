// vgicp_capi_context.inl — part of vgicp_capi.hip.
// The context: creation (device checks, buffers, the environment's switches — read here, once), destruction, error
// text, device info, counters, frame statistics, options.
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
  VG_CREATE(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_fetch_hdr), 640, 0));
  std::memset(ctx->h_fetch_hdr, 0, 640);
  VG_CREATE(hipMalloc(reinterpret_cast<void**>(&ctx->d_fetch_sums), 65 * sizeof(unsigned long long)));
  VG_CREATE(hipMemset(ctx->d_fetch_sums, 0, 65 * sizeof(unsigned long long)));
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
  (void)hipFree(ctx->d_fetch_sums);
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
}  // extern "C"
