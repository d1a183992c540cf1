// vgicp_capi_upload.inl — part of vgicp_capi.hip.
// The scan upload of vgicp_align / vgicp_scan_upload: the copy crew, the staging memory, pack_arena_kernel's launch;
// vgicp_host_register.
extern "C" {

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
}  // extern "C"
