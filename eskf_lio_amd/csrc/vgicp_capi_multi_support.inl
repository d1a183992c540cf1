// vgicp_capi_multi_support.inl — part of vgicp_capi.hip.
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
