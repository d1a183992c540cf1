// vgicp_capi_registration.inl — part of vgicp_capi.hip.
// vgicp_align / vgicp_align_resident and the single-step hooks (accumulate, solve_step, match, voxel_index).
extern "C" {

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
}  // extern "C"
