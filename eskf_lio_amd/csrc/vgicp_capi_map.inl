// vgicp_capi_map.inl — part of vgicp_capi.hip.
// The device mirror of LocalMap's voxel grid (reset / upsert / erase / size / export) and LocalMap::updateLocalMap on the
// device (insertion of a scan or of the resident scan, eviction).
extern "C" {

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
}  // extern "C"
