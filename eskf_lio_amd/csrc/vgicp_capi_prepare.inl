// vgicp_capi_prepare.inl — part of vgicp_capi.hip.
// The scan preparation (CloudPreprocessor: extrinsic, deskew, down-sampling, 30-NN, covariances): stand-alone entry
// points, the resident / enqueued chain, sweeps staged ahead, the reference-order option, fetch and download.
extern "C" {

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
  ctx->fetch_sums_valid = false;
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
  ctx->fetch_sums_valid = false;
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
                           ctx->h_fetch_hdr_dev + 8, ctx->fetch_seq, kFetchPiece, ctx->d_fetch_sums, ctx->h_fetch_hdr_dev + 16));
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
  ctx->fetch_sums_valid = true;   // the kernel has ended (settle synchronised): its last block posted the sums
  return VGICP_OK;
}

int vgicp_scan_fetch_sums(vgicp_ctx* ctx, uint64_t sums[64]) {
  if (!ctx || !sums) return VGICP_ERR_BAD_ARGUMENT;
  if (ctx->multi || !ctx->fetch_sums_valid)
    return fail(ctx, VGICP_ERR_NOT_READY, "no checksums: the last host copy did not come through vgicp_scan_fetch_begin / _end's kernel");
  std::memcpy(sums, ctx->h_fetch_hdr + 16, 64 * sizeof(uint64_t));
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
}  // extern "C"
