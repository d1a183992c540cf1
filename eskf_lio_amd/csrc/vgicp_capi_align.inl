// vgicp_capi_align.inl — part of vgicp_capi.hip.
// ICP::align's launch schedule (reference src/Registration.cpp:15-28): the single persistent launch, the
// launch-per-round loop (RCCL all-reduce between launches with a communicator), the dense record copy, give-up and
// fallback handling.
namespace {
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
