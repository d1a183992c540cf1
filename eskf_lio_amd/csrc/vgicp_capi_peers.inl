// vgicp_capi_peers.inl — part of vgicp_capi.hip.
// Multi-GPU with one process per GPU: the hand-wired peer mailboxes and the RCCL communicator.
extern "C" {

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
