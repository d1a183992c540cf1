/* vgicp_hip.h — C ABI of the MI355X (gfx950) Voxelized-GICP registration module.
 *
 * This is the drop-in boundary for ONE path of LimHaeryong/ESKF_LIO (reference @ 2024_10_08):
 * the scan-to-local-map registration ICP::align() and the hash-voxel lookup it drives.  Every
 * entry point names the reference interface it replaces (file:line relative to the reference
 * checkout).  Plain C: opaque handle, raw pointers and sizes, int status codes; no C++ types, no
 * exceptions, nothing from PyTorch.  The C++ shim that keeps the reference's class signatures on top
 * of this ABI is include/eskf_lio_shim/; the binding recipe is INTEGRATION.md.
 *
 * Memory conventions (identical to what the reference's containers expose through .data()):
 *   points   n x 3 doubles, xyz per point            (std::vector<Eigen::Vector3d>)
 *   covs     n x 9 doubles, COLUMN-major 3x3 each     (std::vector<Eigen::Matrix3d>)
 *   pose     16 doubles, COLUMN-major 4x4            (Eigen::Isometry3d::matrix().data())
 *   keys     n x 3 int32 voxel indices               (Eigen::Vector3i)
 * The caller owns every host buffer, for the duration of the call only.  The context owns all
 * device memory.  A context serves one caller thread at a time (the reference's align/update are
 * never concurrent: src/main.cpp:68-70); distinct contexts are independent.
 */
#ifndef VGICP_HIP_H_
#define VGICP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGICP_ABI_VERSION 6

typedef struct vgicp_ctx vgicp_ctx;

/* Status codes. The reference has no error reporting at all (no exceptions, no status; M == 0 and
 * singular systems are unguarded: src/Registration.cpp:78), so every code but OK is an addition.
 * Non-convergence is NOT an error (stats.converged = 0, pose still returned), as in the reference
 * (src/Registration.cpp:30-34). */
enum {
  VGICP_OK = 0,
  VGICP_ERR_BAD_ARGUMENT = 1,
  VGICP_ERR_HIP = 2,          /* a HIP runtime call failed; text in vgicp_last_error() */
  VGICP_ERR_RCCL = 3,         /* RCCL missing or a collective failed */
  VGICP_ERR_TABLE_FULL = 4,   /* the device voxel table could not grow */
  VGICP_ERR_DEGENERATE = 5,   /* the solved pose is not finite (singular normal equations) */
  VGICP_ERR_NO_DEVICE = 6,    /* no usable gfx950 device */
  VGICP_ERR_NOT_READY = 7,    /* align before a map / scan exists */
  VGICP_ERR_TIMEOUT = 8       /* a bounded host-side wait ran out (a copy thread of the upload that never delivered) */
};

/* Replaces the three YAML keys ICP's constructor reads (include/ESKF_LIO/Registration.hpp:23-28,
 * config/hilti_config.yaml:50-53). */
typedef struct vgicp_params {
  int32_t max_iteration;
  int32_t chunk_iterations;          /* iterations enqueued between host convergence checks;
                                        0 = library default; >= max_iteration = never look */
  double translation_sq_threshold;
  double cosine_threshold;
  uint32_t flags;                    /* VGICP_FLAG_* */
  uint32_t reserved;
} vgicp_params;

#define VGICP_FLAG_PROFILE 1u        /* bracket every iteration launch with HIP events (kernel_ms) */
#define VGICP_FLAG_NO_PERSISTENT 2u  /* one launch per iteration instead of the single persistent launch
                                        (always the case with a communicator or VGICP_FLAG_PROFILE) */

/* What the reference only prints or drops: per-call convergence (its converged_ member is sticky,
 * include/ESKF_LIO/Registration.hpp:50) and the per-iteration correspondence counts. */
typedef struct vgicp_stats {
  int32_t iterations;        /* rounds executed, the converging round included */
  int32_t converged;         /* this call only */
  int32_t world_size;        /* ranks that contributed (1 without a communicator) */
  int32_t launches;          /* iteration kernels enqueued (>= iterations with chunking) */
  double seconds;            /* host wall time of the call */
  double device_seconds;     /* HIP events on the module's stream around all iteration launches */
  uint64_t* corr_count;      /* optional, max_iteration entries: matched points per iteration,
                                summed over all ranks */
  double* normal_eq;         /* optional, max_iteration x 27: 21 lower-triangle entries of JTJ
                                (row by row) then the 6 entries of JTr, summed over all ranks */
  float* kernel_ms;          /* optional, max_iteration entries, filled under VGICP_FLAG_PROFILE */
} vgicp_stats;

/* ---- context ------------------------------------------------------------------------------- */
int vgicp_abi_version(void);
/* Bind a context to one HIP device (one process per GPU; device_id is the LOCAL ordinal). */
int vgicp_create(int device_id, vgicp_ctx** out);
/* One context that drives SEVERAL devices from the one caller thread the reference has (src/main.cpp:68-70 runs
 * Odometry::run on the main thread; src/ErrorStateKF.cpp:130 is where align is called), SURVEY.md 8(b): the handle is
 * used exactly like a single-device one — ICP::align / LocalMap of the shim do not change — and
 *   - the voxel map is replicated: upsert / erase / insert / evict batches go to every device;
 *   - vgicp_align shards the scan by contiguous point blocks in the order of device_ids (rank r owns
 *     [r n / G, (r + 1) n / G)), every device uploads its shard over its own link and runs the single persistent
 *     launch; the devices' 28-double rows of a round cross xGMI through mailboxes the kernels write themselves
 *     (plain peer pointers after hipDeviceEnablePeerAccess: no second process, no IPC handle, no RCCL);
 *   - scan preparation runs on device_ids[0] and the prepared scan is dealt out by peer copies before the align;
 *   - vgicp_comm_* and vgicp_peer_* are refused (they are for hosts that run one process per GPU).
 * device_ids may name a device several times ({0, 0}): those sub-contexts split the device's compute units — how the
 * path is exercised on a single-GPU box.  n_devices == 1 returns an ordinary context.  At most 16 devices.
 * When an in-kernel wait for another device gives up, that align (and the next few) runs one launch per round with
 * the rows added on the host; vgicp_get_counter(VGICP_COUNTER_PERSISTENT_FALLBACKS) counts it. */
int vgicp_create_multi(const int* device_ids, int n_devices, vgicp_ctx** out);
int vgicp_destroy(vgicp_ctx* ctx);
/* Text of the last failure on this context (never NULL; "" when none). ctx may be NULL for
 * failures of vgicp_create itself.  A failure of vgicp_sweep_stage / vgicp_sweep_stage_cloud2 (which another thread
 * than the context's owner may call) is kept with the thread that made the call and returned to that thread. */
const char* vgicp_last_error(const vgicp_ctx* ctx);
/* name[0..name_len) receives the device's gcnArchName; cu_count its compute units. */
int vgicp_device_info(const vgicp_ctx* ctx, char* name, size_t name_len, int32_t* cu_count,
                      uint64_t* hbm_bytes);

/* Diagnostics the reference has no counterpart for. PERSISTENT_LAUNCHES: aligns attempted as ONE kernel
 * launch for the whole loop; PERSISTENT_FALLBACKS: how many of them gave up waiting for a workgroup that was
 * not resident (another process or stream held CUs) and were re-run with one launch per iteration — the
 * result is the same, the align is slower, the first occurrence also prints one line on stderr, and the
 * context retries the single launch after 8 aligns; UPLOAD_*: bytes and time of host-to-device scan copies. */
enum {
  VGICP_COUNTER_PERSISTENT_LAUNCHES = 0,
  VGICP_COUNTER_PERSISTENT_FALLBACKS = 1,
  VGICP_COUNTER_UPLOAD_BYTES = 2,
  VGICP_COUNTER_UPLOAD_NANOSECONDS = 3,
  VGICP_COUNTER_PREP_INDEFINITE = 4,  /* kept points of the LAST vgicp_preprocess / vgicp_scan_prepare whose
                                         regularised covariance has a negative eigenvalue (see vgicp_preprocess) */
  VGICP_COUNTER_SCAN_GENERATION = 5   /* goes up whenever the RESIDENT scan is replaced (vgicp_align, vgicp_scan_upload,
                                         vgicp_scan_prepare*, vgicp_accumulate): a host object that remembers "my cloud is
                                         the resident scan" (the shim's CloudPreprocessor / ICP pair) checks it; no
                                         synchronisation */,
  VGICP_COUNTER_UPLOAD_SLOW = 6       /* staged uploads whose copy threads were held up for so long (> 0.1 s) that the
                                         packing was repeated behind the launch that reads the staging memory */
};
int vgicp_get_counter(const vgicp_ctx* ctx, int which, uint64_t* value);

/* ---- device mirror of LocalMap's voxel grid -------------------------------------------------
 * Replaces the read side of LocalMap::VoxelGrid (include/ESKF_LIO/LocalMap.hpp:25-26,63-89).  The
 * host LocalMap stays authoritative for save()/visualise; after each updateLocalMap
 * (src/LocalMap.cpp:47-72) it forwards the touched voxels as ONE upsert batch and the evicted keys
 * as ONE erase batch so the mirror is current before the next align. */
int vgicp_map_reset(vgicp_ctx* ctx, double voxel_size, size_t capacity_hint);
/* Insert or overwrite voxels {key -> (mean, covariance)}.  Keys must be unique inside one batch. */
int vgicp_map_upsert(vgicp_ctx* ctx, size_t n, const int32_t* keys, const double* means,
                     const double* covs);
/* Remove voxels (src/LocalMap.cpp:60-72); absent keys are ignored. */
int vgicp_map_erase(vgicp_ctx* ctx, size_t n, const int32_t* keys);
int vgicp_map_size(const vgicp_ctx* ctx, size_t* voxels, size_t* table_slots);

/* ---- device-side map maintenance -------------------------------------------
 * The insertion loop of LocalMap::updateLocalMap on the device (src/LocalMap.cpp:15,44-58 with
 * Voxel's constructor and Voxel::addPoint, include/ESKF_LIO/LocalMap.hpp:72-87): the n points (scan
 * frame) are moved by `transform` exactly as cloud->Transform() does, then inserted IN SCAN ORDER —
 * a missing voxel is constructed from the point, an existing one takes the running-mean update while
 * it holds fewer than max_points_per_voxel points.  Means, covariances and counts come out bit for
 * bit as the reference's serial loop leaves them.  Do not mix with vgicp_map_upsert on one voxel
 * (an upserted voxel starts at count 1).  new_voxels (optional) receives the number of voxels created. */
int vgicp_map_insert_scan(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                          const double transform[16], size_t max_points_per_voxel, size_t* new_voxels);
/* The same for the scan that is already resident (vgicp_scan_upload / the last vgicp_align): the
 * per-frame sequence of src/Odometry.cpp:79,86 — register, then insert with the pose found — without a
 * second upload.  With a communicator every rank holds only its shard, so use vgicp_map_insert_scan
 * with the whole scan there (the map is replicated). */
int vgicp_map_insert_resident(vgicp_ctx* ctx, const double transform[16], size_t max_points_per_voxel,
                              size_t* new_voxels);
/* The eviction loop (src/LocalMap.cpp:60-72, needsPointRemoval :149-154): erase every voxel whose
 * centre (index + 0.5) * voxel_size is farther than distance_threshold from position. */
int vgicp_map_evict(vgicp_ctx* ctx, const double position[3], double distance_threshold,
                    size_t* removed);
/* Read the mirror back (order unspecified): keys n x 3, means n x 3, covs n x 9, counts n.
 * `written` receives min(voxels, capacity). For tests and for LocalMap::save(). */
int vgicp_map_export(vgicp_ctx* ctx, size_t capacity, int32_t* keys, double* means, double* covs,
                     uint64_t* counts, size_t* written);

/* ---- registration ---------------------------------------------------------------------------
 * vgicp_align replaces ICP::align(cloud, localMap, guess) (src/Registration.cpp:7-35; declared
 * include/ESKF_LIO/Registration.hpp:30-32) including the hot loop it drives:
 * LocalMap::correspondenceMatching (src/LocalMap.cpp:78-118), ICP::computeTransform and
 * computeJTJAndJTr (src/Registration.cpp:52-102), Open3D PointCloud::Transform (call sites
 * src/Registration.cpp:13,27), Utils::se3ToSE3 (src/Utils.cpp:40-63) and ICP::convergenceCheck
 * (src/Registration.cpp:37-50).  Inputs are not modified (the reference deep-copies the cloud,
 * src/Registration.cpp:11; here the scan is copied to the device instead).
 * Lifetime of `points` / `covs`: free again on return — the reference frees its cloud every frame
 * (src/Odometry.cpp:84-87), and the upload is built for exactly that caller: the scan is copied by this thread and two
 * helpers (VGICP_UPLOAD_THREADS in the environment = threads in all, default 3; they make no HIP call) into page-locked
 * staging memory of the context, unit by unit, while ONE kernel launch reads the staged units over PCIe behind them
 * and packs them — the runtime never registers the caller's pages (a registered range that is freed takes every queue
 * of the process off the device for ~20 ms).  A unit (2 048 points) whose covariances are all bitwise symmetric — what
 * the reference produces — crosses the link as six doubles per covariance and is mirrored on the device (72 instead of
 * 96 bytes per point; one asymmetric covariance and its unit travels whole; the resident scan is the caller's bit for
 * bit either way): a 100 000-point scan reaches the device in 0.15 - 0.17 ms.  Page-locked buffers (vgicp_host_register) are read in place.  VGICP_OPTION_UPLOAD_STAGE_KB sets
 * the size up to which scans are staged (default 512 MB; 0 = hand every scan to the runtime in place).
 * With a communicator (below) every rank passes ITS shard of the scan and all ranks return the same
 * pose. */
int vgicp_align(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                const double guess[16], const vgicp_params* params, double out_pose[16],
                vgicp_stats* stats);
/* The same in two steps, for callers that keep a scan resident across several aligns (and for
 * timing the path with its inputs already in HBM). */
int vgicp_scan_upload(vgicp_ctx* ctx, size_t n, const double* points, const double* covs);
int vgicp_align_resident(vgicp_ctx* ctx, const double guess[16], const vgicp_params* params,
                         double out_pose[16], vgicp_stats* stats);

/* Optional, for callers that keep their clouds in buffers which live across frames (a pool): page-lock such a buffer
 * once and the copy engine reads it in place, no CPU copy at all (9.6 MB in 0.19 ms).  The reference allocates a fresh
 * cloud per frame (src/Registration.cpp:11), for which there is nothing to register: that case is what the staged
 * upload inside vgicp_align is for, and the resident chain (vgicp_scan_prepare*) avoids the upload of the covariances
 * altogether.  Unregister before the buffer is freed. */
int vgicp_host_register(vgicp_ctx* ctx, const void* buffer, size_t bytes);
int vgicp_host_unregister(vgicp_ctx* ctx, const void* buffer);

/* ---- single-step hooks (API parity + tests) -------------------------------------------------
 * One iteration's normal equations at a given total pose, no solve: what the accumulation loop of
 * ICP::computeTransform (src/Registration.cpp:60-76) leaves in JTJ / JTr, plus the match count.
 * JTJ is written as a full column-major 6x6 (mirrored from its lower triangle). Local rank only. */
int vgicp_accumulate(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                     const double pose[16], double JTJ[36], double JTr[6], uint64_t* count);
/* The tail of one round on given normal equations, run on the device by the same code the loop kernels
 * inline: se3 = JTJ.ldlt().solve(-JTr) (src/Registration.cpp:78), step = Utils::se3ToSE3(se3)
 * (src/Registration.cpp:79, src/Utils.cpp:40-63; column-major 4x4) and ICP::convergenceCheck(step)
 * (src/Registration.cpp:37-50).  JTJ is a column-major 6x6 of which only the lower triangle is read (as
 * Eigen's LDLT does).  The device first tries an unpivoted LDL^T that is valid for safely positive
 * definite systems and otherwise runs the pivoted, Eigen-faithful one (pseudo-inverted D);
 * VGICP_SOLVE_FORCE_PIVOTED selects the latter unconditionally.  used_pivoted / converged are optional. */
#define VGICP_SOLVE_FORCE_PIVOTED 1u
int vgicp_solve_step(vgicp_ctx* ctx, const double JTJ[36], const double JTr[6], double cosine_threshold,
                     double translation_sq_threshold, uint32_t flags, double se3[6], double step[16],
                     int32_t* used_pivoted, int32_t* converged);
/* LocalMap::correspondenceMatching (src/LocalMap.cpp:78-112) on points/covs given in the MAP frame:
 * materialises (srcPoints, srcCovs, mapPoints, mapCovs) in ascending point order. Output arrays hold
 * n entries each; src_index (optional) receives the matched point indices. */
int vgicp_match(vgicp_ctx* ctx, size_t n, const double* points, const double* covs,
                double* src_points, double* src_covs, double* map_points, double* map_covs,
                uint64_t* src_index, size_t* matched);
/* LocalMap::getVoxelIndex (src/LocalMap.cpp:114-118) evaluated on the device. */
int vgicp_voxel_index(vgicp_ctx* ctx, size_t n, const double* points, int32_t* keys);

/* ---- scan preparation -----------------------------------------------------
 * CloudPreprocessor::voxelDownsampleAndEstimateCovariances (src/CloudPreprocessor.cpp:76-127): keep the
 * first point of every voxel of size voxel_size, and give each kept point the covariance of its knn
 * nearest neighbours in the WHOLE scan (the point itself included; Open3D's cumulant estimate),
 * regularised to svd.matrixU() diag(1, 1, 1e-2) svd.matrixV()^T with Eigen's JacobiSVD in its published operation
 * order.  For a symmetric matrix that is sum_k f_k sign(eigenvalue_k) q_k q_k^T ordered by |eigenvalue|: the
 * cumulant estimate E[xx^T] - E[x]E[x]^T of an exactly planar, collinear or repeated neighbourhood far from the
 * origin has a rounding-level smallest eigenvalue of either sign, and the reference — and therefore this call —
 * then returns an INDEFINITE matrix (-1e-2 on the normal).  vgicp_get_counter(VGICP_COUNTER_PREP_INDEFINITE) says
 * how many kept points of the last call were affected, so a caller can see it (the registration's Mahalanobis
 * weight of such a point is still finite: the voxel's covariance is added before the inverse).
 * knn is KDTreeSearchParamKNN's (30 in the reference), at most 32 here.  The neighbour search is exact (a Morton-ordered multi-level cell grid searched until the
 * k-th distance is certified), ties broken by the lower point index.
 * out_points (capacity x 3), out_covs (capacity x 9, column-major) and out_index (capacity, optional:
 * the kept points' indices in the input) are written in ascending input order — the reference emits
 * them in unordered_map iteration order, which callers must not rely on.  *kept receives the number
 * of voxels occupied; if it exceeds capacity nothing is written and VGICP_ERR_BAD_ARGUMENT is
 * returned with *kept set, so the caller can retry (capacity = n always suffices).
 * The search grid spans +-2^17 voxel_size per axis (39 km at 0.3 m); a scan with a finite coordinate beyond
 * it is refused with VGICP_ERR_BAD_ARGUMENT (*kept = 0, nothing written).  Non-finite coordinates are not
 * detected: such a point never enters a neighbourhood and its own covariance is garbage, as in the
 * reference's KD-tree. */
int vgicp_preprocess(vgicp_ctx* ctx, size_t n, const double* points, double voxel_size, int knn,
                     size_t capacity, double* out_points, double* out_covs, uint64_t* out_index,
                     size_t* kept);

/* CloudPreprocessor::deskew (src/CloudPreprocessor.cpp:25-74, Utils::interpolateSE3 / transformPoints
 * src/Utils.cpp:13-20,65-75), SURVEY.md 8(f) N4: every point taken before an IMU state's timestamp is moved
 * by (pose at the last point's time)^-1 * (that state's pose), in place.  point_time: n capture times
 * (the reference's LidarMeasurement::pointTime); states: num_states x 8 doubles = timestamp, position xyz,
 * attitude quaternion in Eigen's coefficient order x y z w, ascending in time (what ErrorStateKF::getStates
 * returns, reference Types.hpp:31-36).  The walk over point_time is the reference's sequential one, so the
 * points after the last state at or before the end of the sweep stay as they are.  *transformed receives
 * the number of leading points moved, or -1 with the points untouched where the reference would leave its
 * state queue (no state at or before the last point's time, or none after it). */
int vgicp_deskew(vgicp_ctx* ctx, size_t n, double* points, const double* point_time, size_t num_states,
                 const double* states, int64_t* transformed);

/* CloudPreprocessor::process (src/CloudPreprocessor.cpp:8-23) in one call with the result LEFT ON THE
 * DEVICE as the resident scan: extrinsic (Open3D Transform of the points by the column-major 4x4, NULL =
 * none), deskew (skipped when num_states is 0, like process() with an empty state queue), down-sampling
 * and covariances.  vgicp_align_resident / vgicp_map_insert_resident then work on it without the scan
 * ever returning to the host: per frame one upload of 32 bytes per raw point and the pose back — the
 * frame sequence of src/Odometry.cpp:73-87.  *kept receives the size of the prepared scan, *deskewed
 * (optional) what vgicp_deskew reports.  Unlike vgicp_deskew, states that do not bracket the end of the
 * sweep are an error here (VGICP_ERR_BAD_ARGUMENT, *deskewed = -1, no scan resident).  Not available on
 * a communicator (the resident scan of a rank is a shard there). */
int vgicp_scan_prepare(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time,
                       size_t num_states, const double* states, const double extrinsic[16],
                       double voxel_size, int knn, size_t* kept, int64_t* deskewed);
/* The frame sequence of src/Odometry.cpp:73-87 with ONE host synchronisation per frame:
 *   vgicp_scan_prepare_async      CloudPreprocessor::process (:73-75): as vgicp_scan_prepare, but only ENQUEUED — the
 *                                 tables and grids of the preparation are sized from n, the number of kept points
 *                                 stays on the device, nothing is waited for and nothing is returned;
 *   vgicp_align_resident          ICP::align (:79, through ErrorStateKF::update): the single launch reads the scan's
 *                                 size from the device; its synchronisation is the frame's only one and also
 *                                 brings back what the preparation found (a refused scan fails this call) and
 *                                 the counts of the previous frame's map insertion;
 *   vgicp_map_insert_resident_async   LocalMap::updateLocalMap (:86): enqueued, counts read at the next
 *                                 synchronisation (a failure of the insertion is reported by the next call that
 *                                 synchronises: vgicp_align_resident, vgicp_map_size, ...).
 * vgicp_scan_info returns what the last preparation found (it synchronises if that is still pending). Any other
 * entry point first brings a pending preparation / insertion up to date.
 * Lifetime of the caller's buffers: a sweep of up to 16 MB (points + capture times; VGICP_STAGE_LIMIT bytes in the
 * environment) is copied into page-locked staging memory of the context before the call returns — the caller's
 * buffers are free again at once, and the runtime never registers the caller's pages.  (A registered range that the
 * caller frees takes every queue of the process off the device for ~20 ms: the fate of a caller that allocates and
 * frees its clouds per frame, as the reference does.  For the same reason every synchronous entry point of this
 * header moves buffers of 0.5 - 16 MB through a page-locked arena of the context, and vgicp_align stages its scan;
 * page-locked buffers, vgicp_host_register, go directly.)  A LARGER
 * sweep is handed to the runtime in place and the call waits for those copies (not for the kernels behind them), so
 * `points` / `point_time` are free again on return whatever the size.  `states` and `extrinsic` are always copied
 * before the call returns.  At most 16 000 IMU states may fall inside one sweep (VGICP_ERR_BAD_ARGUMENT beyond; vgicp_deskew
 * has no such limit). */
int vgicp_scan_prepare_async(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time,
                             size_t num_states, const double* states, const double extrinsic[16],
                             double voxel_size, int knn);
int vgicp_scan_info(vgicp_ctx* ctx, size_t* kept, int64_t* deskewed, uint64_t* indefinite);
/* A sweep handed over WHEN IT ARRIVES: the reference's lidar callback (include/ESKF_LIO/Subscriber.hpp:80-103) builds
 * the cloud and its capture times long before Odometry::run pops the measurement (src/Odometry.cpp:43-48) and prepares
 * it (:74).  vgicp_sweep_stage copies the raw sweep (points n x 3, point_time n or NULL) into page-locked memory of the
 * context with the CPU and returns a ticket; vgicp_scan_prepare_staged_async is vgicp_scan_prepare_async for that
 * sweep: the device reads the staged bytes where they lie, the frame's first stage no longer waits for a host copy.
 * vgicp_sweep_stage / _cloud2 / vgicp_sweep_unstage are the entry points that another thread may call while the
 * context's owner thread is inside a call (they have a mutex of their own); the buffers are free again on return.
 * They launch nothing and copy with the CPU, but they are not free of runtime calls: a slot that has to grow is
 * re-allocated (hipSetDevice + hipHostFree + hipHostMalloc — hipHostFree may wait for work in flight on the device:
 * once per slot and sweep size, three slots), and a slot whose last preparation may still be reading it is asked for
 * with hipEventQuery (outside the mutex).  At most three sweeps can be staged ahead (VGICP_ERR_NOT_READY beyond); a
 * ticket is used once — by vgicp_scan_prepare_staged_async, or by vgicp_sweep_unstage for a sweep that is dropped
 * unprepared (a measurement the caller discards: src/Odometry.cpp:43-48 may pop several and keep one), which frees its
 * slot; a ticket that is never used either way keeps its slot for the life of the context. */
int vgicp_sweep_stage(vgicp_ctx* ctx, size_t n, const double* points, const double* point_time, uint64_t* ticket);
int vgicp_sweep_unstage(vgicp_ctx* ctx, uint64_t ticket);
int vgicp_scan_prepare_staged_async(vgicp_ctx* ctx, uint64_t ticket, size_t num_states, const double* states,
                                    const double extrinsic[16], double voxel_size, int knn);
/* The same for the sensor's WIRE format: the payload of a sensor_msgs/PointCloud2 as it arrives (little-endian), n =
 * height x width records of point_step bytes (a multiple of 4, 12 .. 64), float32 x / y / z at byte offsets off_x /
 * off_y / off_z and a float64 capture time at off_time (SIZE_MAX: none) -- the four fields the reference's callback
 * reads one point at a time and widens on the host (include/ESKF_LIO/Subscriber.hpp:89-97).  Here the records are
 * copied as they are and the DEVICE picks the floats out and widens them (float -> double is exact: the prepared scan
 * is bit-identical to the one made from the widened cloud); the host's loop over the points is gone.  Prepared with
 * vgicp_scan_prepare_staged_async like any staged sweep. */
int vgicp_sweep_stage_cloud2(vgicp_ctx* ctx, size_t n, const void* data, size_t point_step, size_t off_x, size_t off_y,
                             size_t off_z, size_t off_time, uint64_t* ticket);
int vgicp_map_insert_resident_async(vgicp_ctx* ctx, const double transform[16], size_t max_points_per_voxel);
/* The host copy CloudPreprocessor::process leaves behind (src/CloudPreprocessor.cpp:8-23 ends with the prepared scan in
 * the caller's cloud) for a preparation that was only enqueued, without a copy command and without waiting twice:
 *   vgicp_scan_fetch_begin   enqueues one kernel behind the pending preparation that writes the prepared scan into
 *                            page-locked memory of the context piece by piece, and returns as soon as the down-sampling
 *                            has reported how many points it keeps (*kept) — while the neighbour search and the
 *                            covariances are still running: the caller sizes its vectors in that time;
 *   vgicp_scan_fetch_end     copies the pieces into points (kept x 3) / covs (kept x 9, column-major) as they arrive and
 *                            brings the context up to date as vgicp_scan_info does (a refused scan fails this call).
 * Without a pending preparation (or on a multi-device context) the pair is vgicp_scan_info + vgicp_scan_download. */
int vgicp_scan_fetch_begin(vgicp_ctx* ctx, size_t* kept);
int vgicp_scan_fetch_end(vgicp_ctx* ctx, size_t capacity, double* points, double* covs, size_t* n);
/* Checksums of what the last vgicp_scan_fetch_end delivered, made BY THE DEVICE while it wrote the pieces — for a caller
 * that wants a fingerprint of the two arrays without reading 96 bytes per point again (the shim's "is this host cloud
 * still the resident scan" stamp).  The 8-byte words of each array are dealt to 16 lanes by their index (lane = index
 * mod 16, k = index / 16); sums[32 a + lane] = the sum of the lane's words, sums[32 a + 16 + lane] = the sum of
 * (m_lane - k) x word with m_lane the lane's word count, both mod 2^64; a = 0 the points, 1 the covariances.  That is
 * what the loop  s1 += w; s2 += s1  over a lane's words ends with, up to its start values.  VGICP_ERR_NOT_READY when
 * the last host copy did not come through the fetch kernel (nothing pending, a multi-device context). */
int vgicp_scan_fetch_sums(vgicp_ctx* ctx, uint64_t sums[64]);

/* What the calls of THIS HOST THREAD into the module (whatever the context) have cost the host since this context's
 * counters were last reset (reset != 0 resets them):
 * kernels launched (library sorts counted by their launch formula), copies / memsets enqueued, host
 * synchronisations; and, with VGICP_OPTION_STAGE_EVENTS on (or VGICP_STAGE_EVENTS=1 in the environment), the device
 * spans (HIP events on the module's stream, microseconds; -1 when not recorded) of the LAST preparation (upload of
 * the raw sweep included), the last single-launch align and the last deferred map insertion — the three stage
 * timers of src/Odometry.cpp:73-87. */
typedef struct vgicp_frame_stats {
  uint64_t kernel_launches;
  uint64_t copies;
  uint64_t host_syncs;
  double prepare_us;       /* upload of the raw sweep + extrinsic + deskew + down-sampling + 30-NN + covariances */
  double align_us;
  double insert_us;
  double prepare_head_us;  /* the first part of prepare_us: upload + extrinsic + deskew (+ Morton codes) */
} vgicp_frame_stats;
int vgicp_get_frame_stats(vgicp_ctx* ctx, vgicp_frame_stats* out, int reset);
#define VGICP_OPTION_STAGE_EVENTS 1
/* value = KiB: scans of up to this size handed to vgicp_align / vgicp_scan_upload are staged through page-locked
 * memory of the context by the copy threads (see vgicp_align; default 524288 = 512 MB;
 * VGICP_UPLOAD_STAGE_LIMIT=bytes in the environment sets the default).  0: every scan is handed to the runtime in
 * place, which registers the caller's pages with the driver — for callers that never free those buffers. */
#define VGICP_OPTION_UPLOAD_STAGE_KB 2
/* value != 0: every scan preparation (vgicp_preprocess, vgicp_scan_prepare*) emits its kept points in the sequence the
 * REFERENCE emits them — the iteration order of its std::unordered_map<Eigen::Vector3i, int, hash_eigen> filled in scan
 * order (src/CloudPreprocessor.cpp:85-99; libstdc++'s node order, i.e. the order on the reference's platform) — instead
 * of ascending input index.  The kept set and every covariance are the same; what changes is what an order-dependent
 * consumer makes of them: Voxel::addPoint's running mean (include/ESKF_LIO/LocalMap.hpp:79-87) and the order the
 * registration's sums are taken in.  With it a frame chain on this module reproduces the reference's chain, not only
 * its per-call results.  A parity mode: the container is replayed on the host (~1.5 ms per 27 000
 * kept points) and an enqueued preparation is waited for; off by default. */
#define VGICP_OPTION_REFERENCE_ORDER 3
int vgicp_set_option(vgicp_ctx* ctx, int option, int value);

/* Copies the resident scan (vgicp_scan_upload / vgicp_scan_prepare) to the host: points n x 3, covs n x 9
 * column-major, e.g. for a host-side map or for saving. capacity in points; with both pointers NULL only
 * *n is written (size query). */
int vgicp_scan_download(vgicp_ctx* ctx, size_t capacity, double* points, double* covs, size_t* n);

/* ---- multi-GPU: one process per GPU, RCCL all-reduce of the normal equations ----------------
 * Replaces the thread merge of ICP::computeTransform (src/Registration.cpp:71-75) across devices:
 * per iteration one all-reduce (sum) of 28 doubles (21 + 6 + match count) over xGMI.  Rank 0 calls
 * vgicp_comm_unique_id, the host side ships the 128 bytes to every rank (any transport), then every
 * rank calls vgicp_comm_init.  The voxel map is replicated: every rank applies the same
 * upsert/erase batches. */
#define VGICP_UNIQUE_ID_BYTES 128
int vgicp_comm_unique_id(vgicp_ctx* ctx, void* id128);
/* Besides the RCCL communicator, vgicp_comm_init sets up the device-initiated exchange below by itself (the
 * mailbox handles travel through one RCCL all-gather); where that is not possible, or with
 * VGICP_PEER_EXCHANGE=0 in the environment, the communicator uses one RCCL all-reduce per iteration. */
int vgicp_comm_init(vgicp_ctx* ctx, int world_size, int rank, const void* id128);
int vgicp_comm_destroy(vgicp_ctx* ctx);

/* ---- multi-GPU: device-initiated exchange over xGMI --------------------------------------------
 * The same merge (src/Registration.cpp:71-75) without a host-enqueued collective between kernel launches:
 * every rank owns a small mailbox in fine-grained device memory, maps the mailboxes of all ranks (HIP IPC),
 * and the ONE persistent kernel launch that runs the whole ICP::align loop on each GPU stores the rank's
 * 28-double row of every iteration straight into all mailboxes and adds the rows it receives in one fixed order
 * (identical bits on every rank, hence the same pose and the same break decision everywhere).
 * Hand-wiring for hosts that do not use vgicp_comm_init: every rank exports its handle, the host side ships
 * the world_size x 64 bytes to every rank in rank order (any transport), every rank connects, and the host
 * side runs a barrier of its own before the first align (a mailbox must be initialised before a peer's kernel
 * writes into it).  At most 16 ranks.  If a launch ever gives up waiting for a peer, the align is re-run
 * through the RCCL communicator when there is one (and the communicator stays on it), else it fails with
 * VGICP_ERR_RCCL. */
#define VGICP_PEER_HANDLE_BYTES 64
int vgicp_peer_export(vgicp_ctx* ctx, void* handle64);
int vgicp_peer_connect(vgicp_ctx* ctx, int world_size, int rank, const void* handles);
int vgicp_peer_disconnect(vgicp_ctx* ctx);
/* "" while the device-initiated exchange carries the per-iteration merge (or there is nothing to merge: one device);
 * otherwise why it does not — a peer mapping that was refused, VGICP_PEER_EXCHANGE=0, a launch that gave up waiting
 * for a peer — and what is used instead.  For a multi-device context and for a communicator alike.  Never NULL. */
const char* vgicp_peer_status(const vgicp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VGICP_HIP_H_ */
