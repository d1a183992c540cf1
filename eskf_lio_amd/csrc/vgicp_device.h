// vgicp_device.h — device-side data layout and kernel launch entry points (gfx950 only).
//
// HBM layout (DESIGN.md "Data layout"):
//   scan      12 SoA planes of `stride` doubles: x y z c00 c10 c20 c01 c11 c21 c02 c12 c22
//             (a point costs 96 B per iteration, every load is a coalesced 512-B wave access)
//   table     open-addressing hash table of 128-byte voxel records, power-of-two slot count,
//             linear probing, load <= 1/4: a lookup is one cache line, a hit needs no second hop
//   rows      two buffers of one 256-byte row of 32 doubles per workgroup (21 JTJ + 6 JTr + count +
//             4 pad): launch j writes buffer j&1 and folds buffer (j+1)&1 in its prologue
//   state     two AlignStates (ping-pong like the rows): total pose, thresholds, iteration counter,
//             done/converged flags
//   log       per-iteration 32-double rows (reduced normal equations + count), read back once
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgicp_math.h"

namespace vgicp {

constexpr int kSlots = 32;      // doubles per partial row
constexpr int kNormalEq = 27;   // 21 lower-triangle JTJ entries + 6 JTr entries
constexpr int kCountSlot = 27;  // match count travels as an exact double
constexpr int kScanPlanes = 12;
constexpr int kMaxIterBlocks = 512;  // grid cap (every workgroup folds all rows: keep them few); larger scans grid-stride

enum : int32_t { SLOT_EMPTY = 0, SLOT_FULL = 1, SLOT_TOMB = 2, SLOT_LOCKED = 3 };

// One voxel of the reference's LocalMap::Voxel as the path reads it: key, mean, covariance
// (reference include/ESKF_LIO/LocalMap.hpp:63-70; numPoints / points stay on the host).
struct alignas(128) VoxelRecord {
  int32_t key[3];
  int32_t state;
  double mean[3];
  double cov[9];  // column-major
  uint64_t count;     // numPoints of the reference's Voxel (device-side insertion freezes it at the cap)
  uint64_t reserved;
};
static_assert(sizeof(VoxelRecord) == 128, "one voxel = one 128-byte line");

struct AlignState {
  double pose[12];  // total transform: R column-major (9) then t (3)
  double step[12];  // last increment
  double cosine_threshold;
  double translation_sq_threshold;
  int32_t max_iteration;
  int32_t iteration;  // rounds executed so far
  int32_t done;       // set on convergence or when iteration == max_iteration
  int32_t converged;
  uint32_t seq;       // persistent launch: echo of PersistArgs::seq, written last (0 from the per-launch loop)
  uint32_t pad;
  // persistent launch only (the host zeroes both before the launch)
  uint32_t abort_seq; // = PersistArgs::seq when ANY workgroup gave up waiting (whatever workgroup 0 concluded)
  uint32_t outcome;   // workgroup 0's verdict: kOutcome*
};
// AlignState::outcome of a persistent launch
constexpr uint32_t kOutcomeNone = 0;         // workgroup 0 gave up inside a round (or never ran)
constexpr uint32_t kOutcomeCommitted = 1;    // the loop ran to its end (several GPUs: on every rank)
constexpr uint32_t kOutcomeAgreedAbort = 2;  // several GPUs: some rank gave up, every rank knows it
constexpr uint32_t kOutcomeNoAgreement = 3;  // several GPUs: a peer's verdict never arrived

struct IterArgs {
  const double* scan;  // SoA planes
  uint64_t stride;
  uint32_t n;
  uint32_t mask;  // slots - 1
  const VoxelRecord* table;
  double voxel_size;
  double* rows;        // [grid][kSlots]: partial rows this launch writes
  const double* prev;  // rows the previous launch wrote (or the all-reduced single row)
  uint32_t prev_rows;  // how many; 0 = first round, nothing to solve yet
  uint32_t pad;
  const AlignState* state_in;
  AlignState* state_out;
  double* log;        // [max_iteration][kSlots]
  uint64_t* stamps;   // diagnostic aid (VGICP_DEBUG_STAMPS=1): phase times of workgroup 0 in 10 ns
                      // ticks, summed over launches; nullptr in normal operation
  // round 6: what the persistent launch keeps on chip, kept in HBM between the launches of one align
  int4* memo;               // [n] per point: voxel key of the last round + its record's slot (index in `dense` when that
                            // is set) / kMemoMiss; written by the launch that finds a point's key changed (all points in
                            // an align's first launch), nullptr = look every point up in every launch
  uint32_t memo_valid;      // != 0: an earlier launch of THIS align filled the memos
  uint32_t scan_seq;        // with asym_dev: *asym_dev != scan_seq = every covariance of the scan is bitwise symmetric
  const uint32_t* asym_dev; // (nine planes are read instead of twelve); nullptr = unknown, read all
  const VoxelRecord* dense; // dense copy of the FULL records (tables beyond the caches' reach), or nullptr
};

// Arguments of the persistent single-launch align (single GPU): every round of the loop runs inside
// one kernel.  Workgroups exchange their partial rows through `rows` and `parts` (three buffers each, by
// round modulo 3) with no flag and no counter: every 8-byte word of a buffer holds the bit pattern
// kRowUnset until its producer stores the value (one write-through 8-byte store per word, never torn), and
// consumers poll the words themselves.  Two levels, in the summation order of iterate_kernel<512>:
//   rows   [3][256][kSlots]      workgroup b's sums of the round (21 + 6 + count) at row (b % 16) * 16 + b / 16
//   parts  [3][kFolders][kSlots] folder g (= workgroup g < kFolders) adds the rows b = g, g + 16, g + 32 ...
//                                in ascending order; every workgroup then adds the parts in ascending g
// A producer re-arms its word (stores kRowUnset) one round after everyone has consumed it and two rounds
// before it is written again; DESIGN.md section 4 has the ordering argument.
constexpr int kExchangeRows = 256;                          // 16 folders x 16 rows: the largest persistent grid
constexpr int kMaxRanks = 16;                               // mailbox rows; = kFolders so one poll routine serves both
constexpr int kFolders = 16;                                // = 512 / kSlots, the groups of iterate_kernel<512>'s fold
constexpr unsigned long long kRowUnset = ~0ull;             // a NaN pattern no fp64 operation produces
constexpr unsigned long long kRowNaN = 0x7FF8000000000000ull;  // what a computed NaN is published as
constexpr size_t kMailRowWords = 3 * (size_t)kMaxRanks * kSlots;  // the rows of a mailbox
constexpr size_t kMailWords = kMailRowWords + kSlots;          // + the verdict words (kMaxRanks used)
struct PersistArgs {
  const double* scan;  // SoA planes
  uint64_t stride;
  uint32_t n;          // points of the scan — an UPPER BOUND when n_dev is set (the launch plan is made from it)
  uint32_t mask;
  const VoxelRecord* table;
  double voxel_size;
  double* rows;         // [3][kExchangeRows][kSlots]; kRowUnset where a workgroup publishes (between launches: all but the
                        // buffer of the last round, which the next launch's first round re-arms), else +0.0
  double* parts;        // [3][kFolders][kSlots], likewise
  AlignState* state;    // out: final state; state->seq == seq tells the host the loop ran to its end
  double* log;          // [max_iteration][kSlots]
  uint32_t spin_limit;  // an in-kernel wait longer than this many polls gives up (host falls back to launches)
  uint32_t seq;
  // inputs travel with the dispatch packet: no host-to-device copy and no memset per align
  double pose0[12];     // guess: R column-major (9) then t (3)
  double cosine_threshold;
  double translation_sq_threshold;
  int32_t max_iteration;
  uint32_t stash_points;   // at most this many extra points per thread are kept in LDS across rounds (scans larger
                           // than the grid); the kernel parks as many of them as fit stash_bytes
  uint32_t memo_points;    // extra points per thread whose last key + table slot are remembered in LDS
  uint32_t round0;         // rounds executed on this context before this launch (mod 3 matters): the exchange
                           // buffers rotate with round0 + it, so a launch leaves nothing to tidy up
  // multi-GPU (world > 1): the rank totals travel through peer-mapped mailboxes, written by the ranks'
  // kernels themselves over xGMI (no host-enqueued collective between launches)
  uint32_t world, rank;
  uint32_t mail_round0;    // rounds all ranks have executed on this communicator before this launch
  uint32_t mail_seq;       // aligns attempted through the mailboxes on this communicator, this one included (same on
                           // every rank, >= 1): the verdict words at the end of a launch carry it
  double* const* mail;     // device array of kMaxRanks pointers; mail[r]: rank r's mailbox as mapped into THIS
                           // process, [3][kMaxRanks][kSlots] words: by round % 3, row = sender rank (rows >=
                           // world hold +0.0 for good), then kMaxRanks VERDICT words (index = sender rank):
                           // mail_seq << 1 | 1 "my loop ran to its end", mail_seq << 1 "I gave up" — written by
                           // workgroup 0 of the sender at the end of its launch, so that all ranks commit an align
                           // or none does
  const uint32_t* asym_dev;  // nullptr, or the word pack_scan_kernel sets to scan_seq when some covariance of the scan is
  uint32_t scan_seq;         // NOT bitwise symmetric; while it differs, the planes above the diagonal are not read
  uint32_t stash_bytes;      // LDS behind the memos that may hold parked points; how many fit depends on the planes
                             // a parked point needs (9 or 12) and on the point-carrying threads (448 or 512)
  const VoxelRecord* dense;  // nullptr, or the FULL records of the table packed densely (launch_table_dense), for tables
                             // far larger than the caches' and TLBs' reach: a record's index there sits in the upper
                             // half of its spare word; a point that stays in its voxel then reads its payload from
                             // this array (1.3 GB at 10M voxels) instead of from the sparse table (8.6 GB)
  const uint32_t* n_dev;   // nullptr, or where the device holds the scan's size (a scan prepared on the device whose
                           // kept count the host has not read yet: no host round trip between preparation and align)
  double prefetch_margin;  // > 0 (only with memo_points == stash_points == 0): a point closer than this many
                           // voxel sizes to a face of its voxel has the neighbour behind that face looked up
                           // into LDS while the workers wait for the exchange
  uint64_t* stamps;
};

// ---- launchers (defined in vgicp_kernels.hip) ----
// The whole ICP::align loop in one launch (512-thread workgroups, at most one per CU).
hipError_t launch_persistent(hipStream_t s, const PersistArgs& args, uint32_t grid);
// How launch_persistent splits the CU's LDS for a scan of n points on `grid` workgroups: points per
// thread beyond the first that get a memo (last key + slot, 16 B) and that are parked whole (96 B).
void persistent_lds_plan(uint32_t n, uint32_t grid, uint32_t* memo_points, uint32_t* stash_points, uint32_t* stash_bytes,
                         uint32_t lds_budget = 0);
// Sizes (8-byte words) of the rows / parts exchange buffers, and their content between launches.
size_t persistent_rows_words();
size_t persistent_parts_words();
void persistent_exchange_image(uint32_t grid, unsigned long long* rows_words, unsigned long long* parts_words);
// Whether one 512-thread workgroup of the persistent kernel with this much dynamic LDS fits a CU of the
// current device: *max_grid = cu_count then, else 0 (the in-kernel exchange needs every workgroup resident).
hipError_t persistent_prepare_device();  // once per context, on its device: every instantiation may use the whole CU's LDS
hipError_t persistent_max_resident(uint32_t dyn_lds_bytes, int cu_count, uint32_t* max_grid);
uint32_t persistent_dyn_lds_bytes(uint32_t memo_points, uint32_t stash_bytes);
uint32_t persistent_max_dyn_lds_bytes();  // the most a launch plan ever asks for
// One VGICP round over the resident scan: prologue folds args.prev and advances the pose, body
// accumulates this round's rows. block = 256 / 512 / 1024 threads per workgroup.
hipError_t launch_iterate(hipStream_t s, const IterArgs& args, uint32_t grid, int block);
// The prologue-only launch that closes the last round (single workgroup).
hipError_t launch_close(hipStream_t s, const IterArgs& args, int block);
// Multi-GPU: fold nrows rows into sums[kSlots] (the 256-byte message of the all-reduce).
hipError_t launch_fold_rows(hipStream_t s, const double* rows, uint32_t nrows, const AlignState* state,
                            double* sums);
// Test hook: solve + exponential + convergence test of one round on given normal equations (one wave).
hipError_t launch_solve_step(hipStream_t s, const double* packed27, double cosine_threshold,
                             double translation_sq_threshold, int force_pivoted, double* out20);
// out[o] = in[perm[o]], o < m, for a scan's AoS arrays (idx / out_idx optional): VGICP_OPTION_REFERENCE_ORDER
hipError_t launch_gather_scan(hipStream_t s, const uint32_t* perm, uint32_t m, const double* pts, const double* covs,
                              const unsigned long long* idx, double* out_pts, double* out_covs, unsigned long long* out_idx);
// asym: one device word that receives seq when a covariance is not bitwise symmetric (PersistArgs::asym_dev)
hipError_t launch_pack_scan(hipStream_t s, const double* points_aos, const double* covs_aos,
                            uint32_t n, double* soa, uint64_t stride, uint32_t* asym, uint32_t seq);
// The same packing straight out of page-locked HOST staging memory that host threads are still filling, unit by unit
// (pack_arena_unit() points each; flags[16 * u] == seq publishes unit u, flags[16 * u + 1] says in which form its
// covariances were staged; wait == false: all there).  Also leaves the AoS copy on the device.  See pack_arena_kernel.
uint32_t pack_arena_unit();
constexpr uint32_t kArenaFull = 2u;      // a unit's covariances as the caller holds them: 72 bytes per point
constexpr uint32_t kArenaCompact = 1u;   // all of them bitwise symmetric: c00 c10 c20 c11 c21 c22, 48 bytes per point
hipError_t launch_pack_arena(hipStream_t s, const void* arena_points, const void* arena_covs, uint32_t n,
                             const uint32_t* flags, bool wait, uint32_t seq, uint32_t spin_limit, double* aos_pts, double* aos_cov,
                             double* soa, uint64_t stride, uint32_t* asym);
hipError_t launch_table_clear(hipStream_t s, VoxelRecord* table, uint64_t slots);
// Pack the FULL records of `table` into `dense` in slot order (128 bytes each) and leave every record's index there in
// the upper half of its spare word.  block_counts: scratch of table_dense_blocks(slots) + 1 words.
uint32_t table_dense_blocks(uint64_t slots);
hipError_t launch_table_dense(hipStream_t s, VoxelRecord* table, uint64_t slots, VoxelRecord* dense, uint64_t dense_capacity,
                              uint32_t* block_counts);
// claimed: scratch of n words (upsert) / old_slots words (rehash): claim launch -> write launch
hipError_t launch_upsert(hipStream_t s, VoxelRecord* table, uint32_t mask, uint32_t n,
                         const int32_t* keys, const double* means, const double* covs,
                         uint32_t* counters /* [0] new inserts, [1] failures */, uint32_t* claimed);
hipError_t launch_erase(hipStream_t s, VoxelRecord* table, uint32_t mask, uint32_t n,
                        const int32_t* keys, uint32_t* counters /* [0] erased */);
hipError_t launch_rehash(hipStream_t s, const VoxelRecord* old_table, uint64_t old_slots,
                         VoxelRecord* table, uint32_t mask, uint32_t* counters, uint32_t* claimed);
hipError_t launch_voxel_index(hipStream_t s, const double* points_aos, uint32_t n, double voxel_size,
                              int32_t* keys);
// Correspondence materialisation in ascending point order (three small passes).
hipError_t launch_match(hipStream_t s, const double* points_aos, const double* covs_aos, uint32_t n,
                        const VoxelRecord* table, uint32_t mask, double voxel_size,
                        uint32_t* block_counts, uint32_t* total, double* src_points,
                        double* src_covs, double* map_points, double* map_covs, uint64_t* src_index);
uint32_t match_blocks(uint32_t n);

// ---- vgicp_mapupdate.hip: LocalMap::updateLocalMap's insert / evict loops on the device ----
// Scratch the insertion needs for n points (bytes), and the insertion itself: transform, find-or-claim
// the voxel of every point, order the points of a voxel by scan index, apply the reference's
// constructor / addPoint rule sequentially per voxel. counters[0] receives the number of new voxels.
size_t map_insert_scratch_bytes(uint32_t n);

// vgicp_preprocess.hip — voxel down-sampling + k-NN covariances (CloudPreprocessor.cpp:76-127)
// Words of the context's counter block (device; a copy travels to pinned host memory after a preparation):
//   [0] kept points  [1] octree cells over all levels  [2..7] developer counts  [8..71] developer histograms
//   [72] kept points whose regularised covariance is indefinite (src/CloudPreprocessor.cpp:119-123)
//   [73] = the call's epoch when a point lies beyond the search grid (the scan is refused)
//   [74] what the deskew reports (leading points moved)   [75] = epoch when a device-wide scan gave up waiting
//   [76], [77] tile tickets of the two single-launch scans
//   [78], [79] queries of the neighbour search that start first (sparse neighbourhood) / after them
//   [80] workgroup ticket of the prologue (zero between launches: the workgroup that draws the last one clears it)
constexpr int kCounterWords = 96;
constexpr int kIndefiniteCounter = 72;
constexpr int kBeyondGrid = 73;
constexpr int kDeskewedCounter = 74;
constexpr int kScanTimeout = 75;
constexpr int kTicketA = 76;
constexpr int kTicketB = 77;
constexpr int kHeavyQueries = 78;
constexpr int kLightQueries = 79;
constexpr int kTicketP = 80;
// the prologue finds the deskew's segments itself (no launch for the bounds) for ordered state queues up to this many
// states (8 bytes of LDS each) and sweeps up to this many workgroups of 256 points (a look-back slot each)
constexpr uint32_t kFusedBoundsStatesMax = 4096;
constexpr uint32_t kFusedBoundsBlocksMax = 4096;
constexpr uint32_t kDeskewMaxStates = 4096;   // the parallel deskew bounds keep 12 bytes per state in LDS; longer (or unordered) queues take the serial walk
constexpr uint32_t kPrepareMaxStates = 16000;  // IMU states that can own points of ONE sweep in the fused preparation (LDS)
constexpr uint32_t kMaxScanTiles = 4096;  // x 2 048 points: scans up to 8 M points
size_t preprocess_scratch_bytes(uint32_t n);
int preprocess_max_knn();
uint64_t preprocess_cell_entries(uint32_t cells);
size_t preprocess_cell_bytes(uint64_t entries);
size_t preprocess_tile_bytes();  // the tile slots of the two device-wide scans (allocated once per context, zeroed)
// CloudPreprocessor::process after the upload, enqueued as ONE sequence with no host round trip in it:
// [deskew bounds] -> extrinsic + deskew + Morton codes (+ clears) -> sort -> runs / kept points / query list ->
// output slots -> octree cells -> exact k-NN -> covariances.  Everything the host does not know yet (kept points,
// cells) stays on the device: tables and grids are sized from n.  pts: n x 3 on the device, moved in place by the
// extrinsic and the deskew.  Outputs: AoS points / covariances / indices (capacity n) and, when soa != nullptr, the
// 12 SoA planes the registration reads.  counters: the context's counter block (layout above); epoch: a number
// that differs from call to call (never 0).
struct PrepareArgs {
  double* pts;
  uint32_t n;
  double voxel_size;
  int knn;
  const double* extrinsic16;     // host, column-major 4x4, or nullptr
  const double* point_time;      // device, n (deskew) or nullptr
  const double* state_time;      // device, `states` timestamps
  const double* poses;           // device, 12 doubles per state
  uint32_t states;               // 0 = no deskew
  bool ordered_states;
  bool max_hits_known;           // max_hits below is set (the host has walked the capture times)
  uint32_t max_hits;             // the largest number of states whose timestamp is not above a point's capture time (all: a NaN time)
  uint32_t* ends;                // device scratch, deskew_scratch_words(states)
  void* scratch;                 // preprocess_scratch_bytes(n)
  void* cell_table;              // preprocess_cell_bytes(preprocess_cell_entries_for(n))
  uint64_t table_entries;
  double* out_pts;
  double* out_covs;
  unsigned long long* out_idx;
  double* soa;                   // optional
  uint64_t soa_stride;
  uint32_t* counters;
  unsigned long long* host_kept; // optional, page-locked host memory as the device addresses it: receives epoch << 32 | kept
                                 // points as soon as the down-sampling knows it (vgicp_scan_fetch_begin waits for that)
  void* tiles;                   // preprocess_tile_bytes()
  uint32_t epoch;
  int debug;
  hipEvent_t ev_after_prologue;  // optional: recorded behind the prologue (extrinsic + deskew + codes)
  // The raw points may still be ARRIVING in page-locked host memory (the frame chain's upload without copy commands, as
  // pack_arena_kernel does it for vgicp_align): src_points != nullptr = n x 24 bytes of staging memory that host threads
  // fill in units of src_unit points (a multiple of 256), publishing unit u by storing src_seq into src_flags[16 * u];
  // the prologue then reads its points from there (a unit's workgroups wait for its flag, at most src_spin polls; a
  // wait that runs out is reported like a scan's: counters[kScanTimeout] = epoch) and writes them to `pts`.
  // src_flags == nullptr with src_points set: everything is staged already.
  const char* src_points;
  const uint32_t* src_flags;
  uint32_t src_seq, src_unit, src_spin;
  // src_step != 0: the staging memory holds the sensor's own records (a PointCloud2 payload, vgicp_sweep_stage_cloud2):
  // src_step bytes per point (a multiple of 4, at most 64), float32 x, y, z at byte offsets src_off[0..2]; the prologue
  // widens them (float -> double is exact).  src_step == 0: n x 3 doubles.
  uint32_t src_step, src_off[3];
};
// The prepared scan (AoS on the device: kept x 3 doubles, kept x 9 doubles; kept and the refusal flags read from the
// counter block) written by a KERNEL into page-locked host memory in pieces of piece_bytes, every piece published by
// storing seq into flags[16 * piece] — the download of vgicp_scan_fetch_end, the mirror image of pack_arena_kernel.
// Layout of `stage`: the points, padded to a multiple of 256 bytes, then the covariances.  hdr_done receives
// seq << 32 | 1 (refused / nothing prepared) or seq << 32 first thing.  sums (device, 65 words, zero on entry and on
// exit) / host_sums (page-locked, 64 words): position-weighted word sums of what was copied, see fetch_kernel.
hipError_t launch_fetch(hipStream_t s, const double* aos_pts, const double* aos_cov, const uint32_t* counters, uint32_t epoch,
                        uint32_t n_cap, char* stage, uint32_t* flags, unsigned long long* hdr_done, uint32_t seq,
                        uint32_t piece_bytes, unsigned long long* sums, unsigned long long* host_sums);
hipError_t launch_prepare(hipStream_t s, const PrepareArgs& a);        // = head + tail
// head: the kernels that read the raw sweep (deskew bounds from the times, prologue); tail: everything behind them.
// A host that stages the sweep itself launches the head, finishes staging and launches the tail, so that the device
// reads the first units while the later ones are still being copied.
hipError_t launch_prepare_head(hipStream_t s, const PrepareArgs& a);
hipError_t launch_prepare_tail(hipStream_t s, const PrepareArgs& a);
// true: the head finds the deskew's segments inside the prologue (PrepareArgs::max_hits must be known), and a workgroup
// reads the capture times of ITS points only, behind the wait for their unit: a host that stages the sweep itself may
// then stage the times unit by unit with the points instead of before the launch
bool prepare_bounds_fused(uint32_t n, uint32_t states, bool ordered_states);
uint64_t preprocess_cell_entries_for(uint32_t n);  // from the number of points alone (no host round trip)
// kernels enqueued by the launchers of this module since the counter was last reset (per host thread)
extern thread_local uint64_t g_kernel_launches;
hipError_t launch_transform_points(hipStream_t s, double* pts, uint32_t n, const double T16[16]);
// CloudPreprocessor::deskew: ends = scratch of deskew_scratch_words(states) words (the first `states` are the
// segment ends); poses = 12 doubles per state (R column-major, t); ordered_states: the host has checked that the
// state times are finite and non-decreasing (the parallel bounds; otherwise the reference's walk, state by state)
size_t deskew_scratch_words(uint32_t states);
hipError_t launch_deskew(hipStream_t s, double* pts, uint32_t n, const double* point_time, const double* state_time,
                         uint32_t states, const double* poses, uint32_t* ends, bool ordered_states);
hipError_t launch_map_insert(hipStream_t s, VoxelRecord* table, uint32_t mask, double voxel_size,
                             const double* points_aos, const double* covs_aos, uint32_t n,
                             const double pose12[12], uint64_t max_points, void* scratch,
                             size_t scratch_bytes, uint32_t* counters, bool short_lists = false);
// short_lists: no sort — every voxel's points hang on a list (the record's spare word) that the voxel's first point
// walks in scan order.  For scans that put a handful of points into a voxel at most (a scan the device down-sampled
// itself); correct for any scan, quadratic in the points of one voxel beyond eight.
// Erase every voxel whose centre is farther than `distance` from `position`; counters[0] += erased.
hipError_t launch_map_evict(hipStream_t s, VoxelRecord* table, uint64_t slots, double voxel_size,
                            const double position[3], double distance, uint32_t* counters);
// Append every FULL record to the output arrays (unordered); counters[0] = records written.
hipError_t launch_map_export(hipStream_t s, const VoxelRecord* table, uint64_t slots, uint32_t capacity,
                             int32_t* keys, double* means, double* covs, uint64_t* counts,
                             uint32_t* counters);

}  // namespace vgicp
