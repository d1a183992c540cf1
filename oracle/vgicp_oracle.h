/* vgicp_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's VGICP hot path, used as
 * the checker in tests/, in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  The
 * product (eskf_lio_amd/, include/) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (LimHaeryong/ESKF_LIO @ 2024_10_08) ships no tests, golden
 * vectors or fixtures for this path, and it cannot be compiled here (Eigen, Open3D, yaml-cpp and
 * ROS 2 are absent; no network), so the oracle is pinned only by analytic known-answer tests and by
 * an independent numpy restatement (oracle/vgicp_numpy.py); see DESIGN.md "Oracle".
 *
 * All matrices are column-major, all points are xyz triples of doubles, exactly the memory the
 * reference's std::vector<Eigen::Vector3d> / std::vector<Eigen::Matrix3d> expose through .data().
 */
#ifndef VGICP_ORACLE_H_
#define VGICP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_map oracle_map;

/* reference: LocalMap(double voxelSize, size_t maxNumPointsPerVoxel, bool) — LocalMap.hpp:54-61 */
oracle_map* oracle_map_create(double voxel_size, size_t max_points_per_voxel);
void oracle_map_destroy(oracle_map* map);
size_t oracle_map_size(const oracle_map* map);

/* The insertion loop of LocalMap::updateLocalMap (LocalMap.cpp:47-58) with Voxel's constructor and
 * Voxel::addPoint (LocalMap.hpp:72-87): points are taken as already in the world frame. */
void oracle_map_insert(oracle_map* map, size_t n, const double* points, const double* covs);
/* reference src/LocalMap.cpp:60-72,149-154: erase every voxel whose centre is farther than distance_threshold from
 * position; returns the number of voxels removed */
size_t oracle_map_evict(oracle_map* map, const double position[3], double distance_threshold);

/* Dump every voxel: keys n x 3 int32, means n x 3, covs n x 9, counts n (numPoints). Order is the
 * container's iteration order (unspecified). Returns the number of voxels written (<= capacity). */
size_t oracle_map_export(const oracle_map* map, size_t capacity, int32_t* keys, double* means,
                         double* covs, uint64_t* counts);

/* LocalMap::getVoxelIndex (LocalMap.cpp:114-118): floor(point / voxelSize) cast to int32. */
void oracle_voxel_index(double voxel_size, size_t n, const double* points, int32_t* keys);

/* LocalMap::correspondenceMatching (LocalMap.cpp:78-112), results in ascending point order.
 * Output arrays must hold n entries; returns M. src_index (optional) receives the point indices. */
size_t oracle_match(const oracle_map* map, size_t n, const double* points, const double* covs,
                    double* src_points, double* src_covs, double* map_points, double* map_covs,
                    uint64_t* src_index);

/* ICP::computeJTJAndJTr (Registration.cpp:83-102) for one correspondence; cov = srcCov + mapCov. */
void oracle_jtj_jtr(const double src_point[3], const double map_point[3], const double cov[9],
                    double JTJ[36], double JTr[6]);

/* One iteration's normal equations for points/covs ALREADY transformed into the map frame:
 * correspondenceMatching + the accumulation loop of ICP::computeTransform (Registration.cpp:52-76),
 * summed in ascending point order (deterministic). Returns M. */
size_t oracle_accumulate(const oracle_map* map, size_t n, const double* points, const double* covs,
                         double JTJ[36], double JTr[6]);

/* JTJ.ldlt().solve(-JTr) then Utils::se3ToSE3 (Registration.cpp:78-79, Utils.cpp:40-63). */
void oracle_solve_step(const double JTJ[36], const double JTr[6], double se3[6], double step16[16]);
void oracle_se3_to_SE3(const double se3[6], double out16[16]);
int oracle_convergence_check(const double step16[16], double cosine_threshold,
                             double translation_sq_threshold);
/* Open3D PointCloud::Transform: p <- (T [p;1]).xyz / w, C <- R C R^T, in place. */
void oracle_transform(size_t n, double* points, double* covs, const double T16[16]);

typedef struct oracle_align_stats {
  int32_t iterations;       /* rounds executed (the converging round included) */
  int32_t converged;        /* per call, not sticky */
  int32_t threads;          /* OpenMP threads used */
  int32_t reserved;
  double seconds;           /* omp_get_wtime() around the whole align */
  uint64_t* corr_count;     /* optional, max_iteration entries */
  double* JTJ;              /* optional, max_iteration x 36 */
  double* JTr;              /* optional, max_iteration x 6 */
} oracle_align_stats;

enum { ORACLE_DETERMINISTIC = 0, ORACLE_FAITHFUL = 1 };

/* ICP::align (Registration.cpp:7-35).
 * mode ORACLE_DETERMINISTIC: single pass in point order (bit-reproducible, thread-count free).
 * mode ORACLE_FAITHFUL: the reference's structure — OpenMP regions with thread-private buffers and
 *   critical-section merges, materialised correspondences, serial in-place transforms. */
int oracle_align(const oracle_map* map, size_t n, const double* points, const double* covs,
                 const double guess16[16], int max_iteration, double translation_sq_threshold,
                 double cosine_threshold, int mode, double out_pose16[16],
                 oracle_align_stats* stats);

/* CloudPreprocessor::voxelDownsampleAndEstimateCovariances (src/CloudPreprocessor.cpp:76-127):
 * keep the first point (lowest index) of every voxel of size voxel_size; for every kept point the knn
 * nearest points of the FULL cloud (the point itself included, Open3D KDTreeSearchParamKNN, default 30),
 * Open3D's ComputeCovariance over them (cumulants: E[x x^T] - E[x] E[x]^T), then the regularisation
 * svd.matrixU() * diag(1, 1, 1e-2) * svd.matrixV()^T with Eigen's JacobiSVD restated in its published operation
 * order (two-sided Jacobi, negative diagonal entries folded into U, selection sort).  For a symmetric input the
 * k-th term carries sign(eigenvalue_k): the cumulant covariance of an exactly planar / collinear / repeated
 * neighbourhood far from the origin has a rounding-level smallest eigenvalue of either sign, and the reference then
 * returns an INDEFINITE matrix (-1e-2 on the normal).  The restatement reproduces that class; which noise-level
 * eigenvalues come out negative depends on every rounding of Eigen's compiled code and is not pinned (parity
 * unpinned).  Fewer than 3 neighbours -> identity before regularising.  A non-finite covariance -> NaNs.
 * Output order: ascending original index (the reference emits unordered_map iteration order).
 * out_points m x 3, out_covs m x 9 column-major, out_index m; returns m. Brute-force neighbour search.
 * _ex: *indefinite (optional) receives the number of kept points with a column k where U.col(k) . V.col(k) < 0
 * (a negative eigenvalue: the returned matrix is indefinite). */
size_t oracle_preprocess(size_t n, const double* points, double voxel_size, int knn,
                         double* out_points, double* out_covs, uint64_t* out_index);
size_t oracle_preprocess_ex(size_t n, const double* points, double voxel_size, int knn,
                            double* out_points, double* out_covs, uint64_t* out_index, uint64_t* indefinite);
/* The same with the output order chosen: ascending input index (the default everywhere else) or the iteration order
 * of the reference's std::unordered_map (src/CloudPreprocessor.cpp:85-99) as libstdc++ produces it. */
enum { ORACLE_ORDER_ASCENDING = 0, ORACLE_ORDER_REFERENCE_HASH = 1 };
size_t oracle_preprocess_ordered(size_t n, const double* points, double voxel_size, int knn, int order,
                                 double* out_points, double* out_covs, uint64_t* out_index, uint64_t* indefinite);
/* Test hooks. oracle_jacobi_svd3: the JacobiSVD restatement on ANY real 3x3 (column-major in and out):
 * A = U diag(sv) V^T, sv descending; returns the number of columns with U.col(k) . V.col(k) < 0 (negative
 * eigenvalues of a symmetric A), -1 for a non-finite input.
 * oracle_regularize: U diag(1, 1, 1e-2) V^T of a 3x3, same return value. */
int oracle_jacobi_svd3(const double A[9], double U[9], double V[9], double sv[3]);
int oracle_regularize(const double cov[9], double out[9]);

/* CloudPreprocessor::deskew (src/CloudPreprocessor.cpp:25-74) with Utils::interpolateSE3 and
 * Utils::transformPoints (src/Utils.cpp:13-20,65-75): every point taken before an IMU state's timestamp is
 * moved by (pose at the last point's time)^-1 * (that state's pose); the scan over point_time is the
 * reference's sequential one (a state whose search reaches the end of the scan moves nothing, so the points
 * after the last state at or before the end time stay as they are). states: num_states x 8 doubles =
 * timestamp, position xyz, attitude quaternion in Eigen's coefficient order x y z w.  points are changed in
 * place.  Returns the number of leading points transformed, or -1 (nothing touched) where the reference
 * would leave its state queue: no state at or before the last point's time, or none after it. */
int64_t oracle_deskew(size_t n, double* points, const double* point_time, size_t num_states,
                      const double* states);

int oracle_max_threads(void);
/* OpenMP thread count of the following calls (the CPU-baseline thread sweep of bench.py). */
void oracle_set_threads(int threads);

#ifdef __cplusplus
}
#endif
#endif /* VGICP_ORACLE_H_ */
