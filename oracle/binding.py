"""ctypes access to the CPU oracle (oracle/libvgicp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package eskf_lio_amd/.  PARITY UNPINNED (see vgicp_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvgicp_oracle.so")
DETERMINISTIC, FAITHFUL = 0, 1


class _Stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("threads", C.c_int32),
                ("reserved", C.c_int32), ("seconds", C.c_double),
                ("corr_count", C.POINTER(C.c_uint64)), ("JTJ", C.POINTER(C.c_double)),
                ("JTr", C.POINTER(C.c_double))]


_lib = None


def build() -> str:
    """Compile the oracle with the reference's flags (oracle/Makefile)."""
    subprocess.run(["make", "-C", _HERE, "libvgicp_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, dp, sz = C.c_void_p, C.POINTER(C.c_double), C.c_size_t
    ip, up = C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
    lib.oracle_map_create.restype = vp
    lib.oracle_map_create.argtypes = [C.c_double, sz]
    lib.oracle_map_destroy.argtypes = [vp]
    lib.oracle_map_size.restype = sz
    lib.oracle_map_size.argtypes = [vp]
    lib.oracle_map_insert.argtypes = [vp, sz, dp, dp]
    lib.oracle_map_evict.argtypes = [vp, dp, C.c_double]
    lib.oracle_map_evict.restype = sz
    lib.oracle_map_export.restype = sz
    lib.oracle_map_export.argtypes = [vp, sz, ip, dp, dp, up]
    lib.oracle_voxel_index.argtypes = [C.c_double, sz, dp, ip]
    lib.oracle_match.restype = sz
    lib.oracle_match.argtypes = [vp, sz, dp, dp, dp, dp, dp, dp, up]
    lib.oracle_jtj_jtr.argtypes = [dp, dp, dp, dp, dp]
    lib.oracle_accumulate.restype = sz
    lib.oracle_accumulate.argtypes = [vp, sz, dp, dp, dp, dp]
    lib.oracle_solve_step.argtypes = [dp, dp, dp, dp]
    lib.oracle_se3_to_SE3.argtypes = [dp, dp]
    lib.oracle_convergence_check.restype = C.c_int
    lib.oracle_convergence_check.argtypes = [dp, C.c_double, C.c_double]
    lib.oracle_transform.argtypes = [sz, dp, dp, dp]
    lib.oracle_align.restype = C.c_int
    lib.oracle_align.argtypes = [vp, sz, dp, dp, dp, C.c_int, C.c_double, C.c_double, C.c_int, dp,
                                 C.POINTER(_Stats)]
    lib.oracle_preprocess.restype = sz
    lib.oracle_preprocess.argtypes = [sz, dp, C.c_double, C.c_int, dp, dp, up]
    lib.oracle_preprocess_ex.restype = sz
    lib.oracle_preprocess_ex.argtypes = [sz, dp, C.c_double, C.c_int, dp, dp, up, up]
    lib.oracle_preprocess_ordered.restype = sz
    lib.oracle_preprocess_ordered.argtypes = [sz, dp, C.c_double, C.c_int, C.c_int, dp, dp, up, up]
    lib.oracle_jacobi_svd3.restype = C.c_int
    lib.oracle_jacobi_svd3.argtypes = [dp, dp, dp, dp]
    lib.oracle_regularize.restype = C.c_int
    lib.oracle_regularize.argtypes = [dp, dp]
    lib.oracle_deskew.restype = C.c_int64
    lib.oracle_deskew.argtypes = [sz, dp, dp, sz, dp]
    lib.oracle_max_threads.restype = C.c_int
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_set_threads.restype = None
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a, tail):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(-1, tail) if a.size else a.reshape(0, tail)


def _pose_in(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(16)


def _pose_out(v):
    return v.reshape(4, 4).T.copy()


@dataclass
class OracleAlign:
    pose: np.ndarray
    iterations: int
    converged: bool
    threads: int
    seconds: float
    corr_count: np.ndarray
    JTJ: np.ndarray  # iterations x 6 x 6
    JTr: np.ndarray  # iterations x 6


class OracleMap:
    """The reference's LocalMap as far as the path reads it (LocalMap.hpp:54-89, LocalMap.cpp:47-58,78-118)."""

    def __init__(self, voxel_size: float, max_points_per_voxel: int = 1):
        self._lib = load()
        self.voxel_size = float(voxel_size)
        self._h = self._lib.oracle_map_create(self.voxel_size, int(max_points_per_voxel))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.oracle_map_destroy(self._h)
            self._h = None

    def __len__(self):
        return self._lib.oracle_map_size(self._h)

    def insert(self, points, covs):
        points, covs = _f64(points, 3), _f64(covs, 9)
        self._lib.oracle_map_insert(self._h, points.shape[0], _dp(points), _dp(covs))

    def evict(self, position, distance_threshold: float) -> int:
        """LocalMap::updateLocalMap's eviction loop (src/LocalMap.cpp:60-72): -> voxels removed."""
        pos = np.ascontiguousarray(position, dtype=np.float64).reshape(3)
        return int(self._lib.oracle_map_evict(self._h, _dp(pos), float(distance_threshold)))

    def export(self):
        n = len(self)
        keys = np.zeros((n, 3), dtype=np.int32)
        means, covs = np.zeros((n, 3)), np.zeros((n, 9))
        counts = np.zeros(n, dtype=np.uint64)
        w = self._lib.oracle_map_export(self._h, n, keys.ctypes.data_as(C.POINTER(C.c_int32)), _dp(means),
                                        _dp(covs), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert w == n
        return keys, means, covs, counts

    def match(self, points, covs):
        points, covs = _f64(points, 3), _f64(covs, 9)
        n = points.shape[0]
        sp, sc, mp, mc = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros((n, 3)), np.zeros((n, 9))
        ix = np.zeros(n, dtype=np.uint64)
        m = self._lib.oracle_match(self._h, n, _dp(points), _dp(covs), _dp(sp), _dp(sc), _dp(mp), _dp(mc),
                                   ix.ctypes.data_as(C.POINTER(C.c_uint64)))
        return sp[:m], sc[:m], mp[:m], mc[:m], ix[:m]

    def accumulate(self, points, covs):
        """Normal equations for points/covs already in the map frame -> (JTJ 6x6, JTr 6, count)."""
        points, covs = _f64(points, 3), _f64(covs, 9)
        JTJ, JTr = np.zeros(36), np.zeros(6)
        m = self._lib.oracle_accumulate(self._h, points.shape[0], _dp(points), _dp(covs), _dp(JTJ), _dp(JTr))
        return JTJ.reshape(6, 6).T.copy(), JTr, int(m)

    def align(self, points, covs, guess, max_iteration, translation_sq_threshold, cosine_threshold,
              mode: int = DETERMINISTIC) -> OracleAlign:
        points, covs = _f64(points, 3), _f64(covs, 9)
        cap = max(int(max_iteration), 1)
        counts = np.zeros(cap, dtype=np.uint64)
        JTJ, JTr = np.zeros((cap, 36)), np.zeros((cap, 6))
        st = _Stats()
        st.corr_count = counts.ctypes.data_as(C.POINTER(C.c_uint64))
        st.JTJ, st.JTr = _dp(JTJ), _dp(JTr)
        out = np.zeros(16)
        g = _pose_in(guess)
        self._lib.oracle_align(self._h, points.shape[0], _dp(points), _dp(covs), _dp(g), int(max_iteration),
                               float(translation_sq_threshold), float(cosine_threshold), int(mode), _dp(out),
                               C.byref(st))
        it = st.iterations
        return OracleAlign(_pose_out(out), it, bool(st.converged), st.threads, st.seconds, counts[:it].copy(),
                           JTJ[:it].reshape(it, 6, 6).transpose(0, 2, 1).copy(), JTr[:it].copy())


def voxel_index(voxel_size, points):
    points = _f64(points, 3)
    keys = np.zeros((points.shape[0], 3), dtype=np.int32)
    load().oracle_voxel_index(float(voxel_size), points.shape[0], _dp(points),
                              keys.ctypes.data_as(C.POINTER(C.c_int32)))
    return keys


def jtj_jtr(src_point, map_point, cov):
    p = np.ascontiguousarray(src_point, dtype=np.float64)
    m = np.ascontiguousarray(map_point, dtype=np.float64)
    c = np.ascontiguousarray(np.asarray(cov, dtype=np.float64).T).reshape(9)  # 3x3 (r,c) -> column-major
    JTJ, JTr = np.zeros(36), np.zeros(6)
    load().oracle_jtj_jtr(_dp(p), _dp(m), _dp(c), _dp(JTJ), _dp(JTr))
    return JTJ.reshape(6, 6).T.copy(), JTr


def solve_step(JTJ, JTr):
    J = np.ascontiguousarray(np.asarray(JTJ, dtype=np.float64).T).reshape(36)
    r = np.ascontiguousarray(JTr, dtype=np.float64)
    se3, step = np.zeros(6), np.zeros(16)
    load().oracle_solve_step(_dp(J), _dp(r), _dp(se3), _dp(step))
    return se3, _pose_out(step)


def se3_to_SE3(xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    out = np.zeros(16)
    load().oracle_se3_to_SE3(_dp(xi), _dp(out))
    return _pose_out(out)


def convergence_check(step, cosine_threshold, translation_sq_threshold) -> bool:
    s = _pose_in(step)
    return bool(load().oracle_convergence_check(_dp(s), float(cosine_threshold), float(translation_sq_threshold)))


def transform(points, covs, T):
    points, covs = _f64(points, 3).copy(), _f64(covs, 9).copy()
    t = _pose_in(T)
    load().oracle_transform(points.shape[0], _dp(points), _dp(covs), _dp(t))
    return points, covs


def max_threads() -> int:
    return load().oracle_max_threads()


def set_threads(threads: int) -> None:
    load().oracle_set_threads(int(threads))


def preprocess(points, voxel_size: float, knn: int = 30):
    """voxelDownsampleAndEstimateCovariances: (kept points m x 3, covariances m x 9, original indices)."""
    points = _f64(points, 3)
    n = points.shape[0]
    op, oc = np.zeros((n, 3)), np.zeros((n, 9))
    ix = np.zeros(n, dtype=np.uint64)
    m = load().oracle_preprocess(n, _dp(points), float(voxel_size), int(knn), _dp(op), _dp(oc),
                                 ix.ctypes.data_as(C.POINTER(C.c_uint64)))
    return op[:m].copy(), oc[:m].copy(), ix[:m].copy()


ORDER_ASCENDING, ORDER_REFERENCE_HASH = 0, 1


def preprocess_ordered(points, voxel_size: float, knn: int = 30, order: int = ORDER_ASCENDING):
    """preprocess() with the output order chosen: ORDER_ASCENDING (input index; what the HIP path emits) or
    ORDER_REFERENCE_HASH (the iteration order of the reference's unordered_map, src/CloudPreprocessor.cpp:85-99, as
    libstdc++ produces it).  Same kept set and covariances, another sequence."""
    points = _f64(points, 3)
    n = points.shape[0]
    op, oc = np.zeros((n, 3)), np.zeros((n, 9))
    ix = np.zeros(n, dtype=np.uint64)
    bad = C.c_uint64(0)
    m = load().oracle_preprocess_ordered(n, _dp(points), float(voxel_size), int(knn), int(order), _dp(op), _dp(oc),
                                         ix.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(bad))
    return op[:m].copy(), oc[:m].copy(), ix[:m].copy()


def preprocess_ex(points, voxel_size: float, knn: int = 30):
    """preprocess() plus the number of kept points whose covariance came out INDEFINITE (the SVD negated a
    column of U: a negative eigenvalue of the cumulant covariance, src/CloudPreprocessor.cpp:119-123)."""
    points = _f64(points, 3)
    n = points.shape[0]
    op, oc = np.zeros((n, 3)), np.zeros((n, 9))
    ix = np.zeros(n, dtype=np.uint64)
    bad = C.c_uint64(0)
    m = load().oracle_preprocess_ex(n, _dp(points), float(voxel_size), int(knn), _dp(op), _dp(oc),
                                    ix.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(bad))
    return op[:m].copy(), oc[:m].copy(), ix[:m].copy(), int(bad.value)


def jacobi_svd3(A):
    """Eigen::JacobiSVD<Matrix3d> restated: A (3x3, row/col numpy) -> (U, sv, V, negated columns); A = U diag(sv) V^T."""
    a = np.ascontiguousarray(np.asarray(A, dtype=np.float64).reshape(3, 3).T).reshape(9)
    U, V, sv = np.zeros(9), np.zeros(9), np.zeros(3)
    neg = load().oracle_jacobi_svd3(_dp(a), _dp(U), _dp(V), _dp(sv))
    return U.reshape(3, 3).T.copy(), sv, V.reshape(3, 3).T.copy(), int(neg)


def regularize(cov):
    """svd.matrixU() * diag(1, 1, 1e-2) * svd.matrixV()^T of a 3x3 -> (3x3, negated columns)."""
    a = np.ascontiguousarray(np.asarray(cov, dtype=np.float64).reshape(3, 3).T).reshape(9)
    out = np.zeros(9)
    neg = load().oracle_regularize(_dp(a), _dp(out))
    return out.reshape(3, 3).T.copy(), int(neg)


def deskew(points, point_time, states):
    """CloudPreprocessor::deskew: states S x 8 (timestamp, position, quaternion xyzw) -> (points, count);
    count is -1 (points unchanged) where the reference would run off its state queue."""
    points = _f64(points, 3).copy()
    t = np.ascontiguousarray(point_time, dtype=np.float64).reshape(-1)
    st = _f64(states, 8)
    assert t.shape[0] == points.shape[0]
    done = load().oracle_deskew(points.shape[0], _dp(points), _dp(t), st.shape[0], _dp(st))
    return points, int(done)
