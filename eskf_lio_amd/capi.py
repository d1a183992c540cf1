"""ctypes binding of the C ABI in include/vgicp_hip.h (libvgicp_hip.so).

This is plumbing for tests and bench.py: it adds no compute and no fallback.  If the HIP module has
not been built, importing the library raises; if there is no gfx950 device, `Context()` raises
`VgicpError` carrying the library's message.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGICP_LIB_PATH: a developer's A/B aid (two builds of the module timed in one session on one box)
LIB_PATH = os.environ.get("VGICP_LIB_PATH") or os.path.join(_HERE, "lib", "libvgicp_hip.so")

OK, ERR_BAD_ARGUMENT, ERR_HIP, ERR_RCCL, ERR_TABLE_FULL, ERR_DEGENERATE, ERR_NO_DEVICE, ERR_NOT_READY, ERR_TIMEOUT = range(9)
FLAG_PROFILE = 1
FLAG_NO_PERSISTENT = 2
SOLVE_FORCE_PIVOTED = 1
UNIQUE_ID_BYTES = 128
PEER_HANDLE_BYTES = 64

# every symbol include/vgicp_hip.h declares
EXPORTS = (
    "vgicp_abi_version", "vgicp_create", "vgicp_create_multi", "vgicp_destroy", "vgicp_last_error", "vgicp_device_info",
    "vgicp_get_counter",
    "vgicp_map_reset", "vgicp_map_upsert", "vgicp_map_erase", "vgicp_map_size",
    "vgicp_map_insert_scan", "vgicp_map_insert_resident", "vgicp_map_evict", "vgicp_map_export",
    "vgicp_align", "vgicp_scan_upload", "vgicp_align_resident",
    "vgicp_accumulate", "vgicp_solve_step", "vgicp_match", "vgicp_voxel_index", "vgicp_preprocess", "vgicp_deskew",
    "vgicp_scan_prepare", "vgicp_scan_download",
    "vgicp_peer_status", "vgicp_scan_prepare_async", "vgicp_sweep_stage", "vgicp_sweep_stage_cloud2", "vgicp_sweep_unstage", "vgicp_scan_fetch_begin", "vgicp_scan_fetch_end", "vgicp_scan_fetch_sums", "vgicp_scan_prepare_staged_async", "vgicp_scan_info", "vgicp_map_insert_resident_async", "vgicp_get_frame_stats",
    "vgicp_set_option", "vgicp_host_register", "vgicp_host_unregister",
    "vgicp_comm_unique_id", "vgicp_comm_init", "vgicp_comm_destroy",
    "vgicp_peer_export", "vgicp_peer_connect", "vgicp_peer_disconnect",
)


class VgicpError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"vgicp status {code}: {message}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("max_iteration", C.c_int32), ("chunk_iterations", C.c_int32),
                ("translation_sq_threshold", C.c_double), ("cosine_threshold", C.c_double),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("kernel_launches", C.c_uint64), ("copies", C.c_uint64), ("host_syncs", C.c_uint64),
                ("prepare_us", C.c_double), ("align_us", C.c_double), ("insert_us", C.c_double),
                ("prepare_head_us", C.c_double)]


OPTION_STAGE_EVENTS = 1
OPTION_UPLOAD_STAGE_KB = 2   # scans up to this many KiB are staged through page-locked memory of the context; 0 = in place (vgicp_hip.h)
OPTION_REFERENCE_ORDER = 3   # != 0: scan preparations emit the kept points in the reference's unordered_map order (vgicp_hip.h)
COUNTER_UPLOAD_SLOW = 6


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("world_size", C.c_int32),
                ("launches", C.c_int32), ("seconds", C.c_double), ("device_seconds", C.c_double),
                ("corr_count", C.POINTER(C.c_uint64)), ("normal_eq", C.POINTER(C.c_double)),
                ("kernel_ms", C.POINTER(C.c_float))]


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen libvgicp_hip.so and declare its prototypes. Raises if the module is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP module first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C eskf_lio_amd/csrc). "
            "There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    sz = C.c_size_t
    lib.vgicp_abi_version.restype = C.c_int
    lib.vgicp_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.vgicp_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.vgicp_destroy.argtypes = [vp]
    lib.vgicp_last_error.argtypes = [vp]
    lib.vgicp_last_error.restype = C.c_char_p
    lib.vgicp_device_info.argtypes = [vp, C.c_char_p, sz, ip, C.POINTER(C.c_uint64)]
    lib.vgicp_get_counter.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    lib.vgicp_map_reset.argtypes = [vp, C.c_double, sz]
    lib.vgicp_map_upsert.argtypes = [vp, sz, ip, dp, dp]
    lib.vgicp_map_erase.argtypes = [vp, sz, ip]
    lib.vgicp_map_size.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    lib.vgicp_map_insert_scan.argtypes = [vp, sz, dp, dp, dp, sz, C.POINTER(sz)]
    lib.vgicp_map_insert_resident.argtypes = [vp, dp, sz, C.POINTER(sz)]
    lib.vgicp_map_evict.argtypes = [vp, dp, C.c_double, C.POINTER(sz)]
    lib.vgicp_map_export.argtypes = [vp, sz, ip, dp, dp, C.POINTER(C.c_uint64), C.POINTER(sz)]
    lib.vgicp_align.argtypes = [vp, sz, dp, dp, dp, C.POINTER(Params), dp, C.POINTER(Stats)]
    lib.vgicp_scan_upload.argtypes = [vp, sz, dp, dp]
    lib.vgicp_align_resident.argtypes = [vp, dp, C.POINTER(Params), dp, C.POINTER(Stats)]
    lib.vgicp_accumulate.argtypes = [vp, sz, dp, dp, dp, dp, dp, C.POINTER(C.c_uint64)]
    lib.vgicp_solve_step.argtypes = [vp, dp, dp, C.c_double, C.c_double, C.c_uint32, dp, dp, ip, ip]
    lib.vgicp_match.argtypes = [vp, sz, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_uint64), C.POINTER(sz)]
    lib.vgicp_voxel_index.argtypes = [vp, sz, dp, ip]
    lib.vgicp_scan_prepare.argtypes = [vp, sz, dp, dp, sz, dp, dp, C.c_double, C.c_int, C.POINTER(sz),
                                       C.POINTER(C.c_int64)]
    lib.vgicp_scan_download.argtypes = [vp, sz, dp, dp, C.POINTER(sz)]
    lib.vgicp_scan_prepare_async.argtypes = [vp, sz, dp, dp, sz, dp, dp, C.c_double, C.c_int]
    lib.vgicp_sweep_stage.argtypes = [vp, sz, dp, dp, C.POINTER(C.c_uint64)]
    lib.vgicp_sweep_stage_cloud2.argtypes = [vp, sz, vp, sz, sz, sz, sz, sz, C.POINTER(C.c_uint64)]
    lib.vgicp_sweep_unstage.argtypes = [vp, C.c_uint64]
    lib.vgicp_scan_fetch_begin.argtypes = [vp, C.POINTER(sz)]
    lib.vgicp_scan_fetch_end.argtypes = [vp, sz, dp, dp, C.POINTER(sz)]
    lib.vgicp_scan_fetch_sums.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.vgicp_peer_status.argtypes = [vp]
    lib.vgicp_peer_status.restype = C.c_char_p
    lib.vgicp_scan_prepare_staged_async.argtypes = [vp, C.c_uint64, sz, dp, dp, C.c_double, C.c_int]
    lib.vgicp_scan_info.argtypes = [vp, C.POINTER(sz), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    lib.vgicp_map_insert_resident_async.argtypes = [vp, dp, sz]
    lib.vgicp_get_frame_stats.argtypes = [vp, C.POINTER(FrameStats), C.c_int]
    lib.vgicp_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.vgicp_host_register.argtypes = [vp, vp, sz]
    lib.vgicp_host_unregister.argtypes = [vp, vp]
    lib.vgicp_deskew.argtypes = [vp, sz, dp, dp, sz, dp, C.POINTER(C.c_int64)]
    lib.vgicp_preprocess.argtypes = [vp, sz, dp, C.c_double, C.c_int, sz, dp, dp, C.POINTER(C.c_uint64),
                                     C.POINTER(sz)]
    lib.vgicp_comm_unique_id.argtypes = [vp, vp]
    lib.vgicp_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.vgicp_comm_destroy.argtypes = [vp]
    lib.vgicp_peer_export.argtypes = [vp, vp]
    lib.vgicp_peer_connect.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.vgicp_peer_disconnect.argtypes = [vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("vgicp_last_error", "vgicp_peer_status"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def _f64(a, shape_tail) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if out.size and (out.ndim != 2 or out.shape[1] != shape_tail):
        out = out.reshape(-1, shape_tail)
    return out


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def pose_to_abi(T: np.ndarray) -> np.ndarray:
    """4x4 (row, col) numpy pose -> 16 doubles column-major."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(16)


def pose_from_abi(v: np.ndarray) -> np.ndarray:
    return np.asarray(v, dtype=np.float64).reshape(4, 4).T.copy()


@dataclass
class AlignResult:
    pose: np.ndarray  # 4x4
    iterations: int
    converged: bool
    world_size: int
    launches: int
    seconds: float
    device_seconds: float
    corr_count: np.ndarray  # per executed iteration
    normal_eq: np.ndarray  # iterations x 27
    kernel_ms: Optional[np.ndarray] = None
    status: int = OK
    message: str = ""
    _expanded: Optional[tuple] = field(default=None, repr=False)

    def _expand(self):
        if self._expanded is None:   # unpacked on first use: diagnostics, not part of an align
            self._expanded = expand_normal_eq(self.normal_eq)
        return self._expanded

    @property
    def JTJ(self) -> np.ndarray:
        """iterations x 6 x 6 (mirrored from the packed lower triangle)."""
        return self._expand()[0]

    @property
    def JTr(self) -> np.ndarray:
        """iterations x 6."""
        return self._expand()[1]


def expand_normal_eq(rows: np.ndarray):
    """iterations x 27 packed rows -> (iterations x 6 x 6 symmetric JTJ, iterations x 6 JTr)."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 27)
    JTJ = np.zeros((rows.shape[0], 6, 6))
    k = 0
    for r in range(6):
        for c in range(r + 1):
            JTJ[:, r, c] = rows[:, k]
            JTJ[:, c, r] = rows[:, k]
            k += 1
    return JTJ, rows[:, 21:27].copy()


class Context:
    """One vgicp_ctx: bound to one HIP device, or (a list of ordinals) driving several from this one thread."""

    def __init__(self, device_id=0):
        """device_id: one ordinal, or a sequence of ordinals for an in-process multi-device context
        (vgicp_create_multi; an ordinal may repeat: those sub-contexts share the device's compute units)."""
        self._lib = load_library()
        self._h = C.c_void_p()
        if isinstance(device_id, (list, tuple)):
            ids = (C.c_int * len(device_id))(*[int(d) for d in device_id])
            rc = self._lib.vgicp_create_multi(ids, len(device_id), C.byref(self._h))
        else:
            rc = self._lib.vgicp_create(int(device_id), C.byref(self._h))
        if rc != OK:
            msg = self._lib.vgicp_last_error(None).decode()
            self._h = C.c_void_p()
            raise VgicpError(rc, msg)

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.vgicp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, allow=()):
        if rc != OK and rc not in allow:
            raise VgicpError(rc, self._lib.vgicp_last_error(self._h).decode())
        return rc

    def last_error(self) -> str:
        return self._lib.vgicp_last_error(self._h).decode()

    def device_info(self):
        name = C.create_string_buffer(64)
        cu = C.c_int32()
        hbm = C.c_uint64()
        self._check(self._lib.vgicp_device_info(self._h, name, 64, C.byref(cu), C.byref(hbm)))
        return name.value.decode(), cu.value, hbm.value

    def counter(self, which: int) -> int:
        """vgicp_get_counter: 0 persistent launches, 1 persistent fallbacks, 2 upload bytes, 3 upload ns,
        4 indefinite covariances of the last scan preparation."""
        v = C.c_uint64(0)
        self._check(self._lib.vgicp_get_counter(self._h, int(which), C.byref(v)))
        return int(v.value)

    # -- map mirror --
    def map_reset(self, voxel_size: float, capacity_hint: int = 0):
        self._check(self._lib.vgicp_map_reset(self._h, float(voxel_size), int(capacity_hint)))

    def map_upsert(self, keys, means, covs):
        keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 3)
        means = _f64(means, 3)
        covs = _f64(covs, 9)
        n = keys.shape[0]
        if means.shape[0] != n or covs.shape[0] != n:
            raise ValueError("keys / means / covs disagree in length")
        self._check(self._lib.vgicp_map_upsert(self._h, n, keys.ctypes.data_as(C.POINTER(C.c_int32)),
                                               _dp(means), _dp(covs)))

    def map_erase(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 3)
        self._check(self._lib.vgicp_map_erase(self._h, keys.shape[0],
                                              keys.ctypes.data_as(C.POINTER(C.c_int32))))

    def map_size(self):
        v, s = C.c_size_t(), C.c_size_t()
        self._check(self._lib.vgicp_map_size(self._h, C.byref(v), C.byref(s)))
        return v.value, s.value

    def map_insert_scan(self, points, covs, transform, max_points_per_voxel: int) -> int:
        """LocalMap::updateLocalMap's insertion loop on the device; returns the number of new voxels."""
        points, covs = _f64(points, 3), _f64(covs, 9)
        if points.shape[0] != covs.shape[0]:
            raise ValueError("points / covs disagree in length")
        T = pose_to_abi(transform)
        new = C.c_size_t()
        self._check(self._lib.vgicp_map_insert_scan(self._h, points.shape[0], _dp(points), _dp(covs), _dp(T),
                                                    int(max_points_per_voxel), C.byref(new)))
        return new.value

    def map_insert_resident(self, transform, max_points_per_voxel: int) -> int:
        """Insert the scan that is already resident (the one the last align registered)."""
        T = pose_to_abi(transform)
        new = C.c_size_t()
        self._check(self._lib.vgicp_map_insert_resident(self._h, _dp(T), int(max_points_per_voxel), C.byref(new)))
        return new.value

    def map_evict(self, position, distance_threshold: float) -> int:
        pos = np.ascontiguousarray(position, dtype=np.float64).reshape(3)
        removed = C.c_size_t()
        self._check(self._lib.vgicp_map_evict(self._h, _dp(pos), float(distance_threshold), C.byref(removed)))
        return removed.value

    def map_export(self):
        """(keys, means, covs, counts) of every voxel in the mirror, sorted by key."""
        n = self.map_size()[0]
        keys = np.zeros((n, 3), dtype=np.int32)
        means, covs = np.zeros((n, 3)), np.zeros((n, 9))
        counts = np.zeros(n, dtype=np.uint64)
        w = C.c_size_t()
        self._check(self._lib.vgicp_map_export(self._h, n, keys.ctypes.data_as(C.POINTER(C.c_int32)), _dp(means),
                                               _dp(covs), counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(w)))
        assert w.value == n
        order = np.lexsort(keys.T)
        return keys[order], means[order], covs[order], counts[order]

    # -- registration --
    def scan_upload(self, points, covs):
        points, covs = _f64(points, 3), _f64(covs, 9)
        if points.shape[0] != covs.shape[0]:
            raise ValueError("points / covs disagree in length")
        self._check(self._lib.vgicp_scan_upload(self._h, points.shape[0], _dp(points), _dp(covs)))

    def _run(self, call, max_iteration, translation_sq_threshold, cosine_threshold, chunk, flags,
             allow_degenerate):
        p = Params(int(max_iteration), int(chunk), float(translation_sq_threshold),
                   float(cosine_threshold), int(flags), 0)
        cap = max(int(max_iteration), 1)
        # output buffers are kept per context and reused (a call in a frame loop should not pay for a dozen
        # numpy allocations); the result holds copies of the filled parts
        scratch = getattr(self, "_scratch", None)
        if scratch is None or scratch[0] < cap:
            counts = np.zeros(cap, dtype=np.uint64)
            neq = np.zeros((cap, 27))
            kms = np.zeros(cap, dtype=np.float32)
            st = Stats()
            st.corr_count = counts.ctypes.data_as(C.POINTER(C.c_uint64))
            st.normal_eq = _dp(neq)
            st.kernel_ms = kms.ctypes.data_as(C.POINTER(C.c_float))
            out = np.zeros(16)
            scratch = self._scratch = (cap, counts, neq, kms, st, out, _dp(out))
        _, counts, neq, kms, st, out, out_p = scratch
        rc = call(p, out_p, st)
        allow = (ERR_DEGENERATE,) if allow_degenerate else ()
        self._check(rc, allow)
        it = st.iterations
        return AlignResult(pose=pose_from_abi(out), iterations=it, converged=bool(st.converged),
                           world_size=st.world_size, launches=st.launches, seconds=st.seconds,
                           device_seconds=st.device_seconds, corr_count=counts[:it].copy(),
                           normal_eq=neq[:it].copy(),
                           kernel_ms=kms[:st.launches].copy() if flags & FLAG_PROFILE else None,
                           status=rc, message=self.last_error() if rc else "")

    def align(self, points, covs, guess, max_iteration, translation_sq_threshold, cosine_threshold,
              chunk_iterations: int = 0, flags: int = 0, allow_degenerate: bool = False) -> AlignResult:
        points, covs = _f64(points, 3), _f64(covs, 9)
        if points.shape[0] != covs.shape[0]:
            raise ValueError("points / covs disagree in length")
        g = pose_to_abi(guess)
        return self._run(lambda p, out, st: self._lib.vgicp_align(
            self._h, points.shape[0], _dp(points), _dp(covs), _dp(g), C.byref(p), out, C.byref(st)),
            max_iteration, translation_sq_threshold, cosine_threshold, chunk_iterations, flags,
            allow_degenerate)

    def align_resident(self, guess, max_iteration, translation_sq_threshold, cosine_threshold,
                       chunk_iterations: int = 0, flags: int = 0,
                       allow_degenerate: bool = False) -> AlignResult:
        g = pose_to_abi(guess)
        return self._run(lambda p, out, st: self._lib.vgicp_align_resident(
            self._h, _dp(g), C.byref(p), out, C.byref(st)),
            max_iteration, translation_sq_threshold, cosine_threshold, chunk_iterations, flags,
            allow_degenerate)

    # -- hooks --
    def accumulate(self, points, covs, pose):
        points, covs = _f64(points, 3), _f64(covs, 9)
        g = pose_to_abi(pose)
        JTJ, JTr, cnt = np.zeros(36), np.zeros(6), C.c_uint64()
        self._check(self._lib.vgicp_accumulate(self._h, points.shape[0], _dp(points), _dp(covs), _dp(g),
                                               _dp(JTJ), _dp(JTr), C.byref(cnt)))
        return JTJ.reshape(6, 6).T.copy(), JTr, cnt.value

    def solve_step(self, JTJ, JTr, cosine_threshold: float = 0.9999, translation_sq_threshold: float = 1e-6,
                   force_pivoted: bool = False):
        """ldlt().solve(-JTr) + se3ToSE3 + convergenceCheck on the device -> (se3, step 4x4, used_pivoted,
        converged). JTJ: 6x6 (row, col) numpy; only its lower triangle is read."""
        A = np.ascontiguousarray(np.asarray(JTJ, dtype=np.float64).reshape(6, 6).T).reshape(36)
        b = np.ascontiguousarray(JTr, dtype=np.float64).reshape(6)
        se3, step = np.zeros(6), np.zeros(16)
        piv, conv = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.vgicp_solve_step(self._h, _dp(A), _dp(b), float(cosine_threshold),
                                               float(translation_sq_threshold),
                                               SOLVE_FORCE_PIVOTED if force_pivoted else 0, _dp(se3), _dp(step),
                                               C.byref(piv), C.byref(conv)))
        return se3, pose_from_abi(step), bool(piv.value), bool(conv.value)

    def match(self, points, covs):
        points, covs = _f64(points, 3), _f64(covs, 9)
        n = points.shape[0]
        sp, sc, mp, mc = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros((n, 3)), np.zeros((n, 9))
        ix = np.zeros(n, dtype=np.uint64)
        m = C.c_size_t()
        self._check(self._lib.vgicp_match(self._h, n, _dp(points), _dp(covs), _dp(sp), _dp(sc), _dp(mp),
                                          _dp(mc), ix.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(m)))
        k = m.value
        return sp[:k], sc[:k], mp[:k], mc[:k], ix[:k]

    def voxel_index(self, points) -> np.ndarray:
        points = _f64(points, 3)
        keys = np.zeros((points.shape[0], 3), dtype=np.int32)
        self._check(self._lib.vgicp_voxel_index(self._h, points.shape[0], _dp(points),
                                                keys.ctypes.data_as(C.POINTER(C.c_int32))))
        return keys

    def preprocess(self, points, voxel_size: float, knn: int = 30):
        """CloudPreprocessor::voxelDownsampleAndEstimateCovariances (CloudPreprocessor.cpp:76-127) on the
        device -> (kept points m x 3, covariances m x 9 column-major, indices of the kept points)."""
        points = _f64(points, 3)
        n = points.shape[0]
        op, oc = np.zeros((n, 3)), np.zeros((n, 9))
        ix = np.zeros(n, dtype=np.uint64)
        m = C.c_size_t(0)
        self._check(self._lib.vgicp_preprocess(self._h, n, _dp(points), float(voxel_size), int(knn), n, _dp(op),
                                               _dp(oc), ix.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(m)))
        k = m.value
        return op[:k].copy(), oc[:k].copy(), ix[:k].copy()

    def deskew(self, points, point_time, states):
        """CloudPreprocessor::deskew (CloudPreprocessor.cpp:25-74) on the device. states: S x 8 (timestamp,
        position, quaternion x y z w). -> (points, count); count is -1 with the points unchanged where the
        reference would run off its state queue."""
        pts = _f64(points, 3).copy()
        t = np.ascontiguousarray(point_time, dtype=np.float64).reshape(-1)
        st = _f64(states, 8)
        if t.shape[0] != pts.shape[0]:
            raise ValueError("one capture time per point")
        done = C.c_int64(0)
        self._check(self._lib.vgicp_deskew(self._h, pts.shape[0], _dp(pts), _dp(t), st.shape[0], _dp(st),
                                           C.byref(done)))
        return pts, int(done.value)

    def scan_prepare(self, points, point_time=None, states=None, extrinsic=None, voxel_size: float = 0.3,
                     knn: int = 30):
        """CloudPreprocessor::process on the device, the prepared scan stays resident -> (kept, deskewed)."""
        pts = _f64(points, 3)
        n = pts.shape[0]
        st = _f64(states, 8) if states is not None and len(states) else np.zeros((0, 8))
        t = np.ascontiguousarray(point_time, dtype=np.float64).reshape(-1) if point_time is not None else np.zeros(0)
        if st.shape[0] and t.shape[0] != n:
            raise ValueError("one capture time per point")
        ext = pose_to_abi(extrinsic) if extrinsic is not None else None
        kept, moved = C.c_size_t(0), C.c_int64(0)
        self._check(self._lib.vgicp_scan_prepare(self._h, n, _dp(pts), _dp(t) if t.size else None, st.shape[0],
                                                 _dp(st) if st.size else None, _dp(ext) if ext is not None else None,
                                                 float(voxel_size), int(knn), C.byref(kept), C.byref(moved)))
        return kept.value, int(moved.value)

    def scan_prepare_async(self, points, point_time=None, states=None, extrinsic=None, voxel_size: float = 0.3,
                           knn: int = 30):
        """vgicp_scan_prepare_async: the same, only enqueued (nothing waited for, nothing returned)."""
        pts = _f64(points, 3)
        n = pts.shape[0]
        st = _f64(states, 8) if states is not None and len(states) else np.zeros((0, 8))
        t = np.ascontiguousarray(point_time, dtype=np.float64).reshape(-1) if point_time is not None else np.zeros(0)
        if st.shape[0] and t.shape[0] != n:
            raise ValueError("one capture time per point")
        ext = pose_to_abi(extrinsic) if extrinsic is not None else None
        self._check(self._lib.vgicp_scan_prepare_async(self._h, n, _dp(pts), _dp(t) if t.size else None, st.shape[0],
                                                       _dp(st) if st.size else None, _dp(ext) if ext is not None else None,
                                                       float(voxel_size), int(knn)))

    def peer_status(self) -> str:
        """vgicp_peer_status: "" while the kernels' own mailboxes carry the merge between devices, else why not."""
        return self._lib.vgicp_peer_status(self._h).decode()

    def sweep_stage(self, points, point_time=None) -> int:
        """vgicp_sweep_stage: a raw sweep copied into page-locked memory of the context when it arrives -> ticket."""
        pts = _f64(points, 3)
        t = np.ascontiguousarray(point_time, dtype=np.float64).reshape(-1) if point_time is not None else np.zeros(0)
        if t.size and t.shape[0] != pts.shape[0]:
            raise ValueError("one capture time per point")
        ticket = C.c_uint64(0)
        self._check(self._lib.vgicp_sweep_stage(self._h, pts.shape[0], _dp(pts), _dp(t) if t.size else None, C.byref(ticket)))
        return int(ticket.value)

    def scan_fetch(self):
        """vgicp_scan_fetch_begin + _end: the prepared scan of a pending (enqueued) preparation -> (points, covs)."""
        kept = C.c_size_t(0)
        self._check(self._lib.vgicp_scan_fetch_begin(self._h, C.byref(kept)))
        pts, covs = np.zeros((kept.value, 3)), np.zeros((kept.value, 9))
        n = C.c_size_t(0)
        self._check(self._lib.vgicp_scan_fetch_end(self._h, kept.value, _dp(pts) if kept.value else None,
                                                  _dp(covs) if kept.value else None, C.byref(n)))
        assert n.value == kept.value
        return pts, covs

    def scan_fetch_sums(self) -> np.ndarray:
        """vgicp_scan_fetch_sums: the device-made checksums of the last scan_fetch -> uint64[2][2][16] (array, A / B, lane)."""
        out = np.zeros(64, dtype=np.uint64)
        self._check(self._lib.vgicp_scan_fetch_sums(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out.reshape(2, 2, 16)

    def sweep_unstage(self, ticket: int) -> None:
        """vgicp_sweep_unstage: drop a staged sweep that will not be prepared (its slot is free again)."""
        self._check(self._lib.vgicp_sweep_unstage(self._h, C.c_uint64(int(ticket))))

    def sweep_stage_cloud2(self, data, n: int, point_step: int, off_x: int, off_y: int, off_z: int, off_time=None) -> int:
        """vgicp_sweep_stage_cloud2: the payload of a PointCloud2 (bytes / uint8 array, little-endian records) as it
        arrives; the device widens the float32 coordinates -> ticket."""
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        if buf.size < n * point_step:
            raise ValueError("payload shorter than n * point_step")
        ticket = C.c_uint64(0)
        self._check(self._lib.vgicp_sweep_stage_cloud2(self._h, int(n), buf.ctypes.data_as(C.c_void_p), int(point_step), int(off_x),
                                                       int(off_y), int(off_z), (2 ** 64 - 1) if off_time is None else int(off_time),
                                                       C.byref(ticket)))
        return int(ticket.value)

    def scan_prepare_staged_async(self, ticket: int, states=None, extrinsic=None, voxel_size: float = 0.3, knn: int = 30):
        """vgicp_scan_prepare_staged_async: vgicp_scan_prepare_async for a sweep staged by sweep_stage."""
        st = _f64(states, 8) if states is not None and len(states) else np.zeros((0, 8))
        ext = pose_to_abi(extrinsic) if extrinsic is not None else None
        self._check(self._lib.vgicp_scan_prepare_staged_async(self._h, int(ticket), st.shape[0], _dp(st) if st.size else None,
                                                              _dp(ext) if ext is not None else None, float(voxel_size), int(knn)))

    def scan_info(self):
        """(kept points, what the deskew reports, indefinite covariances) of the last preparation."""
        kept, moved, bad = C.c_size_t(0), C.c_int64(0), C.c_uint64(0)
        self._check(self._lib.vgicp_scan_info(self._h, C.byref(kept), C.byref(moved), C.byref(bad)))
        return kept.value, int(moved.value), int(bad.value)

    def map_insert_resident_async(self, transform, max_points_per_voxel: int):
        T = pose_to_abi(transform)
        self._check(self._lib.vgicp_map_insert_resident_async(self._h, _dp(T), int(max_points_per_voxel)))

    def frame_stats(self, reset: bool = False) -> FrameStats:
        st = FrameStats()
        self._check(self._lib.vgicp_get_frame_stats(self._h, C.byref(st), 1 if reset else 0))
        return st

    def set_option(self, option: int, value: int):
        self._check(self._lib.vgicp_set_option(self._h, int(option), int(value)))

    def host_register(self, array: np.ndarray):
        self._check(self._lib.vgicp_host_register(self._h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array: np.ndarray):
        self._check(self._lib.vgicp_host_unregister(self._h, array.ctypes.data))

    def scan_download(self):
        """The resident scan -> (points n x 3, covs n x 9)."""
        n = C.c_size_t(0)
        self._check(self._lib.vgicp_scan_download(self._h, 0, None, None, C.byref(n)))   # size query
        cap = n.value
        pts, covs = np.zeros((cap, 3)), np.zeros((cap, 9))
        self._check(self._lib.vgicp_scan_download(self._h, cap, _dp(pts), _dp(covs), C.byref(n)))
        return pts[:n.value], covs[:n.value]

    # -- multi-GPU --
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        self._check(self._lib.vgicp_comm_unique_id(self._h, buf))
        return buf.raw

    def comm_init(self, world_size: int, rank: int, unique_id: bytes):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._check(self._lib.vgicp_comm_init(self._h, int(world_size), int(rank), buf))

    def comm_destroy(self):
        self._check(self._lib.vgicp_comm_destroy(self._h))

    def peer_export(self) -> bytes:
        """IPC handle (64 bytes) of this context's mailbox for the device-initiated exchange."""
        buf = C.create_string_buffer(PEER_HANDLE_BYTES)
        self._check(self._lib.vgicp_peer_export(self._h, buf))
        return buf.raw

    def peer_connect(self, world_size: int, rank: int, handles: bytes):
        """handles: the world_size exported handles concatenated in rank order. The caller runs a barrier of its
        own transport after every rank has connected, before the first align."""
        if len(handles) != world_size * PEER_HANDLE_BYTES:
            raise ValueError("handles must hold world_size x 64 bytes")
        buf = C.create_string_buffer(handles, len(handles))
        self._check(self._lib.vgicp_peer_connect(self._h, int(world_size), int(rank), buf))

    def peer_disconnect(self):
        self._check(self._lib.vgicp_peer_disconnect(self._h))
