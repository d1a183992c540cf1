"""Python handles on the C++ host mirror (libvgicp_host.so = include/eskf_lio_shim/ compiled).

`LocalMap` and `ICP` here are thin ctypes proxies of the C++ classes ESKF_LIO::LocalMap and
ESKF_LIO::ICP that keep the reference's method names and argument meaning (reference
include/ESKF_LIO/LocalMap.hpp:91-98, include/ESKF_LIO/Registration.hpp:23-32), so parity tests read
like tests of the reference would: build a map with updateLocalMap(), register with align().
No compute happens in Python and there is no fallback: errors of the HIP module surface as
RuntimeError with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvgicp_host.so")
_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    capi.load_library()  # libvgicp_hip.so first (RTLD_GLOBAL), the host layer links against it
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C eskf_lio_amd/host")
    lib = C.CDLL(LIB_PATH)
    vp, dp, sz = C.c_void_p, C.POINTER(C.c_double), C.c_size_t
    lib.host_last_error.restype = C.c_char_p
    lib.host_localmap_create_config.restype = vp
    lib.host_localmap_create_config.argtypes = [C.c_double, sz, C.c_double, C.c_double, C.c_int, C.c_double,
                                                C.c_double, C.c_int, C.c_int]
    lib.host_localmap_create.restype = vp
    lib.host_localmap_create.argtypes = [C.c_double, sz]
    lib.host_localmap_destroy.argtypes = [vp]
    lib.host_localmap_size.restype = sz
    lib.host_localmap_size.argtypes = [vp]
    lib.host_hash_helpers.restype = None
    lib.host_hash_helpers.argtypes = [C.c_int]
    lib.host_localmap_drain.restype = sz
    lib.host_localmap_drain.argtypes = [vp]
    lib.host_localmap_update.argtypes = [vp, sz, dp, dp, dp, C.c_int]
    lib.host_localmap_match.argtypes = [vp, sz, dp, dp, dp, dp, dp, dp, C.POINTER(sz)]
    lib.host_localmap_export.restype = sz
    lib.host_localmap_export.argtypes = [vp, sz, C.POINTER(C.c_int32), dp, dp, C.POINTER(C.c_uint64)]
    lib.host_localmap_save.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.host_icp_create.restype = vp
    lib.host_icp_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int]
    lib.host_icp_destroy.argtypes = [vp]
    lib.host_icp_align.argtypes = [vp, sz, dp, dp, vp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint64), sz]
    lib.host_preprocessor_create.restype = vp
    lib.host_preprocessor_create.argtypes = [C.c_double, dp, C.c_int, C.c_int]
    lib.host_frame_begin.restype = vp
    lib.host_frame_begin.argtypes = [sz, dp, dp, sz, dp]
    lib.host_frame_run.argtypes = [vp, vp, vp, vp, dp, C.c_int, C.c_int, vp, C.c_int]
    lib.host_frame_stage.argtypes = [vp, vp]
    lib.host_frame_stage.restype = C.c_int
    lib.host_frame_end.argtypes = [vp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                   C.POINTER(sz), sz, dp, dp]
    lib.host_preprocessor_process.argtypes = [vp, sz, dp, dp, sz, dp, dp, dp, C.POINTER(sz)]
    lib.host_preprocessor_destroy.argtypes = [vp]
    lib.host_preprocessor_downsample.argtypes = [vp, sz, dp, dp, dp, C.POINTER(sz)]
    lib.host_math_ldlt6_solve.argtypes = [dp, dp, dp]
    lib.host_math_ldlt6_solve.restype = None
    lib.host_math_se3_exp.argtypes = [dp, dp]
    lib.host_math_se3_exp.restype = None
    _lib = lib
    return lib


def math_ldlt6_solve(JTJ, b) -> np.ndarray:
    """csrc/vgicp_math.h's pivoted LDLT (the kernels' solve) compiled for the host: A x = b with A a
    6x6 (row, col) array of which the lower triangle is read."""
    A = np.asarray(JTJ, dtype=np.float64).reshape(6, 6)
    low = np.ascontiguousarray([A[r, c] for r in range(6) for c in range(r + 1)], dtype=np.float64)
    rhs = np.ascontiguousarray(b, dtype=np.float64).reshape(6)
    x = np.zeros(6)
    load_library().host_math_ldlt6_solve(_dp(low), _dp(rhs), _dp(x))
    return x


def math_se3_exp(xi) -> np.ndarray:
    v = np.ascontiguousarray(xi, dtype=np.float64).reshape(6)
    out = np.zeros(16)
    load_library().host_math_se3_exp(_dp(v), _dp(out))
    return out.reshape(4, 4).T.copy()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.host_last_error().decode())


class LocalMap:
    """ESKF_LIO::LocalMap.  config: the YAML keys + device_resident (default True: the grid the registration reads lives
    on the device) + keep_raw_points (default True: a host-side shadow grid keeps every raw point for save())."""

    def __init__(self, voxelSize: float, maxNumPointsPerVoxel: int, config: Optional[dict] = None):
        self._lib = load_library()
        if config is None:
            self._h = self._lib.host_localmap_create(float(voxelSize), int(maxNumPointsPerVoxel))
        else:
            self._h = self._lib.host_localmap_create_config(
                float(voxelSize), int(maxNumPointsPerVoxel), float(config["translation_sq_threshold"]),
                float(config["cosine_threshold"]), int(bool(config["remove_distant_points"])),
                float(config["distance_threshold"]), float(config["removing_period"]),
                int(bool(config.get("device_resident", True))), int(bool(config.get("keep_raw_points", True))))
        if not self._h:
            raise RuntimeError(self._lib.host_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.host_localmap_destroy(self._h)
            self._h = None

    def __len__(self):
        return self._lib.host_localmap_size(self._h)

    def drain(self) -> int:
        """Waits for the shadow grid's worker (LocalMap::grid()); the host grid's voxel count."""
        return self._lib.host_localmap_drain(self._h)

    def updateLocalMap(self, points, covs, transform, initialize: bool = False):
        """Returns the cloud moved into the world frame (the reference mutates it in place)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        cvs = np.ascontiguousarray(covs, dtype=np.float64).reshape(-1, 9).copy()
        T = capi.pose_to_abi(transform)
        _check(self._lib, self._lib.host_localmap_update(self._h, pts.shape[0], _dp(pts), _dp(cvs), _dp(T),
                                                         int(initialize)))
        return pts, cvs

    def correspondenceMatching(self, points, covs):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        cvs = np.ascontiguousarray(covs, dtype=np.float64).reshape(-1, 9)
        n = pts.shape[0]
        sp, sc, mp, mc = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros((n, 3)), np.zeros((n, 9))
        m = C.c_size_t()
        _check(self._lib, self._lib.host_localmap_match(self._h, n, _dp(pts), _dp(cvs), _dp(sp), _dp(sc),
                                                        _dp(mp), _dp(mc), C.byref(m)))
        k = m.value
        return sp[:k], sc[:k], mp[:k], mc[:k]

    def export(self):
        n = len(self)
        keys = np.zeros((n, 3), dtype=np.int32)
        means, covs = np.zeros((n, 3)), np.zeros((n, 9))
        counts = np.zeros(n, dtype=np.uint64)
        w = self._lib.host_localmap_export(self._h, n, keys.ctypes.data_as(C.POINTER(C.c_int32)), _dp(means),
                                           _dp(covs), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert w == n
        return keys, means, covs, counts

    def save(self, cloud_path: str, trajectory_path: str):
        _check(self._lib, self._lib.host_localmap_save(self._h, cloud_path.encode(), trajectory_path.encode()))


class ICP:
    """ESKF_LIO::ICP: config keys as in registration.* of the reference's YAML."""

    def __init__(self, max_iteration: int, translation_sq_threshold: float, cosine_threshold: float,
                 chunk_iterations: int = 0):
        self._lib = load_library()
        self.max_iteration = int(max_iteration)
        self._h = self._lib.host_icp_create(self.max_iteration, float(translation_sq_threshold),
                                            float(cosine_threshold), int(chunk_iterations))
        self.iterations = 0
        self.converged = False
        self.correspondence_counts = np.zeros(0, dtype=np.uint64)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.host_icp_destroy(self._h)
            self._h = None

    def align(self, points, covs, localMap: LocalMap, guess) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        cvs = np.ascontiguousarray(covs, dtype=np.float64).reshape(-1, 9)
        g = capi.pose_to_abi(guess)
        out = np.zeros(16)
        it, conv = C.c_int32(), C.c_int32()
        cap = max(self.max_iteration, 1)
        counts = np.zeros(cap, dtype=np.uint64)
        _check(self._lib, self._lib.host_icp_align(self._h, pts.shape[0], _dp(pts), _dp(cvs), localMap._h, _dp(g),
                                                   _dp(out), C.byref(it), C.byref(conv),
                                                   counts.ctypes.data_as(C.POINTER(C.c_uint64)), cap))
        self.iterations, self.converged = it.value, bool(conv.value)
        self.correspondence_counts = counts[:it.value].copy()
        return capi.pose_from_abi(out)


class CloudPreprocessor:
    """ESKF_LIO::CloudPreprocessor's scan-preparation half (include/eskf_lio_shim/CloudPreprocessor.hpp;
    reference include/ESKF_LIO/CloudPreprocessor.hpp:35-36, src/CloudPreprocessor.cpp:76-127)."""

    def __init__(self, voxel_size: float, T_il=None, host_copy: Optional[str] = None, resident_check: str = "full"):
        """host_copy: "eager" (process() leaves the prepared scan in the host cloud, as the reference does),
        "deferred" (it stays on the device until somebody materialises it), None = the shim's default.
        resident_check: "full" (default: every byte of the host cloud is hashed before the resident scan is trusted) or
        "sampled" (CloudPreprocessorConfig::residentCheck = Sampled)."""
        self._lib = load_library()
        t = capi.pose_to_abi(np.eye(4) if T_il is None else T_il)
        mode = {None: -1, "eager": 0, "deferred": 1}[host_copy]
        self._h = self._lib.host_preprocessor_create(float(voxel_size), _dp(t), mode, 1 if resident_check == "sampled" else 0)
        if not self._h:
            raise RuntimeError(self._lib.host_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.host_preprocessor_destroy(self._h)
            self._h = None

    def voxelDownsampleAndEstimateCovariances(self, points):
        """cloud.points_ in -> (cloud.points_, cloud.covariances_) out."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        n = pts.shape[0]
        op, oc = np.zeros((n, 3)), np.zeros((n, 9))
        kept = C.c_size_t(0)
        _check(self._lib, self._lib.host_preprocessor_downsample(self._h, n, _dp(pts), _dp(op), _dp(oc),
                                                                 C.byref(kept)))
        return op[:kept.value].copy(), oc[:kept.value].copy()

    def process(self, states, points, pointTime):
        """process(states, lidarMeas): extrinsic -> deskew -> down-sampling + covariances
        (reference src/CloudPreprocessor.cpp:8-23). states: S x 8 (timestamp, position, quaternion xyzw)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(pointTime, dtype=np.float64).reshape(-1)
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 8)
        n = pts.shape[0]
        op, oc = np.zeros((n, 3)), np.zeros((n, 9))
        kept = C.c_size_t(0)
        _check(self._lib, self._lib.host_preprocessor_process(self._h, n, _dp(pts), _dp(t), st.shape[0], _dp(st),
                                                              _dp(op), _dp(oc), C.byref(kept)))
        return op[:kept.value].copy(), oc[:kept.value].copy()


class Frame:
    """One LiDAR frame through the C++ drop-in classes the way src/Odometry.cpp:73-87 is written:
    process(states, meas) -> icp.align(*meas.cloud, localMap, guess) -> localMap.updateLocalMap(meas.cloud, T).
    begin (builds the measurement object) / run (the three calls; what a caller would time) / end (results)."""

    def __init__(self, points, pointTime, states):
        self._lib = load_library()
        self._pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        self._t = np.ascontiguousarray(pointTime, dtype=np.float64).reshape(-1)
        self._st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 8)
        self._h = self._lib.host_frame_begin(self._pts.shape[0], _dp(self._pts), _dp(self._t), self._st.shape[0],
                                             _dp(self._st))

    def stage(self, preprocessor: "CloudPreprocessor") -> bool:
        """CloudPreprocessor::stage on this frame's measurement: what a lidar callback does when the sweep arrives."""
        rc = self._lib.host_frame_stage(self._h, preprocessor._h)
        if rc < 0:
            raise RuntimeError("CloudPreprocessor::stage failed")
        return bool(rc)

    def run(self, preprocessor: "CloudPreprocessor", icp: "ICP", localMap: "LocalMap", guess, first_frame=False,
            mutate: int = 0, stage_next: "Frame" = None, move_cloud: bool = False):
        """stage_next: a later Frame whose sweep 'arrives' during this one (staged right after this frame's process()).
        move_cloud: hand the cloud to updateLocalMap as src/Odometry.cpp:86 does (std::move: end() then has no cloud)."""
        g = capi.pose_to_abi(guess)
        _check(self._lib, self._lib.host_frame_run(self._h, preprocessor._h, icp._h, localMap._h, _dp(g),
                                                   1 if first_frame else 0, int(mutate),
                                                   stage_next._h if stage_next is not None else None, int(bool(move_cloud))))

    def end(self, want_cloud: bool = False):
        """-> dict(pose, iterations, used_resident, corr0, host_points[, points, covs])."""
        out = np.zeros(16)
        it, res = C.c_int32(), C.c_int32()
        corr0 = C.c_uint64()
        hp = C.c_size_t()
        n = self._pts.shape[0]
        pts = np.zeros((n, 3)) if want_cloud else None
        cvs = np.zeros((n, 9)) if want_cloud else None
        _check(self._lib, self._lib.host_frame_end(self._h, _dp(out), C.byref(it), C.byref(res), C.byref(corr0),
                                                   C.byref(hp), n, _dp(pts) if want_cloud else None,
                                                   _dp(cvs) if want_cloud else None))
        self._h = None
        r = dict(pose=capi.pose_from_abi(out), iterations=it.value, used_resident=bool(res.value), corr0=int(corr0.value),
                 host_points=int(hp.value))
        if want_cloud:
            r["points"], r["covs"] = pts[:hp.value].copy(), cvs[:hp.value].copy()
        return r
