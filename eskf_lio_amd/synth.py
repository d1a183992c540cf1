"""Synthetic voxel maps and scans for the VGICP registration path.

Everything here is generated from a counter-based splitmix64 stream using only IEEE-exact
operations (+, -, *, /, sqrt) on the per-element path, so the same seed produces the same bits on
any host: fixtures under tests/golden/ store only *outputs* and regenerate their inputs from seeds.

Shapes follow the reference's boundary types: points are N x 3 float64 rows (the memory of a
std::vector<Eigen::Vector3d>), covariances are N x 9 float64 rows holding a COLUMN-major 3x3 each
(the memory of a std::vector<Eigen::Matrix3d>) — reference include/ESKF_LIO/Registration.hpp:18-21.
Covariances are `R diag(1, 1, 1e-2) R^T` for a uniformly random rotation R, the family the
reference's preprocessing emits (reference include/ESKF_LIO/CloudPreprocessor.hpp:30-31,
src/CloudPreprocessor.cpp:121-123); that equals `I - 0.99 n n^T` for a uniform unit vector n.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

MAP_SEED = 0x4D41505F31  # "MAP_1"
SCAN_SEED = 0x5343414E5F31  # "SCAN_1"
VOXEL_SIZE = 0.3  # reference config/hilti_config.yaml:37
OCCUPANCY = 0.5

_U64 = np.uint64
_GAMMA = _U64(0x9E3779B97F4A7C15)
_M1 = _U64(0xBF58476D1CE4E5B9)
_M2 = _U64(0x94D049BB133111EB)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step applied element-wise to uint64 counters."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=_U64) + _GAMMA).astype(_U64)
        z = (z ^ (z >> _U64(30))) * _M1
        z = (z ^ (z >> _U64(27))) * _M2
        return z ^ (z >> _U64(31))


def rand_u64(seed: int, stream: int, index: np.ndarray) -> np.ndarray:
    """u64(seed, stream, i) = splitmix64(splitmix64(seed + stream) + i)."""
    base = splitmix64(np.array([(seed + stream) & 0xFFFFFFFFFFFFFFFF], dtype=_U64))[0]
    with np.errstate(over="ignore"):
        return splitmix64(base + np.asarray(index, dtype=_U64))


def rand_unit(seed: int, stream: int, index: np.ndarray) -> np.ndarray:
    """Uniform doubles in [0, 1): top 53 bits scaled by 2^-53."""
    return (rand_u64(seed, stream, index) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _unit_vectors(seed: int, stream0: int, index: np.ndarray, tries: int = 24) -> np.ndarray:
    """Marsaglia (1972) uniform points on S^2 without trigonometry: first accepted of `tries` draws."""
    n = index.shape[0]
    out = np.zeros((n, 3))
    out[:, 2] = 1.0
    pending = np.ones(n, dtype=bool)
    for k in range(tries):
        if not pending.any():
            break
        idx = index[pending]
        a = 2.0 * rand_unit(seed, stream0 + 2 * k, idx) - 1.0
        b = 2.0 * rand_unit(seed, stream0 + 2 * k + 1, idx) - 1.0
        s = a * a + b * b
        ok = (s < 1.0) & (s > 0.0)
        root = np.sqrt(np.maximum(1.0 - s, 0.0))
        vec = np.stack([2.0 * a * root, 2.0 * b * root, 1.0 - 2.0 * s], axis=1)
        where = np.flatnonzero(pending)[ok]
        out[where] = vec[ok]
        pending[where] = False
    return out


def disc_covariances(seed: int, stream0: int, index: np.ndarray) -> np.ndarray:
    """N x 9 exactly symmetric covariances I - 0.99 n n^T (column-major == row-major here)."""
    nrm = _unit_vectors(seed, stream0, index)
    cov = np.empty((index.shape[0], 9))
    for r in range(3):
        for c in range(r, 3):
            v = (1.0 if r == c else 0.0) - 0.99 * (nrm[:, r] * nrm[:, c])
            cov[:, r + 3 * c] = v
            cov[:, c + 3 * r] = v
    return cov


def se3_to_SE3(xi) -> np.ndarray:
    """4x4 pose from [rho; phi] by the reference's rule (reference src/Utils.cpp:40-63).

    Scalar libm calls only; used to build guesses / ground-truth poses, not per-point data."""
    rho = [float(v) for v in xi[:3]]
    phi = [float(v) for v in xi[3:]]
    sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]
    angle = math.sqrt(sq)
    k = phi if sq == 0.0 else [p / angle for p in phi]
    s, c = math.sin(angle), math.cos(angle)
    T = np.eye(4)
    ck = [(1.0 - c) * v for v in k]
    sk = [s * v for v in k]
    T[0, 1] = ck[0] * k[1] - sk[2]
    T[1, 0] = ck[0] * k[1] + sk[2]
    T[0, 2] = ck[0] * k[2] + sk[1]
    T[2, 0] = ck[0] * k[2] - sk[1]
    T[1, 2] = ck[1] * k[2] - sk[0]
    T[2, 1] = ck[1] * k[2] + sk[0]
    for i in range(3):
        T[i, i] = ck[i] * k[i] + c
    if angle < 1e-6:
        J = np.eye(3)
    else:
        f1 = s / angle
        f2 = (1.0 - c) / angle
        K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
        J = f1 * np.eye(3) + (1.0 - f1) * np.outer(k, k) + f2 * K
    T[:3, 3] = J @ np.array(rho)
    return T


GUESS_XI = (0.05, -0.03, 0.02, 0.004, -0.003, 0.005)  # 5 cm / ~0.4 deg
TRUE_XI = (0.06, -0.04, 0.03, 0.003, -0.002, 0.004)


def default_guess() -> np.ndarray:
    return se3_to_SE3(GUESS_XI)


@dataclass
class VoxelMap:
    voxel_size: float
    side: int  # cells per axis of the occupied cube
    keys: np.ndarray  # V x 3 int32
    means: np.ndarray  # V x 3 float64
    covs: np.ndarray  # V x 9 float64, column-major 3x3

    @property
    def lo(self) -> float:
        return -(self.side // 2) * self.voxel_size

    @property
    def hi(self) -> float:
        return (self.side - self.side // 2) * self.voxel_size


def map_side(num_voxels: int, occupancy: float = OCCUPANCY) -> int:
    side = int(math.ceil((num_voxels / occupancy) ** (1.0 / 3.0)))
    while side ** 3 < num_voxels / occupancy:
        side += 1
    return side


def make_map(num_voxels: int, seed: int = MAP_SEED, voxel_size: float = VOXEL_SIZE) -> VoxelMap:
    """The 'fixed synthetic voxel map': the `num_voxels` cells of a centred side^3 cube whose
    splitmix64 hash is smallest are occupied; one mean inside each cell, one disc covariance."""
    side = map_side(num_voxels)
    cells = np.arange(side ** 3, dtype=np.uint64)
    order = np.argsort(rand_u64(seed, 0, cells), kind="stable")[:num_voxels]
    cell = np.sort(order).astype(np.int64)
    half = side // 2
    keys = np.stack([cell % side - half, (cell // side) % side - half, cell // (side * side) - half],
                    axis=1).astype(np.int32)
    idx = cell.astype(np.uint64)
    u = np.stack([0.1 + 0.8 * rand_unit(seed, 1 + a, idx) for a in range(3)], axis=1)
    means = (keys.astype(np.float64) + u) * voxel_size
    covs = disc_covariances(seed, 16, idx)
    return VoxelMap(voxel_size, side, keys, np.ascontiguousarray(means), covs)


def make_uniform_scan(num_points: int, vmap: VoxelMap, seed: int = SCAN_SEED):
    """The headline 'uniform-random' scan: points ~ U[lo, hi)^3 over the map extent."""
    idx = np.arange(num_points, dtype=np.uint64)
    span = vmap.hi - vmap.lo
    pts = np.stack([vmap.lo + span * rand_unit(seed, 1 + a, idx) for a in range(3)], axis=1)
    return np.ascontiguousarray(pts), disc_covariances(seed, 16, idx)


def _apply(T: np.ndarray, pts: np.ndarray) -> np.ndarray:
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    return np.stack([T[r, 0] * x + T[r, 1] * y + T[r, 2] * z + T[r, 3] for r in range(3)], axis=1)


def _conjugate(R: np.ndarray, covs: np.ndarray) -> np.ndarray:
    """R C R^T on N x 9 column-major covariances with a fixed evaluation order."""
    C = covs.reshape(-1, 3, 3).transpose(0, 2, 1)  # C[n, r, c]
    RC = np.stack([np.stack([R[r, 0] * C[:, 0, c] + R[r, 1] * C[:, 1, c] + R[r, 2] * C[:, 2, c]
                             for c in range(3)], axis=1) for r in range(3)], axis=1)
    out = np.stack([np.stack([RC[:, r, 0] * R[c, 0] + RC[:, r, 1] * R[c, 1] + RC[:, r, 2] * R[c, 2]
                              for c in range(3)], axis=1) for r in range(3)], axis=1)
    return np.ascontiguousarray(out.transpose(0, 2, 1).reshape(-1, 9))


def invert_pose(T: np.ndarray) -> np.ndarray:
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -(T[:3, :3].T @ T[:3, 3])
    return Ti


def make_structured_scan(num_points: int, vmap: VoxelMap, seed: int = SCAN_SEED,
                         noise: float = 0.01, true_xi=TRUE_XI):
    """Parity/KAT scan: p_i = T_true^-1 (mean_v(i) + eps), C_i = R_true^T cov_v(i) R_true, with
    v(i) drawn without replacement (num_points <= V) and eps ~ Irwin-Hall(12) * noise."""
    V = vmap.keys.shape[0]
    if num_points > V:
        raise ValueError("structured scan needs num_points <= number of voxels")
    vid = np.arange(V, dtype=np.uint64)
    pick = np.sort(np.argsort(rand_u64(seed, 64, vid), kind="stable")[:num_points])
    idx = np.arange(num_points, dtype=np.uint64)
    eps = np.zeros((num_points, 3))
    for a in range(3):
        acc = np.zeros(num_points)
        for k in range(12):
            acc = acc + rand_unit(seed, 100 + 12 * a + k, idx)
        eps[:, a] = (acc - 6.0) * noise
    T_true = se3_to_SE3(true_xi)
    T_inv = invert_pose(T_true)
    pts = _apply(T_inv, vmap.means[pick] + eps)
    covs = _conjugate(T_inv[:3, :3], vmap.covs[pick])
    return np.ascontiguousarray(pts), covs, T_true


def make_lidar_scan(num_points: int, seed: int = SCAN_SEED, extent: float = 40.0, noise: float = 0.02):
    """Raw-scan stand-in for the scan preparation (CloudPreprocessor.cpp:76-127): a ground plane, two
    walls and a sparse cloud of clutter, denser near the sensor like a spinning LiDAR's returns, with
    Irwin-Hall(4) range noise. Deterministic in (num_points, seed)."""
    idx = np.arange(num_points, dtype=np.uint64)
    u = [rand_unit(seed, 200 + k, idx) for k in range(8)]
    kind = u[0]
    r = extent * u[1] ** 2 + 0.5                    # quadratic: density falls with range
    phi = 2.0 * np.pi * u[2]
    x, y = r * np.cos(phi), r * np.sin(phi)
    z = np.zeros(num_points)
    wall_a = (kind >= 0.55) & (kind < 0.75)
    wall_b = (kind >= 0.75) & (kind < 0.92)
    clutter = kind >= 0.92
    z = np.where(wall_a | wall_b, 6.0 * u[3], z)
    y = np.where(wall_a, 7.5 + 0.0 * y, y)
    x = np.where(wall_a, extent * (2.0 * u[4] - 1.0), x)
    x = np.where(wall_b, -11.0 + 0.0 * x, x)
    y = np.where(wall_b, extent * (2.0 * u[4] - 1.0), y)
    x = np.where(clutter, extent * (2.0 * u[4] - 1.0), x)
    y = np.where(clutter, extent * (2.0 * u[5] - 1.0), y)
    z = np.where(clutter, 8.0 * u[3], z)
    eps = np.zeros((num_points, 3))
    for a in range(3):
        acc = np.zeros(num_points)
        for k in range(4):
            acc = acc + rand_unit(seed, 220 + 4 * a + k, idx)
        eps[:, a] = (acc - 2.0) * noise
    return np.ascontiguousarray(np.stack([x, y, z], axis=1) + eps)


def make_imu_states(num_states: int, t0: float = 100.0, rate_hz: float = 400.0, seed: int = SCAN_SEED):
    """IMU-rate state queue for the deskew (ErrorStateKF::getStates, reference Types.hpp:31-40): a smooth
    motion sampled at rate_hz. -> num_states x 8: timestamp, position xyz, attitude quaternion x y z w."""
    k = np.arange(num_states, dtype=np.float64)
    t = t0 + k / rate_hz
    tau = t - t0
    idx = np.arange(6, dtype=np.uint64)
    a = rand_unit(seed, 300, idx) - 0.5
    pos = np.stack([2.0 * tau + 0.3 * a[0] * np.sin(7.0 * tau), 0.5 * a[1] * tau + 0.1 * np.sin(5.0 * tau),
                    0.05 * np.sin(11.0 * tau + a[2])], axis=1)
    rv = np.stack([0.2 * a[3] * tau, 0.1 * np.sin(3.0 * tau) * (0.5 + a[4]), 0.6 * tau * (0.5 + a[5])], axis=1)
    ang = np.linalg.norm(rv, axis=1)
    axis = np.where(ang[:, None] > 0, rv / np.maximum(ang, 1e-300)[:, None], np.array([[0.0, 0.0, 1.0]]))
    quat = np.concatenate([axis * np.sin(0.5 * ang)[:, None], np.cos(0.5 * ang)[:, None]], axis=1)
    return np.ascontiguousarray(np.concatenate([t[:, None], pos, quat], axis=1))


def make_point_times(num_points: int, t_first: float, t_last: float, seed: int = SCAN_SEED, jitter: float = 0.0):
    """Per-point capture times of one sweep: ascending from t_first to t_last; jitter > 0 perturbs them
    (firing-order noise), which makes the sequence locally non-monotonic like a real multi-beam sensor."""
    idx = np.arange(num_points, dtype=np.uint64)
    t = t_first + (t_last - t_first) * (np.arange(num_points) / max(num_points - 1, 1))
    if jitter > 0.0:
        t = t + jitter * (rand_unit(seed, 310, idx) - 0.5)
        t[-1] = t_last
    return np.ascontiguousarray(t)


# ---- a moving sensor: IMU + LiDAR stream for the replay harness (eskf_lio_amd/replay.py) -------------
STREAM_T0 = 1000.0          # absolute time of the end of the first sweep
STREAM_GRAVITY = np.array([0.0, 0.0, 9.805])   # the reference's convention: v' = R a + g (ErrorStateKF.cpp:96-98)


def stream_pose(tau):
    """Pose of the IMU in the world, tau seconds after the end of the first sweep (at rest before it).
    -> (R [..., 3, 3], p [..., 3], world acceleration [..., 3]); vectorised over tau."""
    tau = np.asarray(tau, dtype=np.float64)
    s = np.maximum(tau, 0.0)
    moving = (tau > 0.0).astype(np.float64)
    amp = np.array([1.5, 0.5, 0.05])
    om = np.array([2.0, 1.4, 2.6])
    p = amp * (1.0 - np.cos(om * s[..., None]))
    acc = amp * om * om * np.cos(om * s[..., None]) * moving[..., None]
    yaw, pitch, roll = 0.25 * (1 - np.cos(1.6 * s)), 0.03 * (1 - np.cos(2.2 * s)), 0.02 * (1 - np.cos(1.4 * s))
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.empty(tau.shape + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = -sp, cp * sr, cp * cr
    return R, p, acc


def make_sensor_stream(frames: int = 5, points_per_frame: int = 8_000, seed: int = SCAN_SEED,
                       world_points: int = 60_000, imu_rate: float = 400.0, sweep: float = 0.1):
    """-> (events, truth): events = [(arrival time, ("imu", t, gyro, accel) | ("lidar", points float32-exact,
    pointTime))] sorted by arrival; truth = [(end time, 4x4 pose)] per LiDAR frame. The sensor sits still
    until the first sweep ends, then moves along stream_pose(); every LiDAR point is seen from the pose at
    its own capture time, so later sweeps are motion-distorted like a real spinning sensor's."""
    world = make_lidar_scan(world_points, seed=seed, extent=25.0)
    world[:, 2] -= 1.2                                            # the IMU rides 1.2 m above the ground
    events, truth = [], []
    n_imu = int(round((0.02 + sweep * (frames - 1) + 0.0125) * imu_rate)) + 1
    tau_imu = -0.0213 + np.arange(n_imu) / imu_rate               # never exactly on a sweep boundary
    R, _, acc = stream_pose(tau_imu)
    d = 1e-5
    Rm, _, _ = stream_pose(tau_imu - d)
    Rp, _, _ = stream_pose(tau_imu + d)
    dR = np.einsum("nji,njk->nik", Rm, Rp)                        # R(t-d)^T R(t+d) ~ I + 2 d [w]x
    gyro = np.stack([dR[:, 2, 1] - dR[:, 1, 2], dR[:, 0, 2] - dR[:, 2, 0], dR[:, 1, 0] - dR[:, 0, 1]], axis=1) / (4 * d)
    accel = np.einsum("nji,nj->ni", R, acc - STREAM_GRAVITY)      # R^T (a - g)
    for t, w, a in zip(tau_imu, gyro, accel):
        events.append((STREAM_T0 + t, ("imu", STREAM_T0 + t, w.copy(), a.copy())))
    idx_all = np.arange(world_points, dtype=np.uint64)
    for k in range(frames):
        end = sweep * k
        pick = np.sort(np.argsort(rand_u64(seed + 17 * (k + 1), 400, idx_all), kind="stable")[:points_per_frame])
        tau = np.linspace(end - sweep + 1e-4, end, points_per_frame)
        Rk, pk, _ = stream_pose(tau)
        local = np.einsum("nji,nj->ni", Rk, world[pick] - pk)     # R(t_i)^T (W_i - p(t_i))
        local = local.astype(np.float32).astype(np.float64)       # what a PointCloud2 carries
        events.append((STREAM_T0 + end + 5e-4, ("lidar", np.ascontiguousarray(local), STREAM_T0 + tau)))
        Re, pe, _ = stream_pose(np.array(end))
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = Re, pe
        truth.append((STREAM_T0 + end, T))
    events.sort(key=lambda e: e[0])
    return events, truth


# The configurations BASELINE.json names (C1, C2, C5): points, voxels.
CONFIGS = {
    "C1": (5_000, 50_000),
    "C2": (100_000, 1_000_000),
    "C5": (1_000_000, 10_000_000),
}


# ---- a drive along a street seen by a 32-ring spinning LiDAR: the stand-in for BASELINE config C4 ----------------
# (HILTI exp21 is a Hesai PandarXT-32 carried through a building site: config/hilti_config.yaml:20-26; the recording is
# not available offline.)  32 rings x 2 000 azimuth steps at 10 Hz = 64 000 rays per sweep, every ray cast from the pose
# the sensor has AT ITS OWN FIRING TIME into a world of axis-aligned boxes over a ground plane, returned in the LiDAR
# frame as float32 with range noise: ~60 000 returns per sweep, motion-distorted like a real sweep.  The sensor drives
# > 200 m in 30 s, so a local map with the reference's 100 m eviction radius grows, is evicted from, and rehashes.
DRIVE_RINGS, DRIVE_AZIMUTHS = 32, 2000
DRIVE_SENSOR_HEIGHT = 1.6


def hilti_lidar_extrinsic() -> np.ndarray:
    """sensors.lidar.extrinsics of config/hilti_config.yaml:20-25 as the 4x4 the reference builds from them
    (include/ESKF_LIO/CloudPreprocessor.hpp:20-28): quaternion (x, y, z, w) = (0.7071068, -0.7071068, 0, 0), normalised."""
    q = np.array([0.7071068, -0.7071068, 0.0, 0.0])
    x, y, z, w = q / np.linalg.norm(q)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [-0.001, -0.00855, 0.055]
    return T


def drive_world(seed: int = SCAN_SEED, length: float = 290.0) -> np.ndarray:
    """-> boxes [B, 6] (xmin, ymin, zmin, xmax, ymax, zmax), ground at z = -DRIVE_SENSOR_HEIGHT: buildings of varying
    depth, height and setback on both sides of a street along +x, parked cars, poles, kiosks."""
    g = -DRIVE_SENSOR_HEIGHT
    boxes = []
    idx = np.arange(4096, dtype=np.uint64)
    u = [rand_unit(seed, 500 + k, idx) for k in range(10)]
    for side in (1.0, -1.0):
        x, k = -40.0, (0 if side > 0 else 2048)
        while x < length:
            w, gap = 8.0 + 14.0 * u[0][k], 1.5 + 4.5 * u[1][k]
            setback, depth, height = 8.0 + 4.0 * u[2][k], 6.0 + 10.0 * u[3][k], 5.0 + 11.0 * u[4][k]
            y0, y1 = sorted((side * setback, side * (setback + depth)))
            boxes.append([x, y0, g, x + w, y1, g + height])
            x += w + gap
            k += 1
    for j in range(int(length / 11.0) + 4):                   # parked cars, alternating sides, irregular spacing
        side = 1.0 if j % 2 == 0 else -1.0
        cx = -35.0 + 11.0 * j + 4.0 * u[5][j]
        cy = side * (5.2 + 0.8 * u[6][j])
        boxes.append([cx, cy - 0.9, g, cx + 4.2, cy + 0.9, g + 1.45 + 0.3 * u[7][j]])
    for j in range(int(length / 23.0) + 3):                   # poles and kiosks
        cx = -30.0 + 23.0 * j + 6.0 * u[8][j]
        boxes.append([cx, 6.9, g, cx + 0.3, 7.2, g + 7.0])
        boxes.append([cx + 9.0, -7.2, g, cx + 9.3, -6.9, g + 7.0])
        if j % 3 == 1:
            boxes.append([cx + 4.0, 6.2 + u[9][j], g, cx + 6.5, 7.9 + u[9][j], g + 2.6])
    return np.ascontiguousarray(np.array(boxes, dtype=np.float64))


def drive_pose(tau):
    """Pose of the IMU in the world tau seconds after the end of the first sweep: at rest before, then accelerating to
    7.5 m/s along the street with a lateral sway and small attitude oscillations.  -> (R [..., 3, 3], p [..., 3])."""
    tau = np.asarray(tau, dtype=np.float64)
    s = np.maximum(tau, 0.0)
    v, T = 7.5, 2.0
    p = np.stack([v * (s - T * (1.0 - np.exp(-s / T))), 1.2 * (1.0 - np.cos(0.35 * s)), 0.05 * (1.0 - np.cos(1.1 * s))], axis=-1)
    yaw, pitch, roll = 0.12 * (1 - np.cos(0.3 * s)), 0.02 * (1 - np.cos(0.9 * s)), 0.015 * (1 - np.cos(0.7 * s))
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.empty(tau.shape + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = -sp, cp * sr, cp * cr
    return R, p


def _cast(origin, direction, boxes, max_range):
    """Nearest hit of rays (origin [N, 3], unit direction [N, 3]) with the ground plane z = -DRIVE_SENSOR_HEIGHT and the
    boxes; -> range [N], inf where nothing is hit within max_range.  The sensor drives between the two rows of boxes
    (|y| < 3 m, boxes at |y| > 4 m), so a box is only tested against the rays that point to its side."""
    n = origin.shape[0]
    best = np.full(n, np.inf)
    dz = direction[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (-DRIVE_SENSOR_HEIGHT - origin[:, 2]) / dz
    best = np.where((dz < 0.0) & (tg > 0.0), tg, best)
    d = np.where(np.abs(direction) < 1e-12, 1e-12, direction)
    inv = 1.0 / d
    lo_x, hi_x = origin[:, 0].min() - max_range, origin[:, 0].max() + max_range
    near = boxes[(boxes[:, 3] >= lo_x) & (boxes[:, 0] <= hi_x)]
    for side in (1.0, -1.0):
        rays = np.nonzero(direction[:, 1] * side > 0.0)[0]
        o, iv, b_best = origin[rays], inv[rays], best[rays]
        for b in near[(near[:, 1] if side > 0 else -near[:, 4]) > 3.0]:
            t1 = (b[:3] - o) * iv
            t2 = (b[3:] - o) * iv
            tmin = np.minimum(t1, t2).max(axis=1)
            tmax = np.maximum(t1, t2).min(axis=1)
            b_best = np.where((tmax >= tmin) & (tmin > 0.0) & (tmin < b_best), tmin, b_best)
        best[rays] = b_best
    return np.where(best <= max_range, best, np.inf)


def drive_truth(frames: int, sweep: float = 0.1):
    """[(end time, 4x4 IMU pose)] of the drive's sweeps."""
    out = []
    for k in range(frames):
        Re, pe = drive_pose(np.array(sweep * k))
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = Re, pe
        out.append((STREAM_T0 + sweep * k, T))
    return out


def iter_drive_stream(frames: int = 300, seed: int = SCAN_SEED, imu_rate: float = 400.0, sweep: float = 0.1,
                      max_range: float = 60.0, noise: float = 0.01, extrinsic=None):
    """Yields (arrival time, ("imu", t, gyro, accel) | ("lidar", points float32-exact in the LiDAR frame, pointTime)) in
    arrival order (a generator: 300 sweeps of 57 000 points are 0.5 GB).  extrinsic: LiDAR -> IMU (default:
    hilti_lidar_extrinsic()).  Deterministic in its arguments."""
    T_il = hilti_lidar_extrinsic() if extrinsic is None else np.asarray(extrinsic, dtype=np.float64)
    boxes = drive_world(seed)
    n_imu = int(round((0.02 + sweep * (frames - 1) + 0.0125) * imu_rate)) + 1
    tau_imu = -0.0213 + np.arange(n_imu) / imu_rate
    R, p0 = drive_pose(tau_imu)
    d = 1e-5
    Rm, _ = drive_pose(tau_imu - d)
    Rp, _ = drive_pose(tau_imu + d)
    dR = np.einsum("nji,njk->nik", Rm, Rp)
    gyro = np.stack([dR[:, 2, 1] - dR[:, 1, 2], dR[:, 0, 2] - dR[:, 2, 0], dR[:, 1, 0] - dR[:, 0, 1]], axis=1) / (4 * d)
    da = 1e-3                                                     # world acceleration by a central second difference
    _, pm = drive_pose(tau_imu - da)
    _, pp = drive_pose(tau_imu + da)
    acc = (pp - 2.0 * p0 + pm) / (da * da)
    accel = np.einsum("nji,nj->ni", R, acc - STREAM_GRAVITY)
    # the firing pattern: azimuth column j fires all 32 rings at once (elevations -16 .. +15 degrees)
    el = np.deg2rad(np.arange(DRIVE_RINGS, dtype=np.float64) - 16.0)
    az = 2.0 * np.pi * np.arange(DRIVE_AZIMUTHS, dtype=np.float64) / DRIVE_AZIMUTHS
    d_l = np.stack([np.cos(el)[None, :] * np.cos(az)[:, None], np.cos(el)[None, :] * np.sin(az)[:, None],
                    np.broadcast_to(np.sin(el)[None, :], (DRIVE_AZIMUTHS, DRIVE_RINGS))], axis=-1)   # [A, 32, 3]
    ray = np.arange(DRIVE_AZIMUTHS * DRIVE_RINGS, dtype=np.uint64)
    next_imu = 0
    for k in range(frames):
        end = sweep * k
        arrival = STREAM_T0 + end + 5e-4
        while next_imu < n_imu and STREAM_T0 + tau_imu[next_imu] <= arrival:
            t = STREAM_T0 + tau_imu[next_imu]
            yield t, ("imu", t, gyro[next_imu].copy(), accel[next_imu].copy())
            next_imu += 1
        tau = end - sweep + sweep * (np.arange(DRIVE_AZIMUTHS) + 1.0) / DRIVE_AZIMUTHS     # column j fires at tau[j]
        Rk, pk = drive_pose(tau)
        Rw = np.einsum("aij,jk->aik", Rk, T_il[:3, :3])                                       # LiDAR -> world, per column
        origin = np.einsum("aij,j->ai", Rk, T_il[:3, 3]) + pk
        d_w = np.einsum("aij,arj->ari", Rw, d_l)
        rng = _cast(np.repeat(origin, DRIVE_RINGS, axis=0), d_w.reshape(-1, 3), boxes, max_range)
        keep = np.isfinite(rng) & (rng > 0.5)
        eps = (rand_unit(seed + 31 * (k + 1), 520, ray) + rand_unit(seed + 31 * (k + 1), 521, ray) +
               rand_unit(seed + 31 * (k + 1), 522, ray) + rand_unit(seed + 31 * (k + 1), 523, ray) - 2.0) * noise
        local = d_l.reshape(-1, 3)[keep] * (rng[keep] + eps[keep])[:, None]
        local = local.astype(np.float32).astype(np.float64)       # what a PointCloud2 carries
        t_pts = np.repeat(STREAM_T0 + tau, DRIVE_RINGS)[keep]
        yield arrival, ("lidar", np.ascontiguousarray(local), np.ascontiguousarray(t_pts))
    while next_imu < n_imu:
        t = STREAM_T0 + tau_imu[next_imu]
        yield t, ("imu", t, gyro[next_imu].copy(), accel[next_imu].copy())
        next_imu += 1
