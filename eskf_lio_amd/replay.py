"""ROS-free replay of the reference's frame loop.

The reference runs as three ROS 2 nodes fed by a rosbag2 player (launch/eskf_lio.launch.py:11-13):
`/alphasense/imu` (sensor_msgs/Imu, 400 Hz) and `/hesai/pandar` (sensor_msgs/PointCloud2 with float32
x, y, z and a float64 `timestamp` per point) go through Subscriber.hpp:38-52,80-103 into two queues that
Odometry::run drains (src/Odometry.cpp:9-110).  Neither ROS 2 nor the HILTI bag exist here, so this
module provides

  * the wire formats: a reader (and, for fixtures, a writer) of the rosbag2 sqlite3 storage with the CDR
    encodings of those two message types, and a TUM trajectory writer,
  * the estimator as the reference has it on the host: the 18-state error-state Kalman filter of
    src/ErrorStateKF.cpp, restated with numpy — it is not on the hot path and stays on the CPU,
  * a single-threaded frame loop with the ordering of src/Odometry.cpp:24-87,

all written against a small backend interface (scan preparation, registration, map update), so that the
SAME loop can be driven by the MI355X path (`GpuBackend`: the C++ mirror of CloudPreprocessor / ICP /
LocalMap over libvgicp_hip.so) and, in tests/, by the CPU oracle; the two trajectories are then compared
frame by frame.  Nothing here imports the oracle.
"""
from __future__ import annotations

import os
import sqlite3
import struct
import time
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Tuple

import numpy as np

IMU_TOPIC = "/alphasense/imu"      # config/hilti_config.yaml:3
LIDAR_TOPIC = "/hesai/pandar"      # config/hilti_config.yaml:20
IMU_TYPE = "sensor_msgs/msg/Imu"
CLOUD_TYPE = "sensor_msgs/msg/PointCloud2"
_FLOAT32, _FLOAT64 = 7, 8          # sensor_msgs/PointField datatypes


# ---- messages as the subscribers hand them to the queues (Types.hpp:14-29) ---------------------------
@dataclass
class ImuMeasurement:
    timestamp: float
    angularVelocity: np.ndarray
    acceleration: np.ndarray


@dataclass
class LidarMeasurement:
    points: np.ndarray             # N x 3 float64 (cloud->points_)
    pointTime: np.ndarray          # N float64
    startTime: float = 0.0
    endTime: float = 0.0
    covariances: Optional[np.ndarray] = None

    def __post_init__(self):
        if len(self.pointTime):
            self.startTime = float(self.pointTime[0])      # Subscriber.hpp:97-98
            self.endTime = float(self.pointTime[-1])


# ---- CDR (little endian, XCDR1 as rosbag2 stores it) ------------------------------------------------
class _CdrReader:
    def __init__(self, blob: bytes):
        # encapsulation header: representation identifier {0x00, 0x00 = CDR_BE | 0x01 = CDR_LE}, then two option bytes.
        # Parameter-list CDR (0x0002 / 0x0003) and the XCDR2 identifiers (0x0006 ...) are other wire formats: refused.
        if len(blob) < 4 or blob[0] != 0 or blob[1] not in (0, 1):
            raise ValueError("not a plain-CDR payload (encapsulation header %s)" % bytes(blob[:4]).hex())
        self.little = blob[1] == 1
        self.buf = memoryview(blob)[4:]
        self.pos = 0

    def _align(self, size: int):
        self.pos = (self.pos + size - 1) & ~(size - 1)

    def scalar(self, fmt: str):
        size = struct.calcsize(fmt)
        self._align(size)
        (v,) = struct.unpack_from(("<" if self.little else ">") + fmt, self.buf, self.pos)
        self.pos += size
        return v

    def string(self) -> str:
        n = self.scalar("I")
        raw = bytes(self.buf[self.pos:self.pos + n])
        self.pos += n
        return raw[:-1].decode() if n else ""

    def array(self, dtype: str, count: int) -> np.ndarray:
        dt = np.dtype(dtype).newbyteorder("<" if self.little else ">")
        self._align(dt.itemsize)
        out = np.frombuffer(self.buf, dtype=dt, count=count, offset=self.pos)
        self.pos += dt.itemsize * count
        return out

    def header(self) -> Tuple[float, str]:
        sec, nsec = self.scalar("i"), self.scalar("I")
        return sec + 1e-9 * nsec, self.string()            # rclcpp::Time(stamp).seconds()


class _CdrWriter:
    def __init__(self):
        self.buf = bytearray()

    def _align(self, size: int):
        self.buf.extend(b"\0" * ((-len(self.buf)) % size))

    def scalar(self, fmt: str, v):
        self._align(struct.calcsize(fmt))
        self.buf.extend(struct.pack("<" + fmt, v))

    def string(self, s: str):
        raw = s.encode() + b"\0"
        self.scalar("I", len(raw))
        self.buf.extend(raw)

    def array(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        self._align(a.dtype.itemsize)
        self.buf.extend(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())

    def header(self, stamp: float, frame_id: str):
        sec = int(np.floor(stamp))
        self.scalar("i", sec)
        self.scalar("I", min(int(round((stamp - sec) * 1e9)), 999_999_999))
        self.string(frame_id)

    def payload(self) -> bytes:
        return b"\x00\x01\x00\x00" + bytes(self.buf)


def decode_imu(blob: bytes) -> ImuMeasurement:
    """sensor_msgs/Imu -> ImuMeasurement as ImuSubscriber::imuCallback does (Subscriber.hpp:38-52)."""
    r = _CdrReader(blob)
    stamp, _ = r.header()
    r.array("f8", 4)               # orientation
    r.array("f8", 9)
    w = r.array("f8", 3).astype(np.float64)
    r.array("f8", 9)
    a = r.array("f8", 3).astype(np.float64)
    return ImuMeasurement(stamp, w.copy(), a.copy())


def encode_imu(m: ImuMeasurement, frame_id: str = "imu") -> bytes:
    w = _CdrWriter()
    w.header(m.timestamp, frame_id)
    w.array(np.array([0.0, 0.0, 0.0, 1.0]))
    w.array(np.zeros(9))
    w.array(np.asarray(m.angularVelocity, dtype=np.float64))
    w.array(np.zeros(9))
    w.array(np.asarray(m.acceleration, dtype=np.float64))
    w.array(np.zeros(9))
    return w.payload()


def decode_pointcloud2(blob: bytes) -> LidarMeasurement:
    """sensor_msgs/PointCloud2 -> LidarMeasurement as LidarSubscriber::cloudCallback does
    (Subscriber.hpp:80-103): float32 x, y, z widened to double, float64 `timestamp` per point."""
    r = _CdrReader(blob)
    r.header()
    height, width = r.scalar("I"), r.scalar("I")
    fields = {}
    for _ in range(r.scalar("I")):
        name = r.string()
        offset, datatype, count = r.scalar("I"), r.scalar("B"), r.scalar("I")
        fields[name] = (offset, datatype, count)
    big = bool(r.scalar("B"))
    point_step, _row_step = r.scalar("I"), r.scalar("I")
    data = r.array("u1", r.scalar("I"))
    n = height * width
    order = ">" if big else "<"
    raw = np.frombuffer(data, dtype=np.uint8, count=n * point_step).reshape(n, point_step)

    def column(name, want, dtype):
        if name not in fields or fields[name][1] != want:
            raise ValueError(f"PointCloud2 has no {dtype} field '{name}'")
        off = fields[name][0]
        width_b = np.dtype(dtype).itemsize
        return np.ascontiguousarray(raw[:, off:off + width_b]).view(np.dtype(dtype).newbyteorder(order)).reshape(n)

    pts = np.stack([column(a, _FLOAT32, "f4").astype(np.float64) for a in ("x", "y", "z")], axis=1)
    t = column("timestamp", _FLOAT64, "f8").astype(np.float64)
    return LidarMeasurement(np.ascontiguousarray(pts), np.ascontiguousarray(t))


def pointcloud2_payload(blob: bytes):
    """sensor_msgs/PointCloud2 -> (payload uint8 array, n, point_step, off_x, off_y, off_z, off_timestamp): what
    vgicp_sweep_stage_cloud2 takes -- the records stay as the sensor wrote them, the DEVICE widens the float32
    coordinates (Subscriber.hpp:89-97 does it on the host, one point at a time).  Little-endian messages only."""
    r = _CdrReader(blob)
    r.header()
    height, width = r.scalar("I"), r.scalar("I")
    fields = {}
    for _ in range(r.scalar("I")):
        name = r.string()
        offset, datatype, count = r.scalar("I"), r.scalar("B"), r.scalar("I")
        fields[name] = (offset, datatype, count)
    if bool(r.scalar("B")):
        raise ValueError("big-endian PointCloud2: decode_pointcloud2 converts it on the host")
    point_step, _row_step = r.scalar("I"), r.scalar("I")
    data = r.array("u1", r.scalar("I"))
    for name, want in (("x", _FLOAT32), ("y", _FLOAT32), ("z", _FLOAT32), ("timestamp", _FLOAT64)):
        if name not in fields or fields[name][1] != want:
            raise ValueError(f"PointCloud2 has no field '{name}' of the expected type")
    n = height * width
    payload = np.frombuffer(data, dtype=np.uint8, count=n * point_step)
    return payload, n, point_step, fields["x"][0], fields["y"][0], fields["z"][0], fields["timestamp"][0]


def encode_pointcloud2(points: np.ndarray, point_time: np.ndarray, frame_id: str = "PandarXT-32") -> bytes:
    """The layout of the Hesai driver's cloud as far as the reference reads it: x y z float32 at 0/4/8,
    intensity float32 at 12 (unused), timestamp float64 at 16, ring uint16 at 24 (unused), 32-byte points."""
    n = len(points)
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"),
                             ("timestamp", "<f8"), ("ring", "<u2"), ("pad", "V6")])
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    rec["timestamp"] = point_time
    w = _CdrWriter()
    w.header(float(point_time[-1]) if n else 0.0, frame_id)
    w.scalar("I", 1)
    w.scalar("I", n)
    layout = [("x", 0, _FLOAT32), ("y", 4, _FLOAT32), ("z", 8, _FLOAT32), ("intensity", 12, _FLOAT32),
              ("timestamp", 16, _FLOAT64), ("ring", 24, 4)]
    w.scalar("I", len(layout))
    for name, off, dt in layout:
        w.string(name)
        w.scalar("I", off)
        w.scalar("B", dt)
        w.scalar("I", 1)
    w.scalar("B", 0)
    w.scalar("I", 32)
    w.scalar("I", 32 * n)
    w.scalar("I", 32 * n)
    w.buf.extend(rec.tobytes())
    w.scalar("B", 1)
    return w.payload()


# ---- rosbag2 (sqlite3 storage plugin: tables `topics` and `messages`) --------------------------------
def write_rosbag2(path: str, events: Iterable[Tuple[float, object]], imu_topic: str = IMU_TOPIC,
                  lidar_topic: str = LIDAR_TOPIC):
    """events: (arrival time in seconds, ImuMeasurement | LidarMeasurement) in any order."""
    if os.path.exists(path):
        os.remove(path)
    db = sqlite3.connect(path)
    db.execute("CREATE TABLE topics(id INTEGER PRIMARY KEY, name TEXT NOT NULL, type TEXT NOT NULL, "
               "serialization_format TEXT NOT NULL, offered_qos_profiles TEXT NOT NULL)")
    db.execute("CREATE TABLE messages(id INTEGER PRIMARY KEY, topic_id INTEGER NOT NULL, "
               "timestamp INTEGER NOT NULL, data BLOB NOT NULL)")
    db.execute("INSERT INTO topics VALUES (1, ?, ?, 'cdr', '')", (imu_topic, IMU_TYPE))
    db.execute("INSERT INTO topics VALUES (2, ?, ?, 'cdr', '')", (lidar_topic, CLOUD_TYPE))
    rows = []
    for arrival, m in events:
        if isinstance(m, ImuMeasurement):
            rows.append((1, int(round(arrival * 1e9)), encode_imu(m)))
        else:
            rows.append((2, int(round(arrival * 1e9)), encode_pointcloud2(m.points, m.pointTime)))
    db.executemany("INSERT INTO messages(topic_id, timestamp, data) VALUES (?, ?, ?)", rows)
    db.commit()
    db.close()


def read_rosbag2(path: str, imu_topic: str = IMU_TOPIC, lidar_topic: str = LIDAR_TOPIC):
    """-> list of (arrival time in seconds, ImuMeasurement | LidarMeasurement), in the order a player
    publishes them (bag timestamp, then insertion order)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    db = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    topics = {tid: (name, typ) for tid, name, typ in db.execute("SELECT id, name, type FROM topics")}
    out = []
    for tid, stamp, data in db.execute("SELECT topic_id, timestamp, data FROM messages ORDER BY timestamp, id"):
        name, typ = topics.get(tid, (None, None))
        if name == imu_topic and typ == IMU_TYPE:
            out.append((stamp * 1e-9, decode_imu(data)))
        elif name == lidar_topic and typ == CLOUD_TYPE:
            out.append((stamp * 1e-9, decode_pointcloud2(data)))
    db.close()
    return out


def write_tum(path: str, trajectory: List[Tuple[float, np.ndarray]]):
    """timestamp tx ty tz qx qy qz qw per line."""
    with open(path, "w") as f:
        for stamp, T in trajectory:
            q = matrix_to_quat(T[:3, :3])
            f.write(f"{stamp:.9f} {T[0, 3]:.9f} {T[1, 3]:.9f} {T[2, 3]:.9f} "
                    f"{q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}\n")


# ---- rotations with Eigen's formulas (quaternions as x, y, z, w) ------------------------------------
def quat_to_matrix(q):
    x, y, z, w = q
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def matrix_to_quat(m):
    """Eigen::Quaterniond(Matrix3d)."""
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def quat_multiply(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def angle_axis_to_quat(angle, axis):
    return np.concatenate([np.sin(0.5 * angle) * axis, [np.cos(0.5 * angle)]])


def _normalized(v):
    z = float(v @ v)
    return v / np.sqrt(z) if z > 0.0 else v                 # Eigen's normalized()


def rotation_vector_to_quat(r):
    """Utils::rotationVectorToQuaternion (src/Utils.cpp:34-38)."""
    return angle_axis_to_quat(np.linalg.norm(r), _normalized(r))


def rotation_matrix_to_vector(R):
    """Utils::rotationMatrixToVector (src/Utils.cpp:22-26): Eigen::AngleAxisd(R)."""
    q = matrix_to_quat(R)
    n = np.linalg.norm(q[:3])
    if n == 0.0:
        return np.zeros(3)
    angle = 2.0 * np.arctan2(n, abs(q[3]))
    if q[3] < 0.0:
        n = -n
    return angle * (q[:3] / n)


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


# ---- the estimator (src/ErrorStateKF.cpp) ---------------------------------------------------------
@dataclass
class State:                                                # Types.hpp:31-40
    timestamp: float = 0.0
    position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    attitude: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0, 1.0]))
    biasAccel: np.ndarray = field(default_factory=lambda: np.zeros(3))
    biasGyro: np.ndarray = field(default_factory=lambda: np.zeros(3))
    gravity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    P: np.ndarray = field(default_factory=lambda: 1e-3 * np.eye(18))

    def copy(self):
        return State(self.timestamp, self.position.copy(), self.velocity.copy(), self.attitude.copy(),
                     self.biasAccel.copy(), self.biasGyro.copy(), self.gravity.copy(), self.P.copy())

    def pose(self):
        T = np.eye(4)
        T[:3, :3] = quat_to_matrix(self.attitude)
        T[:3, 3] = self.position
        return T


DEFAULT_CONFIG = {                                          # config/hilti_config.yaml
    "imu": dict(update_rate=400.0, bias_a=[0.0, 0.0, 0.0], bias_g=[0.0, 0.0, 0.0], gravity=[0.0, 0.0, 9.805],
                accel_noise_density=[105.0, 105.0, 135.0], accel_zero_g_offset=20.0, gyro_noise_density=0.014,
                gyro_zero_rate_offset=1.0),
    "lidar_extrinsic": np.eye(4),
    "kalman_filter": dict(translation_noise=1.0e-6, rotation_noise=1.0e-6),
    "local_map": dict(voxel_size=0.3, max_num_points_per_voxel=1000, translation_sq_threshold=1.0e-2,
                      cosine_threshold=0.985, remove_distant_points=False, distance_threshold=100.0,
                      removing_period=10.0),
    "cloud_preprocessor": dict(voxel_size=0.3),
    "registration": dict(max_iteration=100, translation_sq_threshold=1.0e-6, cosine_threshold=0.9999),
}


class ErrorStateKF:
    """src/ErrorStateKF.cpp, include/ESKF_LIO/ErrorStateKF.hpp:15-53; `align` is ICP::align bound to a map."""

    def __init__(self, config: dict, align):
        imu = config["imu"]
        G = 9.81
        rate = float(imu["update_rate"])
        init = State()
        init.biasAccel = np.array(imu["bias_a"], dtype=np.float64)
        init.biasGyro = np.array(imu["bias_g"], dtype=np.float64)
        init.gravity = np.array(imu["gravity"], dtype=np.float64)
        self.states_: List[State] = [init]
        self.imu_: List[ImuMeasurement] = []
        sigma_an = np.array(imu["accel_noise_density"], dtype=np.float64) * G * np.sqrt(rate)   # :30-32, as written
        sigma_gn = imu["gyro_noise_density"] * np.sqrt(rate) * np.pi / 180.0
        sigma_aw = imu["accel_zero_g_offset"] * np.sqrt(rate) * 1e-3 * G
        sigma_gw = imu["gyro_zero_rate_offset"] * np.sqrt(rate) * np.pi / 180.0
        self.Q_ = np.zeros((12, 12))
        self.Q_[0:3, 0:3] = np.diag(sigma_an ** 2)
        self.Q_[3:6, 3:6] = sigma_gn ** 2 * np.eye(3)
        self.Q_[6:9, 6:9] = sigma_aw ** 2 * np.eye(3)
        self.Q_[9:12, 9:12] = sigma_gw ** 2 * np.eye(3)
        self.F_i_ = np.zeros((18, 12))
        self.F_i_[3:15, 0:12] = np.eye(12)
        kf = config["kalman_filter"]
        self.V_ = np.zeros((6, 6))
        self.V_[0:3, 0:3] = kf["translation_noise"] * np.eye(3)
        self.V_[3:6, 3:6] = kf["rotation_noise"] * np.eye(3)
        self.H_ = np.zeros((6, 18))
        self.H_[0:3, 0:3] = np.eye(3)
        self.H_[3:6, 6:9] = np.eye(3)
        self.G_ = np.eye(18)
        self.F_x_ = np.eye(18)
        self.align_ = align

    def getStates(self):
        return self.states_

    def getLastStateTime(self):
        return self.states_[-1].timestamp

    def feedImu(self, imu):
        self.imu_.append(imu)

    def initialize(self, lidarEndTime):                     # :60-75
        self.states_[0].timestamp = lidarEndTime
        while self.imu_ and self.imu_[0].timestamp < lidarEndTime:
            self.imu_.pop(0)
        for imu in self.imu_:
            self.process(imu)

    def process(self, imu):                                 # :77-114
        prev = self.states_[-1]
        dt = imu.timestamp - prev.timestamp
        if dt < 0.0:
            return
        new = prev.copy()
        new.timestamp = imu.timestamp
        R = quat_to_matrix(prev.attitude)
        acc = imu.acceleration - prev.biasAccel
        w = imu.angularVelocity - prev.biasGyro
        dq = angle_axis_to_quat(np.linalg.norm(w) * dt, _normalized(w))
        dt2 = dt * dt
        world_acc = R @ acc + prev.gravity
        new.position = prev.position + prev.velocity * dt + 0.5 * world_acc * dt2
        new.velocity = prev.velocity + world_acc * dt
        new.attitude = quat_multiply(prev.attitude, dq)
        Q_i = self.Q_.copy()
        Q_i[0:6, 0:6] *= dt2
        Q_i[6:12, 6:12] *= dt
        F = self.F_x_
        F[0:3, 3:6] = np.eye(3) * dt
        F[3:6, 6:9] = -R @ skew(acc) * dt
        F[3:6, 9:12] = -R * dt
        F[3:6, 15:18] = np.eye(3) * dt
        F[6:9, 6:9] = quat_to_matrix(dq * np.array([-1.0, -1.0, -1.0, 1.0]))
        F[6:9, 12:15] = -np.eye(3) * dt
        new.P = F @ prev.P @ F.T + self.F_i_ @ Q_i @ self.F_i_.T
        self.states_.append(new)

    def update(self, lidar: LidarMeasurement):              # :116-164
        end = lidar.endTime
        while self.states_ and self.states_[-1].timestamp > end:
            self.states_.pop()
        prev = self.states_[-1]
        new = prev.copy()
        new.timestamp = end
        guess = prev.pose()
        observation = self.align_(lidar.points, lidar.covariances, guess)
        residual = np.zeros(6)
        residual[:3] = observation[:3, 3] - guess[:3, 3]
        residual[3:] = rotation_matrix_to_vector(guess[:3, :3].T @ observation[:3, :3])
        S = self.H_ @ prev.P @ self.H_.T + self.V_
        K = prev.P @ self.H_.T @ np.linalg.inv(S)
        err = K @ residual
        new.P = (np.eye(18) - K @ self.H_) @ prev.P
        # injectError (:166-174)
        new.position = new.position + err[0:3]
        new.velocity = new.velocity + err[3:6]
        new.attitude = quat_multiply(new.attitude, rotation_vector_to_quat(err[6:9]))
        new.biasAccel = new.biasAccel + err[9:12]
        new.biasGyro = new.biasGyro + err[12:15]
        new.gravity = new.gravity + err[15:18]
        # reset (:176-182)
        self.G_[6:9, 6:9] = np.eye(3) - 0.5 * skew(err[6:9])
        new.P = self.G_ @ new.P @ self.G_.T
        self.states_.append(new)
        while self.imu_ and self.imu_[0].timestamp < end:
            self.imu_.pop(0)
        for imu in self.imu_:
            self.process(imu)
        return new.pose()


def pack_states(states: List[State], earliest_point_time: Optional[float] = None) -> np.ndarray:
    """std::deque<State> -> S x 8 (timestamp, position, quaternion x y z w): what vgicp_deskew takes.
    The reference never trims its deque (400 states per second pile up); a state whose timestamp is not above
    the earliest capture time of the sweep cannot take a point in the deskew's walk, so with
    earliest_point_time given all but the last of those leading states are left out (same result)."""
    if earliest_point_time is not None:
        first = 0
        for k in range(len(states) - 1, -1, -1):             # states are in time order: scan from the back
            if states[k].timestamp <= earliest_point_time:
                first = k
                break
        states = states[first:]
    out = np.zeros((len(states), 8))
    for k, s in enumerate(states):
        out[k, 0] = s.timestamp
        out[k, 1:4] = s.position
        out[k, 4:8] = s.attitude
    return out


# ---- the frame loop (src/Odometry.cpp:9-110) ---------------------------------------------------------
class Odometry:
    """Single-threaded replay of Odometry::run: events are taken in arrival order, one per loop turn.
    backend: preprocess(states S x 8 | None, points, pointTime) -> (points, covs)   CloudPreprocessor::process
             align(points, covs, guess 4x4) -> 4x4                                   ICP::align
             update_map(points, covs, transform 4x4, initialize: bool)               LocalMap::updateLocalMap"""

    def __init__(self, config: dict, backend):
        self.backend = backend
        self.filter = ErrorStateKF(config, backend.align)
        self.initialized = False
        self.pending: List[LidarMeasurement] = []
        self.lidar: Optional[LidarMeasurement] = None
        self.trajectory: List[Tuple[float, np.ndarray]] = []
        self.frames = 0
        # the reference's three stage timers (Odometry.cpp:11-15,73-96): sums and maxima in seconds
        self.stage_seconds = {"cloud preprocessor": [0.0, 0.0], "filter update": [0.0, 0.0], "map update": [0.0, 0.0]}

    def _clock(self, stage: str, started: float):
        dt = time.perf_counter() - started
        self.stage_seconds[stage][0] += dt
        self.stage_seconds[stage][1] = max(self.stage_seconds[stage][1], dt)

    def report(self) -> str:
        """The lines Odometry::run prints at exit (Odometry.cpp:98-109)."""
        out = []
        for stage, (total, worst) in self.stage_seconds.items():
            out.append(f"{stage} average elapsed time = {1e3 * total / max(self.frames, 1):.6f} ms")
            out.append(f"{stage} max elapsed time = {1e3 * worst:.6f} ms")
        return "\n".join(out)

    def _try_frame(self) -> bool:
        if self.lidar is None and self.pending:
            self.lidar = self.pending.pop(0)
        if self.lidar is None:
            return False
        meas = self.lidar
        if not self.initialized:                            # :55-63
            self.initialized = True
            self.filter.initialize(meas.endTime)
            self.lidar = None
            pts, covs = self.backend.preprocess(None, meas.points, meas.pointTime)
            # Odometry.cpp:61 leaves `initialize` at its default (false); the first update inserts because no
            # previous transform exists yet (the reference reads prevTransform_ uninitialised there)
            self.backend.update_map(pts, covs, np.eye(4), False)
            self.trajectory.append((meas.endTime, np.eye(4)))
            return True
        if self.filter.getLastStateTime() < meas.endTime:   # :66-70: wait for the next IMU sample
            return False
        started = time.perf_counter()
        states = pack_states(self.filter.getStates(), float(np.min(meas.pointTime)))
        meas.points, meas.covariances = self.backend.preprocess(states, meas.points, meas.pointTime)
        self._clock("cloud preprocessor", started)
        started = time.perf_counter()
        transform = self.filter.update(meas)
        self._clock("filter update", started)
        started = time.perf_counter()
        self.lidar = None
        self.backend.update_map(meas.points, meas.covariances, transform, False)
        self._clock("map update", started)
        self.trajectory.append((meas.endTime, transform))
        self.frames += 1
        return True

    def run(self, events):
        for _, m in events:
            if isinstance(m, ImuMeasurement):               # :27-43
                if self.initialized:
                    self.filter.process(m)
                self.filter.feedImu(m)
            else:
                self.pending.append(m)
            while self._try_frame():
                pass
        return self.trajectory


class GpuBackend:
    """The MI355X path behind the C++ mirror of the reference's classes (eskf_lio_amd/host.py)."""

    def __init__(self, config: dict, device_resident_map: bool = False):
        from . import host
        lm = dict(config["local_map"])
        self.map = host.LocalMap(lm["voxel_size"], lm["max_num_points_per_voxel"],
                                 dict(translation_sq_threshold=lm["translation_sq_threshold"],
                                      cosine_threshold=lm["cosine_threshold"],
                                      remove_distant_points=lm["remove_distant_points"],
                                      distance_threshold=lm["distance_threshold"],
                                      removing_period=lm["removing_period"],
                                      device_resident=device_resident_map))
        reg = config["registration"]
        self.icp = host.ICP(reg["max_iteration"], reg["translation_sq_threshold"], reg["cosine_threshold"])
        self.pre = host.CloudPreprocessor(config["cloud_preprocessor"]["voxel_size"], config["lidar_extrinsic"])
        self.iterations: List[int] = []

    def preprocess(self, states, points, pointTime):
        if states is None:
            states = np.zeros((0, 8))
        return self.pre.process(states, points, pointTime)

    def align(self, points, covs, guess):
        T = self.icp.align(points, covs, self.map, guess)
        self.iterations.append(self.icp.iterations)
        return T

    def update_map(self, points, covs, transform, initialize):
        self.map.updateLocalMap(points, covs, transform, initialize)


class DeviceBackend:
    """The frame chain with the scan never leaving the GPU: vgicp_scan_prepare (extrinsic, deskew, preparation)
    -> vgicp_align_resident -> vgicp_map_insert_resident, straight on the C ABI (eskf_lio_amd/capi.py). Per
    frame the host sends 32 bytes per raw point and the IMU states, and receives the pose. The map is the
    device-resident voxel grid; the motion gate of LocalMap::needsMapUpdate (src/LocalMap.cpp:132-147) is a
    few flops and stays on the host."""

    def __init__(self, config: dict, device=0):
        """device: one ordinal, or a list of ordinals for ONE multi-device context (vgicp_create_multi)."""
        from . import capi
        self.ctx = capi.Context(device)
        lm = config["local_map"]
        self.ctx.map_reset(lm["voxel_size"], 0)
        self.cap = int(lm["max_num_points_per_voxel"])
        self.gate = (lm["translation_sq_threshold"], lm["cosine_threshold"])
        self.evict = (bool(lm["remove_distant_points"]), float(lm["distance_threshold"]), float(lm["removing_period"]))
        # the reference's period is wall-clock time (omp_get_wtime, src/LocalMap.cpp:60,70); a replay that has to be
        # repeatable counts map updates instead: local_map.remove_every_updates (10 s of sweeps at 10 Hz = 100)
        self.evict_every = int(lm.get("remove_every_updates", 0))
        self.updates_since_evict = 0
        self.removed: List[int] = []
        self.last_evict = time.perf_counter()
        self.prev: Optional[np.ndarray] = None
        self.reg = config["registration"]
        self.voxel = config["cloud_preprocessor"]["voxel_size"]
        self.T_il = np.asarray(config["lidar_extrinsic"], dtype=np.float64)
        self.iterations: List[int] = []
        self.kept: List[int] = []

    def preprocess(self, states, points, pointTime):
        kept, _ = self.ctx.scan_prepare(points, pointTime, states, self.T_il, self.voxel, 30)
        self.kept.append(kept)
        return None, None                                   # the prepared scan is on the device

    def align(self, points, covs, guess):
        r = self.ctx.align_resident(guess, self.reg["max_iteration"], self.reg["translation_sq_threshold"],
                                    self.reg["cosine_threshold"])
        self.iterations.append(r.iterations)
        return r.pose

    def update_map(self, points, covs, transform, initialize):
        if not initialize and self.prev is not None:
            moved = np.linalg.inv(self.prev) @ transform
            cosine = 0.5 * (np.trace(moved[:3, :3]) - 1.0)
            if not (cosine < self.gate[1] or float(moved[:3, 3] @ moved[:3, 3]) > self.gate[0]):
                self.prev = transform.copy()
                return
        self.ctx.map_insert_resident(transform, self.cap)
        self.updates_since_evict += 1
        due = (self.updates_since_evict >= self.evict_every) if self.evict_every else \
            (time.perf_counter() - self.last_evict > self.evict[2])
        if self.evict[0] and due:
            self.removed.append(self.ctx.map_evict(transform[:3, 3], self.evict[1]))
            self.last_evict = time.perf_counter()
            self.updates_since_evict = 0
        self.prev = transform.copy()
