"""Backends for the replay harness used by the tests: the CPU oracle behind the interface that
eskf_lio_amd/replay.py's Odometry drives (test infrastructure; the product's backend is replay.GpuBackend)."""
import numpy as np


def stream_events(replay, events):
    """synth.make_sensor_stream's tuples -> the harness's measurement objects."""
    out = []
    for arrival, e in events:
        if e[0] == "imu":
            out.append((arrival, replay.ImuMeasurement(e[1], e[2], e[3])))
        else:
            out.append((arrival, replay.LidarMeasurement(e[1].copy(), e[2].copy())))
    return out


class OracleBackend:
    """CloudPreprocessor::process, ICP::align and LocalMap::updateLocalMap as chains of oracle calls."""

    def __init__(self, config, oracle):
        self.o = oracle
        lm = config["local_map"]
        self.map = oracle.OracleMap(lm["voxel_size"], lm["max_num_points_per_voxel"])
        self.gate = (lm["translation_sq_threshold"], lm["cosine_threshold"])
        self.prev = None
        self.reg = config["registration"]
        self.voxel = config["cloud_preprocessor"]["voxel_size"]
        self.T_il = np.asarray(config["lidar_extrinsic"], dtype=np.float64)
        self.iterations = []
        self.kept = []
        # eviction (src/LocalMap.cpp:60-72) by a count of map updates instead of the reference's wall-clock period
        self.evict = (bool(lm["remove_distant_points"]), float(lm["distance_threshold"]), int(lm.get("remove_every_updates", 0)))
        self.updates_since_evict = 0
        self.removed = []

    def preprocess(self, states, points, pointTime):
        eye = np.tile(np.eye(3).reshape(9), (len(points), 1))
        pts, _ = self.o.transform(points, eye, self.T_il)               # cloud->Transform(T_il)
        if states is not None and len(states):
            pts, done = self.o.deskew(pts, pointTime, states)
            assert done >= 0, "IMU states do not bracket the sweep"
        p, c, _ = self.o.preprocess(pts, self.voxel, 30)
        self.kept.append(len(p))
        return p, c

    def align(self, points, covs, guess):
        r = self.map.align(points, covs, guess, self.reg["max_iteration"], self.reg["translation_sq_threshold"],
                           self.reg["cosine_threshold"])
        self.iterations.append(r.iterations)
        return r.pose

    def update_map(self, points, covs, transform, initialize):
        if not initialize and self.prev is not None:                     # LocalMap::needsMapUpdate
            moved = np.linalg.inv(self.prev) @ transform
            cosine = 0.5 * (np.trace(moved[:3, :3]) - 1.0)
            if not (cosine < self.gate[1] or float(moved[:3, 3] @ moved[:3, 3]) > self.gate[0]):
                self.prev = transform.copy()
                return
        wp, wc = self.o.transform(points, covs, transform)
        self.map.insert(wp, wc)
        self.updates_since_evict += 1
        if self.evict[0] and self.evict[2] and self.updates_since_evict >= self.evict[2]:
            self.removed.append(self.map.evict(transform[:3, 3], self.evict[1]))
            self.updates_since_evict = 0
        self.prev = transform.copy()
