"""Generates tests/golden/drive_c4.npz: the ORACLE-driven replay of the street drive (synth.iter_drive_stream: the
stand-in for BASELINE config C4, HILTI exp21) — 300 sweeps of ~57 000 points from a 32-ring sensor that travels > 200 m,
through the replay harness (eskf_lio_amd/replay.py: the ErrorStateKF restatement + Odometry::run's frame loop) with the
CPU oracle behind every stage (tests/replay_backends.py).  The GPU test (tests/test_replay.py::
test_street_drive_c4_surrogate) replays the same stream through the HIP module and compares with this file, so the ten
minutes of brute-force neighbour searches run once, here.

    python tests/golden/make_drive_fixture.py [frames]
"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from eskf_lio_amd import replay, synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from replay_backends import OracleBackend, stream_events  # noqa: E402


def drive_config():
    """config/hilti_config.yaml with the LiDAR extrinsic it names, the eviction on (as there), its period counted in map
    updates (100 = the file's 10 s at 10 Hz) and the radius shortened to the synthetic sensor's 60 m range + margin."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in replay.DEFAULT_CONFIG.items()}
    cfg["lidar_extrinsic"] = synth.hilti_lidar_extrinsic()
    cfg["local_map"].update(remove_distant_points=True, distance_threshold=70.0, remove_every_updates=100)
    return cfg


def lazy_events(frames):
    for arrival, e in synth.iter_drive_stream(frames=frames):
        if e[0] == "imu":
            yield arrival, replay.ImuMeasurement(e[1], e[2], e[3])
        else:
            yield arrival, replay.LidarMeasurement(e[1], e[2])


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg = drive_config()
    backend = OracleBackend(cfg, oracle)
    odo = replay.Odometry(cfg, backend)
    t0 = time.time()
    traj = odo.run(lazy_events(frames))
    print(f"{len(traj)} frames in {time.time() - t0:.0f} s; rounds {sum(backend.iterations)}; removed {backend.removed}; "
          f"map {len(backend.map)} voxels")
    print(odo.report())
    truth = synth.drive_truth(frames)
    err = max(float(np.linalg.norm(T[:3, 3] - G[:3, 3])) for (_, T), (_, G) in zip(traj, truth))
    print(f"largest position error against the generating motion: {err:.4f} m; travelled {np.linalg.norm(truth[-1][1][:3, 3]):.1f} m")
    keys, means, covs, counts = backend.map.export()
    order = np.lexsort(keys.T)
    keys = np.ascontiguousarray(keys[order])
    out = os.path.join(ROOT, "tests", "golden", "drive_c4.npz")
    np.savez_compressed(
        out, frames=frames, stamps=np.array([s for s, _ in traj]), poses=np.array([T for _, T in traj]),
        iterations=np.array(backend.iterations, dtype=np.int32), kept=np.array(backend.kept, dtype=np.int32),
        removed=np.array(backend.removed, dtype=np.int64), map_keys=keys.astype(np.int32),
        map_count_sum=np.int64(counts.sum()), map_mean_centroid=means.mean(axis=0), position_error=err)
    print(f"wrote {out}: {os.path.getsize(out) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
