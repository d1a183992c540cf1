#!/usr/bin/env python3
"""Replays a rosbag2 recording (or a synthetic stream) through the reference's frame loop with the MI355X
path doing the per-frame work, writes the trajectory in TUM format and prints the reference's stage timers.

usage: python tools/replay.py --bag run_0.db3 [--imu-topic /alphasense/imu] [--lidar-topic /hesai/pandar]
       python tools/replay.py --synthetic 20 --points 60000          (GPU box; no recording needed)
options: --out traj.tum   --device-map (keep the voxel grid on the GPU only)   --write-bag file.db3
         --resident (the scan never leaves the GPU between the raw sweep and the pose)
The configuration is the reference's config/hilti_config.yaml as a dict (eskf_lio_amd/replay.py:DEFAULT_CONFIG);
--config file.yaml overrides it with a file of the reference's own layout."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import replay, synth  # noqa: E402


def config_from_yaml(path):
    import yaml
    y = yaml.safe_load(open(path))
    imu = y["sensors"]["imu"]
    par = imu["intrinsics"]["parameters"]
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in replay.DEFAULT_CONFIG.items()}
    cfg["imu"] = dict(update_rate=imu["update_rate"], bias_a=par["bias_a"], bias_g=par["bias_g"], gravity=par["gravity"],
                      accel_noise_density=par["accel_noise_density"], accel_zero_g_offset=par["accel_zero_g_offset"],
                      gyro_noise_density=par["gyro_noise_density"], gyro_zero_rate_offset=par["gyro_zero_rate_offset"])
    ext = y["sensors"]["lidar"]["extrinsics"]
    T = np.eye(4)
    T[:3, :3] = replay.quat_to_matrix(np.array(ext["quaternion"], dtype=np.float64))   # Eigen::Map: x y z w
    T[:3, 3] = ext["translation"]
    cfg["lidar_extrinsic"] = T
    cfg["kalman_filter"] = dict(y["kalman_filter"]["update"])
    lm = y["local_map"]
    cfg["local_map"] = dict(voxel_size=lm["voxel_size"], max_num_points_per_voxel=lm["max_num_points_per_voxel"],
                            translation_sq_threshold=lm["update"]["translation_sq_threshold"],
                            cosine_threshold=lm["update"]["cosine_threshold"],
                            remove_distant_points=lm["remove_distant_points"]["enabled"],
                            distance_threshold=lm["remove_distant_points"]["distance_threshold"],
                            removing_period=lm["remove_distant_points"]["removing_period"])
    cfg["cloud_preprocessor"] = dict(voxel_size=y["cloud_preprocessor"]["voxel_size"])
    cfg["registration"] = dict(y["registration"])
    return cfg, imu["topic_name"], y["sensors"]["lidar"]["topic_name"]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--bag")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic LiDAR frames")
    ap.add_argument("--points", type=int, default=20_000, help="points per synthetic frame")
    ap.add_argument("--config")
    ap.add_argument("--imu-topic", default=None)
    ap.add_argument("--lidar-topic", default=None)
    ap.add_argument("--out", default="trajectory.tum")
    ap.add_argument("--write-bag", default=None, help="also store the synthetic stream as a rosbag2 file")
    ap.add_argument("--device-map", action="store_true")
    ap.add_argument("--resident", action="store_true",
                    help="scan stays on the GPU from the raw sweep to the pose (vgicp_scan_prepare chain)")
    args = ap.parse_args()
    cfg, imu_topic, lidar_topic = replay.DEFAULT_CONFIG, replay.IMU_TOPIC, replay.LIDAR_TOPIC
    if args.config:
        cfg, imu_topic, lidar_topic = config_from_yaml(args.config)
    imu_topic, lidar_topic = args.imu_topic or imu_topic, args.lidar_topic or lidar_topic
    truth = None
    if args.bag:
        events = replay.read_rosbag2(args.bag, imu_topic, lidar_topic)
    elif args.synthetic > 0:
        raw, truth = synth.make_sensor_stream(frames=args.synthetic, points_per_frame=args.points,
                                              world_points=max(3 * args.points, 60_000))
        events = [(a, replay.ImuMeasurement(e[1], e[2], e[3]) if e[0] == "imu" else replay.LidarMeasurement(e[1], e[2]))
                  for a, e in raw]
        if args.write_bag:
            replay.write_rosbag2(args.write_bag, events, imu_topic, lidar_topic)
    else:
        ap.error("one of --bag / --synthetic is required")
    backend = replay.DeviceBackend(cfg) if args.resident else replay.GpuBackend(cfg, device_resident_map=args.device_map)
    odo = replay.Odometry(cfg, backend)
    traj = odo.run(events)
    replay.write_tum(args.out, traj)
    print(f"{len(traj)} poses -> {args.out}; Gauss-Newton rounds per frame: {odo.backend.iterations}")
    print(odo.report())
    if truth is not None:
        err = [np.linalg.norm(T[:3, 3] - G[:3, 3]) for (_, T), (_, G) in zip(traj, truth)]
        print(f"synthetic stream: position error against the generating motion, max {max(err):.4f} m")


if __name__ == "__main__":
    main()
