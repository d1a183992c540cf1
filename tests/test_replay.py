"""the ROS-free replay harness (eskf_lio_amd/replay.py).

CPU part: the wire formats (CDR encodings of sensor_msgs/Imu and sensor_msgs/PointCloud2 inside a rosbag2
sqlite3 file), the estimator restatement and the frame loop, driven by the oracle. GPU part: the same
loop driven by the MI355X path, compared with the oracle-driven run frame by frame."""
import os

import numpy as np
import pytest

from conftest import pose_error
from eskf_lio_amd import replay, synth
from replay_backends import OracleBackend, stream_events


@pytest.fixture(scope="module")
def stream():
    events, truth = synth.make_sensor_stream(frames=5, points_per_frame=6_000)
    return stream_events(replay, events), truth


@pytest.fixture(scope="module")
def oracle_run(stream, oracle):
    events, _ = stream
    backend = OracleBackend(replay.DEFAULT_CONFIG, oracle)
    odo = replay.Odometry(replay.DEFAULT_CONFIG, backend)
    return odo.run([(a, _clone(m)) for a, m in events]), backend


def _clone(m):
    if isinstance(m, replay.ImuMeasurement):
        return replay.ImuMeasurement(m.timestamp, m.angularVelocity.copy(), m.acceleration.copy())
    return replay.LidarMeasurement(m.points.copy(), m.pointTime.copy())


def test_cdr_messages_round_trip():
    imu = replay.ImuMeasurement(1234.567891234, np.array([0.1, -0.2, 0.3]), np.array([9.0, -0.5, 0.25]))
    back = replay.decode_imu(replay.encode_imu(imu))
    assert abs(back.timestamp - imu.timestamp) < 2e-9            # stamp travels as sec + nanosec
    assert np.array_equal(back.angularVelocity, imu.angularVelocity)
    assert np.array_equal(back.acceleration, imu.acceleration)
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(257, 3)).astype(np.float32).astype(np.float64)
    t = 1000.0 + np.sort(rng.uniform(0, 0.1, size=257))
    cloud = replay.decode_pointcloud2(replay.encode_pointcloud2(pts, t))
    assert np.array_equal(cloud.points, pts) and np.array_equal(cloud.pointTime, t)
    assert cloud.startTime == t[0] and cloud.endTime == t[-1]    # Subscriber.hpp:97-98
    with pytest.raises(ValueError):
        replay.decode_imu(b"\x07\x07")


def test_rosbag2_round_trip_and_player_order(stream, tmp_path):
    events, _ = stream
    path = str(tmp_path / "synthetic_0.db3")
    replay.write_rosbag2(path, reversed(events))                  # written out of order on purpose
    back = replay.read_rosbag2(path)
    assert len(back) == len(events)
    assert all(b[0] <= c[0] for b, c in zip(back, back[1:]))      # played in bag-time order
    for (a0, m0), (a1, m1) in zip(events, back):
        assert type(m0) is type(m1) and abs(a0 - a1) < 1e-9
        if isinstance(m0, replay.LidarMeasurement):
            assert np.array_equal(m0.points, m1.points) and np.array_equal(m0.pointTime, m1.pointTime)
        else:
            assert abs(m0.timestamp - m1.timestamp) < 2e-9 and np.array_equal(m0.acceleration, m1.acceleration)
    with pytest.raises(FileNotFoundError):
        replay.read_rosbag2(str(tmp_path / "missing.db3"))


def test_rotation_helpers_are_consistent():
    rng = np.random.default_rng(5)
    for _ in range(20):
        r = rng.normal(size=3)
        r = r / np.linalg.norm(r) * rng.uniform(0.0, 3.0)        # below pi: the vector is recovered as it is
        q = replay.rotation_vector_to_quat(r)
        R = replay.quat_to_matrix(q)
        assert np.allclose(R, synth.se3_to_SE3(np.concatenate([np.zeros(3), r]))[:3, :3], atol=1e-13)
        assert np.allclose(replay.rotation_matrix_to_vector(R), r, atol=1e-12)
        q2 = replay.matrix_to_quat(R)
        assert np.allclose(q2 * np.sign(q2[3]), q * np.sign(q[3]), atol=1e-13)
    a, b = replay.rotation_vector_to_quat(np.array([0.3, 0.1, -0.2])), replay.rotation_vector_to_quat(np.array([-0.1, 0.4, 0.2]))
    assert np.allclose(replay.quat_to_matrix(replay.quat_multiply(a, b)), replay.quat_to_matrix(a) @ replay.quat_to_matrix(b))
    assert np.array_equal(replay.rotation_matrix_to_vector(np.eye(3)), np.zeros(3))


def test_filter_prediction_follows_the_imu():
    """ErrorStateKF::process alone (no update): integrating the stream's IMU reproduces the motion."""
    events, truth = synth.make_sensor_stream(frames=3, points_per_frame=10)
    kf = replay.ErrorStateKF(replay.DEFAULT_CONFIG, align=None)
    kf.initialize(synth.STREAM_T0)
    for _, e in events:
        if e[0] == "imu" and e[1] >= synth.STREAM_T0:
            kf.process(replay.ImuMeasurement(e[1], e[2], e[3]))
    last = kf.getStates()[-1]
    R, p, _ = synth.stream_pose(np.array(last.timestamp - synth.STREAM_T0))
    assert np.linalg.norm(last.position - p) < 5e-3               # 0.2 s of dead reckoning at 400 Hz
    assert np.abs(replay.quat_to_matrix(last.attitude) - R).max() < 1e-3
    assert np.all(np.linalg.eigvalsh(last.P) > 0) and last.P[0, 0] > 1e-3   # covariance grows, stays PD
    before = len(kf.getStates())
    kf.process(replay.ImuMeasurement(last.timestamp - 1.0, np.zeros(3), np.zeros(3)))   # dt < 0: dropped
    assert len(kf.getStates()) == before


def _inject(state, err):
    """ErrorStateKF::injectError (src/ErrorStateKF.cpp:166-174) on a copy."""
    s = state.copy()
    s.position = s.position + err[0:3]
    s.velocity = s.velocity + err[3:6]
    s.attitude = replay.quat_multiply(s.attitude, replay.rotation_vector_to_quat(err[6:9]))
    s.biasAccel = s.biasAccel + err[9:12]
    s.biasGyro = s.biasGyro + err[12:15]
    s.gravity = s.gravity + err[15:18]
    return s


def _error_between(a, b):
    """The 18-vector e with a (+) e = b, to first order."""
    e = np.zeros(18)
    e[0:3] = b.position - a.position
    e[3:6] = b.velocity - a.velocity
    e[6:9] = replay.rotation_matrix_to_vector(replay.quat_to_matrix(a.attitude).T @ replay.quat_to_matrix(b.attitude))
    e[9:12] = b.biasAccel - a.biasAccel
    e[12:15] = b.biasGyro - a.biasGyro
    e[15:18] = b.gravity - a.gravity
    return e


def test_filter_transition_matrix_is_the_jacobian_of_its_own_prediction():
    """An independent check of the estimator restatement (it drives the GPU run AND the oracle run, so an error
    in it would cancel in their comparison): the error-state transition matrix F that ErrorStateKF::process
    uses for the covariance (src/ErrorStateKF.cpp:99-110) must be the Jacobian of the nominal prediction it
    performs on the state (:84-97) — perturb the state by an error vector, predict both, and compare the
    resulting error with F times the perturbation, block by block."""
    rng = np.random.default_rng(3)
    kf = replay.ErrorStateKF(replay.DEFAULT_CONFIG, align=None)
    base = kf.getStates()[0]
    base.timestamp = 10.0
    base.position = rng.normal(size=3)
    base.velocity = rng.normal(size=3)
    base.attitude = replay.rotation_vector_to_quat(np.array([0.3, -0.5, 0.8]))
    base.biasAccel = 0.05 * rng.normal(size=3)
    base.biasGyro = 0.01 * rng.normal(size=3)
    dt = 1.0 / 400.0
    imu = replay.ImuMeasurement(base.timestamp + dt, np.array([0.2, -0.4, 0.3]), np.array([0.5, -1.0, 9.6]))
    kf.process(imu)
    nominal = kf.getStates()[-1]
    F = kf.F_x_.copy()
    assert np.allclose(nominal.P, F @ base.P @ F.T + kf.F_i_ @ (kf.Q_ * np.r_[[dt * dt] * 6, [dt] * 6][:, None] *
                                                                np.eye(12)) @ kf.F_i_.T)
    eps = 1e-6
    J = np.zeros((18, 18))
    for k in range(18):
        e = np.zeros(18)
        e[k] = eps
        other = replay.ErrorStateKF(replay.DEFAULT_CONFIG, align=None)
        other.states_[0] = _inject(base, e)
        other.process(imu)
        J[:, k] = _error_between(nominal, other.getStates()[-1]) / eps
    # F is first order in dt: the prediction's 0.5 a dt^2 terms (position <- attitude / biases / gravity) are not
    # in it; everything else agrees to the finite-difference error
    tol = np.full((18, 18), 2e-6)
    tol[0:3, 6:18] = 0.5 * 12.0 * dt * dt + 2e-6
    tol[3:6, 6:9] += 12.0 * dt * np.linalg.norm(imu.angularVelocity) * dt     # R(t + dt) vs R(t) in the velocity row
    assert (np.abs(J - F) <= tol).all(), np.abs(J - F).max()
    # and the blocks themselves (Sola's error-state kinematics): identities where states do not interact
    assert np.allclose(F[0:3, 3:6], np.eye(3) * dt) and np.allclose(F[3:6, 15:18], np.eye(3) * dt)
    assert np.allclose(F[6:9, 12:15], -np.eye(3) * dt) and np.allclose(F[9:18, 9:18], np.eye(9))


def test_filter_update_moves_the_state_onto_a_trusted_observation():
    """ErrorStateKF::update (src/ErrorStateKF.cpp:116-164): with the measurement noise far below the prior
    covariance the posterior pose is the observed pose, the covariance of the observed states collapses to V,
    unobserved ones shrink only through their correlations, P stays symmetric positive definite — properties
    the restatement must have whatever `align` returns."""
    rng = np.random.default_rng(4)
    T_obs = synth.se3_to_SE3(np.array([0.3, -0.2, 0.1, 0.02, -0.03, 0.04]))
    kf = replay.ErrorStateKF(replay.DEFAULT_CONFIG, align=lambda p, c, guess: T_obs)
    kf.initialize(0.0)
    for k in range(1, 41):                                            # 0.1 s of prediction so that P is full
        kf.process(replay.ImuMeasurement(k / 400.0, 0.05 * rng.normal(size=3), np.array([0.0, 0.0, -9.805]) + 0.1 * rng.normal(size=3)))
    prior = kf.getStates()[-1].copy()
    lidar = replay.LidarMeasurement(np.zeros((1, 3)), np.array([0.1]))
    lidar.covariances = np.zeros((1, 9))
    lidar.endTime = 0.1
    pose = kf.update(lidar)
    post = [s for s in kf.getStates() if s.timestamp == 0.1][-1]
    V = kf.V_
    assert np.allclose(pose[:3, 3], T_obs[:3, 3], atol=1e-4) and np.abs(pose[:3, :3] - T_obs[:3, :3]).max() < 1e-4
    assert np.allclose(post.P, post.P.T, atol=1e-12) and np.all(np.linalg.eigvalsh(0.5 * (post.P + post.P.T)) > 0)
    assert np.all(np.diag(post.P)[0:3] <= 1.01 * (np.diag(V)[0:3] + 1e-12) + 1e-9)          # observed: down to the noise
    assert np.all(np.diag(post.P)[6:9] <= 1.01 * (np.diag(V)[3:6] + 1e-12) + 1e-9)
    assert np.all(np.diag(post.P) <= np.diag(prior.P) * (1 + 1e-9) + 1e-12)                 # nothing grows in an update
    # the gain is the textbook one: K = P H^T (H P H^T + V)^-1 reproduces the correction that was applied
    S = kf.H_ @ prior.P @ kf.H_.T + V
    K = prior.P @ kf.H_.T @ np.linalg.inv(S)
    guess = prior.pose()
    r = np.r_[T_obs[:3, 3] - guess[:3, 3], replay.rotation_matrix_to_vector(guess[:3, :3].T @ T_obs[:3, :3])]
    assert np.allclose(post.velocity - prior.velocity, (K @ r)[3:6], atol=1e-12)


def test_oracle_driven_replay_tracks_the_motion(stream, oracle_run, tmp_path):
    _, truth = stream
    traj, backend = oracle_run
    assert len(traj) == len(truth) and all(2 <= it <= 10 for it in backend.iterations)
    for (stamp, T), (tstamp, G) in zip(traj, truth):
        dt, dr = pose_error(T, G)
        assert stamp == tstamp and dt < 0.02 and dr < 2e-3         # centimetre-level odometry
    path = str(tmp_path / "traj.tum")
    replay.write_tum(path, traj)
    rows = np.loadtxt(path)
    assert rows.shape == (len(traj), 8) and np.allclose(np.linalg.norm(rows[:, 4:], axis=1), 1.0, atol=1e-8)
    assert np.allclose(rows[:, 1:4], [T[:3, 3] for _, T in traj], atol=1e-8)


def test_replay_from_a_bag_equals_replay_from_memory(stream, oracle, oracle_run, tmp_path):
    events, _ = stream
    path = str(tmp_path / "run.db3")
    replay.write_rosbag2(path, events)
    backend = OracleBackend(replay.DEFAULT_CONFIG, oracle)
    traj = replay.Odometry(replay.DEFAULT_CONFIG, backend).run(replay.read_rosbag2(path))
    for (s0, T0), (s1, T1) in zip(oracle_run[0], traj):
        assert abs(s0 - s1) < 1e-9 and np.abs(T0 - T1).max() < 1e-7   # stamps are quantised to nanoseconds


def test_frames_wait_for_the_imu(stream, oracle):
    """Odometry.cpp:66-70: a sweep is not processed until an IMU sample at or after its end has arrived."""
    events, _ = stream
    backend = OracleBackend(replay.DEFAULT_CONFIG, oracle)
    odo = replay.Odometry(replay.DEFAULT_CONFIG, backend)
    lidar_seen = 0
    for arrival, m in events:
        odo.run([(arrival, _clone(m))])
        if isinstance(m, replay.LidarMeasurement):
            lidar_seen += 1
            if lidar_seen == 2:
                # second sweep arrived, the IMU sample covering its end has not: still only the first pose
                assert len(odo.trajectory) == 1 and odo.lidar is not None
    assert len(odo.trajectory) == lidar_seen


@pytest.mark.gpu
@pytest.mark.parametrize("device_map", [False, True])
def test_gpu_driven_replay_matches_the_oracle_driven_one(stream, oracle_run, device_map):
    """The whole per-frame chain on the MI355X (extrinsic -> vgicp_deskew -> vgicp_preprocess -> vgicp_align ->
    map update) against the same chain on the CPU, frame by frame."""
    events, truth = stream
    backend = replay.GpuBackend(replay.DEFAULT_CONFIG, device_resident_map=device_map)
    traj = replay.Odometry(replay.DEFAULT_CONFIG, backend).run([(a, _clone(m)) for a, m in events])
    ref_traj, ref_backend = oracle_run
    assert len(traj) == len(ref_traj) == len(truth)
    assert backend.iterations == ref_backend.iterations               # same Gauss-Newton rounds per frame
    for (s0, T0), (s1, T1) in zip(ref_traj, traj):
        dt, dr = pose_error(T1, T0)
        assert s0 == s1 and dt < 1e-8 and dr < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("device", [0, [0, 0]])
def test_resident_frame_chain_matches_the_oracle_driven_replay(stream, oracle_run, device):
    """vgicp_scan_prepare -> vgicp_align_resident -> vgicp_map_insert_resident: the scan never returns to the
    host between the raw sweep and the pose; trajectory and round counts equal the CPU chain's.  [0, 0]: the same
    loop over ONE multi-device context (vgicp_create_multi, two sub-contexts on device 0): prepared on the first,
    dealt out, registered sharded, inserted into every replica."""
    events, _ = stream
    backend = replay.DeviceBackend(replay.DEFAULT_CONFIG, device)
    traj = replay.Odometry(replay.DEFAULT_CONFIG, backend).run([(a, _clone(m)) for a, m in events])
    ref_traj, ref_backend = oracle_run
    assert backend.iterations == ref_backend.iterations and len(traj) == len(ref_traj)
    for (s0, T0), (s1, T1) in zip(ref_traj, traj):
        dt, dr = pose_error(T1, T0)
        assert s0 == s1 and dt < 1e-8 and dr < 1e-8
    backend.ctx.close()


@pytest.mark.gpu
def test_scan_prepare_equals_the_separate_calls(gpu_ctx, oracle):
    n = 20_000
    st = synth.make_imu_states(48, seed=41)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=41)
    pts = synth.make_lidar_scan(n, seed=41)
    T_il = synth.se3_to_SE3(np.array([0.05, -0.02, 0.1, 0.01, -0.02, 0.03]))
    kept, moved = gpu_ctx.scan_prepare(pts, t, st, T_il, 0.3, 30)
    gp, gc = gpu_ctx.scan_download()
    ext, _ = oracle.transform(pts, np.tile(np.eye(3).reshape(9), (n, 1)), T_il)
    desk, done = oracle.deskew(ext, t, st)
    rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
    assert kept == len(rp) and moved == done
    assert np.array_equal(gp, rp) and np.array_equal(gc, rc)       # the whole chain, bit for bit
    # no IMU states: process() with an empty queue skips the deskew; no extrinsic either
    kept, moved = gpu_ctx.scan_prepare(pts, None, None, None, 0.3, 30)
    gp, gc = gpu_ctx.scan_download()
    rp, rc, _ = oracle.preprocess(pts, 0.3, 30)
    assert moved == 0 and np.array_equal(gp, rp) and np.array_equal(gc, rc)
    # states that do not bracket the sweep are an error and leave no scan behind
    from eskf_lio_amd.capi import VgicpError
    with pytest.raises(VgicpError):
        gpu_ctx.scan_prepare(pts, t, st[:3], None, 0.3, 30)
    with pytest.raises(VgicpError):
        gpu_ctx.scan_download()
    assert gpu_ctx.scan_prepare(np.zeros((0, 3)), None, None, None, 0.3, 30) == (0, 0)
    assert gpu_ctx.scan_download()[0].shape == (0, 3)


# ---- wire formats against fixtures that the package's own writer never touched ------------------------------------
def test_wire_format_readers_against_hand_assembled_fixtures():
    """tests/golden/make_wire_fixtures.py lays a sensor_msgs/Imu, a sensor_msgs/PointCloud2 (32-byte point step with
    padding, `intensity` and `ring` BEFORE `timestamp`, height 2 x width 3) and a minimal rosbag2 sqlite3 file out by
    hand from the OMG CDR and message definitions — it imports nothing from the package.  The readers
    (reference include/ESKF_LIO/Subscriber.hpp:38-52,80-103) must return exactly what was put in: little and big
    endian; a float32 widened to double as the subscriber does; the bag in player order with the unrelated topic skipped."""
    import json
    from eskf_lio_amd import replay
    gold = os.path.join(os.path.dirname(__file__), "golden")
    want = json.load(open(os.path.join(gold, "wire_expected.json")))

    def blob(name):
        return open(os.path.join(gold, name), "rb").read()
    for name in ("imu_le.cdr", "imu_be.cdr"):
        m = replay.decode_imu(blob(name))
        assert m.timestamp == want["imu"]["timestamp"]
        assert m.angularVelocity.tolist() == want["imu"]["angular_velocity"]
        assert m.acceleration.tolist() == want["imu"]["linear_acceleration"]
    for name in ("cloud_le.cdr", "cloud_be.cdr"):
        c = replay.decode_pointcloud2(blob(name))
        assert c.points.dtype == np.float64 and c.points.tolist() == want["cloud"]["points"]
        assert c.points[2, 0] == float(np.float32(0.1)) != 0.1          # widened from float32, not re-parsed
        assert c.pointTime.tolist() == want["cloud"]["point_time"]
        assert c.startTime == want["cloud"]["point_time"][0] and c.endTime == want["cloud"]["point_time"][-1]
    with pytest.raises(ValueError):
        replay.decode_pointcloud2(blob("cloud_bad_datatype.cdr"))       # x declared FLOAT64
    with pytest.raises(ValueError):
        replay.decode_pointcloud2(blob("cloud_pl_cdr.cdr"))             # parameter-list CDR: another wire format
    with pytest.raises(ValueError):
        replay.decode_imu(b"\x00\x07\x00\x00" + b"\x00" * 64)           # XCDR2 identifier
    with pytest.raises(Exception):
        replay.decode_imu(blob("imu_le.cdr")[:100])                     # truncated
    events = replay.read_rosbag2(os.path.join(gold, "mini_bag.db3"))
    kinds = [["imu" if isinstance(m, replay.ImuMeasurement) else "cloud", round(t, 6)] for t, m in events]
    assert kinds == want["bag_order"]
    assert events[1][1].acceleration.tolist() == want["imu"]["linear_acceleration"]
    assert events[0][1].points.tolist() == want["cloud"]["points"]


def test_the_package_writer_agrees_with_the_hand_assembled_layout():
    """The other direction: what replay's own encoder produces for the same Imu content is byte for byte the
    hand-assembled little-endian message (so the writer that builds the synthetic bags speaks the same format)."""
    import json
    from eskf_lio_amd import replay
    gold = os.path.join(os.path.dirname(__file__), "golden")
    want = json.load(open(os.path.join(gold, "wire_expected.json")))
    hand = open(os.path.join(gold, "imu_le.cdr"), "rb").read()
    m = replay.ImuMeasurement(want["imu"]["timestamp"], np.array(want["imu"]["angular_velocity"]),
                              np.array(want["imu"]["linear_acceleration"]))
    mine = replay.encode_imu(m, frame_id="imu_sensor_frame")
    # encapsulation + seconds, then (skipping the nanoseconds: a double holds 1.6e9 s to ~2e-7 s only) frame id + padding
    assert len(mine) == len(hand) and mine[:8] == hand[:8] and mine[12:36] == hand[12:36]
    import struct
    assert abs(struct.unpack("<I", mine[8:12])[0] - struct.unpack("<I", hand[8:12])[0]) <= 250
    # the fields the subscriber reads sit at the same byte offsets: angular_velocity at 140, linear_acceleration at 236
    assert mine[140:164] == hand[140:164] and mine[236:260] == hand[236:260]
    back = replay.decode_imu(mine)
    assert back.timestamp == m.timestamp and back.angularVelocity.tolist() == m.angularVelocity.tolist()


# ---- the stand-in for BASELINE config C4 (HILTI exp21 through Odometry.cpp end to end) -----------------------------
def _drive_config():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_drive_fixture", os.path.join(os.path.dirname(__file__), "golden", "make_drive_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.drive_config(), mod.lazy_events


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("launch", ["one point per thread", "several points per thread + dense record copy"])
def test_street_drive_c4_surrogate(capsys, monkeypatch, launch):
    """300 sweeps of ~57 000 points from a 32-ring spinning sensor that drives > 200 m down a street (synth.
    iter_drive_stream; every ray cast from the pose at its own firing time), through Odometry::run's frame loop with the
    HIP module behind every stage (vgicp_scan_prepare -> vgicp_align_resident -> vgicp_map_insert_resident ->
    vgicp_map_evict every 100 updates), against the ORACLE-driven replay of the same stream (tests/golden/drive_c4.npz,
    made once by tests/golden/make_drive_fixture.py: minutes of CPU): the trajectory (to rounding while no voxel
    assignment has flipped between the two chains, to a fraction of the sensor noise after), the Gauss-Newton round
    counts, the kept points, what every eviction removed, and the final map's voxel set.  The map grows from
    nothing through several table sizes, is evicted from three times (tombstones, then rehashes under them), and the
    registration runs on whatever the table looks like at that frame.  Prints the three stage timers the way
    src/Odometry.cpp:98-109 does.
    Second parametrisation: the same drive with the persistent launch on 120 workgroups and the dense-copy threshold at
    4 096 slots (VGICP_PERSIST_GRID, VGICP_DENSE_SLOTS: read when the context is created) — every align is then the
    several-points-per-thread instantiation (57 000 raw points > 120 x 448) reading remembered voxel payloads from a dense
    copy of the occupied records that is REBUILT EVERY FRAME, because every frame's insertion (and every eviction, and
    every rehash) changes the map: the long-run exercise of that path that C5 (one map, never mutated) cannot give."""
    if launch != "one point per thread":
        monkeypatch.setenv("VGICP_PERSIST_GRID", "120")
        monkeypatch.setenv("VGICP_DENSE_SLOTS", "4096")
    fixture = os.path.join(os.path.dirname(__file__), "golden", "drive_c4.npz")
    ref = np.load(fixture)
    frames = int(ref["frames"])
    assert frames >= 300
    cfg, lazy_events = _drive_config()
    backend = replay.DeviceBackend(cfg, 0)
    odo = replay.Odometry(cfg, backend)
    slots_seen = set()
    orig_update = backend.update_map

    def update_and_watch(points, covs, transform, initialize):
        orig_update(points, covs, transform, initialize)
        slots_seen.add(backend.ctx.map_size()[1])
    backend.update_map = update_and_watch
    traj = odo.run(lazy_events(frames))
    with capsys.disabled():
        print("\n[street drive, %d frames, GPU-driven]\n%s" % (len(traj), odo.report()))
    assert len(traj) == frames
    # The trajectory.  Closed loop, the two chains cannot stay bit-close for 300 frames: they differ by rounding (1e-13)
    # until, in some frame, a point sits within that distance of a voxel face and the two sides bin it differently --
    # one correspondence more or less moves the pose by ~1e-5 m, the next frame's deskew and map inherit that, and from
    # there on the runs are two equally valid registrations of the same noisy sweeps (1 cm range noise): millimetres
    # apart, both on the generating motion.  So: rounding-level agreement while no flip has happened (the first
    # frames), then agreement to a fraction of the sensor noise, and the SAME accuracy against the truth.
    truth = synth.drive_truth(frames)
    dev = np.array([pose_error(T1, T0)[0] for (_, T1), T0 in zip(traj, ref["poses"])])
    assert all(s1 == s0 for (s1, _), s0 in zip(traj, ref["stamps"]))
    assert dev[:10].max() < 1e-9, dev[:10]
    if launch != "one point per thread":
        assert backend.ctx.counter(0) >= frames - 1          # every align was ONE persistent launch
    assert dev.max() < 0.02 and np.median(dev) < 0.006, (dev.max(), np.median(dev))
    err_gpu = np.array([np.linalg.norm(T[:3, 3] - G[:3, 3]) for (_, T), (_, G) in zip(traj, truth)])
    err_ref = np.array([np.linalg.norm(T[:3, 3] - G[:3, 3]) for T, (_, G) in zip(ref["poses"], truth)])
    assert err_gpu.max() < 0.25 and abs(err_gpu.max() - err_ref.max()) < 0.02 and abs(err_gpu[-1] - err_ref[-1]) < 0.02
    # rounds per frame: the same distribution (one frame of the fixture never converges: Gauss-Newton without damping
    # cycles between two voxel assignments until max_iteration = 100 stops it -- the reference's behaviour)
    it_gpu, it_ref = np.array(backend.iterations), ref["iterations"]
    cap = cfg["registration"]["max_iteration"]
    assert len(it_gpu) == len(it_ref) and np.array_equal(it_gpu[:10], it_ref[:10])
    assert abs(np.mean(it_gpu[it_gpu < cap]) - np.mean(it_ref[it_ref < cap])) < 0.3 and (it_gpu >= cap).sum() <= 3
    # kept points per frame: equal while the chains are bit-close, within half a per cent after
    kept_gpu, kept_ref = np.array(backend.kept), ref["kept"]
    assert len(kept_gpu) == len(kept_ref) and np.array_equal(kept_gpu[:10], kept_ref[:10])
    assert (np.abs(kept_gpu - kept_ref) / kept_ref).max() < 0.005
    # evictions fired, and removed what the oracle's loop removed (to the voxels the maps differ in)
    assert len(backend.removed) == len(ref["removed"]) >= 2 and max(backend.removed) > 10_000
    assert (np.abs(np.array(backend.removed) - ref["removed"]) / ref["removed"]).max() < 0.01
    # the table grew through several sizes on the way, and tombstones were left behind by the evictions
    assert len(slots_seen) >= 3, slots_seen
    # the final map: the same voxels but for the fringe the flips decide
    keys, means, covs, counts = backend.ctx.map_export()
    a = {tuple(k) for k in keys.tolist()}
    b = {tuple(k) for k in ref["map_keys"].tolist()}
    assert len(a ^ b) < 0.05 * len(b), (len(a), len(b), len(a ^ b))     # millimetres of pose difference move the surface voxels of a 0.3 m grid
    assert abs(int(counts.sum()) - int(ref["map_count_sum"])) < 0.01 * int(ref["map_count_sum"])
    assert np.abs(means.mean(axis=0) - ref["map_mean_centroid"]).max() < 0.05
    assert backend.ctx.counter(1) == 0      # no persistent launch gave up
    backend.ctx.close()
