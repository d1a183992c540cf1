#!/usr/bin/env python3
"""Soak of the drop-in classes' full-hash check on helper threads (include/eskf_lio_shim: shim::HashCrew; ICP::align
registers the resident scan WHILE the helpers hash the host cloud and keeps the result only if the hash matches).

The same frames — eager host copy, a caller that leaves the prepared cloud alone in most frames and edits one element /
resizes it in the others (host_frame_run's `mutate`) — run once with the caller hashing alone (0 helper threads: the check
before the call, as rounds 1-5 and the reference's order of things) and once per helper count 1..3: every frame's pose,
round count and "found its cloud resident" verdict must be the same bits, untouched clouds must be found resident and
edited ones must not.

    python tools/soak_dropin.py [seconds] [sweep points]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eskf_lio_amd import host, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
points = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
lib = host.load_library()
lib.host_hash_helpers.argtypes = [C.c_int]
st = synth.make_imu_states(48, seed=5)
ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
tt = synth.make_point_times(points, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=7)
CFG = dict(translation_sq_threshold=-1.0, cosine_threshold=2.0, remove_distant_points=False, distance_threshold=1e9,
           removing_period=1e9, device_resident=True, keep_raw_points=True)


def chain(sweeps, mutates, helpers):
    pre = host.CloudPreprocessor(0.3, ext, "eager")
    lib.host_hash_helpers(helpers)           # (the constructor set the configuration's default)
    icp = host.ICP(30, 1e-6, 0.9999)
    lmap = host.LocalMap(0.3, 20, CFG)
    fr = host.Frame(sweeps[0], tt, st)
    fr.run(pre, icp, lmap, np.eye(4), first_frame=True)
    fr.end()
    out, pose = [], np.eye(4)
    for sweep, mutate in zip(sweeps[1:], mutates):
        fr = host.Frame(sweep, tt, st)
        fr.run(pre, icp, lmap, pose, mutate=mutate)
        got = fr.end()
        pose = got["pose"]
        out.append((pose.copy(), got["iterations"], bool(got["used_resident"])))
    lmap.drain()
    return out


t_end = time.time() + seconds
rounds = frames = edited = 0
rng = np.random.default_rng(2026)
while time.time() < t_end:
    n_frames = 24
    sweeps = [synth.make_lidar_scan(points, seed=int(rng.integers(1, 1 << 30))) for _ in range(n_frames + 1)]
    mutates = [int(m) for m in rng.choice([0, 0, 0, 1, 2, 3], size=n_frames)]
    want = chain(sweeps, mutates, 0)
    for f, ((_, _, resident), mutate) in enumerate(zip(want, mutates)):
        if resident != (mutate == 0):
            print(f"round {rounds} frame {f}: mutate {mutate} but used_resident = {resident} (caller hashing alone)")
            sys.exit(1)
    for helpers in (1, 2, 3):
        got = chain(sweeps, mutates, helpers)
        for f, (a, b) in enumerate(zip(want, got)):
            if not (np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]):
                print(f"round {rounds} frame {f} (mutate {mutates[f]}), {helpers} helper threads: pose / rounds / verdict differ "
                      f"from the caller hashing alone: {a[1:]} vs {b[1:]}, |dpose| = {np.abs(a[0] - b[0]).max():.3e}")
                sys.exit(1)
    rounds += 1
    frames += n_frames * 4
    edited += sum(1 for m in mutates if m) * 4
print(f"[soak drop-in] {rounds} rounds, {frames} frames of {points}-point sweeps ({edited} with an edited or resized cloud) through the classes "
      f"with 0, 1, 2 and 3 hash helper threads: poses, round counts and resident verdicts identical, every untouched cloud found "
      f"resident, every edited one registered from the host data")
