#!/usr/bin/env python3
"""Where the host time of a frame goes inside the drop-in classes (CloudPreprocessor::process -> ICP::align ->
LocalMap::updateLocalMap), per host-copy mode and resident check.

    python tools/probe_eager.py [frames] [sweep points] [drain|-] [hash helper threads]

"drain": the shadow grid's worker is waited for between frames, outside the timed region (a sensor's pace, where a frame
never meets the previous one's host work) — without it the frames run back to back.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eskf_lio_amd import host, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
points = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
drain = len(sys.argv) > 3 and sys.argv[3] == "drain"
helpers = int(sys.argv[4]) if len(sys.argv) > 4 else None
SLOTS = ("process: enqueue", "process: wait (scan_info)", "process: resize", "process: download", "process: stamp",
         "align: verify", "align: call", "update: verify", "update: rest", "(of it: insert call", "shadow push)")
lib = host.load_library()
lib.host_trace.argtypes = [C.c_int, C.POINTER(C.c_double)]
lib.host_hash_helpers.argtypes = [C.c_int]

h, cap = 0.3, 20
st = synth.make_imu_states(48, seed=5)
ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
sweeps = [synth.make_lidar_scan(points, seed=40 + k) for k in range(frames + 2)]
tt = synth.make_point_times(points, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=7)


def run(host_copy, check):
    pre = host.CloudPreprocessor(h, ext, host_copy, resident_check=check)
    if helpers is not None:
        lib.host_hash_helpers(helpers)     # the constructor set the configuration's default (2)
    icp = host.ICP(30, 1e-6, 0.9999)
    cfg = dict(translation_sq_threshold=-1.0, cosine_threshold=2.0, remove_distant_points=False, distance_threshold=1e9,
               removing_period=1e9, device_resident=True, keep_raw_points=True)
    lmap = host.LocalMap(h, cap, cfg)
    fr = host.Frame(sweeps[0], tt, st)
    fr.run(pre, icp, lmap, np.eye(4), first_frame=True)
    fr.end()
    pose, wall = np.eye(4), 0.0
    nxt = host.Frame(sweeps[1], tt, st)
    for f in range(1, frames + 1):
        fr = nxt
        nxt = host.Frame(sweeps[f + 1], tt, st) if f < frames else None
        t0 = time.perf_counter()
        fr.run(pre, icp, lmap, pose, move_cloud=True)
        wall += time.perf_counter() - t0
        pose = fr.end()["pose"]
        if drain:
            lmap.drain()
    return wall / frames * 1e3


for host_copy in ("deferred", "eager"):
    for check in ("sampled", "full"):
        run(host_copy, check)   # warm-up
        lib.host_trace(1, None)
        ms = run(host_copy, check)
        out = (C.c_double * (2 * len(SLOTS)))()
        lib.host_trace(0, out)
        parts = ", ".join(f"{name} {out[k] / max(out[len(SLOTS) + k], 1) * 1e6:.0f}" for k, name in enumerate(SLOTS)
                          if out[len(SLOTS) + k] > 0)
        print(f"{host_copy:8s} {check:7s}: {ms:.3f} ms per frame | us per call: {parts}", flush=True)
