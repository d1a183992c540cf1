#!/usr/bin/env python3
"""Developer soak for the map insertion of device-prepared scans (LocalMap::updateLocalMap, src/LocalMap.cpp:44-72):
random sweeps, scan-voxel / map-voxel ratios on both sides of the switch between per-voxel lists and the sort, caps and
poses; several frames into one map (existing voxels take addPoint, the cap freezes them), the synchronous and the
non-waiting call mixed.  The device's map must equal the serial reference loop's bit for bit (keys, means, covariances,
counts) after every sequence.
usage (GPU box): python tools/soak_insert.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402  (the checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t_end = time.time() + budget
runs = bad = frames = lists = 0
while time.time() < t_end:
    map_voxel = float(rng.choice([0.2, 0.3, 0.5, 1.0]))
    scan_voxel = float(rng.choice([0.1, 0.15, 0.3, 0.5]))
    cap = int(rng.choice([1, 2, 5, 20, 1000]))
    short = (np.ceil(map_voxel / scan_voxel) + 1) ** 3 <= 64
    om = oracle.OracleMap(map_voxel, cap)
    with capi.Context(0) as ctx:
        ctx.map_reset(map_voxel, 0)
        for f in range(int(rng.integers(1, 5))):
            n = int(rng.integers(200, 40_000))
            raw = synth.make_lidar_scan(n, seed=int(rng.integers(1 << 30)), extent=float(rng.uniform(5, 50)))
            T = synth.se3_to_SE3(rng.normal(size=6) * np.array([1.0, 1.0, 0.2, 0.05, 0.05, 0.3]))
            kept, _ = ctx.scan_prepare(raw, None, None, None, scan_voxel, int(rng.choice([5, 30])))
            gp, gc = ctx.scan_download()
            om.insert(*oracle.transform(gp, gc, T))
            if rng.random() < 0.5:
                ctx.map_insert_resident_async(T, cap)
            else:
                ctx.map_insert_resident(T, cap)
            frames += 1
            lists += int(short)
        got = ctx.map_export()
    keys, means, covs, counts = om.export()
    order = np.lexsort(keys.T)
    ref = (keys[order], means[order], covs[order], counts[order])
    ok = len(got[0]) == len(ref[0]) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    runs += 1
    if not ok:
        bad += 1
        print(f"MISMATCH map {map_voxel} scan {scan_voxel} cap {cap}", flush=True)
print(f"[soak insert] {runs} maps, {frames} frames ({lists} through per-voxel lists), {bad} mismatches")
sys.exit(1 if bad else 0)
