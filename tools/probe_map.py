#!/usr/bin/env python3
"""Developer probe for the N1 row: device-side LocalMap insertion / eviction vs the CPU loop (GPU box).
usage: python tools/probe_map.py [n_points] [n_voxels]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402  (timing comparison only)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
v = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
vmap = synth.make_map(v)
pts, covs = synth.make_uniform_scan(n, vmap)
T = synth.default_guess()
with capi.Context(0) as ctx:
    ctx.map_reset(vmap.voxel_size, v)
    t0 = time.perf_counter()
    ctx.map_insert_scan(vmap.means, vmap.covs, np.eye(4), 20)
    print(f"[map] device: build {v}-voxel map by insertion {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    times = []
    for rep in range(6):
        shift = synth.se3_to_SE3([0.01 * rep, 0, 0, 0, 0, 0.001 * rep])
        t0 = time.perf_counter()
        new = ctx.map_insert_scan(pts, covs, shift @ T, 20)
        times.append(time.perf_counter() - t0)
    print(f"[map] device: insert {n}-point scan (upload + transform + insert, host wall): "
          f"{1e3 * np.median(times[1:]):.3f} ms (first {1e3 * times[0]:.3f}), last created {new} voxels", flush=True)
    t0 = time.perf_counter()
    removed = ctx.map_evict(np.zeros(3), 15.0)
    print(f"[map] device: evict beyond 15 m: {1e3 * (time.perf_counter() - t0):.3f} ms, removed {removed}", flush=True)
om = oracle.OracleMap(vmap.voxel_size, 20)
t0 = time.perf_counter()
om.insert(vmap.means, vmap.covs)
print(f"[map] cpu oracle (the reference's serial loop): build map {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
times = []
for rep in range(4):
    shift = synth.se3_to_SE3([0.01 * rep, 0, 0, 0, 0, 0.001 * rep])
    wp, wc = oracle.transform(pts, covs, shift @ T)
    t0 = time.perf_counter()
    om.insert(wp, wc)
    times.append(time.perf_counter() - t0)
print(f"[map] cpu oracle: insert {n}-point scan (insertion loop only): {1e3 * np.median(times):.2f} ms", flush=True)
