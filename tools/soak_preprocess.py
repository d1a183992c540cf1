#!/usr/bin/env python3
"""Developer soak for the scan preparation: random scans of varied size, density, voxel size and k (lidar-like
sweeps, uniform boxes, thin lines, clumps with duplicates, lattices) against the oracle, bit for bit.
usage (GPU box): python tools/soak_preprocess.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402  (the checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(kind, n):
    if kind == 0:
        return synth.make_lidar_scan(n, seed=int(rng.integers(1 << 30)), extent=float(rng.uniform(5, 60)))
    if kind == 1:
        return rng.uniform(-1, 1, size=(n, 3)) * rng.uniform(0.5, 30, size=3)
    if kind == 2:   # thin structures: lines and a plane
        t = rng.uniform(-20, 20, size=(n, 1))
        d = rng.normal(size=(1, 3))
        line = t * d / np.linalg.norm(d) + rng.normal(size=(n, 3)) * 1e-3
        plane = np.concatenate([rng.uniform(-8, 8, size=(n, 2)), rng.normal(size=(n, 1)) * 1e-4], axis=1)
        pick = rng.random(n) < 0.5
        return np.where(pick[:, None], line, plane)
    if kind == 3:   # clumps with exact duplicates
        centres = rng.uniform(-10, 10, size=(int(rng.integers(3, 40)), 3))
        p = centres[rng.integers(len(centres), size=n)] + rng.normal(size=(n, 3)) * rng.uniform(1e-3, 0.3)
        dup = rng.random(n) < 0.3
        p[dup] = p[rng.integers(n, size=int(dup.sum()))]
        return p
    g = np.arange(-12, 13) * float(rng.choice([0.0625, 0.125, 0.25]))
    lat = np.stack(np.meshgrid(g, g, g[:int(rng.integers(1, 6))], indexing="ij"), axis=-1).reshape(-1, 3)
    return lat[rng.permutation(len(lat))[:n]]


t_end = time.time() + budget
runs = bad = 0
with capi.Context(0) as ctx:
    while time.time() < t_end:
        kind = int(rng.integers(5))
        n = int(rng.integers(40, 12_000))
        h = float(rng.choice([0.05, 0.1, 0.3, 0.5, 1.0, 2.5]))
        k = int(rng.choice([3, 7, 12, 20, 30, 32]))
        pts = np.ascontiguousarray(make(kind, n) + rng.uniform(-50, 50, size=3))
        gp, gc, gi = ctx.preprocess(pts, h, k)
        rp, rc, ri = oracle.preprocess(pts, h, k)
        order = np.argsort(ri, kind="stable")
        ok = (len(gi) == len(ri) and np.array_equal(gi, ri[order]) and np.array_equal(gp, rp[order])
              and np.array_equal(gc, rc[order]))
        runs += 1
        if not ok:
            bad += 1
            print(f"MISMATCH kind {kind} n {len(pts)} h {h} k {k}", flush=True)
            np.save(f"gpurun_out/soak_prep_fail_{runs}.npy", pts)
    # the deskew's segment bounds: random point times (ordered, jittered, shuffled, with repeats), state queues of
    # varied length, in time order and not
    d_runs = d_bad = 0
    t_end = time.time() + 0.25 * budget
    while time.time() < t_end:
        ns = int(rng.integers(6, 300))
        st = synth.make_imu_states(ns, seed=int(rng.integers(1 << 30)))
        n = int(rng.integers(10, 40_000))
        pts = synth.make_lidar_scan(n, seed=int(rng.integers(1 << 30)))
        lo, hi = st[int(rng.integers(0, ns // 2)), 0], st[int(rng.integers(ns // 2, ns - 2)), 0]
        t = synth.make_point_times(n, lo + 1e-4, hi + 1e-3, seed=int(rng.integers(1 << 30)))
        mode = int(rng.integers(4))
        if mode == 1:
            t = t + rng.normal(size=n) * float(rng.uniform(1e-4, 5e-2))
        elif mode == 2:
            t = rng.permutation(t)
        elif mode == 3:
            t[rng.integers(n, size=n // 10)] = st[rng.integers(ns, size=n // 10), 0]
        t[-1] = hi + 1e-3
        if rng.random() < 0.2:
            i, j = rng.integers(ns, size=2)
            st[[i, j]] = st[[j, i]]
        gp, done = ctx.deskew(pts, t, st)
        rp, rdone = oracle.deskew(pts, t, st)
        d_runs += 1
        if done != rdone or not np.array_equal(gp, rp):
            d_bad += 1
            print(f"DESKEW MISMATCH n {n} states {ns} mode {mode}", flush=True)
    print(f"[soak deskew] {d_runs} sweeps, {d_bad} mismatches", flush=True)
    bad += d_bad
print(f"[soak prep] {runs} scans, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
