#!/usr/bin/env python3
"""Developer probe: would starting the expensive queries of the neighbour search first shorten the kernel?  Reads the
per-query trace of a -DVGICP_PREP_TRACE=4|5|6 build (the 10-bit field carries a density predictor: points in the query's
own level-5 / level-4 cell, or the finest level whose own cell holds K points) and list-schedules the measured
durations on as many slots as ran at once: as dispatched, predicted-heavy first (for a range of thresholds), and the
oracle orders (by start level, longest first).   usage: VGICP_LIB_PATH=... python tools/probe_knn_order.py [n] [h]"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000
h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
pts = synth.make_lidar_scan(n, seed=0x46524D, extent=25.0) if os.environ.get("VGICP_TRACE_SCENE", "frame") == "frame" \
    else synth.make_lidar_scan(n, seed=11)
with capi.Context(0) as ctx:
    for rep in range(3):
        kp, kc, ki = ctx.preprocess(pts, h, 30)
rec = ki.astype(np.uint64)
dt = (rec >> np.uint64(44)).astype(np.float64) * 0.01
field = ((rec >> np.uint64(34)) & np.uint64(1023)).astype(np.int64)
level = ((rec >> np.uint64(30)) & np.uint64(15)).astype(np.int64)
t0 = (rec & np.uint64(0x3FFFFFFF)).astype(np.float64) * 0.1
t0 -= t0.min()
end = t0 + dt
slots = int(((t0 <= 0.5 * end.max()) & (end > 0.5 * end.max())).sum())


def makespan(order):
    heap = [0.0] * slots
    heapq.heapify(heap)
    last = 0.0
    for i in order:
        t = heapq.heappop(heap) + dt[i]
        last = max(last, t)
        heapq.heappush(heap, t)
    return last


by_start = np.argsort(t0, kind="stable")
print(f"[order] {len(rec)} queries, span {end.max():.1f} us, {slots} slots, work {dt.sum() / slots:.1f} us; as dispatched {makespan(by_start):.1f}, "
      f"by start level descending {makespan(by_start[np.argsort(-level[by_start], kind='stable')]):.1f}, longest first {makespan(np.argsort(-dt)):.1f}")
f = field[by_start]
print(f"[order] predictor field: min {f.min()} median {int(np.median(f))} max {f.max()}; mean time by field decile: "
      + " ".join(f"{dt[by_start][(f >= lo) & (f <= hi)].mean():.1f}" for lo, hi in zip(np.percentile(f, range(0, 100, 10)), np.percentile(f, range(10, 101, 10)))))
mode = os.environ.get("VGICP_ORDER_HEAVY", "low")   # low: small field = heavy (cell counts); high: large field = heavy (home level)
for thr in sorted(set(np.percentile(f, [5, 10, 15, 20, 25, 30, 40, 50]).astype(int).tolist())):
    heavy = f < thr if mode == "low" else f > thr
    order = np.concatenate([by_start[heavy], by_start[~heavy]])
    print(f"[order] heavy = field {'<' if mode == 'low' else '>'} {thr}: {int(heavy.sum())} queries ({100.0 * heavy.mean():.0f} %) first -> {makespan(order):.1f} us")
# and the other end of the list: the queries predicted CHEAP last (what is still running when the list runs out is what
# the kernel waits for), keeping the dispatched order otherwise
for thr in sorted(set(np.percentile(f, [50, 60, 70, 80, 90]).astype(int).tolist())):
    cheap = f > thr if mode == "low" else f < thr
    order = np.concatenate([by_start[~cheap], by_start[cheap]])
    print(f"[order] cheap = field {'>' if mode == 'low' else '<'} {thr}: {int(cheap.sum())} queries ({100.0 * cheap.mean():.0f} %) last -> {makespan(order):.1f} us "
          f"(mean time of those {dt[by_start][cheap].mean():.1f} us, max {dt[by_start][cheap].max() if cheap.any() else 0:.1f})")
