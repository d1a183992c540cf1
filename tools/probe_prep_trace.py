#!/usr/bin/env python3
"""Developer probe: per-query trace of the neighbour search from a -DVGICP_PREP_TRACE build
(tools/ab_build.sh trace -DVGICP_PREP_TRACE; VGICP_LIB_PATH=eskf_lio_amd/lib_ab/trace/libvgicp_hip.so).
The index output of that build carries, per kept point: time in the search, cells taken, start level, start time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
h = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
pts = synth.make_lidar_scan(n, seed=11)
if os.environ.get("VGICP_TRACE_SCENE") == "frame":  # the sweep of bench.py's frame chain
    pts = synth.make_lidar_scan(n, seed=0x46524D, extent=25.0)
with capi.Context(0) as ctx:
    for rep in range(3):
        kp, kc, ki = ctx.preprocess(pts, h, 30)
rec = ki.astype(np.uint64)
dt = (rec >> np.uint64(44)).astype(np.float64) * 0.01          # us
pops = ((rec >> np.uint64(34)) & np.uint64(1023)).astype(np.int64)
level = ((rec >> np.uint64(30)) & np.uint64(15)).astype(np.int64)
t0 = (rec & np.uint64(0x3FFFFFFF)).astype(np.float64) * 0.1    # us
t0 -= t0.min()
end = t0 + dt
print(f"[trace] {len(rec)} queries; kernel span {end.max():.1f} us; time per query mean {dt.mean():.1f} median {np.median(dt):.1f} "
      f"p90 {np.percentile(dt, 90):.1f} p99 {np.percentile(dt, 99):.1f} max {dt.max():.1f} us")
print(f"[trace] last start {t0.max():.1f} us; queries still running at 60/70/80/90 % of the span: "
      + " ".join(str(int(((t0 <= f * end.max()) & (end > f * end.max())).sum())) for f in (0.6, 0.7, 0.8, 0.9)))
for lv in sorted(set(level.tolist())):
    k = level == lv
    print(f"[trace] level {lv}: {k.sum()} queries, cells taken mean {pops[k].mean():.1f} max {pops[k].max()}, "
          f"time mean {dt[k].mean():.1f} max {dt[k].max():.1f} us, us per cell {dt[k].sum() / max(1, pops[k].sum()):.2f}")
late = np.argsort(end)[-10:]
print("[trace] the ten last to finish: " + "; ".join(f"start {t0[i]:.0f} dt {dt[i]:.0f} cells {pops[i]} level {level[i]}" for i in late))
if os.environ.get("VGICP_TRACE_XCD"):  # a -DVGICP_PREP_TRACE=3 build: the level field carries blockIdx % 8 (the XCD)
    for x in range(8):
        k = level == x
        print(f"[trace] xcd {x}: {k.sum()} queries, busy sum {dt[k].sum() / 1e3:.1f} ms, first start {t0[k].min():.1f}, last start {t0[k].max():.1f}, last end {end[k].max():.1f} us")
if os.environ.get("VGICP_TRACE_SIM"):  # list scheduling of the measured durations on as many slots as ran at once
    import heapq
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
    print(f"[trace] {slots} slots; makespan as dispatched {makespan(by_start):.1f} us, sum/slots {dt.sum() / slots:.1f} us, "
          f"by level descending {makespan(by_start[np.argsort(-level[by_start], kind='stable')]):.1f} us, "
          f"by the level field ascending {makespan(by_start[np.argsort(level[by_start], kind='stable')]):.1f} us, "
          f"by cells taken descending {makespan(np.argsort(-pops, kind='stable')):.1f} us, "
          f"longest first {makespan(np.argsort(-dt)):.1f} us")
