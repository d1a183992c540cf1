#!/usr/bin/env python3
"""bench.py's frame-chain record with this process (and the threads it starts later) bound to the CPUs of ONE NUMA node:
    python tools/probe_frame_numa.py <node> [frames] [sweep points]
The GPU of these boxes hangs off node 0 (/sys/class/drm/card*/device/numa_node); memory is placed by first touch."""
import json
import os
import sys

node = int(sys.argv[1])
cpus = set()
for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
    lo, _, hi = part.partition("-")
    cpus.update(range(int(lo), int(hi or lo) + 1))
os.sched_setaffinity(0, cpus)          # before anything touches the GPU or starts a thread

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
points = int(sys.argv[3]) if len(sys.argv) > 3 else 60_000
d = bench.frame_chain_leg(0, frames=frames, sweep_points=points, cpu=False)
print(json.dumps({k: round(v, 4) for k, v in d.items() if "ms_per_frame" in k and "host_auth" not in k}))
