#!/usr/bin/env python3
"""The frame-chain record of bench.py on its own (for rocprofv3 --kernel-trace --stats: per-kernel times of one frame).

    python tools/probe_frame.py [frames] [sweep points]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
points = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
print(json.dumps(bench.frame_chain_leg(0, frames=frames, sweep_points=points, cpu=False)))
