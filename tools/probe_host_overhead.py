#!/usr/bin/env python3
"""Developer probe: where the host time of one resident align goes (GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

vmap = synth.make_map(1_000_000)
pts, covs = synth.make_uniform_scan(100_000, vmap)
g = synth.default_guess()
with capi.Context(0) as ctx:
    ctx.map_reset(vmap.voxel_size, 1_000_000)
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    ctx.scan_upload(pts, covs)
    for _ in range(10):
        ctx.align_resident(g, 20, 1e-6, 2.0, chunk_iterations=20)
    wall, cwall, dev = [], [], []
    for _ in range(200):
        t0 = time.perf_counter()
        r = ctx.align_resident(g, 20, 1e-6, 2.0, chunk_iterations=20)
        wall.append(time.perf_counter() - t0)
        cwall.append(r.seconds)
        dev.append(r.device_seconds)
    print(f"python wall {1e6 * np.median(wall):.1f} us | C entry point wall {1e6 * np.median(cwall):.1f} us | "
          f"device event span {1e6 * np.median(dev):.1f} us")
