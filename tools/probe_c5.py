#!/usr/bin/env python3
"""A/B of context-level knobs at BASELINE config C5 (1M points vs 10M voxels, 20 forced rounds, scan resident) in ONE
process on ONE box: the map and the scan are generated once, every setting gets its own context.

    python tools/probe_c5.py VGICP_WARM_POINTS=0 VGICP_WARM_POINTS=4 VGICP_WARM_POINTS=8
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eskf_lio_amd import capi, synth  # noqa: E402


def main():
    settings = sys.argv[1:] or ["VGICP_WARM_POINTS=0", "VGICP_WARM_POINTS=4"]
    n, v = synth.CONFIGS["C5"]
    vmap = synth.make_map(v)
    pts, covs = synth.make_uniform_scan(n, vmap)
    g = synth.default_guess()
    ref = None
    for rep in range(2):
        for s in settings:
            for kv in s.split(","):
                k, val = kv.split("=")
                os.environ[k] = val
            with capi.Context(0) as ctx:
                ctx.map_reset(vmap.voxel_size, v)
                ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
                ctx.scan_upload(pts, covs)
                for _ in range(3):
                    r = ctx.align_resident(g, 20, 1e-6, 2.0, chunk_iterations=20)
                dev = []
                for _ in range(10):
                    r = ctx.align_resident(g, 20, 1e-6, 2.0, chunk_iterations=20)
                    dev.append(r.device_seconds)
                if ref is None:
                    ref = r
                same = bool(np.array_equal(ref.corr_count, r.corr_count)) and float(np.abs(ref.pose - r.pose).max()) < 1e-11
                print(f"{s:40s} {np.median(dev) / 20 * 1e6:7.2f} us/round (min {min(dev) / 20 * 1e6:.2f}) launches {r.launches} "
                      f"same result {same}", flush=True)
            for kv in s.split(","):
                os.environ.pop(kv.split("=")[0], None)


if __name__ == "__main__":
    main()
