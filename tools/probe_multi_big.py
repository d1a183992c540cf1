"""Developer probe: sub-contexts sharing device 0 with shards larger than a sub-context's grid (the
several-points-per-thread instantiation): which variants complete the single launch.  usage: probe_multi_big.py N"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

n = int(sys.argv[1])
vmap = synth.make_map(50_000)
pts, covs = synth.make_uniform_scan(5_000, vmap)
big_pts, big_covs = synth.make_uniform_scan(140_000, vmap, seed=77)
g = synth.default_guess()


def fresh():
    ctx = capi.Context([0] * n)
    ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    return ctx


def run(label, fn):
    with fresh() as ctx:
        t0 = time.perf_counter()
        out = fn(ctx)
        print(f"{label}: launches {out} fallbacks {ctx.counter(1)} in {time.perf_counter()-t0:.2f} s", flush=True)


run("big align first, x3", lambda c: [c.align(big_pts, big_covs, g, 6, 1e-6, 2.0).launches for _ in range(3)])
run("upload then resident x3", lambda c: (c.scan_upload(big_pts, big_covs), [c.align_resident(g, 6, 1e-6, 2.0).launches for _ in range(3)])[1])
run("small x3 then big x2", lambda c: [c.align(pts, covs, g, 20, 1e-6, 2.0).launches for _ in range(3)] + [c.align(big_pts, big_covs, g, 6, 1e-6, 2.0).launches for _ in range(2)])
run("big 20 rounds x3", lambda c: [c.align(big_pts, big_covs, g, 20, 1e-6, 2.0).launches for _ in range(3)])
mid_pts, mid_covs = big_pts[:n * 14_000], big_covs[:n * 14_000]     # fits the grids: one point per thread
run("mid (one point per thread) x3", lambda c: [c.align(mid_pts, mid_covs, g, 6, 1e-6, 2.0).launches for _ in range(3)])
