import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth
n = int(sys.argv[1])
vmap = synth.make_map(50_000)
pts, covs = synth.make_uniform_scan(5_000, vmap)
big_pts, big_covs = synth.make_uniform_scan(70_000 * n // 2 if n <= 2 else 140_000, vmap, seed=77)
g = synth.default_guess()
def fresh():
    ctx = capi.Context([0] * n)
    ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0]); ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    return ctx
def run(label, fn):
    with fresh() as ctx:
        t0 = time.perf_counter(); out = fn(ctx)
        print(f"{label}: launches {out} fallbacks {ctx.counter(1)} in {time.perf_counter()-t0:.2f} s", flush=True)
S = lambda c, r: c.align(pts, covs, g, r, 1e-6, 2.0).launches
B = lambda c, r: c.align(big_pts, big_covs, g, r, 1e-6, 2.0).launches
run("small20 big6", lambda c: [S(c, 20), B(c, 6)])
run("small1 big6", lambda c: [S(c, 1), B(c, 6)])
run("small2 big6", lambda c: [S(c, 2), B(c, 6)])
run("small3 big6", lambda c: [S(c, 3), B(c, 6)])
run("big6 small20 big6", lambda c: [B(c, 6), S(c, 20), B(c, 6)])
run("big6 big6 small3 small3", lambda c: [B(c, 6), B(c, 6), S(c, 3), S(c, 3)])
