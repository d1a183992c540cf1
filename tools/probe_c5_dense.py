"""Developer probe: C5 (1M points / 10M voxels) persistent launch with and without the dense record copy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth
n, v = synth.CONFIGS["C5"]
vmap = synth.make_map(v)
pts, covs = synth.make_uniform_scan(n, vmap)
g = synth.default_guess()
res = {}
for label, env in (("dense", None), ("plain", "0"), ("dense again", None)):
    if env is not None:
        os.environ["VGICP_DENSE_SLOTS"] = env
    else:
        os.environ.pop("VGICP_DENSE_SLOTS", None)
    with capi.Context(0) as ctx:
        ctx.map_reset(vmap.voxel_size, v)
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        ctx.scan_upload(pts, covs)
        t0 = time.perf_counter()
        r = ctx.align_resident(g, 20, 1e-6, 2.0)
        first = time.perf_counter() - t0
        dev = []
        for _ in range(12):
            r = ctx.align_resident(g, 20, 1e-6, 2.0)
            dev.append(r.device_seconds)
        res[label] = r
        print(f"{label}: first align {first*1e3:.2f} ms (incl. a dense build), then {np.median(dev)/20*1e6:.2f} us per round (min {min(dev)/20*1e6:.2f}), fallbacks {ctx.counter(1)}", flush=True)
print("same bits:", np.array_equal(res["dense"].pose, res["plain"].pose), np.array_equal(res["dense"].normal_eq, res["plain"].normal_eq))
