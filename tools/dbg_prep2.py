import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from eskf_lio_amd import capi, synth
from oracle import binding as ob
for n, h in ((15_000, 2.0), (100_000, 0.3)):
    pts = synth.make_lidar_scan(n, seed=5, extent=10.0 if n < 50_000 else 40.0)
    rp, rc, ri = ob.preprocess(pts, h, 30)
    with capi.Context(0) as ctx:
        gp, gc, gi = ctx.preprocess(pts, h, 30)
    print(n, h, "kept equal", np.array_equal(gi, ri), "bad covs", int((np.abs(gc - rc).max(axis=1) > 0).sum()))
