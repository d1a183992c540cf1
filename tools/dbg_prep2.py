import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from eskf_lio_amd import capi, synth
from oracle import binding as ob
pts = synth.make_lidar_scan(15_000, seed=5, extent=10.0)
rp, rc, ri = ob.preprocess(pts, 2.0, 30)
with capi.Context(0) as ctx:
    for dbg in ("1000", "1"):
        os.environ["VGICP_DEBUG_PREP"] = dbg
        gp, gc, gi = ctx.preprocess(pts, 2.0, 30)
        print(dbg, "bad", int((np.abs(gc - rc).max(axis=1) > 0).sum()))
