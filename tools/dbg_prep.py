import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from eskf_lio_amd import capi, synth
pts = synth.make_lidar_scan(15_000, seed=5, extent=10.0)
h = 2.0
lists = []
with capi.Context(0) as ctx:
    for off in (0, 9, 18, 21):
        os.environ["VGICP_DEBUG_PREP"] = str(2 + off)
        gp, gc, gi = ctx.preprocess(pts, h, 30)
        lists.append((off, gc.astype(np.int64)))
full = np.zeros((len(gi), 30), dtype=np.int64)
for off, g in lists:
    full[:, off:off + 9] = g
nbad = 0
for o in range(len(gi)):
    q = pts[int(gi[o])]
    d = ((pts - q) ** 2).sum(axis=1)
    order = np.lexsort((np.arange(len(pts)), d))[:30]
    if not np.array_equal(order, full[o]):
        nbad += 1
        if nbad <= 4:
            print("query", o, int(gi[o]), q)
            print(" truth", order.tolist())
            print(" got  ", full[o].tolist())
            print(" d truth", d[order][[0, 1, 2, 27, 28, 29]], "d got", d[full[o]][[0, 1, 2, 27, 28, 29]])
            miss = sorted(set(order.tolist()) - set(full[o].tolist()))
            print(" missing", miss, pts[miss], d[miss])
print("bad", nbad, "of", len(gi))
