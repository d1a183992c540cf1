"""Developer probe: vgicp_map_upsert of the C2 (1M voxels) and C5 (10M voxels) maps, a growth (rehash) and a re-upsert
(update in place); run under rocprofv3 --kernel-trace --stats to read upsert_kernel / rehash_kernel times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

for voxels in (1_000_000, 10_000_000):
    vmap = synth.make_map(voxels)
    with capi.Context(0) as ctx:
        ctx.map_reset(vmap.voxel_size, voxels)
        t0 = time.perf_counter()
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        t1 = time.perf_counter()
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)             # every record exists: update in place
        t2 = time.perf_counter()
        assert ctx.map_size()[0] == voxels
        print(f"{voxels} voxels: upsert {1e3*(t1-t0):.2f} ms, re-upsert {1e3*(t2-t1):.2f} ms (host wall incl. the copies)")
    with capi.Context(0) as ctx:                                     # growth from a small table: rehashes on the way
        ctx.map_reset(vmap.voxel_size, 0)
        step = voxels // 8
        for k in range(8):
            ctx.map_upsert(vmap.keys[k * step:(k + 1) * step], vmap.means[k * step:(k + 1) * step], vmap.covs[k * step:(k + 1) * step])
        assert ctx.map_size()[0] == step * 8
        k, m, c, cnt = ctx.map_export()
        o = np.lexsort(k.T)
        o2 = np.lexsort(vmap.keys[:step * 8].T)
        assert np.array_equal(k[o], vmap.keys[:step * 8][o2]) and np.array_equal(m[o], vmap.means[:step * 8][o2])
        assert np.array_equal(c[o], vmap.covs[:step * 8][o2]) and (cnt == 1).all()
        print(f"{voxels} voxels in 8 growing batches: table {ctx.map_size()}, export equal")
