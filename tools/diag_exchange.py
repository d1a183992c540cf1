#!/usr/bin/env python3
"""Developer diagnostic: persistent launch vs per-launch loop, per-round differences of the 27 sums."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

vmap = synth.make_map(50_000)
big_p, big_c = synth.make_uniform_scan(300_000, vmap, seed=99)
g = synth.default_guess()
with capi.Context(0) as ctx:
    ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    for n, rounds in ((5_000, 6), (20_000, 6), (60_000, 6), (60_000, 3), (60_000, 12), (114_688, 8), (100_000, 20)):
        ctx.scan_upload(big_p[:n], big_c[:n])
        loop = ctx.align_resident(g, rounds, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
        bad_rounds = {}
        self_diff = 0
        prev = None
        for rep in range(20):
            one = ctx.align_resident(g, rounds, 1e-6, 2.0)
            d = np.abs(one.normal_eq - loop.normal_eq)
            scale = np.abs(loop.normal_eq).max(axis=1, keepdims=True)
            rel = (d / scale).max(axis=1)
            for r in np.nonzero(rel > 0)[0]:
                bad_rounds.setdefault(int(r), []).append(float(rel[r]))
            if prev is not None and not np.array_equal(prev.normal_eq, one.normal_eq):
                self_diff += 1
            prev = one
        print(f"n={n} rounds={rounds}: rounds that differ from the loop (round: count, max rel diff): "
              f"{ {r: (len(v), max(v)) for r, v in sorted(bad_rounds.items())} } | persistent runs that differ from the "
              f"previous one: {self_diff}/19 | counts equal: {np.array_equal(one.corr_count, loop.corr_count)} | "
              f"fallbacks {ctx.counter(1)}", flush=True)
