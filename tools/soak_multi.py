#!/usr/bin/env python3
"""Soak test of the in-process multi-device context (GPU box; the sub-contexts share device 0): for SECONDS seconds,
random scan sizes (0 .. 300k points: empty shards, one-point shards, shards that fit a sub-context's grid and shards
that give every thread several points), random round counts, converging and forced runs, host buffers and resident
scans, interleaved with map mutations (replicated insert / erase) and the resident frame calls — every align compared
with a single-device context on the same map: identical correspondence counts, normal equations and pose to rounding
(sharding only regroups the sums), and the same bits when the multi-device align is repeated.
usage: GPU_MAX_HW_QUEUES=24 python tools/soak_multi.py [seconds] [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
vmap = synth.make_map(200_000)
big_p, big_c = synth.make_uniform_scan(300_000, vmap, seed=5)
spts, scovs, _ = synth.make_structured_scan(20_000, vmap)
g = synth.default_guess()
rng = np.random.default_rng(4242)
with capi.Context(0) as one, capi.Context([0] * world) as ctx:
    for c in (one, ctx):
        c.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        c.map_upsert(vmap.keys, vmap.means, vmap.covs)
    t0 = time.time()
    aligns = mismatches = repeats = mutations = 0
    while time.time() - t0 < seconds:
        kind = int(rng.integers(0, 12))
        if kind == 0:                                   # a converging run: every device must stop in the same round
            a = ctx.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
            b = one.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
        elif kind == 1:                                 # the map changes on every replica
            k = int(rng.integers(0, 190_000))
            sel = slice(k, k + 2_000)
            if rng.integers(0, 2):
                for c in (one, ctx):
                    c.map_erase(vmap.keys[sel])
            else:
                for c in (one, ctx):
                    c.map_upsert(vmap.keys[sel], vmap.means[sel] + 0.001, vmap.covs[sel])
            mutations += 1
            assert one.map_size()[0] == ctx.map_size()[0]
            continue
        else:
            n = int(rng.integers(0, 300_000)) if kind < 9 else int(rng.choice([0, 1, world - 1, world, world + 1, 448 * world, 448 * world + 1]))
            rounds = int(rng.integers(1, 13))
            if kind % 2:
                a = ctx.align(big_p[:n], big_c[:n], g, rounds, 1e-6, 2.0, allow_degenerate=True)
            else:
                ctx.scan_upload(big_p[:n], big_c[:n])
                a = ctx.align_resident(g, rounds, 1e-6, 2.0, allow_degenerate=True)
                if rng.integers(0, 4) == 0:             # the same again: the same bits
                    a2 = ctx.align_resident(g, rounds, 1e-6, 2.0, allow_degenerate=True)
                    repeats += 1
                    if not (np.array_equal(a.pose, a2.pose, equal_nan=True) and np.array_equal(a.normal_eq, a2.normal_eq, equal_nan=True)):
                        mismatches += 1
                        print(f"NOT REPEATABLE at align {aligns}: n {n} rounds {rounds}", flush=True)
            b = one.align(big_p[:n], big_c[:n], g, rounds, 1e-6, 2.0, allow_degenerate=True)
        # a handful of correspondences makes the 6x6 system (near-)singular: the step then amplifies the last-bit
        # differences of the regrouped sums, so only round 0 (same pose on both sides) is comparable there
        few = a.corr_count.size == 0 or a.corr_count.min() < 200
        rounds_cmp = 1 if few else a.iterations
        same = (a.status == b.status or few) and a.iterations == b.iterations and a.corr_count[:rounds_cmp].tolist() == b.corr_count[:rounds_cmp].tolist()
        same = same and np.allclose(a.normal_eq[:rounds_cmp], b.normal_eq[:rounds_cmp], rtol=1e-9, atol=1e-6, equal_nan=True)
        if not few:
            same = same and np.allclose(a.pose, b.pose, rtol=0, atol=1e-9, equal_nan=True)
        aligns += 1
        if not same:
            mismatches += 1
            n_now = locals().get("n", -1)
            print(f"MISMATCH at align {aligns}: kind {kind} n {n_now} status {a.status}/{b.status} iterations {a.iterations}/{b.iterations} "
                  f"launches {a.launches} counts {a.corr_count[:4].tolist()} vs {b.corr_count[:4].tolist()} "
                  f"pose delta {np.nanmax(np.abs(a.pose - b.pose)):.2e}", flush=True)
    print(f"[soak multi] N = {world}: {aligns} aligns ({repeats} repeated for bits) and {mutations} map mutations in {time.time() - t0:.0f} s: "
          f"{mismatches} mismatches; {ctx.counter(0)} aligns tried as single launches, {ctx.counter(1)} fell back to the host-summed loop",
          flush=True)
    sys.exit(1 if mismatches else 0)
