#!/usr/bin/env python3
"""Soak test of the in-kernel row exchange of the persistent launch (GPU box): for SECONDS seconds, random scan
sizes (1 .. 300k points: idle, single-point and multi-point workgroups), random round counts (every buffer takes
its turn as the last one), converging and forced runs, each compared with the one-launch-per-round loop — bit for
bit where both partition the points alike, to rounding otherwise.  With --load a second stream keeps the device
busy with unrelated kernels of uneven length (torch matmuls), which takes compute units away from the persistent
launch now and then: it must then give up cleanly, the align must still return the loop's result, and the single
launch must come back afterwards.
usage: python tools/soak_exchange.py [seconds] [--load]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--load" in sys.argv:
    import torch  # noqa: E402,F401  BEFORE libvgicp_hip.so is loaded: one HIP runtime per process
from eskf_lio_amd import capi, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 60.0
load = "--load" in sys.argv
vmap = synth.make_map(200_000)
big_p, big_c = synth.make_uniform_scan(300_000, vmap, seed=5)
spts, scovs, _ = synth.make_structured_scan(20_000, vmap)
g = synth.default_guess()
rng = np.random.default_rng(12345)
bg = None
if load:
    import torch
    side = torch.cuda.Stream()
    mats = [torch.randn(s, s, device="cuda") for s in (256, 1024, 4096)]
with capi.Context(0) as ctx:
    ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    t0 = time.time()
    aligns = mismatches = single = 0
    while time.time() - t0 < seconds:
        if load:
            with torch.cuda.stream(side):
                for _ in range(int(rng.integers(1, 6))):
                    m = mats[int(rng.integers(0, 3))]
                    (m @ m).sum()
        kind = int(rng.integers(0, 10))
        if kind == 0:
            a = ctx.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
            b = ctx.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999, flags=capi.FLAG_NO_PERSISTENT)
            same = a.iterations == b.iterations and np.array_equal(a.normal_eq, b.normal_eq) and a.converged == b.converged
        else:
            n = int(rng.integers(1, 300_000)) if kind < 8 else int(rng.choice([1, 448, 449, 114_688, 114_689]))
            rounds = int(rng.integers(1, 13))
            ctx.scan_upload(big_p[:n], big_c[:n])
            a = ctx.align_resident(g, rounds, 1e-6, 2.0, allow_degenerate=True)
            b = ctx.align_resident(g, rounds, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT, allow_degenerate=True)
            if n <= 114_688:
                same = np.array_equal(a.normal_eq, b.normal_eq, equal_nan=True)
            else:
                same = np.allclose(a.normal_eq, b.normal_eq, rtol=1e-10, atol=1e-6, equal_nan=True)
            same = same and np.array_equal(a.corr_count, b.corr_count)
        aligns += 1
        single += a.launches == 1
        if not same:
            mismatches += 1
            print(f"MISMATCH at align {aligns}: kind {kind}", flush=True)
    print(f"[soak] {aligns} aligns in {time.time() - t0:.0f} s ({'with' if load else 'without'} competing load): "
          f"{mismatches} mismatches, {single} ran as a single launch, {ctx.counter(1)} gave up and fell back "
          f"(of {ctx.counter(0)} single launches tried)", flush=True)
    sys.exit(1 if mismatches else 0)
