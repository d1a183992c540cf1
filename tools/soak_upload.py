#!/usr/bin/env python3
"""Soak of the staged scan upload (copy crew + pack_arena_kernel): random scan sizes, fresh buffers that are freed right
after the call, every result compared bit for bit with the same scan handed to the runtime in place
(VGICP_OPTION_UPLOAD_STAGE_KB = 0) on a second context.     python tools/soak_upload.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eskf_lio_amd import capi, synth  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
vmap = synth.make_map(200_000)
rng = np.random.default_rng(7)
g = synth.default_guess()
pool_p, pool_c = synth.make_uniform_scan(300_000, vmap, seed=11)
t_end, n_runs, slow, worst = time.time() + seconds, 0, 0, 0.0
with capi.Context(0) as a, capi.Context(0) as b:
    for c in (a, b):
        c.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        c.map_upsert(vmap.keys, vmap.means, vmap.covs)
    b.set_option(capi.OPTION_UPLOAD_STAGE_KB, 0)
    while time.time() < t_end:
        n = int(rng.choice([rng.integers(1, 3000), rng.integers(2700, 70_000), rng.integers(70_000, 300_000)]))
        lo = int(rng.integers(0, 300_000 - n + 1))
        p, c = pool_p[lo:lo + n].copy(), pool_c[lo:lo + n].copy()
        t0 = time.perf_counter()
        ra = a.align(p, c, g, 4, 1e-6, 2.0)
        worst = max(worst, time.perf_counter() - t0)
        del p, c
        rb = b.align(pool_p[lo:lo + n], pool_c[lo:lo + n], g, 4, 1e-6, 2.0)
        if not (np.array_equal(ra.pose, rb.pose) and np.array_equal(ra.corr_count, rb.corr_count) and np.array_equal(ra.normal_eq, rb.normal_eq)):
            print(f"MISMATCH at run {n_runs}: n = {n}")
            sys.exit(1)
        n_runs += 1
    slow = a.counter(capi.COUNTER_UPLOAD_SLOW)
print(f"staged upload soak: {n_runs} aligns of 1 .. 300 000 points in {seconds:.0f} s, every one bit-equal to the in-place upload; "
      f"uploads repeated because the copy threads were held up: {slow}; slowest staged align {worst * 1e3:.2f} ms; "
      f"VGICP_UPLOAD_THREADS = {os.environ.get('VGICP_UPLOAD_THREADS', '3 (default)')}")
