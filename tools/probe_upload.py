#!/usr/bin/env python3
"""Developer probe: host-to-device scan upload (vgicp_scan_upload) and the whole vgicp_align with the scan in
ordinary host buffers. (Round 2 also tried staging through a pinned buffer with 2-12 copy threads: 26-31 GB/s
against 46 GB/s for the plain hipMemcpyAsync of pageable memory, gpurun_out/r02b; dropped.)
usage: python tools/probe_upload.py [C2|C5] [reps]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from eskf_lio_amd import capi, synth
    cfg, reps = sys.argv[2], int(sys.argv[3])
    n, v = synth.CONFIGS[cfg]
    vmap = synth.make_map(v)
    pts, covs = synth.make_uniform_scan(n, vmap)
    guess = synth.default_guess()
    with capi.Context(0) as ctx:
        ctx.map_reset(vmap.voxel_size, v)
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for _ in range(3):
            ctx.scan_upload(pts, covs)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.scan_upload(pts, covs)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts)
        mb = 96.0 * n / 1e6
        for _ in range(3):
            ctx.align(pts, covs, guess, 20, 1e-6, 2.0)
        ws, ds = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = ctx.align(pts, covs, guess, 20, 1e-6, 2.0)
            ws.append(time.perf_counter() - t0)
            ds.append(r.device_seconds)
        ws = np.array(ws)
        print(f"upload {mb:.1f} MB median {np.median(ts) * 1e3:.3f} ms "
              f"(min {ts.min() * 1e3:.3f}) = {mb / 1e3 / np.median(ts):.1f} GB/s | align incl. upload median {np.median(ws) * 1e3:.3f} ms "
              f"(min {ws.min() * 1e3:.3f}), kernel span {np.median(ds) * 1e3:.3f} ms -> {n * 20 / np.median(ws) / 1e9:.2f} G points/s",
              flush=True)
    sys.exit(0)

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = sys.argv[2] if len(sys.argv) > 2 else "30"
subprocess.run([sys.executable, __file__, "--child", cfg, reps], check=False)
