#!/usr/bin/env python3
"""Developer probe: per-launch timing of the iteration kernel on a synthetic config (GPU box only).

usage: python tools/probe.py [C1|C2|C5] [reps]
With VGICP_DEBUG_STAMPS=1 the library also prints the last workgroup's phase times at close.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402



def digest(r):
    """pose bits + correspondence counts + normal equations of the last align, to compare builds in an A/B"""
    import hashlib
    h = hashlib.sha1(r.pose.tobytes() + r.corr_count.tobytes())
    return h.hexdigest()[:12]


cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, v = synth.CONFIGS[cfg]
t0 = time.time()
vmap = synth.make_map(v)
pts, covs = synth.make_uniform_scan(n, vmap)
guess = synth.default_guess()
print(f"[probe] {cfg}: inputs in {time.time() - t0:.1f}s", flush=True)
with capi.Context(0) as ctx:
    ctx.map_reset(vmap.voxel_size, int(os.environ.get("PROBE_MAP_HINT", v)))   # PROBE_MAP_HINT: size the table for fewer voxels (a fuller, smaller table)
    t0 = time.time()
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    print(f"[probe] map upsert {time.time() - t0:.3f}s size {ctx.map_size()}", flush=True)
    t0 = time.time()
    ctx.scan_upload(pts, covs)
    print(f"[probe] scan upload {(time.time() - t0) * 1e3:.3f} ms", flush=True)
    IT = 20
    for _ in range(3):
        ctx.align_resident(guess, IT, 1e-6, 2.0, chunk_iterations=IT)
    spans, walls = [], []
    for _ in range(reps):
        r = ctx.align_resident(guess, IT, 1e-6, 2.0, chunk_iterations=IT)
        spans.append(r.device_seconds)
        walls.append(r.seconds)
    print(f"[probe] eager: device span/iter {np.mean(spans) / IT * 1e6:.2f} us (min {np.min(spans) / IT * 1e6:.2f}), "
          f"host wall/align {np.mean(walls) * 1e3:.3f} ms; result digest {digest(r)}", flush=True)
    kms = []
    for _ in range(reps):
        r = ctx.align_resident(guess, IT, 1e-6, 2.0, flags=capi.FLAG_PROFILE)
        kms.append(r.kernel_ms)
    kms = np.array(kms) * 1e3
    print(f"[probe] per-launch events: mean {kms.mean():.2f} us, min {kms.min():.2f}, first-iter mean {kms[:, 0].mean():.2f}, "
          f"later mean {kms[:, 1:].mean():.2f}", flush=True)
    m = float(r.corr_count.mean())
    b = 112.0 * n + 96.0 * m
    print(f"[probe] algorithmic bytes/launch {b / 1e6:.2f} MB -> {b / (kms.mean() * 1e-6) / 1e9:.0f} GB/s (events), "
          f"{b / (np.mean(spans) / IT) / 1e9:.0f} GB/s (span)", flush=True)
