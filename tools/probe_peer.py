#!/usr/bin/env python3
"""Developer probe: what the mailbox hop of the device-initiated exchange adds per round, measured with RANKS
processes sharing ONE device (the round's boxes have one GPU; on a multi-GPU node the same stores cross xGMI).
Every rank registers its shard of a 40k-point scan for 20 forced rounds on VGICP_PERSIST_GRID workgroups, through
the mailboxes and — the same shard, the same grid — alone.
usage: python tools/probe_peer.py [ranks] [grid]      (spawns the ranks itself)"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def wait_for(paths, seconds=60.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > seconds:
            raise SystemExit(f"timed out waiting for {paths}")
        time.sleep(0.005)


def child(rank, world, d):
    from eskf_lio_amd import capi, synth
    from eskf_lio_amd.distributed import shard_bounds
    vmap = synth.make_map(200_000)
    pts, covs = synth.make_uniform_scan(40_000, vmap, seed=77)
    g = synth.default_guess()
    lo, hi = shard_bounds(pts.shape[0], world, rank)
    with capi.Context(0) as alone, capi.Context(0) as ctx:
        for c in (alone, ctx):
            c.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            c.map_upsert(vmap.keys, vmap.means, vmap.covs)
            c.scan_upload(pts[lo:hi], covs[lo:hi])
        open(os.path.join(d, f"h{rank}.tmp"), "wb").write(ctx.peer_export())
        os.rename(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}"))
        wait_for([os.path.join(d, f"h{r}") for r in range(world)])
        ctx.peer_connect(world, rank, b"".join(open(os.path.join(d, f"h{r}"), "rb").read() for r in range(world)))
        open(os.path.join(d, f"c{rank}"), "w").close()
        wait_for([os.path.join(d, f"c{r}") for r in range(world)])
        spans = {"mailboxes": [], "alone": []}
        for k in range(60):
            r = ctx.align_resident(g, 20, 1e-6, 2.0)
            assert r.launches == 1 and r.world_size == world
            if k >= 10:
                spans["mailboxes"].append(r.device_seconds)
        open(os.path.join(d, f"m{rank}"), "w").close()
        wait_for([os.path.join(d, f"m{r}") for r in range(world)])       # nobody times "alone" beside a peer's exchange
        for k in range(60):
            r = alone.align_resident(g, 20, 1e-6, 2.0)
            if k >= 10:
                spans["alone"].append(r.device_seconds)
        print(f"rank {rank}/{world}: {hi - lo} points on {os.environ.get('VGICP_PERSIST_GRID')} workgroups: "
              f"{np.median(spans['mailboxes']) / 20 * 1e6:.2f} us per round through the mailboxes, "
              f"{np.median(spans['alone']) / 20 * 1e6:.2f} us alone (other ranks running too), gave up {ctx.counter(1)}", flush=True)
        open(os.path.join(d, f"d{rank}"), "w").close()
        wait_for([os.path.join(d, f"d{r}") for r in range(world)])
        ctx.peer_disconnect()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    grid = sys.argv[2] if len(sys.argv) > 2 else str(200 // world)
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, VGICP_PERSIST_GRID=grid)
        procs = [subprocess.Popen([sys.executable, __file__, "--child", str(r), str(world), d], env=env) for r in range(world)]
        sys.exit(max(p.wait() for p in procs))
