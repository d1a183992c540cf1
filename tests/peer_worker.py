#!/usr/bin/env python3
"""Child process of test_peer_exchange_two_processes_one_device: one rank of a device-initiated exchange.

usage: peer_worker.py <rank> <world> <dir> <points> <rounds> [giveup]
"giveup": rank 0 runs with a tiny spin limit, the other ranks start their (one-round) align half a second late: rank 0
gives up waiting, the late ranks find its row already in their mailboxes — and must NOT return success on their own:
the outcome of an align is collective (verdict words at the end of the launch). There is no RCCL communicator in this
set-up, so every rank has to fail with VGICP_ERR_RCCL; exit code 0 = it did.
Every rank is its own process with its own vgicp context on device 0 (the 1-GPU box has one device; on a
multi-GPU node the ranks would sit on different GPUs and the mailbox stores would cross xGMI).  The mailbox
handles travel through files in <dir>.  Exit code 0 = the sharded aligns completed through the mailboxes and
agree with the whole scan registered alone; anything else (a launch that gave up, a mismatch) is non-zero.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eskf_lio_amd import capi, synth  # noqa: E402
from eskf_lio_amd.distributed import shard_bounds  # noqa: E402


def wait_for(paths, seconds=60.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > seconds:
            raise SystemExit(f"timed out waiting for {paths}")
        time.sleep(0.01)


def giveup_case(rank, world, d, n):
    if rank == 0:
        os.environ["VGICP_SPIN_LIMIT"] = "100"       # x20 between ranks: a few milliseconds
    vmap = synth.make_map(50_000)
    pts, covs = synth.make_uniform_scan(n, vmap, seed=4242)
    g = synth.default_guess()
    lo, hi = shard_bounds(n, world, rank)
    with capi.Context(0) as ctx:
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        with open(os.path.join(d, f"handle{rank}.tmp"), "wb") as f:
            f.write(ctx.peer_export())
        os.rename(os.path.join(d, f"handle{rank}.tmp"), os.path.join(d, f"handle{rank}"))
        wait_for([os.path.join(d, f"handle{r}") for r in range(world)])
        ctx.peer_connect(world, rank, b"".join(open(os.path.join(d, f"handle{r}"), "rb").read() for r in range(world)))
        open(os.path.join(d, f"connected{rank}"), "w").close()
        wait_for([os.path.join(d, f"connected{r}") for r in range(world)])
        if rank != 0:
            time.sleep(0.5)
        code = capi.OK
        try:
            ctx.align(pts[lo:hi], covs[lo:hi], g, 1, 1e-6, 2.0)      # ONE round: the round the align ends in
        except capi.VgicpError as e:
            code = e.code
        with open(os.path.join(d, f"code{rank}"), "w") as f:
            f.write(str(code))
        open(os.path.join(d, f"done{rank}"), "w").close()
        wait_for([os.path.join(d, f"done{r}") for r in range(world)])
        ctx.peer_disconnect()
    if code != capi.ERR_RCCL:
        raise SystemExit(f"rank {rank}: status {code}, expected {capi.ERR_RCCL} on every rank")
    print(f"rank {rank}: gave up together with its peers")


def main():
    rank, world, d, n, rounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    if len(sys.argv) > 6 and sys.argv[6] == "giveup":
        return giveup_case(rank, world, d, n)
    vmap = synth.make_map(50_000)
    pts, covs = synth.make_uniform_scan(n, vmap, seed=4242)
    g = synth.default_guess()
    lo, hi = shard_bounds(n, world, rank)
    with capi.Context(0) as alone, capi.Context(0) as ctx:
        for c in (alone, ctx):
            c.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            c.map_upsert(vmap.keys, vmap.means, vmap.covs)
        want = alone.align(pts, covs, g, rounds, 1e-6, 2.0)            # the whole scan, no exchange between ranks
        assert want.launches == 1
        handle = ctx.peer_export()
        tmp = os.path.join(d, f"handle{rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(handle)
        os.rename(tmp, os.path.join(d, f"handle{rank}"))
        wait_for([os.path.join(d, f"handle{r}") for r in range(world)])
        handles = b"".join(open(os.path.join(d, f"handle{r}"), "rb").read() for r in range(world))
        ctx.peer_connect(world, rank, handles)
        open(os.path.join(d, f"connected{rank}"), "w").close()
        wait_for([os.path.join(d, f"connected{r}") for r in range(world)])   # the barrier the header asks for
        results = []
        for k in range(6):
            r = ctx.align(pts[lo:hi], covs[lo:hi], g, rounds if k % 2 == 0 else max(1, rounds // 2), 1e-6, 2.0)
            if r.launches != 1 or ctx.counter(1) != 0:
                raise SystemExit(f"rank {rank}: align {k} did not complete through the mailboxes "
                                 f"(launches {r.launches}, gave up {ctx.counter(1)})")
            assert r.world_size == world
            results.append(r)
        got = results[0]
        assert got.iterations == want.iterations == rounds
        assert np.array_equal(got.corr_count, want.corr_count), (got.corr_count, want.corr_count)
        scale = np.abs(want.normal_eq).max()
        assert np.allclose(got.normal_eq, want.normal_eq, rtol=1e-10, atol=1e-12 * scale)
        assert np.abs(got.pose - want.pose).max() < 1e-11
        assert np.array_equal(results[2].pose, got.pose) and np.array_equal(results[4].normal_eq, got.normal_eq)
        np.savez(os.path.join(d, f"result{rank}.npz"), pose=got.pose, normal_eq=got.normal_eq, half_pose=results[1].pose)
        # nobody leaves (and unmaps its mailbox) while a peer may still be writing into it
        open(os.path.join(d, f"done{rank}"), "w").close()
        wait_for([os.path.join(d, f"done{r}") for r in range(world)])
        ctx.peer_disconnect()
    print(f"rank {rank}: ok")


if __name__ == "__main__":
    main()
