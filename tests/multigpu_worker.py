#!/usr/bin/env python3
"""Child process of test_two_gpu_sharded_align: one rank (= one GPU) of a point-sharded align through
vgicp_comm_init — RCCL communicator plus, where the GPUs can map each other's memory, the device-initiated
mailbox exchange — checked against the whole scan on one GPU.
usage: multigpu_worker.py <rank> <world> <port>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from eskf_lio_amd import capi, synth
    from eskf_lio_amd.distributed import shard_bounds, share_unique_id
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    vmap = synth.make_map(200_000)
    pts, covs = synth.make_uniform_scan(60_000, vmap)
    g = synth.default_guess()
    lo, hi = shard_bounds(pts.shape[0], world, rank)
    with capi.Context(rank) as alone, capi.Context(rank) as ctx:
        for c in (alone, ctx):
            c.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            c.map_upsert(vmap.keys, vmap.means, vmap.covs)
        want = alone.align(pts, covs, g, 10, 1e-6, 2.0)
        ctx.comm_init(world, rank, share_unique_id(ctx, rank))
        for flags in (0, capi.FLAG_NO_PERSISTENT):                 # mailboxes (if wired), then RCCL per iteration
            got = ctx.align(pts[lo:hi], covs[lo:hi], g, 10, 1e-6, 2.0, flags=flags)
            assert got.world_size == world and got.iterations == 10
            assert np.array_equal(got.corr_count, want.corr_count)
            assert np.abs(got.pose - want.pose).max() < 1e-11
            poses = [torch.zeros(16, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(poses, torch.from_numpy(got.pose.reshape(16).copy()).cuda())
            assert all(torch.equal(poses[0], p) for p in poses)    # every rank: the same bits
            print(f"rank {rank}: flags {flags} launches {got.launches} ok", flush=True)
        dist.barrier()
        ctx.comm_destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
