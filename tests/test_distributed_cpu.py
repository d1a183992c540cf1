"""The N > 1 path on CPU: two gloo ranks, contiguous point shards, exchange of the 28-double row.

Without GPUs the per-rank accumulation is done by the oracle (test-only stand-in for the HIP kernel);
what is under test is the host logic the multi-GPU path relies on: shard ownership, the unique-id and
mailbox-handle hand-offs, and that "per-shard normal equations -> exchange -> identical solve on every rank"
reproduces the single-rank result (pose within 1e-12, identical correspondence counts), both with a sum
all-reduce (the RCCL fallback's shape) and with the mailbox rule of the device-initiated exchange: every
rank receives every rank's row and adds them in rank order starting from +0.0.
The HIP sharded path itself is covered on the GPU: tests/test_gpu_parity.py::test_rccl_path_world_size_one,
::test_sharded_loop_on_one_device_matches_the_single_gpu_log, ::test_peer_exchange_two_processes_one_device."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_points, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eskf_lio_amd import synth
    from eskf_lio_amd.distributed import gather_bytes, shard_bounds, share_bytes
    from oracle import binding as oracle

    # 1. the 128-byte id travels from rank 0 to everyone; the 64-byte mailbox handles from everyone to everyone
    uid = share_bytes(lambda: bytes(range(128)), rank, 128)
    assert uid == bytes(range(128))
    handles = gather_bytes(bytes([rank * 7 + k & 0xFF for k in range(64)]), world)
    assert handles == b"".join(bytes([r * 7 + k & 0xFF for k in range(64)]) for r in range(world))

    # 2. sharded registration loop
    vmap = synth.make_map(20_000)
    pts, covs = synth.make_uniform_scan(n_points, vmap)
    lo, hi = shard_bounds(n_points, world, rank)
    om = oracle.OracleMap(vmap.voxel_size, 1)      # map replicated on every rank
    om.insert(vmap.means, vmap.covs)
    for mailbox in (False, True):
        total = synth.default_guess()
        counts = []
        my_pts, my_covs = oracle.transform(pts[lo:hi], covs[lo:hi], total)
        for _ in range(6):
            JTJ, JTr, m = om.accumulate(my_pts, my_covs)
            row = torch.zeros(28, dtype=torch.float64)
            k = 0
            for r in range(6):
                for c in range(r + 1):
                    row[k] = JTJ[r, c]
                    k += 1
            row[21:27] = torch.from_numpy(JTr)
            row[27] = m
            if mailbox:                                      # every rank's row to every rank, added in rank order
                rows = [torch.zeros(28, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(rows, row)
                row = torch.zeros(28, dtype=torch.float64)
                for r in range(world):
                    row = row + rows[r]
            else:
                dist.all_reduce(row, op=dist.ReduceOp.SUM)   # the RCCL all-reduce's stand-in
            full = np.zeros((6, 6))
            k = 0
            for r in range(6):
                for c in range(r + 1):
                    full[r, c] = full[c, r] = row[k].item()
                    k += 1
            _, step = oracle.solve_step(full, row[21:27].numpy())
            total = step @ total
            counts.append(int(row[27].item()))
            my_pts, my_covs = oracle.transform(my_pts, my_covs, step)
        # every rank ends with the same bits (identical solve on identical exchanged rows)
        gathered = [torch.zeros(16, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(total.reshape(16).copy()))
        assert all(torch.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            ref = om.align(pts, covs, synth.default_guess(), 6, 1e-6, 2.0)
            key = "mailbox" if mailbox else "allreduce"
            ret[f"pose_delta_{key}"] = float(np.abs(ref.pose - total).max())
            ret[f"counts_equal_{key}"] = bool(np.array_equal(ref.corr_count, np.array(counts, dtype=np.uint64)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_loop_matches_single_rank():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, 3001, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        for key in ("allreduce", "mailbox"):
            assert ret[f"counts_equal_{key}"]
            assert ret[f"pose_delta_{key}"] < 1e-12
