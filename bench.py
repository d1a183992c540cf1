#!/usr/bin/env python3
"""bench.py — registered points/s per VGICP iteration on MI355X (BASELINE.json metric).

A "step" is one whole ICP::align over the resident scan: 20 VGICP iterations (cosine_threshold 2.0
forces all of them) of config C2 — a 100k-point synthetic uniform scan against a
1M-voxel synthetic map.  With --gpus N (launched by torch.distributed.run, one rank per GPU) there are
two ways to use the ranks, and the warm-up measures both:
  * SHARD: the one scan is split in contiguous blocks over the ranks, the map is replicated, every
    iteration ends in one RCCL all-reduce of the 28-double normal-equation row ("scaling": "strong");
  * REPLICAS: every rank registers its OWN 100k-point scan (same generator, seed + rank) against the
    replicated map with the single-launch loop and no communication ("scaling": "weak": per-GPU work is
    fixed, `value` counts the points of all N scans).
A C2 round takes ≈10 us on one GPU, less than one host-enqueued all-reduce, so sharding a 100k-point scan
cannot pay (it does from about C5); the timed region therefore runs whichever of the two
delivers more registered points per second (`config.sharding`, `config.sharding_autotune` report the
choice and both rates; BENCH_SHARDING=shard|replicate forces one; `multi_gpu_parity` always exercises and
checks the sharded path against the single-GPU result).

Timed region: inputs already resident in HBM (scan uploaded, map built) — barrier +
torch.cuda.synchronize() on both sides, K steps, max over ranks.  `roofline` is measured live over
the same timed region: every align brackets its iteration launches with a HIP event pair on the
module's own stream (stats.device_seconds), so launch time = event span / body launches, kernel
boundaries included; a second pass with an event pair around EVERY launch (VGICP_FLAG_PROFILE) is
reported beside it.  `roofline.traffic` is the PMC-measured HBM traffic per launch of the newest
profiles/*_summary.json (tools/profile_gpu.sh), or null.  `cpu_baseline` times the CPU oracle's
reference-faithful mode (OpenMP, all host cores) on the same inputs, rank 0 at N=1 only.

PyTorch is plumbing here (torch.distributed rendezvous/barrier, device sync); the path itself is
the C-ABI HIP module.  The oracle is touched only by the cpu_baseline leg.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from eskf_lio_amd import capi, synth  # noqa: E402
from eskf_lio_amd.distributed import shard_bounds, share_unique_id  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s achievable)
ITERATIONS = 20


def algorithmic_bytes(n_points: int, matches: float) -> float:
    """112 B per point-iteration (24 point + 72 covariance + 16 hash slot) plus
    96 B per matched point (24 voxel mean + 72 voxel covariance)."""
    return 112.0 * n_points + 96.0 * matches


def measured_traffic(n_points: int, world: int, kernel: str):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary taken at this scan
    size on one GPU (profiles/*_summary.json, tools/profile_gpu.sh + tools/summarize_profile.py)."""
    import glob
    if world != 1:
        return None
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            t = d.get("traffic")
            points = d.get("bench", {}).get("config", {}).get("points", 100_000)
        except Exception:
            continue
        key = f"{kernel}_total_calibrated"
        if t and points == n_points and key in t and t[key] == t[key]:
            best = t[key]
    return best


def cpu_baseline(vmap, pts, covs, guess, budget_s: float):
    """The reference-faithful CPU path (oracle, OpenMP) on the same inputs; bounded wall time."""
    from oracle import binding as oracle
    omap = oracle.OracleMap(vmap.voxel_size, 1)  # maxNumPointsPerVoxel = 1
    omap.insert(vmap.means, vmap.covs)
    times = []
    t_start = time.time()
    reps = 0
    while True:
        r = omap.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0, mode=oracle.FAITHFUL)
        reps += 1
        if reps > 2:  # two warm-ups
            times.append(r.seconds)
        if (len(times) >= 10) or (time.time() - t_start > budget_s and len(times) >= 3):
            break
    med = float(np.median(times))
    n = pts.shape[0]
    return {
        "value": n * r.iterations / med,
        "unit": "points/s",
        "cores": int(r.threads),
        "kind": "port",
        "sample": f"full workload: {n}-pt scan x {r.iterations} iterations vs {len(omap)}-voxel map, "
                  f"median of {len(times)} aligns after 2 warm-ups ({med * 1e3:.1f} ms each), "
                  f"oracle faithful mode (OpenMP, -O3, no -march)",
    }, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with python -m torch.distributed.run "
                             "--nproc-per-node N (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_COMM=1 runs the multi-GPU code path (process group, unique-id hand-off, RCCL
    # all-reduce per iteration) with however many ranks there are — on one GPU a way to exercise it
    force_comm = os.environ.get("BENCH_FORCE_COMM", "0") == "1"
    use_dist = world > 1 or force_comm
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n_points, n_voxels = synth.CONFIGS[args.config]
    vmap = synth.make_map(n_voxels)
    pts, covs = synth.make_uniform_scan(n_points, vmap)
    guess = synth.default_guess()
    lo, hi = shard_bounds(n_points, world, rank)

    ctx = capi.Context(local_rank)
    ctx.map_reset(vmap.voxel_size, n_voxels)
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    shard_error = None
    if use_dist:
        # The module's own RCCL communicator (the sharded loop's all-reduce). If it cannot be set up on this
        # box the independent-scans mode still runs; every rank has to see the same verdict.
        try:
            ctx.comm_init(world, rank, share_unique_id(ctx, rank))
        except Exception as e:  # noqa: BLE001 - reported in the JSON line
            shard_error = f"{type(e).__name__}: {e}"
        bad = torch.tensor([1 if shard_error else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) and shard_error is None:
            shard_error = "communicator set-up failed on another rank"
            ctx.comm_destroy()
    if shard_error is None:
        ctx.scan_upload(pts[lo:hi], covs[lo:hi])

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # With several ranks there are two ways to register one scan: SHARD it (each rank a contiguous
    # block of points, one RCCL all-reduce of the 28-double row per iteration) or REPLICATE it (every
    # rank registers the whole scan with the single-launch loop, no communication).  Sharding only pays
    # once a shard is large enough to amortise the all-reduce, so the host side measures both during
    # the warm-up and times the faster one (BENCH_SHARDING=shard|replicate forces a choice).
    run_ctx, n_local, mode, tuning = ctx, hi - lo, "single", None
    if use_dist:
        solo = capi.Context(local_rank)
        solo.map_reset(vmap.voxel_size, n_voxels)
        solo.map_upsert(vmap.keys, vmap.means, vmap.covs)
        own_pts, own_covs = synth.make_uniform_scan(n_points, vmap, seed=synth.SCAN_SEED + rank)
        solo.scan_upload(own_pts, own_covs)              # rank 0 keeps the common scan (seed + 0)
        timing = {}
        modes = (("replicate", solo),) if shard_error else (("shard", ctx), ("replicate", solo))
        for name, c in modes:
            for _ in range(2):
                c.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)
            fence()
            t0 = time.perf_counter()
            for _ in range(max(args.warmup, 3)):
                c.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)
            fence()
            t = torch.tensor([(time.perf_counter() - t0) / max(args.warmup, 3)], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)     # every rank sees the same numbers -> same choice
            timing[name] = float(t.item())
        want = os.environ.get("BENCH_SHARDING", "auto")
        rate = {"replicate": world * n_points * ITERATIONS / timing["replicate"]}   # one scan per rank per step
        if "shard" in timing:
            rate["shard"] = n_points * ITERATIONS / timing["shard"]                  # one scan per step
        mode = want if want in rate else max(rate, key=rate.get)
        tuning = {"policy": want, "ms_per_step_replicate": timing["replicate"] * 1e3,
                  "points_per_s_replicate": rate["replicate"],
                  "ms_per_step_shard": timing["shard"] * 1e3 if "shard" in timing else None,
                  "points_per_s_shard": rate.get("shard"), "shard_error": shard_error}
        if mode == "replicate":
            run_ctx, n_local = solo, n_points

    def step(flags=0):
        return run_ctx.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS, flags=flags)

    for _ in range(args.warmup):
        res = step()
    fence()
    t0 = time.perf_counter()
    dev_s = 0.0
    for _ in range(args.steps):
        res = step()
        dev_s += res.device_seconds
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert res.iterations == ITERATIONS, res.iterations

    # roofline pass: same steps, every iteration launch bracketed by HIP events on the module's stream
    kernel_ms = []
    for _ in range(max(3, min(args.steps, 20))):
        r = step(capi.FLAG_PROFILE)
        kernel_ms.append(r.kernel_ms[:ITERATIONS])
    kernel_ms = np.array(kernel_ms)
    bracketed_s = float(kernel_ms.mean()) * 1e-3
    # matched points handled by this rank per round (corr_count is summed over the communicator)
    matches = float(res.corr_count.mean()) / (world if mode == "shard" else 1)
    bytes_per_round = algorithmic_bytes(n_local, matches)
    persistent = res.launches == 1                   # single GPU: the whole align is ONE launch
    rounds_per_launch = ITERATIONS if persistent else 1
    launches = args.steps * (1 if persistent else ITERATIONS)
    span_s = dev_s / launches                        # timed region: HIP-event span per launch
    bytes_per_launch = bytes_per_round * rounds_per_launch
    achieved = bytes_per_launch / span_s / 1e9

    sharded = None
    if use_dist and shard_error is None:  # a collective: every rank takes part (compared on rank 0 below)
        sharded = ctx.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)

    out = None
    if rank == 0:
        scans_per_step = world if mode == "replicate" else 1
        value = scans_per_step * n_points * ITERATIONS * args.steps / elapsed
        out = {
            "metric": "registered points/sec per VGICP iteration (100k-pt scan vs 1M-voxel map)",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # the label of the 1/2/4/8-GPU series: at N > 1 what the timed region ran; at N = 1 what that
            # series runs at this size (replicas up to C2, where a round is shorter than any exchange)
            "scaling": ("weak" if mode == "replicate" else "strong") if use_dist
                       else ("weak" if n_points <= 100_000 else "strong"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {n_points}-pt uniform-random scan vs {n_voxels}-voxel map "
                            f"(voxel 0.3 m, occupancy 0.5), {ITERATIONS} VGICP iterations per align "
                            f"(cosine_threshold 2.0 forces all), scan resident in HBM",
                "points": n_points, "voxels": n_voxels, "iterations": ITERATIONS,
                "sharding": {"single": "single GPU",
                             "shard": f"contiguous point shards over {world} rank(s), replicated map, RCCL "
                                      f"all-reduce of 28 doubles per iteration",
                             "replicate": f"{world} independent {n_points}-point scans, one per rank, replicated "
                                          f"map, no communication: more points/s than point-sharding ONE scan "
                                          f"at this size (see sharding_autotune)"}[mode],
                "sharding_autotune": tuning,
                "matches_per_iteration": float(res.corr_count.mean()),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n_points, world,
                                            "persistent_kernel" if persistent else "iterate_kernel"),
                "kernel": "vgicp::persistent_kernel" if persistent else "vgicp::iterate_kernel",
                "rounds_per_launch": rounds_per_launch,
                "bytes_per_launch": bytes_per_launch,
                "bytes_per_round": bytes_per_round,
                "launch_us": span_s * 1e6,
                "us_per_round": span_s / rounds_per_launch * 1e6,
                "launch_us_source": "HIP event pair on the module's stream around the launch(es) of every "
                                    f"timed align, {launches} launches",
                "per_launch_variant_us_per_round": bracketed_s * 1e6,
                "per_launch_variant_source": "iterate_kernel (one launch per round, the multi-GPU path), HIP event "
                                             f"pair around each of {kernel_ms.size} launches, ~2 us event overhead each",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            base, ref = cpu_baseline(vmap, pts, covs, guess, args.cpu_budget)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_over_cpu"] = value / base["value"]
            # the timed GPU result is also the parity-checked one
            same_counts = bool((ref.corr_count == res.corr_count).all())
            dt = float(np.linalg.norm(ref.pose[:3, 3] - res.pose[:3, 3]))
            out["parity"] = {"identical_counts": same_counts, "pose_delta_m": dt}
        if use_dist and sharded is not None:
            # evidence that the sharded, all-reduced loop computes what one GPU computes
            one = solo.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0)
            out["multi_gpu_parity"] = {
                "identical_counts": bool((one.corr_count == sharded.corr_count).all()),
                "pose_delta": float(np.abs(one.pose - sharded.pose).max()),
                "against": "sharded + all-reduced loop vs the whole scan on one GPU (persistent launch)",
            }
    if use_dist:
        solo.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes last: flush whatever native libraries (RCCL's banner) still hold in
        # C stdio buffers first, so nothing can follow it on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
