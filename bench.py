#!/usr/bin/env python3
"""bench.py — registered points/s per VGICP iteration on MI355X (BASELINE.json metric).

A "step" is one whole ICP::align of config C2 — a 100k-point synthetic uniform scan against a 1M-voxel
synthetic map, 20 VGICP iterations (cosine_threshold 2.0 forces all of them) — called the
way the reference calls it: with the scan in ordinary HOST buffers (reference src/Registration.cpp:11 deep-
copies the cloud; here that copy is the upload).  The timed region therefore holds, per step, the
host-to-device copy of the 96·N-byte scan, its packing and the 20 rounds ("a pre-resident
scan is not allowed for the headline number").  The map is resident before the timed region starts (it
belongs to updateLocalMap).  `value` = N · 20 · steps / wall; `value_resident` is the same loop over a scan
that is already in HBM (vgicp_align_resident), reported beside it, never as `value`.

--gpus N (launched by torch.distributed.run, one rank per GPU): the ONE scan is split into contiguous point
shards (BASELINE config C3), every rank uploads and registers its shard against the replicated map, every
iteration ends in the exchange of the 28-double normal-equation row; all ranks return the same pose.
`value` = N_points · 20 · steps / wall of that sharded align ("scaling": "strong": the total work is fixed).
N independent scans, one per rank (no communication), are measured too and reported as
`replicas_aggregate` — a throughput figure, not the metric.  `multi_gpu_parity` compares the sharded result
with the whole scan on one GPU.

`roofline` is measured live over the timed region: every align brackets its iteration launch(es) with a HIP
event pair on the module's own stream (stats.device_seconds); `achieved` = ALGORITHMIC bytes per launch
(112 B per point-iteration + 96 B per matched point) ÷ that span.  On one GPU a launch is
the persistent kernel = all 20 rounds.  `traffic` is the PMC-measured HBM traffic per launch from the newest
profiles/*_summary.json taken at this scan size (tools/profile_gpu.sh), or null; the persistent launch keeps
scan and voxel records on chip, so its real HBM traffic is far BELOW the algorithmic bytes and `frac` is a
statement about time per algorithmic byte, not about HBM utilisation (`traffic_frac` is the latter).
`cpu_baseline` times the CPU oracle's reference-faithful mode (OpenMP) on the same inputs at several thread
counts, rank 0 at N=1 only, and reports the best.

PyTorch is plumbing here (torch.distributed rendezvous/barrier, device sync); the path itself is the C-ABI
HIP module.  The oracle is touched only by the cpu_baseline leg.
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
METRIC = "registered points/sec per VGICP iteration (100k-pt scan vs 1M-voxel map), 1/2/4/8 GPU"


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
    """The reference-faithful CPU path (oracle, OpenMP) on the same inputs, at OpenMP thread counts
    {1, 8, 32, all host cores}; bounded wall time; the best rate is the baseline."""
    from oracle import binding as oracle
    omap = oracle.OracleMap(vmap.voxel_size, 1)  # maxNumPointsPerVoxel = 1
    omap.insert(vmap.means, vmap.covs)
    n = pts.shape[0]
    all_cores = oracle.max_threads()
    counts = sorted({c for c in (1, 8, 32, all_cores) if c <= all_cores})
    per_setting = budget_s / len(counts)
    sweep, best, ref = [], None, None
    for threads in counts:
        oracle.set_threads(threads)
        t_start = time.time()
        times = []
        r = omap.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0, mode=oracle.FAITHFUL)  # warm-up
        while len(times) < 5 and (time.time() - t_start < per_setting or len(times) < 2):
            r = omap.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0, mode=oracle.FAITHFUL)
            times.append(r.seconds)
        med = float(np.median(times))
        row = {"threads": int(r.threads), "ms_per_align": med * 1e3, "points_per_s": n * r.iterations / med,
               "aligns_timed": len(times)}
        sweep.append(row)
        if best is None or row["points_per_s"] > best["points_per_s"]:
            best = row
        ref = r
    oracle.set_threads(all_cores)
    return {
        "value": best["points_per_s"],
        "unit": "points/s",
        "cores": best["threads"],
        "kind": "port",
        "sample": f"full workload: {n}-pt scan x {ref.iterations} iterations vs {len(omap)}-voxel map; oracle "
                  f"faithful mode (OpenMP, -O3, no -march), median of {best['aligns_timed']} aligns after 1 warm-up "
                  f"at each of {counts} threads; best = {best['threads']} threads ({best['ms_per_align']:.1f} ms per align)",
        "thread_sweep": sweep,
        "host_cores": all_cores,
    }, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)      # 500 x 0.37 ms: a timed region long enough to be seen
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the secondary C5 (1M points / 10M voxels) leg")
    ap.add_argument("--resident", action="store_true",
                    help="time vgicp_align_resident (scan already in HBM) as the step: profiling aid, the JSON "
                         "line then says so in config.workload")
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
    # BENCH_FORCE_COMM=1 runs the multi-GPU code path (process group, unique-id hand-off, per-iteration
    # exchange) with however many ranks there are — on one GPU a way to exercise it
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
    my_pts, my_covs = np.ascontiguousarray(pts[lo:hi]), np.ascontiguousarray(covs[lo:hi])

    ctx = capi.Context(local_rank)
    ctx.map_reset(vmap.voxel_size, n_voxels)
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    if use_dist:
        ctx.comm_init(world, rank, share_unique_id(ctx, rank))     # every rank or none: a failure ends the run

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def step_host(flags=0):
        """ICP::align as the reference calls it: scan (this rank's shard) in host buffers."""
        return ctx.align(my_pts, my_covs, guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS, flags=flags)

    def step_resident(flags=0):
        return ctx.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS, flags=flags)

    def timed(step, steps):
        import gc
        fence()
        gc.collect()
        gc.disable()                  # no collector pauses of the host language inside the timed region
        t0 = time.perf_counter()
        dev_s, res = 0.0, None
        for _ in range(steps):
            res = step()
            dev_s += res.device_seconds
        fence()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, dev_s, res

    headline = step_resident if args.resident else step_host
    if args.resident:
        ctx.scan_upload(my_pts, my_covs)
    for _ in range(args.warmup):
        res = headline()
    elapsed, dev_s, res = timed(headline, args.steps)              # ---- the timed region ----
    assert res.iterations == ITERATIONS, res.iterations

    # secondary: the same loop over the scan already resident (kernel + launch + copy-back only)
    if args.resident:
        elapsed_res = elapsed
    else:
        for _ in range(2):
            step_resident()
        elapsed_res, _, _ = timed(step_resident, args.steps)
    upload_ns, upload_bytes = ctx.counter(3), ctx.counter(2)

    # per-launch variant: every iteration launch bracketed by HIP events on the module's stream
    kernel_ms = []
    for _ in range(max(3, min(args.steps, 20))):
        r = step_resident(capi.FLAG_PROFILE)
        kernel_ms.append(r.kernel_ms[:ITERATIONS])
    kernel_ms = np.array(kernel_ms)
    bracketed_s = float(kernel_ms.mean()) * 1e-3
    n_local = hi - lo
    matches = float(res.corr_count.mean()) / world                 # corr_count is summed over the communicator
    bytes_per_round = algorithmic_bytes(n_local, matches)
    persistent = res.launches == 1                                 # single GPU: the whole align is ONE launch
    rounds_per_launch = ITERATIONS if persistent else 1
    launches = args.steps * (1 if persistent else ITERATIONS)
    span_s = dev_s / launches                                      # timed region: HIP-event span per launch
    bytes_per_launch = bytes_per_round * rounds_per_launch
    achieved = bytes_per_launch / span_s / 1e9
    fallbacks = ctx.counter(1)

    # N > 1, secondary: N independent scans, one per rank, no communication (throughput, not the metric)
    replicas = None
    solo = None
    if use_dist:
        solo = capi.Context(local_rank)
        solo.map_reset(vmap.voxel_size, n_voxels)
        solo.map_upsert(vmap.keys, vmap.means, vmap.covs)
        own_pts, own_covs = synth.make_uniform_scan(n_points, vmap, seed=synth.SCAN_SEED + rank)

        def step_replica():
            return solo.align(own_pts, own_covs, guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)
        for _ in range(3):
            step_replica()
        steps_r = max(3, args.steps // 2)
        el, _, _ = timed(step_replica, steps_r)
        replicas = {"value": world * n_points * ITERATIONS * steps_r / el, "unit": "points/s",
                    "ms_per_step": el / steps_r * 1e3,
                    "what": f"{world} independent {n_points}-point scans, one per rank, host buffers, replicated map, "
                            f"no communication — aggregate throughput, NOT the BASELINE metric"}

    out = None
    if rank == 0:
        kernel = "vgicp::persistent_kernel" if persistent else "vgicp::iterate_kernel"
        traffic = measured_traffic(n_points, world, "persistent_kernel" if persistent else "iterate_kernel")
        workload = (f"{args.config}: {n_points}-pt uniform-random scan vs {n_voxels}-voxel map (voxel 0.3 m, occupancy 0.5), "
                    f"{ITERATIONS} VGICP iterations per align (cosine_threshold 2.0 forces all); ")
        if args.resident:
            workload += "PROFILING MODE --resident: scan already in HBM (not the headline configuration)"
        elif world == 1:
            workload += ("each step = vgicp_align with the scan in host buffers: upload of the 96*N-byte scan + pack + "
                         "20 rounds in one persistent launch; map resident")
        else:
            workload += (f"ONE scan point-sharded over {world} ranks (contiguous shards, replicated map); each step = "
                         f"every rank uploads its shard from host buffers and runs the sharded loop, one exchange of the "
                         f"28-double normal-equation row per iteration")
        out = {
            "metric": METRIC,
            "value": n_points * ITERATIONS * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "value_resident": n_points * ITERATIONS * args.steps / elapsed_res,
            "ms_per_step_resident": elapsed_res / args.steps * 1e3,
            "config": {
                "workload": workload,
                "points": n_points, "voxels": n_voxels, "iterations": ITERATIONS,
                "sharding": "single GPU" if world == 1 and not use_dist else
                            f"contiguous point shards over {world} rank(s), replicated map, RCCL all-reduce of 28 "
                            f"doubles per iteration",
                "matches_per_iteration": float(res.corr_count.mean()),
                "upload": {"bytes_per_step": 96 * n_local,
                           "host_side_ms_per_upload": (upload_ns / 1e6) / max(1, upload_bytes // max(1, 96 * n_local)),
                           "what": "two hipMemcpyAsync from the caller's pageable buffers (pinned on the fly by the runtime, DMA in "
                                   "place) + pack kernel, enqueued in front of the persistent launch"},
                "persistent_fallbacks": fallbacks,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_frac": (traffic / span_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "note": "achieved/frac price ALGORITHMIC bytes (SURVEY.md 8(d)) against the launch's duration; the "
                        "persistent launch keeps scan and voxel records on chip, so its HBM traffic (`traffic`, PMC) is "
                        "far below them and traffic_frac is the real HBM utilisation: at this size the kernel is "
                        "bound by the latency of the per-round reduce/exchange/solve chain, not by bytes",
                "kernel": kernel,
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
        if replicas is not None:
            out["replicas_aggregate"] = replicas
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            base, ref = cpu_baseline(vmap, pts, covs, guess, args.cpu_budget)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / base["value"]
            # the timed GPU result is also the parity-checked one
            same_counts = bool((ref.corr_count == res.corr_count).all())
            dt = float(np.linalg.norm(ref.pose[:3, 3] - res.pose[:3, 3]))
            out["parity"] = {"identical_counts": same_counts, "pose_delta_m": dt}
        if use_dist:
            # evidence that the sharded, exchanged loop computes what one GPU computes
            one = solo.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0)
            out["multi_gpu_parity"] = {
                "identical_counts": bool((one.corr_count == res.corr_count).all()),
                "pose_delta": float(np.abs(one.pose - res.pose).max()),
                "against": "the timed sharded result vs the whole scan on one GPU (persistent launch)",
            }
    if solo is not None:
        solo.close()
        solo = None

    # Secondary leg (never `value`): BASELINE config C5 — 1M-point scan vs 10M-voxel map, the bandwidth regime —
    # on the same ranks: the scan point-sharded, the map replicated, scan resident, 20 forced rounds per align.
    # On one GPU this puts the C5 figures into the same record as the headline; on N GPUs it is the curve of the
    # regime that sharding is for.  Any failure is reported in the field, never raised.
    c5 = None
    if args.config == "C2" and not args.no_c5:
        try:
            n5, v5 = synth.CONFIGS["C5"]
            map5 = synth.make_map(v5)
            pts5, covs5 = synth.make_uniform_scan(n5, map5)
            lo5, hi5 = shard_bounds(n5, world, rank)
            ctx.map_reset(map5.voxel_size, v5)
            ctx.map_upsert(map5.keys, map5.means, map5.covs)
            ctx.scan_upload(np.ascontiguousarray(pts5[lo5:hi5]), np.ascontiguousarray(covs5[lo5:hi5]))
            del map5, pts5, covs5
            for _ in range(3):
                r5 = step_resident()
            steps5 = 10
            el5, dev5, r5 = timed(step_resident, steps5)
            per_rank_bytes = algorithmic_bytes(hi5 - lo5, float(r5.corr_count.mean()) / world) * ITERATIONS
            launches5 = r5.launches
            c5 = {"value": n5 * ITERATIONS * steps5 / el5, "unit": "points/s", "ms_per_step": el5 / steps5 * 1e3,
                  "us_per_round": dev5 / steps5 / ITERATIONS * 1e6,
                  "achieved_GBs_per_gpu": per_rank_bytes / (dev5 / steps5) / 1e9,
                  "frac_of_8TBs_per_gpu": per_rank_bytes / (dev5 / steps5) / 1e9 / HBM_PEAK_GBS,
                  "single_launch": launches5 == 1, "matches_per_iteration": float(r5.corr_count.mean()),
                  "workload": f"C5: {n5}-pt scan vs {v5}-voxel map, {ITERATIONS} rounds per align, scan resident, "
                              f"point-sharded over {world} rank(s); algorithmic bytes of a rank / its event span"}
        except Exception as e:  # noqa: BLE001 - reported in the JSON line
            c5 = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and out is not None:
            out["c5_resident"] = c5
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
