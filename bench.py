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
with the whole scan on one GPU.  `config.sharding.transport` names what carried the per-iteration exchange in the
timed region ("mailbox": the kernels' own stores into peer-mapped mailboxes; "rccl": one launch + one ncclAllReduce
per iteration) and `exchange_us_per_round` what it cost (sharded round minus the same shard registered alone).
BENCH_SHARE_DEVICE=1 is the dress rehearsal of that line on a box with ONE GPU: N ranks as N processes on device 0,
rendezvous over gloo, the mailboxes wired by hand (vgicp_peer_export / _connect), 256 / N workgroups per rank.
IN-PROCESS mode: `python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset) drives all N devices from this one
process and thread through ONE multi-device context (vgicp_create_multi: what the reference's single-threaded caller
would hold, src/main.cpp:68-70) — same sharding, same mailboxes (wired by plain peer pointers), no torch.distributed in
the data path; `config.sharding.wiring` says so.  Under the launcher, rank 0 measures that in-process context too, after
the per-rank contexts are closed, and reports it as `in_process` beside the launcher's own figure.

The upload is measured the way the reference's caller behaves: every timed step gets a FRESH cloud (host memory the
HIP runtime has never seen, src/Registration.cpp:11 deep-copies a fresh cloud per frame) and that cloud is FREED INSIDE
THE TIMED LOOP right after the align, as src/Odometry.cpp:84-87 lets it die with the frame — with free(); a second
loop beside it (`config.upload.unmapped_every_step`) munmaps every cloud instead, so that the pages go back to the
kernel at every step whatever malloc's thresholds are.  (A cloud the runtime had registered would
stall every queue of the process for ~20 ms at that free: vgicp_align therefore stages the scan with its own copy
threads and lets one kernel read the staging memory; `config.upload` reports the step-time distribution, the rate
reached against the link, and beside it one re-used buffer, a buffer the caller page-locked, and the runtime's
in-place path whose buffers must never be freed.)

`roofline` is measured live over the timed region: every align brackets its iteration launch(es) with a HIP
event pair on the module's own stream (stats.device_seconds); `achieved` = ALGORITHMIC bytes per launch
(112 B per point-iteration + 96 B per matched point) ÷ that span.  On one GPU a launch is
the persistent kernel = all 20 rounds.  `traffic` is the PMC-measured HBM traffic per launch from the newest
profiles/*_summary.json taken at this scan size (tools/profile_gpu.sh) — it is NOT measured by this run, and
`traffic_source` says which file / tag / library hash it comes from; a summary taken on another build of the kernels
is not quoted (`traffic: null` with the reason); the persistent launch keeps
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

# the exchange between GPUs maps peer memory through dmabuf IPC handles; the pool's hosts only support that mode.
# The driver's shell exports this already; a bare shell would otherwise fail in hipIpcGetMemHandle (must be set
# before anything initialises the GPU)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# BENCH_SHARE_DEVICE=1 (several ranks / sub-contexts on ONE device): every one of them needs a hardware queue of its own
# for its persistent launch — the runtime's default of 4 per process is too few from 3 sub-contexts on
if os.environ.get("BENCH_SHARE_DEVICE", "0") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from eskf_lio_amd import capi, synth  # noqa: E402

_KEEP_ALIVE = []   # host buffers the HIP runtime has REGISTERED (the in-place comparison legs only) stay mapped until the process ends
from eskf_lio_amd.distributed import gather_bytes, shard_bounds, share_unique_id  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s achievable)
ITERATIONS = 20
METRIC = "registered points/sec per VGICP iteration (100k-pt scan vs 1M-voxel map), 1/2/4/8 GPU"


def algorithmic_bytes(n_points: int, matches: float) -> float:
    """112 B per point-iteration (24 point + 72 covariance + 16 hash slot) plus
    96 B per matched point (24 voxel mean + 72 voxel covariance)."""
    return 112.0 * n_points + 96.0 * matches


def measured_traffic(n_points: int, world: int, kernel: str):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary taken at this scan size on one
    GPU (profiles/*_summary.json, tools/profile_gpu.sh + tools/summarize_profile.py) — quoted only for the library it
    was measured on.  -> (traffic or None, provenance dict): the summary records the sha256 of libvgicp_hip.so and of
    the kernel sources at profile time; a loaded library that matches neither gets `traffic: null` and the reason."""
    import glob
    from eskf_lio_amd import provenance
    if world != 1:
        return None, {"reason": "PMC profiles are taken on one GPU; no per-launch traffic for a sharded run"}
    lib_now, src_now = provenance.library_sha256(capi.LIB_PATH), provenance.kernel_source_sha256()
    best, stale = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
            t = d.get("traffic")
            points = d.get("bench", {}).get("config", {}).get("points", 100_000)
        except Exception:
            continue
        key = f"{kernel}_total_calibrated"
        if not (t and points == n_points and key in t and t[key] == t[key]):
            continue
        src = {"file": os.path.relpath(f, ROOT), "tag": d.get("tag"), "library_sha256": d.get("library_sha256"),
               "kernel_source_sha256": d.get("kernel_source_sha256")}
        if src["library_sha256"] and src["library_sha256"] == lib_now:
            src["match"] = "the loaded libvgicp_hip.so is byte for byte the profiled one"
        elif src["kernel_source_sha256"] and src["kernel_source_sha256"] == src_now:
            src["match"] = "library rebuilt since; the kernel sources and build flags are the profiled ones"
        else:
            stale = src
            continue
        best = (t[key], src)
    if best is not None:
        return best
    reason = ("no PMC summary for this scan size under profiles/" if stale is None else
              f"the newest PMC summary ({stale['file']}) was taken on another build of the kernels: not quoted")
    return None, {"reason": reason, "library_sha256": lib_now, "kernel_source_sha256": src_now}


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


def frame_chain_leg(device: int, frames: int = 30, sweep_points: int = 60_000, cpu: bool = True):
    """Secondary record (never `value`): the frame sequence of reference src/Odometry.cpp:73-87 — raw sweep in host
    memory -> extrinsic + deskew -> down-sampling + 30-NN covariances (CloudPreprocessor::process) -> ICP::align ->
    LocalMap::updateLocalMap — through the no-round-trip entry points (vgicp_scan_prepare_async, vgicp_align_resident,
    vgicp_map_insert_resident_async).  A lidar-like world is seen from a slowly moving sensor; IMU states of a sensor
    at rest drive the deskew (its arithmetic runs in full, the points stay where they are).  Pass 1: host wall per
    frame, kernel launches / copies / host synchronisations per frame.  Pass 2 (fresh context, stage events on):
    device spans per stage, and the parity of every frame's prepared scan, pose and of the final map against the CPU
    oracle chain.  The oracle's time per stage on the last frame stands beside it (`cpu`)."""
    from oracle import binding as oracle
    cap, knn, h = 20, 30, 0.3
    world = synth.make_lidar_scan(sweep_points, seed=0x46524D, extent=25.0)
    st = synth.make_imu_states(48, seed=5)
    st[:, 1:4] = 0.0
    st[:, 4:8] = [0.0, 0.0, 0.0, 1.0]                                # at rest: the deskew moves nothing
    tt = synth.make_point_times(sweep_points, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=5)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    ext_inv = synth.invert_pose(ext)
    truth = [synth.se3_to_SE3([0.05 * f, 0.02 * f, 0.0, 0.0, 0.0, 0.004 * f]) for f in range(frames + 1)]
    rng = np.random.default_rng(12)

    def sweep(f):
        # what the sensor at truth[f] sees, in the LiDAR frame (the extrinsic brings it to the IMU frame)
        Tinv = ext_inv @ synth.invert_pose(truth[f])
        pts = world + rng.normal(scale=0.005, size=world.shape)
        return np.ascontiguousarray(pts @ Tinv[:3, :3].T + Tinv[:3, 3])
    sweeps = [sweep(f) for f in range(frames + 1)]

    def run(ctx, record=None, on_arrival=False):
        """on_arrival: every sweep is handed over when it 'arrives' (vgicp_sweep_stage, right after the previous frame's
        preparation was enqueued: where a lidar callback's thread would be copying beside the device's work) and prepared
        from its ticket; else vgicp_scan_prepare_async copies it first."""
        ctx.map_reset(h, 400_000)
        ctx.scan_prepare_async(sweeps[0], tt, st, ext, h, knn)
        if record is not None:
            record(0, ctx, np.eye(4), None)
        ctx.map_insert_resident_async(np.eye(4), cap)
        ctx.map_size()
        pose, results = np.eye(4), []
        ticket = ctx.sweep_stage(sweeps[1], tt) if on_arrival else 0
        ctx.frame_stats(reset=True)
        t0 = time.perf_counter()
        for f in range(1, frames + 1):
            if on_arrival:
                ctx.scan_prepare_staged_async(ticket, st, ext, h, knn)
                if f < frames:
                    ticket = ctx.sweep_stage(sweeps[f + 1], tt)
            else:
                ctx.scan_prepare_async(sweeps[f], tt, st, ext, h, knn)
            r = ctx.align_resident(pose, 30, 1e-6, 0.9999)
            pose = r.pose
            if record is not None:
                record(f, ctx, pose, r)
            ctx.map_insert_resident_async(pose, cap)
            results.append(r)
        wall = time.perf_counter() - t0
        fs = ctx.frame_stats()
        ctx.map_size()
        return wall, fs, results, pose

    out = {"what": f"{frames} frames of a {sweep_points}-point raw sweep (host memory) -> extrinsic + deskew (48 IMU states) "
                   f"-> down-sampling + {knn}-NN covariances -> ICP::align (thresholds of hilti_config.yaml) -> map insertion "
                   f"(cap {cap}); reference src/Odometry.cpp:73-87; vgicp_scan_prepare_async / vgicp_align_resident / "
                   f"vgicp_map_insert_resident_async"}
    with capi.Context(device) as ctx:
        run(ctx)                                                       # warm-up (allocations, code objects)
        wall, fs, results, _ = run(ctx)
        run(ctx, on_arrival=True)
        wall_arr, fs_arr, results_arr, _ = run(ctx, on_arrival=True)
        out.update({
            "ms_per_frame": wall / frames * 1e3,
            "ms_per_frame_sweeps_staged_on_arrival": wall_arr / frames * 1e3,
            "staged_on_arrival": {
                "what": "the same frames with every sweep handed over when it arrives (vgicp_sweep_stage, called where a lidar "
                        "callback's thread would run: beside the device's work on the frame before) and prepared from its "
                        "ticket (vgicp_scan_prepare_staged_async): the frame's first stage starts from page-locked bytes",
                "poses_bit_equal": bool(all(np.array_equal(a.pose, b.pose) for a, b in zip(results, results_arr))),
                "copies_per_frame": fs_arr.copies / frames, "kernel_launches_per_frame": fs_arr.kernel_launches / frames},
            "kernel_launches_per_frame": fs.kernel_launches / frames,
            "kernel_launches_note": "counted by the library: prologue, the preparation's hand-written sort (one tile sort + one "
                                    "merge launch per factor of four in run length: five at this size), two scans, split, "
                                    "search, covariances, the align's one launch, two for the insertion "
                                    "(profiles/*_frame_kernel_stats.csv shows the same kernels)",
            "copies_per_frame": fs.copies / frames,
            "host_syncs_per_frame": fs.host_syncs / frames,
            "align_rounds_per_frame": float(np.mean([r.iterations for r in results])),
        })
    # the SAME frames through the C++ drop-in classes the way src/Odometry.cpp:73-87 writes them (libvgicp_host.so =
    # include/eskf_lio_shim/ compiled): cloudPreprocessor.process(states, meas) -> icp.align(*meas.cloud, localMap,
    # guess) -> localMap.updateLocalMap(meas.cloud, T).  "deferred": grid on the device, the prepared scan handed from
    # class to class on the device (the chain above behind the reference's interface); "eager": the classes' defaults
    # (host-authoritative map, the host cloud holds the prepared scan after process()).
    try:
        from eskf_lio_amd import host
        chain_poses = [r.pose for r in results]

        def dropin(host_copy, device_resident, on_arrival=False, keep_raw_points=True, sensor_rate=False):
            # sensor_rate: between two frames (outside the timed part) the map's shadow-grid worker is allowed to catch
            # up, as it does at a LiDAR's 10 Hz: the clouds it has finished with leave their storage to the next frame's
            # process(), which a back-to-back loop (frames every 0.8 ms against 3 ms of shadow work each) never sees
            pre = host.CloudPreprocessor(h, ext, host_copy)
            icp = host.ICP(30, 1e-6, 0.9999)
            cfg = dict(translation_sq_threshold=-1.0, cosine_threshold=2.0, remove_distant_points=False,
                       distance_threshold=1e9, removing_period=1e9, device_resident=device_resident,
                       keep_raw_points=keep_raw_points)
            lmap = host.LocalMap(h, cap, cfg)
            fr = host.Frame(sweeps[0], tt, st)
            fr.run(pre, icp, lmap, np.eye(4), first_frame=True)
            fr.end()
            pose, poses, resident, wall = np.eye(4), [], 0, 0.0
            nxt = host.Frame(sweeps[1], tt, st)
            if on_arrival:
                nxt.stage(pre)
            for f in range(1, frames + 1):
                fr = nxt                                           # the measurement object: built outside the timed part
                nxt = host.Frame(sweeps[f + 1], tt, st) if f < frames else None
                t0 = time.perf_counter()
                fr.run(pre, icp, lmap, pose, stage_next=nxt if on_arrival else None, move_cloud=True)   # std::move, as src/Odometry.cpp:86
                wall += time.perf_counter() - t0
                got = fr.end()
                pose = got["pose"]
                poses.append(pose)
                resident += int(got["used_resident"])
                if sensor_rate:
                    lmap.drain()                                   # waits for the shadow grid's worker (untimed)
            voxels = len(lmap)
            return wall / frames * 1e3, poses, resident, voxels
        dropin("deferred", True)                                       # warm-up
        ms_def, poses_def, res_def, vox_def = dropin("deferred", True)
        dropin("eager", True)                                          # warm-up
        ms_eag, poses_eag, res_eag, _ = dropin("eager", True)          # THE CLASSES' DEFAULTS since round 5
        dropin("eager", True, sensor_rate=True)
        ms_eag_rate, _, _, _ = dropin("eager", True, sensor_rate=True)
        dropin("eager", False)                                         # warm-up (first-use allocations: page-locked arena, table growth)
        ms_host, poses_host, res_host, _ = dropin("eager", False)      # host-authoritative map (the defaults of rounds 1-4)
        dropin("deferred", True, on_arrival=True)
        ms_arr, poses_arr, res_arr, _ = dropin("deferred", True, on_arrival=True)
        out["dropin_ms_per_frame"] = ms_def
        out["dropin_ms_per_frame_sweeps_staged_on_arrival"] = ms_arr
        out["dropin_eager_ms_per_frame"] = ms_eag
        out["dropin_eager_at_sensor_rate_ms_per_frame"] = ms_eag_rate
        out["dropin_host_authoritative_ms_per_frame"] = ms_host
        out["dropin"] = {
            "what": "the frames above through ESKF_LIO::CloudPreprocessor::process / ICP::align / LocalMap::updateLocalMap "
                    "(C++ shim, called as src/Odometry.cpp:73-87 does); dropin_ms_per_frame: LocalMapConfig::deviceResident + "
                    "CloudPreprocessorConfig::HostCopy::Deferred (the scan never returns to the host); dropin_eager: the "
                    "classes' DEFAULTS since round 5 (host copy of the prepared scan as the reference leaves it; the grid the "
                    "registration reads on the device; a host-side shadow grid, kept by a worker thread, holds every raw point "
                    "for save()) in a back-to-back loop; dropin_eager_at_sensor_rate: the same three calls per frame with the shadow "
                    "grid's worker allowed to catch up BETWEEN frames (untimed), as at a LiDAR's 10 Hz, so that the storage of the "
                    "clouds it has finished with is there for the next process() — about the same figure: process() is bound by "
                    "the device (the whole preparation, then the scan over the link) either way; "
                    "dropin_host_authoritative: the defaults of rounds 1-4 (host map authoritative, device mirror fed batches)",
            "ratio_to_the_abi_chain": ms_def / out["ms_per_frame"],
            "aligns_that_found_the_scan_resident": f"{res_def} of {frames} (deferred), {res_eag} of {frames} (defaults), {res_host} of {frames} (host-authoritative)",
            "host_authoritative_pose_delta_max": float(max(np.abs(a - b).max() for a, b in zip(poses_host, chain_poses))),
            "poses_bit_equal_to_the_abi_chain": bool(all(np.array_equal(a, b) for a, b in zip(poses_def, chain_poses))),
            "eager_pose_delta_max": float(max(np.abs(a - b).max() for a, b in zip(poses_eag, chain_poses))),
            "map_voxels": vox_def,
            "staged_on_arrival": {"what": "CloudPreprocessor::stage(meas) called where the lidar callback would (here: right after the "
                                          "previous frame's process()); process() then starts from the staged sweep",
                                  "poses_bit_equal": bool(all(np.array_equal(a, b) for a, b in zip(poses_arr, chain_poses))),
                                  "aligns_that_found_the_scan_resident": f"{res_arr} of {frames}"},
        }
    except Exception as e:  # noqa: BLE001 - a secondary record
        out["dropin"] = {"error": f"{type(e).__name__}: {e}"}
    # pass 2: stage events + parity, on a fresh context
    stage = {"prepare_head": [], "prepare": [], "align": [], "insert": []}
    kept_points, mismatches, pose_delta = [], [], 0.0
    omap = oracle.OracleMap(h, cap)
    cpu_times = {}

    def record(f, ctx, pose, r):
        nonlocal pose_delta
        gp, gc = ctx.scan_download()
        moved, _ = oracle.transform(sweeps[f], np.tile(np.eye(3).reshape(9), (sweep_points, 1)), ext)
        last = f == frames
        t0 = time.perf_counter()
        desk, _ = oracle.deskew(moved, tt, st)
        t1 = time.perf_counter()
        if last or f <= 1:                                             # the brute-force oracle search is seconds per frame
            rp, rc, _ = oracle.preprocess(desk, h, knn)
            t2 = time.perf_counter()
            if not (np.array_equal(gp, rp) and np.array_equal(gc, rc)):
                mismatches.append(f)
            if last:
                cpu_times["deskew_ms"] = (t1 - t0) * 1e3
                cpu_times["downsample_knn_cov_ms"] = (t2 - t1) * 1e3
        if r is not None:
            ref = omap.align(gp, gc, prev_pose[0], 30, 1e-6, 0.9999)
            if last:
                t0 = time.perf_counter()
                omap.align(gp, gc, prev_pose[0], 30, 1e-6, 0.9999, mode=oracle.FAITHFUL)
                cpu_times["align_ms"] = (time.perf_counter() - t0) * 1e3
            if ref.iterations != r.iterations or not np.array_equal(ref.corr_count, r.corr_count):
                mismatches.append(-f)
            pose_delta = max(pose_delta, float(np.abs(ref.pose - r.pose).max()))
            fsn = ctx.frame_stats()
            stage["prepare_head"].append(fsn.prepare_head_us)
            stage["prepare"].append(fsn.prepare_us)
            stage["align"].append(fsn.align_us)
        t0 = time.perf_counter()
        wp, wc = oracle.transform(gp, gc, pose)
        omap.insert(wp, wc)
        if last:
            cpu_times["insert_ms"] = (time.perf_counter() - t0) * 1e3
        kept_points.append(len(gp))
        prev_pose[0] = pose
    prev_pose = [np.eye(4)]
    with capi.Context(device) as ctx:
        ctx.set_option(capi.OPTION_STAGE_EVENTS, 1)
        _, _, results2, _ = run(ctx, record)
        fsn = ctx.frame_stats()
        stage["insert"].append(fsn.insert_us)
        gk, gm, gcv, gn = ctx.map_export()
    k, m, c, cnt = omap.export()
    o = np.lexsort(k.T)
    map_equal = bool(np.array_equal(gk, k[o]) and np.array_equal(gn, cnt[o]) and
                     np.abs(gm - m[o]).max() < 1e-9 and np.abs(gcv - c[o]).max() < 1e-9)
    head = float(np.mean(stage["prepare_head"]))
    out.update({
        "kept_points_per_frame": float(np.mean(kept_points)),
        "stage_us": {"upload_extrinsic_deskew": head, "downsample_knn_cov": float(np.mean(stage["prepare"])) - head,
                     "align": float(np.mean(stage["align"])), "map_insert": float(np.mean(stage["insert"])),
                     "source": "HIP events on the module's stream (VGICP_OPTION_STAGE_EVENTS), mean over the frames of a "
                               "second run; map_insert: the last frame's"},
        "parity": {"prepared_scans_bit_equal_to_oracle": not [f for f in mismatches if f >= 0],
                   "align_rounds_and_counts_equal_to_oracle": not [f for f in mismatches if f < 0],
                   "pose_delta_max": pose_delta, "final_map_equal": map_equal,
                   "frames_checked": "prepared scan: frames 0, 1 and the last; align + map: every frame"},
    })
    # what bounds a frame from below: the launches it still makes (each costs the host ~5 us to enqueue and the device
    # ~2-4 us to start: MI355X_MICROARCH.md), and the bytes it has to move at the rates they can move at
    kept_mean = float(np.mean(kept_points))
    launches = out["kernel_launches_per_frame"]
    raw_bytes = 32.0 * sweep_points
    out["floor"] = {
        "launches_per_frame": launches,
        "launch_floor_us": launches * 5.0,
        "launch_floor_what": "kernel launches per frame x ~5 us of host enqueue each (measured: tools/frame_gaps.sh, "
                             "profiles/NOTES_dropped_experiments.md); the device-side start of a dependent launch is 2-4 us",
        "pcie_bytes_per_frame": raw_bytes,
        "pcie_floor_us": raw_bytes / 54e9 * 1e6,
        "hbm_bytes_per_frame_order_of": 10e6,
        "hbm_floor_us": 10e6 / 8e12 * 1e6,
        "what": "a frame moves ~2 MB over PCIe (the raw sweep, 32 B per point) and of the order of 10 MB through HBM: "
                "bytes are not what bounds it; the launches' fixed costs and the neighbour search's instruction count are",
        "achieved_us": out["ms_per_frame"] * 1e3,
    }
    if cpu:
        cpu_times["what"] = ("the oracle's stages on the last frame, OpenMP on all host cores: deskew (serial, as the "
                             "reference), down-sampling + BRUTE-FORCE 30-NN + covariances (the reference uses a KD-tree; "
                             "this is the checker's search, not a baseline for it), faithful-mode align, serial insertion")
        out["cpu_oracle_last_frame"] = cpu_times
    return out


def in_process_leg(device_ids, vmap, pts, covs, guess, steps):
    """Secondary record under the launcher: ONE multi-device context (vgicp_create_multi) over `device_ids`, driven by
    this one thread — the whole scan in host buffers in, the pose out, sharded and exchanged inside the library."""
    n = pts.shape[0]
    with capi.Context(list(device_ids)) as ctx:
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for _ in range(5):
            res = ctx.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0)
        if ctx.counter(1) > 0:
            steps = min(steps, 10)   # the single launch gives up on this node (every give-up waits ~1 s): say so, quickly
        bufs = [(pts.copy(), covs.copy()) for _ in range(min(steps, 50))]   # buffers the runtime has not uploaded from
        _KEEP_ALIVE.append(bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev = 0.0
        for k in range(steps):
            p, c = bufs[k % len(bufs)]
            res = ctx.align(p, c, guess, ITERATIONS, 1e-6, 2.0)
            dev += res.device_seconds
        wall = time.perf_counter() - t0
        ctx.scan_upload(pts, covs)
        t0 = time.perf_counter()
        for _ in range(steps):
            rr = ctx.align_resident(guess, ITERATIONS, 1e-6, 2.0)
        wall_res = time.perf_counter() - t0
        fallbacks, attempts = ctx.counter(1), ctx.counter(0)
    with capi.Context(int(device_ids[0])) as one:
        one.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        one.map_upsert(vmap.keys, vmap.means, vmap.covs)
        ref = one.align(pts, covs, guess, ITERATIONS, 1e-6, 2.0)
    return {
        "what": f"ONE process, ONE caller thread, one multi-device context (vgicp_create_multi) over devices {list(device_ids)}: "
                "vgicp_align of the whole scan from host buffers (sharded upload + single persistent launch per device, rows "
                "exchanged through peer-pointer mailboxes); the reference's caller is single-threaded (src/main.cpp:68-70)",
        "value": n * ITERATIONS * steps / wall, "unit": "points/s", "ms_per_step": wall / steps * 1e3,
        "value_resident": n * ITERATIONS * steps / wall_res, "ms_per_step_resident": wall_res / steps * 1e3,
        "us_per_round": dev / steps / ITERATIONS * 1e6, "steps": steps,
        "single_launch_per_device": res.launches == 1 and rr.launches == 1, "world_size": res.world_size,
        "single_launch_attempts": attempts, "fallbacks_to_host_summed_loop": fallbacks,
        "parity": {"identical_counts": bool((ref.corr_count == res.corr_count).all()),
                   "pose_delta": float(np.abs(ref.pose - res.pose).max()),
                   "against": "the whole scan on one single-device context"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)      # 500 x 0.37 ms: a timed region long enough to be seen
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the secondary C5 (1M points / 10M voxels) leg")
    ap.add_argument("--no-frame-chain", action="store_true", help="skip the secondary frame-chain record")
    ap.add_argument("--frames", type=int, default=30, help="frames of the frame-chain record")
    ap.add_argument("--in-process-leg", default=None, metavar="IDS",
                    help="internal: measure ONE multi-device context over these comma-separated device ordinals and print "
                         "its record as one JSON line (what rank 0 runs in a child process under the launcher)")
    ap.add_argument("--resident", action="store_true",
                    help="time vgicp_align_resident (scan already in HBM) as the step: profiling aid, the JSON "
                         "line then says so in config.workload")
    args = ap.parse_args()

    if args.in_process_leg is not None:
        ids = [int(x) for x in args.in_process_leg.split(",")]
        n_points, n_voxels = synth.CONFIGS[args.config]
        vmap = synth.make_map(n_voxels)
        pts, covs = synth.make_uniform_scan(n_points, vmap)
        print(json.dumps(in_process_leg(ids, vmap, pts, covs, synth.default_guess(), args.steps)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # no launcher but several GPUs asked for: ONE process, ONE thread, one multi-device context (vgicp_create_multi)
    inproc = world == 1 and args.gpus > 1
    if world != args.gpus and not inproc:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    # BENCH_FORCE_COMM=1 runs the multi-GPU code path (process group, unique-id hand-off, per-iteration
    # exchange) with however many ranks there are — on one GPU a way to exercise it
    force_comm = os.environ.get("BENCH_FORCE_COMM", "0") == "1"
    # BENCH_SHARE_DEVICE=1: every rank is a process on device 0 (a box with one GPU): gloo for the host side, the
    # mailboxes of the device-initiated exchange wired by hand, the CUs divided between the ranks
    share_device = os.environ.get("BENCH_SHARE_DEVICE", "0") == "1" and (world > 1 or inproc)
    use_dist = (world > 1 or force_comm) and not inproc
    eff_world = args.gpus if inproc else world                     # devices (or sub-contexts) that share the ONE scan
    if share_device:
        local_rank = 0
        if not inproc:
            os.environ.setdefault("VGICP_PERSIST_GRID", str(max(1, 256 // world)))
    if inproc and not share_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} device(s) visible "
                         "(BENCH_SHARE_DEVICE=1 runs the sub-contexts on one device)")
    torch.cuda.set_device(local_rank)
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if share_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if os.environ.get("VGICP_LIB_PATH") and rank == 0:
        print(f"[bench] VGICP_LIB_PATH override active: measuring {capi.LIB_PATH}", file=sys.stderr)

    n_points, n_voxels = synth.CONFIGS[args.config]
    vmap = synth.make_map(n_voxels)
    pts, covs = synth.make_uniform_scan(n_points, vmap)
    guess = synth.default_guess()
    lo, hi = shard_bounds(n_points, world, rank)
    my_pts, my_covs = np.ascontiguousarray(pts[lo:hi]), np.ascontiguousarray(covs[lo:hi])
    n_local = hi - lo
    if inproc:
        n_local = -(-n_points // eff_world)                        # the largest shard: what a device's launch works on

    device_ids = ([0] * args.gpus if share_device else list(range(args.gpus))) if inproc else None
    ctx = capi.Context(device_ids if inproc else local_rank)
    ctx.map_reset(vmap.voxel_size, n_voxels)
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    if share_device and not inproc:
        ctx.peer_connect(world, rank, gather_bytes(ctx.peer_export(), world))
        dist.barrier()                                             # every mailbox initialised before any kernel writes
    elif use_dist:
        ctx.comm_init(world, rank, share_unique_id(ctx, rank))     # every rank or none: a failure ends the run

    def host_max(x: float) -> float:
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if share_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # a communicator of ONE rank has nothing to exchange and would take the single launch: BENCH_FORCE_COMM at world 1
    # asks for the launch-per-round loop with its (one-rank) ncclAllReduce, which is what it is there to exercise
    base_flags = capi.FLAG_NO_PERSISTENT if (force_comm and world == 1) else 0

    def align_host(p, c, flags=0):
        """ICP::align as the reference calls it: scan (this rank's shard) in host buffers."""
        return ctx.align(p, c, guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS, flags=flags | base_flags)

    def step_resident(flags=0):
        return ctx.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS, flags=flags | base_flags)

    def timed(step, steps):
        """steps calls of step(k) between two fences; -> (max-over-ranks wall, summed device spans, last result,
        per-step host seconds of this rank)."""
        import gc
        per_step = np.zeros(steps)
        fence()
        gc.collect()
        gc.disable()                  # no collector pauses of the host language inside the timed region
        t0 = time.perf_counter()
        dev_s, res, t_prev = 0.0, None, t0
        for k in range(steps):
            res = step(k)
            dev_s += res.device_seconds
            t_now = time.perf_counter()
            per_step[k] = t_now - t_prev
            t_prev = t_now
        fence()
        elapsed = time.perf_counter() - t0
        gc.enable()
        return host_max(elapsed), dev_s, res, per_step

    # ---- the upload, as the reference's caller lives it: a fresh cloud per step, freed inside the timed loop ----
    # (src/Registration.cpp:11 deep-copies a FRESH cloud per frame; src/Odometry.cpp:84-87 lets it die with the frame).
    # As many distinct clouds as steps (500 steps x 9.6 MB = 4.8 GB of host memory at C2), each in an
    # anonymous mapping of its own that is unmapped right after its align: the pages go back to the kernel THEN,
    # whatever malloc's thresholds are.
    import mmap
    scan_bytes = 96 * n_local

    class FreshCloud:
        """One scan in host memory of its own, written once and never read since.  how = "free": numpy arrays from
        malloc, released with free() (what `delete` of the reference's cloud does).
        how = "munmap": an anonymous mapping per array, unmapped by release() — the pages go back to the kernel at
        that very moment, whatever malloc's thresholds are."""
        def __init__(self, how):
            self.how = how
            if how == "munmap":
                self.maps = [mmap.mmap(-1, max(1, a.nbytes), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS) for a in (my_pts, my_covs)]
                self.arrays = [np.frombuffer(m, dtype=np.float64, count=a.size).reshape(a.shape)
                               for m, a in zip(self.maps, (my_pts, my_covs))]
                for dst, src in zip(self.arrays, (my_pts, my_covs)):
                    dst[...] = src
            else:
                self.maps = []
                self.arrays = [my_pts.copy(), my_covs.copy()]

        def release(self):
            self.arrays = None
            for m in self.maps:
                m.close()
            self.maps = []

    if not args.resident and args.steps * scan_bytes > 32e9:
        raise SystemExit(f"--steps {args.steps}: {args.steps * scan_bytes / 1e9:.0f} GB of fresh clouds; use at most {int(32e9 // scan_bytes)} steps")

    # free() hands a cloud back to the ALLOCATOR.  The clouds of all steps exist before the loop starts (their production is
    # not part of the path), next to each other on the heap: left alone, glibc merges them as they are freed and gives
    # the whole block (4.8 GB at 500 steps) back to the kernel inside the free() of the last one -- a quarter of a second
    # that no caller with one cloud alive at a time ever sees.  So glibc is told to keep what is freed
    # (M_TRIM_THRESHOLD); the loop `unmapped_every_step` beside it is the one in which every cloud's pages DO go back
    # to the kernel, at every step.
    try:
        import ctypes
        _libc = ctypes.CDLL("libc.so.6")
        mallopt_ok = bool(_libc.mallopt(-1, -1)) and bool(_libc.mallopt(-3, 32 << 20))  # M_TRIM_THRESHOLD = never, M_MMAP_THRESHOLD = 32 MB
    except Exception:  # noqa: BLE001
        mallopt_ok = False

    def fresh_loop(how, steps, keep=False):
        """steps aligns, each from its own fresh cloud, which is freed right after its align INSIDE the timed loop (unless
        keep) -> timed(...) + (seconds inside vgicp_align per step, seconds inside the free per step)"""
        clouds = [FreshCloud(how) for _ in range(steps)]
        align_s, free_s, up_s, dev_each = np.zeros(steps), np.zeros(steps), np.zeros(steps), np.zeros(steps)

        def step(k):
            cl = clouds[k]
            ns_a = ctx.counter(3)
            t_a = time.perf_counter()
            r = align_host(cl.arrays[0], cl.arrays[1])
            t_b = time.perf_counter()
            if not keep:
                cl.release()
                free_s[k] = time.perf_counter() - t_b
            align_s[k] = t_b - t_a
            up_s[k] = (ctx.counter(3) - ns_a) * 1e-9
            dev_each[k] = r.device_seconds
            return r
        out = timed(step, steps)
        if keep:
            _KEEP_ALIVE.append(clouds)
        worst = [{"step": int(k), "align_ms": float(align_s[k] * 1e3), "of_which_upload_host_ms": float(up_s[k] * 1e3),
                  "persistent_launch_ms": float(dev_each[k] * 1e3), "free_ms": float(free_s[k] * 1e3)}
                 for k in np.argsort(align_s)[::-1][:5]]
        return out + (align_s, free_s, worst)

    def loop_report(el, per_step, align_s, free_s, worst, steps):
        return {
            "slowest_aligns": worst,
            "steps": int(steps),
            "ms_per_step": el / steps * 1e3,
            "step_ms_p50": float(np.percentile(per_step, 50) * 1e3),
            "step_ms_p99": float(np.percentile(per_step, 99) * 1e3),
            "step_ms_max": float(per_step.max() * 1e3),
            "steps_above_1ms": int((per_step > 1e-3).sum()),
            "align_ms_mean": float(align_s.mean() * 1e3),
            "align_ms_p99": float(np.percentile(align_s, 99) * 1e3),
            "align_ms_max": float(align_s.max() * 1e3),
            "aligns_above_1ms": int((align_s > 1e-3).sum()),
            "free_ms_mean": float(free_s.mean() * 1e3),
            "free_ms_max": float(free_s.max() * 1e3),
        }

    if args.resident:
        ctx.scan_upload(my_pts, my_covs)
        for _ in range(args.warmup):
            res = step_resident()
        elapsed, dev_s, res, _ = timed(lambda k: step_resident(), args.steps)
        upload_report = None
        elapsed_res = elapsed
    else:
        for _ in range(args.warmup):                               # warm-up on a buffer of its own
            res = align_host(my_pts, my_covs)
        ns0, slow0 = ctx.counter(3), ctx.counter(capi.COUNTER_UPLOAD_SLOW)
        elapsed, dev_s, res, per_step, align_s, free_s, worst = fresh_loop("free", args.steps)      # ---- the timed region ----
        ns_cold = ctx.counter(3) - ns0
        upload_slow = ctx.counter(capi.COUNTER_UPLOAD_SLOW) - slow0
        side = min(args.steps, 100)
        # the same, every cloud in a mapping of its own that is UNMAPPED right after its align: the pages go back to the
        # kernel at every step (the operating system charges ~0.8 ms per 9.6 MB for that, tools/probe_munmap.py) -- the
        # loop in which a registered buffer would stall every queue of the process for ~20 ms per step
        el_unmap, _, _, per_unmap, align_unmap, free_unmap, worst_unmap = fresh_loop("munmap", side)
        # one buffer over and over (what rounds 1-2 timed)
        ns0 = ctx.counter(3)
        el_reused, _, _, _ = timed(lambda k: align_host(my_pts, my_covs), side)
        ns_reused = ctx.counter(3) - ns0
        # ... the same buffer page-locked once by the caller (vgicp_host_register: a pool of clouds): read in place
        ctx.host_register(my_pts)
        ctx.host_register(my_covs)
        for _ in range(3):
            align_host(my_pts, my_covs)
        el_reg, _, _, _ = timed(lambda k: align_host(my_pts, my_covs), side)
        ctx.host_unregister(my_pts)
        ctx.host_unregister(my_covs)
        # ... and the runtime's in-place path (VGICP_OPTION_UPLOAD_STAGE_KB = 0: the runtime registers the caller's pages):
        # last of the upload legs, its clouds are never freed (freeing one would stall every queue for ~20 ms)
        ctx.set_option(capi.OPTION_UPLOAD_STAGE_KB, 0)
        el_inplace, _, _, _, _, _, _ = fresh_loop("free", side, keep=True)
        ctx.set_option(capi.OPTION_UPLOAD_STAGE_KB, 512 << 10)
        upload_ms = ns_cold / 1e6 / args.steps
        upload_report = {
            "bytes_per_step": scan_bytes,
            "clouds": f"{args.steps} distinct fresh clouds, one per timed step ({args.steps * scan_bytes / 1e6:.0f} MB in all): numpy "
                      "arrays from malloc, written once, handed to vgicp_align, then FREED (free(), as `delete` of the "
                      "reference's cloud does) inside the timed loop",
            "how": f"staged: {os.environ.get('VGICP_UPLOAD_THREADS', '3')} host thread(s) (the caller's included, no HIP calls) copy "
                   "the scan into page-locked memory of the context in units of 2048 points — a unit whose covariances "
                   "are all bitwise symmetric as six doubles per covariance (72 instead of 96 bytes per point cross the "
                   "link; mirrored on the device, the resident scan is the caller's bit for bit); ONE kernel launch reads "
                   "the units over PCIe as they are published and packs them (pack_arena_kernel); the runtime never "
                   "registers the caller's pages",
            **loop_report(elapsed, per_step, align_s, free_s, worst, args.steps),
            "allocator": "glibc keeps freed memory (mallopt: M_TRIM_THRESHOLD = never, M_MMAP_THRESHOLD = 32 MB)" if mallopt_ok else "glibc defaults (mallopt failed)",
            "upload_ms": upload_ms,
            "upload_GBps": scan_bytes / upload_ms / 1e6 if upload_ms > 0 else None,
            "bytes_over_the_link_per_step": 72 * n_local,
            "bytes_over_the_link": "72 per point (24 + six doubles of a bitwise symmetric covariance; VGICP_UPLOAD_COMPACT=0 sends 96): "
                                   "what upload_ms and the two fractions below are priced on since round 6 — upload_ms is the HOST side "
                                   "(the copy threads), the link's own share is ms_per_step - persistent launch - host tail",
            "fraction_of_kernel_read_rate_54GBps": 72 * n_local / upload_ms / 1e6 / 54.0 if upload_ms > 0 else None,
            "fraction_of_pcie5_x16_63GBps": 72 * n_local / upload_ms / 1e6 / 63.0 if upload_ms > 0 else None,
            "link_rate_source": "54 GB/s: a kernel reading page-locked host memory on this node type, "
                                "tools/micro/stage_crew_probe.hip (profiles/r12_stage_crew_probe.txt); 63 GB/s: PCIe 5.0 x16 payload",
            "uploads_repeated_because_the_copy_threads_were_held_up": int(upload_slow),
            "unmapped_every_step": {
                **loop_report(el_unmap, per_unmap, align_unmap, free_unmap, worst_unmap, side),
                "what": "the same loop with every cloud in an anonymous mapping of its own, munmap'ed right after its "
                        "align inside the timed loop: the pages return to the KERNEL at every step.  free_ms is what the "
                        "operating system charges for that (tools/probe_munmap.py: the same without any GPU work); "
                        "align_ms_max / aligns_above_1ms show that no queue stalls (a buffer the runtime had registered "
                        "costs ~20 ms at this point, profiles/r10_sync_stall.txt)"},
            "ms_per_step_reused": el_reused / side * 1e3,
            "upload_ms_reused": ns_reused / 1e6 / side,
            "ms_per_step_registered": el_reg / side * 1e3,
            "ms_per_step_in_place": el_inplace / side * 1e3,
            "what": "upload_ms: host side of the staged upload (the copy threads' time; the pack launch runs beside "
                    "them and ends a few microseconds after the last unit) in front of the persistent launch; `value` is "
                    "ms_per_step.  ms_per_step_in_place: the runtime's pin-on-the-fly path (VGICP_OPTION_UPLOAD_STAGE_KB "
                    "= 0) from fresh clouds that are never freed — freeing one takes every queue of the "
                    "process off the device for ~20 ms (profiles/r10_sync_stall.txt), which is why it is not the default",
        }
        for _ in range(2):
            step_resident()
        elapsed_res, _, _, _ = timed(lambda k: step_resident(), args.steps)
    assert res.iterations == ITERATIONS, res.iterations

    # per-launch variant: every iteration launch bracketed by HIP events on the module's stream
    # (not with BENCH_SHARE_DEVICE: that rehearsal has no RCCL communicator for the launch-per-round loop to use)
    kernel_ms = []
    for _ in range(0 if (share_device and not inproc) else max(3, min(args.steps, 20))):
        r = step_resident(capi.FLAG_PROFILE)
        kernel_ms.append(r.kernel_ms[:ITERATIONS])
    kernel_ms = np.array(kernel_ms)
    bracketed_s = float(kernel_ms.mean()) * 1e-3 if kernel_ms.size else None
    matches = float(res.corr_count.mean()) / eff_world             # corr_count is summed over the communicator
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
    exchange = None
    if use_dist:
        solo = capi.Context(local_rank)
        solo.map_reset(vmap.voxel_size, n_voxels)
        solo.map_upsert(vmap.keys, vmap.means, vmap.covs)
        own_pts, own_covs = synth.make_uniform_scan(n_points, vmap, seed=synth.SCAN_SEED + rank)

        def step_replica(k):
            return solo.align(own_pts, own_covs, guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)
        for _ in range(3):
            step_replica(0)
        steps_r = max(3, args.steps // 2)
        el, _, _, _ = timed(step_replica, steps_r)
        replicas = {"value": world * n_points * ITERATIONS * steps_r / el, "unit": "points/s",
                    "ms_per_step": el / steps_r * 1e3,
                    "what": f"{world} independent {n_points}-point scans, one per rank, host buffers, replicated map, "
                            f"no communication — aggregate throughput, NOT the BASELINE metric"}
        # what the exchange between the ranks costs per round: the sharded round (scan resident) minus the SAME shard
        # registered alone on the same grid (no exchange between ranks); all ranks measure at the same time
        steps_x = max(3, min(args.steps, 50))
        for _ in range(2):
            step_resident()
        _, dev_sh, r_sh, _ = timed(lambda k: step_resident(), steps_x)
        solo.scan_upload(my_pts, my_covs)

        def step_alone(k):
            return solo.align_resident(guess, ITERATIONS, 1e-6, 2.0, chunk_iterations=ITERATIONS)
        for _ in range(2):
            step_alone(0)
        _, dev_al, r_al, _ = timed(step_alone, steps_x)
        us_sharded = host_max(dev_sh / steps_x / ITERATIONS * 1e6)
        us_alone = host_max(dev_al / steps_x / ITERATIONS * 1e6)
        exchange = {"exchange_us_per_round": us_sharded - us_alone, "sharded_us_per_round": us_sharded,
                    "shard_alone_us_per_round": us_alone, "aligns": steps_x,
                    "what": "device span per round of the sharded align (scan resident) minus the same shard registered "
                            "alone on the same workgroups, no exchange between ranks; slowest rank of each"}

    # which transport carried the exchange of the TIMED aligns: one launch with several ranks = the mailboxes
    transport = "mailbox" if (persistent and res.world_size > 1) else ("host-sum" if inproc else "rccl")
    out = None
    if rank == 0:
        kernel = "vgicp::persistent_kernel" if persistent else "vgicp::iterate_kernel"
        traffic, traffic_source = measured_traffic(n_points, eff_world, "persistent_kernel" if persistent else "iterate_kernel")
        workload = (f"{args.config}: {n_points}-pt uniform-random scan vs {n_voxels}-voxel map (voxel 0.3 m, occupancy 0.5), "
                    f"{ITERATIONS} VGICP iterations per align (cosine_threshold 2.0 forces all); ")
        if args.resident:
            workload += "PROFILING MODE --resident: scan already in HBM (not the headline configuration)"
        elif world == 1 and not inproc:
            workload += ("each step = vgicp_align with the scan in host buffers: upload of the 96*N-byte scan + pack + "
                         "20 rounds in one persistent launch; map resident; every step's cloud is fresh host memory and is "
                         "freed inside the timed loop right after its align")
        else:
            workload += (f"ONE scan point-sharded over {eff_world} ranks (contiguous shards, replicated map); each step = "
                         f"every rank uploads its shard from host buffers and runs the sharded loop, one exchange of the "
                         f"28-double normal-equation row per iteration")
        out = {
            "metric": METRIC,
            "value": n_points * ITERATIONS * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": eff_world,
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
                "sharding": "single GPU" if not (use_dist or inproc) else {
                    "layout": f"contiguous point shards over {eff_world} rank(s), replicated map",
                    "transport": transport,
                    "transport_detail": {
                        "mailbox": "device-initiated: the ONE persistent launch of every rank stores its 28-double row "
                                   "of each iteration into all ranks' peer-mapped mailboxes and adds what it receives",
                        "host-sum": "one iterate launch per device and round, the devices' rows added on the host "
                                    "(the in-process context's fallback: the mailboxes could not be wired or gave up)",
                        "rccl": "one iterate launch + one ncclAllReduce of 32 doubles per iteration"}[transport],
                    "wiring": ("IN-PROCESS: one process, one caller thread, ONE multi-device context (vgicp_create_multi) "
                               f"over devices {device_ids}; mailboxes wired by plain peer pointers, no launcher, no RCCL"
                               + ("; BENCH_SHARE_DEVICE=1: the sub-contexts split ONE device (dress rehearsal)" if share_device else ""))
                              if inproc else
                              "BENCH_SHARE_DEVICE=1: all ranks are processes on ONE device (dress rehearsal), gloo "
                              "rendezvous, vgicp_peer_export/_connect by hand, "
                              f"VGICP_PERSIST_GRID={os.environ.get('VGICP_PERSIST_GRID')}" if share_device else
                              "one rank per GPU, torch.distributed (nccl = RCCL) rendezvous, vgicp_comm_init",
                    "peer_status": ctx.peer_status() or "wired: the kernels' own mailboxes carry the per-iteration merge",
                    **(exchange or {}),
                },
                "matches_per_iteration": float(res.corr_count.mean()),
                "upload": upload_report,
                "persistent_fallbacks": fallbacks,
                "library": capi.LIB_PATH if os.environ.get("VGICP_LIB_PATH") else "eskf_lio_amd/lib/libvgicp_hip.so",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
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
                "per_launch_variant_us_per_round": bracketed_s * 1e6 if bracketed_s is not None else None,
                "per_launch_variant_source": "iterate_kernel (one launch per round, the multi-GPU path), HIP event "
                                             f"pair around each of {kernel_ms.size} launches, ~2 us event overhead each",
            },
        }
        if replicas is not None:
            out["replicas_aggregate"] = replicas
        if world == 1 and not use_dist and not inproc and not args.no_cpu_baseline:
            base, ref = cpu_baseline(vmap, pts, covs, guess, args.cpu_budget)
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / base["value"]
            # the timed GPU result is also the parity-checked one
            same_counts = bool((ref.corr_count == res.corr_count).all())
            dt = float(np.linalg.norm(ref.pose[:3, 3] - res.pose[:3, 3]))
            out["parity"] = {"identical_counts": same_counts, "pose_delta_m": dt}
        if world == 1 and not use_dist and not inproc and not args.no_frame_chain:
            try:
                out["frame_chain"] = frame_chain_leg(local_rank, frames=args.frames)
            except Exception as e:  # noqa: BLE001 - a secondary record: reported, never raised
                out["frame_chain"] = {"error": f"{type(e).__name__}: {e}"}
        if inproc:
            solo = capi.Context(0)
            solo.map_reset(vmap.voxel_size, n_voxels)
            solo.map_upsert(vmap.keys, vmap.means, vmap.covs)
        if use_dist or inproc:
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
        r5 = None
        try:
            n5, v5 = synth.CONFIGS["C5"]
            map5 = synth.make_map(v5)
            pts5, covs5 = synth.make_uniform_scan(n5, map5)
            lo5, hi5 = shard_bounds(n5, world, rank)
            rank_points5 = -(-n5 // eff_world) if inproc else hi5 - lo5
            ctx.map_reset(map5.voxel_size, v5)
            ctx.map_upsert(map5.keys, map5.means, map5.covs)
            ctx.scan_upload(np.ascontiguousarray(pts5[lo5:hi5]), np.ascontiguousarray(covs5[lo5:hi5]))
            _KEEP_ALIVE.append((map5, pts5, covs5))   # freed with the process, not in front of the timed aligns (see `cold`)
            ready, why = 1.0, ""
        except Exception as e:  # noqa: BLE001 - reported in the JSON line
            ready, why = 0.0, f"{type(e).__name__}: {e}"
        # the aligns below are collective under --gpus N: every rank runs them or none does (a rank that failed to
        # set up would leave its peers waiting in the exchange)
        all_ready = -host_max(-ready) > 0.5
        if not all_ready:
            c5 = {"error": why or "another rank could not set up the C5 map / scan; leg skipped on every rank"}
        else:
            try:
                for _ in range(3):
                    r5 = step_resident()
                steps5 = 10
                el5, dev5, r5, _ = timed(lambda k: step_resident(), steps5)
                per_rank_bytes = algorithmic_bytes(rank_points5, float(r5.corr_count.mean()) / eff_world) * ITERATIONS
                c5 = {"value": n5 * ITERATIONS * steps5 / el5, "unit": "points/s", "ms_per_step": el5 / steps5 * 1e3,
                      "us_per_round": dev5 / steps5 / ITERATIONS * 1e6,
                      "achieved_GBs_per_gpu": per_rank_bytes / (dev5 / steps5) / 1e9,
                      "frac_of_8TBs_per_gpu": per_rank_bytes / (dev5 / steps5) / 1e9 / HBM_PEAK_GBS,
                      "single_launch": r5.launches == 1, "matches_per_iteration": float(r5.corr_count.mean()),
                      "workload": f"C5: {n5}-pt scan vs {v5}-voxel map, {ITERATIONS} rounds per align, scan resident, "
                                  f"point-sharded over {eff_world} rank(s); algorithmic bytes of a rank / its event span"}
            except Exception as e:  # noqa: BLE001 - a failing collective align is fatal for the run: say so and stop
                if use_dist:
                    raise
                c5 = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and out is not None:
            out["c5_resident"] = c5
    if share_device and not inproc:
        dist.barrier()                 # nobody unmaps a mailbox a peer's kernel may still be writing into
        ctx.peer_disconnect()
    ctx.close()
    if use_dist and world > 1:
        # every rank's own context is closed: rank 0 now drives ALL the devices from its one thread through ONE
        # multi-device context (what the reference's single-threaded caller would hold) — the others wait
        dist.barrier()
        if rank == 0 and out is not None:
            # in a CHILD process with a time limit: this path has never met a multi-GPU node, and whatever it does there —
            # a hang included — must not cost the launcher's own line
            try:
                import subprocess
                ids = [0] * world if share_device else list(range(world))
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                         "VGICP_PERSIST_GRID", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-process-leg", ",".join(map(str, ids)),
                                        "--config", args.config, "--steps", str(max(10, min(args.steps, 100)))],
                                       capture_output=True, text=True, timeout=240, env=env)
                lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
                out["in_process"] = json.loads(lines[-1]) if child.returncode == 0 and lines else {
                    "error": f"child exited with {child.returncode}: {child.stderr[-400:]}"}
            except Exception as e:  # noqa: BLE001 - a secondary record: reported, never raised
                out["in_process"] = {"error": f"{type(e).__name__}: {e}"}
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
