"""-m gpu: the in-process multi-device context (vgicp_create_multi) and the drop-in classes riding
the resident frame chain.  One caller thread, N sub-contexts; on the one-GPU boxes of this pool the sub-contexts share
device 0 (device_ids = [0] * N), which runs the same code — sharded upload, one persistent launch per sub-context,
mailboxes wired by plain pointers, replicated map — except that the stores do not cross xGMI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import TIGHT_POSE_TOL, pose_error

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MULTI_POSE_TOL = 1e-11      # sharding changes the grouping of the sums, nothing else


def _single(vmap):
    from eskf_lio_amd import capi
    ctx = capi.Context(0)
    ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    return ctx


def _sorted_export(ctx):
    k, m, c, n = ctx.map_export()
    o = np.lexsort(k.T)
    return k[o], m[o], c[o], n[o]


def test_multi_device_align_matches_one_device(c1_inputs, c1_oracle_map):
    """Two sub-contexts: identical correspondence counts every round, pose within 1e-11 of the single-device context
    (and of the oracle within the usual bound), one launch per sub-context, the same bits run to run, ragged and
    empty shards, the launch-per-round loop with the rows added on the host (NO_PERSISTENT / PROFILE)."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    with _single(vmap) as one, capi.Context([0, 0]) as ctx:
        assert ctx.device_info()[1] == one.device_info()[1]                      # distinct devices are counted once
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        assert ctx.map_size()[0] == one.map_size()[0] == 50_000
        ref = one.align(pts, covs, g, 20, 1e-6, 2.0)
        got = ctx.align(pts, covs, g, 20, 1e-6, 2.0)
        assert got.world_size == 2 and got.launches == 1 and got.iterations == 20 and not got.converged
        assert np.array_equal(got.corr_count, ref.corr_count)
        assert np.abs(got.pose - ref.pose).max() <= MULTI_POSE_TOL
        assert np.allclose(got.normal_eq, ref.normal_eq, rtol=1e-11, atol=1e-9)
        oracle_ref = c1_oracle_map.align(pts, covs, g, 20, 1e-6, 2.0)
        dt, dr = pose_error(got.pose, oracle_ref.pose)
        assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL and np.array_equal(got.corr_count, oracle_ref.corr_count)
        again = ctx.align(pts, covs, g, 20, 1e-6, 2.0)
        assert np.array_equal(again.pose, got.pose) and np.array_equal(again.normal_eq, got.normal_eq)
        # the resident form
        ctx.scan_upload(pts, covs)
        res = ctx.align_resident(g, 20, 1e-6, 2.0)
        assert np.array_equal(res.pose, got.pose)
        dp, dc = ctx.scan_download()
        assert np.array_equal(dp, pts) and np.array_equal(dc, covs)
        # a converging run stops in the same round everywhere
        spts, scovs, _ = synth.make_structured_scan(5_000, vmap)
        a, b = ctx.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999), one.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
        assert a.converged and a.iterations == b.iterations == 3 and np.abs(a.pose - b.pose).max() <= MULTI_POSE_TOL
        # shards of 0, 1, 2 points and a ragged one
        for n in (0, 1, 3, 449, 4_999):
            a = ctx.align(pts[:n], covs[:n], g, 5, 1e-6, 2.0, allow_degenerate=True)
            b = one.align(pts[:n], covs[:n], g, 5, 1e-6, 2.0, allow_degenerate=True)
            assert a.status == b.status and np.array_equal(a.corr_count, b.corr_count)
            assert np.allclose(a.pose, b.pose, rtol=0, atol=1e-9, equal_nan=True)
        # one launch per round on every sub-context, the rows added on the host in the mailbox order
        for flags in (capi.FLAG_NO_PERSISTENT, capi.FLAG_PROFILE):
            h = ctx.align(pts, covs, g, 20, 1e-6, 2.0, flags=flags)
            assert h.launches == 21 and h.world_size == 2 and np.array_equal(h.corr_count, ref.corr_count)
            assert np.abs(h.pose - ref.pose).max() <= MULTI_POSE_TOL
        assert h.kernel_ms is not None and (h.kernel_ms[:20] > 0).all()
        z = ctx.align(pts, covs, g, 0, 1e-6, 2.0)
        assert z.iterations == 0 and np.array_equal(z.pose, g)
        assert ctx.counter(1) == 0                                               # nothing ever gave up
        # the hooks that work on one device answer from sub-context 0
        J, r, c = ctx.accumulate(pts, covs, g)
        J1, r1, c1 = one.accumulate(pts, covs, g)
        assert c == c1 and np.array_equal(J, J1) and np.array_equal(r, r1)
        with pytest.raises(capi.VgicpError):
            ctx.align_resident(g, 5, 1e-6, 2.0)                                  # the hook replaced the resident scan
        with pytest.raises(capi.VgicpError):
            ctx.comm_unique_id()                                                 # communicators are for one process per GPU


def test_multi_device_many_points_per_thread(c1_inputs):
    """Shards larger than a sub-context's grid (128 workgroups x 448 points): the several-points-per-thread
    instantiation with mailboxes."""
    from eskf_lio_amd import capi, synth
    vmap = c1_inputs[0]
    pts, covs = synth.make_uniform_scan(131_073, vmap, seed=5)
    g = synth.default_guess()
    with _single(vmap) as one, capi.Context([0, 0]) as ctx:
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        a, b = ctx.align(pts, covs, g, 6, 1e-6, 2.0), one.align(pts, covs, g, 6, 1e-6, 2.0)
        assert a.launches == 1 and np.array_equal(a.corr_count, b.corr_count)
        assert np.abs(a.pose - b.pose).max() <= MULTI_POSE_TOL and ctx.counter(1) == 0


def test_multi_device_map_calls_are_replicated(c1_inputs, oracle):
    """upsert / erase / insert_scan / evict on a multi-device context reach every replica: the map every sub-context
    registers against is the single-device one, bit for bit."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    rng = np.random.default_rng(3)
    with _single(vmap) as one, capi.Context([0, 0]) as ctx:
        ctx.map_reset(vmap.voxel_size, 0)
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for c in (one, ctx):
            c.map_erase(vmap.keys[:1000])
            T = synth.se3_to_SE3([0.2, 0.1, 0.0, 0.0, 0.0, 0.02])
            p = vmap.means[rng.choice(50_000, 8_000)] + 0.01
            c.map_insert_scan(p, covs[rng.choice(5_000, 8_000)], T, 5)
            rng = np.random.default_rng(3)                                      # the same scan for both
        removed = [c.map_evict(np.zeros(3), 5.0) for c in (one, ctx)]
        assert removed[0] == removed[1] > 0
        assert one.map_size()[0] == ctx.map_size()[0]
        for a, b in zip(_sorted_export(one), _sorted_export(ctx)):
            assert np.array_equal(a, b)
        g = synth.default_guess()
        a, b = ctx.align(pts, covs, g, 10, 1e-6, 2.0), one.align(pts, covs, g, 10, 1e-6, 2.0)
        assert np.array_equal(a.corr_count, b.corr_count) and np.abs(a.pose - b.pose).max() <= MULTI_POSE_TOL


def test_multi_device_frame_chain(oracle):
    """prepare (device 0) -> deal out -> sharded align -> insertion into every replica, the waiting and the non-waiting
    form, against the same chain on a single-device context: identical prepared scans and maps (bit for bit: the
    insertion is order-exact), identical round counts, poses within 1e-11."""
    from eskf_lio_amd import capi, synth
    frames, n = 4, 30_000
    st = synth.make_imu_states(48, seed=9)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=9)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    raws = [synth.make_lidar_scan(n, seed=40 + f) for f in range(frames)]
    for deferred in (False, True):
        with capi.Context(0) as one, capi.Context([0, 0]) as ctx:
            poses = []
            for c in (one, ctx):
                c.map_reset(0.3, 0)
                pose, track = np.eye(4), []
                for f in range(frames):
                    if deferred:
                        c.scan_prepare_async(raws[f], t, st, ext, 0.3, 30)
                    else:
                        kept, moved = c.scan_prepare(raws[f], t, st, ext, 0.3, 30)
                        assert kept > 1000 and moved > 0
                    if f > 0:
                        r = c.align_resident(pose, 30, 1e-6, 0.9999)
                        pose = r.pose
                        track.append((r.iterations, r.pose, r.corr_count))
                    if deferred:
                        c.map_insert_resident_async(pose, 20)
                    else:
                        c.map_insert_resident(pose, 20)
                poses.append(track)
                c.map_size()
            for (i1, p1, c1), (i2, p2, c2) in zip(*poses):
                assert i1 == i2 and np.array_equal(c1, c2) and np.abs(p1 - p2).max() <= MULTI_POSE_TOL
            k1, k2 = one.scan_info(), ctx.scan_info()
            assert k1 == k2
            for a, b in zip(one.scan_download(), ctx.scan_download()):
                assert np.array_equal(a, b)
            assert one.map_size()[0] == ctx.map_size()[0]
            e1, e2 = _sorted_export(one), _sorted_export(ctx)
            assert np.array_equal(e1[0], e2[0]) and np.array_equal(e1[3], e2[3])
            assert np.abs(e1[1] - e2[1]).max() <= 1e-9 and np.abs(e1[2] - e2[2]).max() <= 1e-9   # poses differ at 1e-16
            assert ctx.counter(1) == 0


def test_multi_device_give_up_is_handled_in_process(c1_inputs, monkeypatch):
    """An in-kernel wait that runs out (forced: poll budget zero) on a multi-device context: every sub-context's
    launch has ended when its thread returns, so the ONE process re-arms all mailboxes and runs the align with the
    rows added on the host — the right result, counted, the next aligns stay on that loop, then the single launch
    is tried again.  No rank is ever left with a pose of its own."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    with _single(vmap) as one:
        ref = one.align(pts, covs, g, 10, 1e-6, 2.0)
    monkeypatch.setenv("VGICP_SPIN_LIMIT", "0")
    with capi.Context([0, 0]) as ctx:
        monkeypatch.delenv("VGICP_SPIN_LIMIT")
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for k in range(12):
            r = ctx.align(pts, covs, g, 10, 1e-6, 2.0)
            assert r.launches == 11 and r.world_size == 2                        # never the single launch: it cannot complete
            assert np.array_equal(r.corr_count, ref.corr_count) and np.abs(r.pose - ref.pose).max() <= MULTI_POSE_TOL
        # 12 aligns: the single launch tried at #0 and, after 8 aligns on the loop, again at #9
        assert ctx.counter(0) == 2 and ctx.counter(1) == 2
    monkeypatch.setenv("VGICP_MULTI_EXCHANGE", "host")                           # never wired: the loop from the start
    with capi.Context([0, 0, 0]) as ctx:
        monkeypatch.delenv("VGICP_MULTI_EXCHANGE")
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        r = ctx.align(pts, covs, g, 10, 1e-6, 2.0)
        assert r.launches == 11 and r.world_size == 3 and np.array_equal(r.corr_count, ref.corr_count)
        assert np.abs(r.pose - ref.pose).max() <= MULTI_POSE_TOL and ctx.counter(0) == 0


def _worker(n, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_worker.py"), str(n)], capture_output=True,
                         text=True, timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]), out.stderr


@pytest.mark.parametrize("world", [4, 8])
def test_multi_device_four_and_eight_sub_contexts(world):
    """N = 4 and 8 sub-contexts on the one device.  Every sub-context's persistent launch needs a hardware queue of
    its own (they wait for each other inside the kernel); the runtime's default is 4 per process, so the child runs
    with GPU_MAX_HW_QUEUES raised — on N real devices every launch has its device's queues to itself."""
    clean = False
    for attempt in range(3):
        d, _ = _worker(world, {"GPU_MAX_HW_QUEUES": "24"})
        # right under every circumstance ...
        assert d["world_size"] == world and d["counts_equal"] and d["pose_delta"] <= MULTI_POSE_TOL and d["repeatable"]
        assert d["big_counts_equal"] and d["big_pose_delta"] <= MULTI_POSE_TOL
        # ... and as ONE launch per sub-context, without a single give-up, when every launch has a queue of its own (how
        # the runtime spreads streams over hardware queues is its business: up to three tries)
        if d["launches"] == [1, 1, 1] and d["fallbacks"] == 0:
            clean = True
            break
    assert clean, d


def test_multi_device_sub_contexts_short_of_hardware_queues_still_return_the_right_pose():
    """The same four sub-contexts with the runtime's default queue count: launches that share a hardware queue cannot
    run side by side, their in-kernel waits give up (bounded), and the context finishes every align on the host-summed
    loop — slower, never wrong."""
    d, err = _worker(4, {"VGICP_SPIN_LIMIT": "400", "GPU_MAX_HW_QUEUES": "4"})
    assert d["world_size"] == 4 and d["counts_equal"] and d["pose_delta"] <= MULTI_POSE_TOL
    assert d["big_counts_equal"] and d["big_pose_delta"] <= MULTI_POSE_TOL
    if d["fallbacks"]:
        assert "added on the host" in err and all(n > 1 for n in d["launches"][:1])


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_shim_classes_and_cpp_example_on_a_multi_device_context(devices):
    """The unchanged drop-in classes over a multi-device context: VGICP_DEVICES selects it for the shim's process-wide
    context, nothing else changes — the host-mirror tests (LocalMap, ICP, CloudPreprocessor through libvgicp_host.so)
    and examples/register_frames pass as they do on one device."""
    env = dict(os.environ, VGICP_DEVICES=devices, GPU_MAX_HW_QUEUES="24")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                          os.path.join(ROOT, "tests", "test_multi_device.py"), "-q", "-x", "-m", "gpu",
                          "-k", "test_host_mirror_ or test_cpp_example_tracks or test_dropin_"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
    exe = os.path.join(ROOT, "examples", "register_frames")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("frame")]
    assert len(lines) == 6 and all("converged 1" in ln for ln in lines[1:])
    one = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    # the same trajectory as on one device, to the printed precision at least
    single = [ln for ln in one.stdout.splitlines() if ln.startswith("frame")]
    assert [ln.split("converged")[1] for ln in lines[1:]] == [ln.split("converged")[1] for ln in single[1:]]


# ---- the drop-in classes riding the resident chain --------------------------------------------------------------
def _frame_inputs(frames=5, n=30_000):
    from eskf_lio_amd import synth
    st = synth.make_imu_states(48, seed=9)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=9)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    raws = [synth.make_lidar_scan(n, seed=60 + f) for f in range(frames)]
    return st, t, ext, raws


def _abi_chain(st, t, ext, raws, cap=20):
    """The frames through vgicp_scan_prepare_async / _align_resident / _map_insert_resident_async."""
    from eskf_lio_amd import capi
    with capi.Context(0) as c:
        c.map_reset(0.3, 0)
        pose, out = np.eye(4), []
        for f, raw in enumerate(raws):
            # the first sweep goes in without a state queue, as src/Odometry.cpp:60 calls process({}, ...)
            c.scan_prepare_async(raw, t if f else None, st if f else None, ext, 0.3, 30)
            if f > 0:
                r = c.align_resident(pose, 30, 1e-6, 0.9999)
                pose = r.pose
                out.append((r.pose, r.iterations, int(r.corr_count[0])))
            c.map_insert_resident_async(pose, cap)
        voxels = c.map_size()[0]
        prepared = c.scan_download()
    return out, voxels, prepared


_NO_GATE = dict(translation_sq_threshold=-1.0, cosine_threshold=2.0, remove_distant_points=False, distance_threshold=1e9,
                removing_period=1e9)


@pytest.mark.parametrize("host_copy,device_resident", [("deferred", True), ("eager", True), ("eager", False), ("deferred", False)])
def test_dropin_classes_ride_the_resident_chain(host_copy, device_resident):
    """CloudPreprocessor::process -> ICP::align -> LocalMap::updateLocalMap through the C++ classes, written as
    src/Odometry.cpp:73-87 writes it.  process() leaves the prepared scan on the device and stamps the host cloud;
    align() finds that cloud resident (no second upload) in every mode; with the grid on the device the map update
    is the resident insertion too, and the poses are the bits of the ABI chain that does not wait."""
    from eskf_lio_amd import host
    st, t, ext, raws = _frame_inputs()
    want, voxels, prepared = _abi_chain(st, t, ext, raws)
    pre = host.CloudPreprocessor(0.3, ext, host_copy)
    icp = host.ICP(30, 1e-6, 0.9999)
    lmap = host.LocalMap(0.3, 20, dict(_NO_GATE, device_resident=device_resident))
    pose = np.eye(4)
    for f, raw in enumerate(raws):
        fr = host.Frame(raw, t, st)
        fr.run(pre, icp, lmap, pose, first_frame=(f == 0))
        got = fr.end(want_cloud=True)
        if f == 0:
            continue
        pose = got["pose"]
        assert got["used_resident"], "align() uploaded a cloud that process() had left resident"
        assert got["iterations"] == want[f - 1][1] and got["corr0"] == want[f - 1][2]
        if device_resident and not os.environ.get("VGICP_DEVICES"):
            assert np.array_equal(pose, want[f - 1][0])                          # the very calls of the ABI chain
        else:
            assert np.abs(pose - want[f - 1][0]).max() <= 1e-9                   # host-built map: the same voxels to rounding
        if host_copy == "deferred" and device_resident:
            assert got["host_points"] == raw.shape[0]                            # the host cloud was never touched
        elif f == len(raws) - 1 and host_copy == "eager" and device_resident:
            assert got["host_points"] == prepared[0].shape[0]                    # it holds the prepared scan (world frame by now)
    assert len(lmap) == voxels


def test_dropin_sweeps_staged_on_arrival_give_the_same_poses():
    """CloudPreprocessor::stage (the optional hook for the lidar callback): every frame's sweep is staged while the
    frame before is being processed; process() picks the staged sweep up by its measurement object.  Poses, round counts
    and the map are those of the chain that copies inside process()."""
    from eskf_lio_amd import host
    st, t, ext, raws = _frame_inputs()
    want, voxels, _ = _abi_chain(st, t, ext, raws)
    pre = host.CloudPreprocessor(0.3, ext, "deferred")
    icp = host.ICP(30, 1e-6, 0.9999)
    lmap = host.LocalMap(0.3, 20, dict(_NO_GATE, device_resident=True))
    pose = np.eye(4)
    frames = [host.Frame(raw, t, st) for raw in raws]
    assert frames[0].stage(pre)
    for f, fr in enumerate(frames):
        nxt = frames[f + 1] if f + 1 < len(frames) else None
        if f == 0:
            fr.run(pre, icp, lmap, pose, first_frame=True)
            if nxt is not None:
                assert nxt.stage(pre)
        else:
            fr.run(pre, icp, lmap, pose, stage_next=nxt)
        got = fr.end()
        if f == 0:
            continue
        pose = got["pose"]
        assert got["used_resident"] and got["iterations"] == want[f - 1][1] and got["corr0"] == want[f - 1][2]
        if not os.environ.get("VGICP_DEVICES"):
            assert np.array_equal(pose, want[f - 1][0])
    assert len(lmap) == voxels


def test_dropin_align_falls_back_when_the_cloud_changed():
    """A caller that edits or resizes the prepared cloud between process() and align() gets what it asked for: the
    stamp no longer matches, align() uploads the cloud as it is now, and the result is the registration of THAT cloud —
    by DEFAULT for an edit of any element (mutate 3: point 1 of thousands, which a 64-sample stamp never looks at: the
    classes hash every byte unless CloudPreprocessorConfig::residentCheck says Sampled), as the reference always reads
    the host cloud (src/Registration.cpp:11, src/LocalMap.cpp:45-58)."""
    from eskf_lio_amd import host
    st, t, ext, raws = _frame_inputs(frames=2)
    for host_copy in ("eager", "deferred"):
        for mutate in (1, 2, 3):
            pre = host.CloudPreprocessor(0.3, ext, host_copy)
            icp = host.ICP(30, 1e-6, 0.9999)
            lmap = host.LocalMap(0.3, 20, dict(_NO_GATE, device_resident=True))
            fr = host.Frame(raws[0], t, st)
            fr.run(pre, icp, lmap, np.eye(4), first_frame=True)
            fr.end()
            # what the edited cloud should register to: the prepared scan, edited the same way, through plain ICP::align
            gp, gc = host.CloudPreprocessor(0.3, ext, "eager").process(st, raws[1], t)
            if mutate == 1:
                gp = gp.copy()
                gp[0, 0] += 1e-3
            elif mutate == 3:
                assert gp.shape[0] > 200
                gp = gp.copy()
                gp[1, 0] += 1e-3
            else:
                gp, gc = gp[:-1], gc[:-1]
            want = host.ICP(30, 1e-6, 0.9999)
            want_pose = want.align(gp, gc, lmap, np.eye(4))
            fr = host.Frame(raws[1], t, st)
            fr.run(pre, icp, lmap, np.eye(4), mutate=mutate)
            got = fr.end()
            assert not got["used_resident"]
            assert got["iterations"] == want.iterations and np.array_equal(got["pose"], want_pose)
    # the opt-in: a 64-sample stamp sees the edit of a sampled element (and a resize), not that of an unsampled one
    for mutate, seen in ((1, True), (3, False)):
        pre = host.CloudPreprocessor(0.3, ext, "eager", resident_check="sampled")
        icp = host.ICP(30, 1e-6, 0.9999)
        lmap = host.LocalMap(0.3, 20, dict(_NO_GATE, device_resident=True))
        fr = host.Frame(raws[0], t, st)
        fr.run(pre, icp, lmap, np.eye(4), first_frame=True)
        fr.end()
        fr = host.Frame(raws[1], t, st)
        fr.run(pre, icp, lmap, np.eye(4), mutate=mutate)
        assert fr.end()["used_resident"] == (not seen)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_line_in_process_multi_device(world):
    """`python bench.py --gpus N` WITHOUT a launcher: one process, one thread, one multi-device context.  On this pool's
    one-GPU boxes the sub-contexts share device 0 (BENCH_SHARE_DEVICE=1); the line keeps the driver's contract, says how
    it was wired, and the sharded result is the single-device one."""
    env = dict(os.environ, BENCH_SHARE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
                          "--no-c5"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == world and d["scaling"] == "strong" and d["steps"] == 6
    sh = d["config"]["sharding"]
    assert "IN-PROCESS" in sh["wiring"] and "vgicp_create_multi" in sh["wiring"]
    assert d["multi_gpu_parity"]["identical_counts"] is True and d["multi_gpu_parity"]["pose_delta"] <= MULTI_POSE_TOL
    if d["config"]["persistent_fallbacks"] == 0:         # every sub-context's launch had a hardware queue of its own
        assert sh["transport"] == "mailbox" and d["roofline"]["rounds_per_launch"] == 20
    else:                                                # it says so, and the result is still the single-device one
        assert sh["transport"] in ("mailbox", "host-sum")
    assert d["roofline"]["traffic"] is None and "reason" in d["roofline"]["traffic_source"]
