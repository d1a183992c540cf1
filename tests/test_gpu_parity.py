"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libvgicp_hip.so, and through
the C++ host mirror libvgicp_host.so), against the CPU oracle and the committed golden fixtures.

Bar (BASELINE.json north_star): final pose within 1e-4 m / 1e-4 rad of the CPU path and IDENTICAL
per-iteration correspondence counts.  The tests also assert the much tighter bound the fp64 kernels
actually reach (1e-9), so drift is caught early.  Integer work (voxel keys, match indices, counts)
is compared bit-exactly.
"""
import os

import numpy as np
import pytest

from conftest import POSE_TOL_M, POSE_TOL_RAD, TIGHT_POSE_TOL, NORMAL_EQ_RTOL, pose_error

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def assert_align_parity(got, ref, tight=TIGHT_POSE_TOL):
    assert got.iterations == ref.iterations
    assert got.converged == ref.converged
    assert np.array_equal(got.corr_count, ref.corr_count)          # identical correspondence counts
    dt, dr = pose_error(got.pose, ref.pose)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD                 # contractual tolerance
    assert dt <= tight and dr <= tight                             # what the HIP path achieves


# ---- device / table ---------------------------------------------------------------------------
def test_device_is_gfx950(gpu_ctx):
    name, cus, hbm = gpu_ctx.device_info()
    assert name.startswith("gfx950") and cus >= 200 and hbm > 200e9


def test_map_upsert_overwrite_erase_and_growth(gpu_ctx, c1_inputs, oracle):
    vmap, pts, covs = c1_inputs
    V = vmap.keys.shape[0]
    gpu_ctx.map_reset(vmap.voxel_size, 0)                          # no hint: forces growth + rehash
    for lo in range(0, V, 7_000):                                  # ragged batches
        gpu_ctx.map_upsert(vmap.keys[lo:lo + 7_000], vmap.means[lo:lo + 7_000], vmap.covs[lo:lo + 7_000])
    voxels, slots = gpu_ctx.map_size()
    assert voxels == V and slots >= 2 * V and slots & (slots - 1) == 0
    gpu_ctx.map_upsert(vmap.keys[:100], vmap.means[:100], vmap.covs[:100])   # pure overwrite
    assert gpu_ctx.map_size()[0] == V
    # lookups see exactly the inserted voxels
    sp, sc, mp, mc, ix = gpu_ctx.match(vmap.means, vmap.covs)
    assert len(ix) == V and np.array_equal(ix, np.arange(V, dtype=np.uint64))
    assert np.array_equal(mp, vmap.means) and np.array_equal(mc, vmap.covs)
    # erase half (with duplicates and absent keys in the batch), lookups follow
    gone = vmap.keys[::2]
    batch = np.concatenate([gone, gone[:50], np.full((10, 3), 10_000, dtype=np.int32)])
    gpu_ctx.map_erase(batch)
    assert gpu_ctx.map_size()[0] == V - gone.shape[0]
    _, _, _, _, ix = gpu_ctx.match(vmap.means, vmap.covs)
    assert np.array_equal(ix, np.arange(1, V, 2, dtype=np.uint64))
    # re-insert with new values: tombstones do not hide or duplicate them
    gpu_ctx.map_upsert(gone, vmap.means[::2] + 0.001, vmap.covs[::2])
    assert gpu_ctx.map_size()[0] == V
    _, _, mp, _, ix = gpu_ctx.match(vmap.means, vmap.covs)
    assert len(ix) == V and np.array_equal(mp[::2], vmap.means[::2] + 0.001) and np.array_equal(mp[1::2], vmap.means[1::2])
    gpu_ctx.map_upsert(np.zeros((0, 3), np.int32), np.zeros((0, 3)), np.zeros((0, 9)))   # empty batch


def test_voxel_index_is_bit_exact(gpu_ctx, oracle):
    gpu_ctx.map_reset(0.3, 0)
    rng = np.random.default_rng(5)
    rnd = rng.uniform(-50, 50, size=(60_000, 3))
    on_face = np.round(rnd / 0.3) * 0.3
    edge = np.array([[-0.1, 0.0, 0.3], [-0.3, 0.29999999999999993, 0.6], [-0.30000000000000004, 0.9, -1e-300],
                     [1e-300, -0.0, 299.99999999999994]])
    for p in (rnd, on_face, edge):
        assert np.array_equal(gpu_ctx.voxel_index(p), oracle.voxel_index(0.3, p))
    assert gpu_ctx.voxel_index(edge)[0].tolist() == [-1, 0, 1]     # floor, not truncation
    # The loop kernels take the key without dividing (x * (1 / h), remainder test, real division only near
    # an integer quotient): coordinates within a few ulps of every kind of cell boundary, for several voxel
    # sizes, must give the reference's floor(x / h) exactly — including where the correctly rounded quotient
    # rounds UP to the next integer.
    for h in (0.3, 0.1, 0.7, 1.0, 0.05, 2.5, 1.0 / 3.0):
        gpu_ctx.map_reset(h, 0)
        k = np.concatenate([np.arange(-2_000, 2_000), rng.integers(-2**29, 2**29, size=4_000),
                            rng.integers(-70_000, 70_000, size=6_000)]).astype(np.float64)
        base = k * h
        cols = [base]
        for ulps in (1, 2, 3, 5, 17):
            up = base.copy()
            dn = base.copy()
            for _ in range(ulps):
                up = np.nextafter(up, np.inf)
                dn = np.nextafter(dn, -np.inf)
            cols += [up, dn]
        near = np.concatenate(cols)
        near = near[: (near.size // 3) * 3].reshape(-1, 3)
        assert np.array_equal(gpu_ctx.voxel_index(near), oracle.voxel_index(h, near)), h
        far = rng.uniform(-1e5, 1e5, size=(30_000, 3))
        assert np.array_equal(gpu_ctx.voxel_index(far), oracle.voxel_index(h, far)), h
    gpu_ctx.map_reset(0.3, 0)


def test_match_equals_oracle_correspondences(c1_gpu, c1_inputs, c1_oracle_map, oracle):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    tp, tc = oracle.transform(pts, covs, synth.default_guess())
    got = c1_gpu.match(tp, tc)
    ref = c1_oracle_map.match(tp, tc)
    assert len(got[4]) == len(ref[4]) > 2000
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)                                # bit-exact, ascending point order
    # ragged sizes around the workgroup width, and the empty scan
    for n in (0, 1, 63, 64, 65, 255, 256, 257, 1000):
        g = c1_gpu.match(tp[:n], tc[:n])
        r = c1_oracle_map.match(tp[:n], tc[:n])
        assert all(np.array_equal(a, b) for a, b in zip(g, r))


# ---- one iteration ----------------------------------------------------------------------------
def test_k1_single_correspondence_on_device(gpu_ctx, oracle):
    gpu_ctx.map_reset(0.3, 0)
    mu = np.array([[0.95, 2.05, 3.1]])
    gpu_ctx.map_upsert(np.floor(mu / 0.3).astype(np.int32), mu, 0.5 * np.eye(3).reshape(1, 9))
    p = np.array([[1.0, 2.0, 3.0]])
    JTJ, JTr, cnt = gpu_ctx.accumulate(p, 0.5 * np.eye(3).reshape(1, 9), np.eye(4))
    assert cnt == 1
    oJ, oR = oracle.jtj_jtr(p[0], mu[0], np.eye(3))
    assert np.allclose(JTJ, oJ, rtol=0, atol=1e-14) and np.allclose(JTr, oR, rtol=0, atol=1e-14)
    assert JTJ[3, 3] == 13.0 and JTJ[0, 4] == 3.0                  # hand-written entries for p = (1,2,3)


def test_accumulate_matches_oracle(c1_gpu, c1_inputs, c1_oracle_map, oracle):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    for pose in (np.eye(4), synth.default_guess(), synth.se3_to_SE3([0.5, -0.4, 0.3, 0.2, -0.1, 0.3])):
        JTJ, JTr, cnt = c1_gpu.accumulate(pts, covs, pose)
        tp, tc = oracle.transform(pts, covs, pose)
        oJ, oR, oc = c1_oracle_map.accumulate(tp, tc)
        assert cnt == oc
        assert np.abs(JTJ - oJ).max() <= NORMAL_EQ_RTOL * np.abs(oJ).max()
        assert np.abs(JTr - oR).max() <= NORMAL_EQ_RTOL * max(np.abs(oR).max(), 1.0)


def test_accumulate_is_linear_over_shards(c1_gpu, c1_inputs):
    """Point sharding is exact up to summation order: the multi-GPU path's premise."""
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    g = synth.default_guess()
    J, r, c = c1_gpu.accumulate(pts, covs, g)
    parts = [c1_gpu.accumulate(pts[lo:hi], covs[lo:hi], g) for lo, hi in ((0, 1700), (1700, 1701), (1701, 5000))]
    assert sum(p[2] for p in parts) == c
    assert np.abs(sum(p[0] for p in parts) - J).max() <= 1e-12 * np.abs(J).max()
    assert np.abs(sum(p[1] for p in parts) - r).max() <= 1e-12 * np.abs(J).max()


# ---- the whole loop ---------------------------------------------------------------------------
def test_align_c1_forced_20_iterations(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    guess = synth.default_guess()
    ref = c1_oracle_map.align(pts, covs, guess, 20, 1e-6, 2.0)
    got = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0)
    assert got.iterations == 20 and not got.converged
    assert_align_parity(got, ref)
    scale = np.abs(ref.JTJ).max(axis=(1, 2), keepdims=True)
    assert (np.abs(got.JTJ - ref.JTJ) <= 1e-8 * scale).all()


def test_align_structured_converges_like_the_reference(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    vmap, _, _ = c1_inputs
    pts, covs, T_true = synth.make_structured_scan(5_000, vmap)
    ref = c1_oracle_map.align(pts, covs, np.eye(4), 100, 1e-6, 0.9999)   # shipped thresholds
    got = c1_gpu.align(pts, covs, np.eye(4), 100, 1e-6, 0.9999)
    assert got.converged and got.iterations == 3
    assert_align_parity(got, ref)
    dt, dr = pose_error(got.pose, T_true)
    assert dt < 2e-3 and dr < 1e-3


@pytest.mark.parametrize("name", ["c1_uniform", "c1_structured"])
def test_align_against_golden_fixture(c1_gpu, c1_inputs, name):
    from eskf_lio_amd import synth
    vmap, pts, covs = c1_inputs
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if name == "c1_structured":
        pts, covs, _ = synth.make_structured_scan(5_000, vmap)
    max_it, tsq, cos = g["params"]
    got = c1_gpu.align(pts, covs, g["guess"], int(max_it), tsq, cos)
    assert got.iterations == int(g["iterations"]) and got.converged == bool(g["converged"])
    assert np.array_equal(got.corr_count, g["corr_count"])
    dt, dr = pose_error(got.pose, g["pose"])
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL


def test_align_tiny_hermetic_fixture(gpu_ctx):
    g = np.load(os.path.join(GOLDEN, "tiny.npz"))
    gpu_ctx.map_reset(float(g["voxel_size"]), 0)
    gpu_ctx.map_upsert(g["keys"], g["means"], g["covs"])
    max_it, tsq, cos = g["params"]
    got = gpu_ctx.align(g["points"], g["point_covs"], g["guess"], int(max_it), tsq, cos)
    assert np.array_equal(got.corr_count, g["corr_count"])
    dt, dr = pose_error(got.pose, g["pose"])
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL


def test_chunking_and_workgroup_schedule_do_not_change_results(c1_gpu, c1_inputs):
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    guess = synth.default_guess()
    NP = capi.FLAG_NO_PERSISTENT                                   # chunking belongs to the per-launch loop
    base = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0, chunk_iterations=20, flags=NP)
    for chunk in (1, 3, 7, 64):
        r = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0, chunk_iterations=chunk, flags=NP)
        assert np.array_equal(r.pose, base.pose) and np.array_equal(r.corr_count, base.corr_count)
    r = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0, flags=capi.FLAG_PROFILE)
    assert np.array_equal(r.pose, base.pose) and r.kernel_ms is not None and (r.kernel_ms[:20] > 0).all()
    # early exit: converging run, every chunk size reports the same iteration count and pose
    spts, scovs, _ = synth.make_structured_scan(5_000, vmap)
    ref = c1_gpu.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999, chunk_iterations=100, flags=NP)
    for chunk in (1, 2, 4, 5):
        r = c1_gpu.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999, chunk_iterations=chunk, flags=NP)
        assert r.iterations == ref.iterations == 3 and r.converged
        assert np.array_equal(r.pose, ref.pose) and r.launches < 100
    # bit-reproducible run to run (the reference is not: SURVEY.md F10)
    again = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0, chunk_iterations=20, flags=NP)
    assert np.array_equal(again.pose, base.pose) and np.array_equal(again.normal_eq, base.normal_eq)


def test_persistent_and_per_launch_variants_return_the_same_bits(c1_gpu, c1_inputs):
    """The single-launch persistent loop and the one-launch-per-round loop share arithmetic, summation
    order and workgroup geometry; only the way rows travel differs (in-kernel sc1 hand-off vs kernel
    boundary). A stale or torn row in the in-kernel exchange would show up here as a bit difference."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    per_launch = c1_gpu.align(pts, covs, g, 20, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
    assert per_launch.launches == 21
    for _ in range(40):                                            # many epochs over the same row buffers
        one = c1_gpu.align(pts, covs, g, 20, 1e-6, 2.0)
        assert one.launches == 1 and one.iterations == 20
        assert np.array_equal(one.pose, per_launch.pose)
        assert np.array_equal(one.normal_eq, per_launch.normal_eq)
        assert np.array_equal(one.corr_count, per_launch.corr_count)
    spts, scovs, _ = synth.make_structured_scan(5_000, vmap)
    a = c1_gpu.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
    b = c1_gpu.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999, flags=capi.FLAG_NO_PERSISTENT)
    assert a.converged and b.converged and a.iterations == b.iterations == 3
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_eq, b.normal_eq)
    for n in (1, 449, 4999):                                       # ragged grids
        a = c1_gpu.align(pts[:n], covs[:n], g, 5, 1e-6, 2.0, allow_degenerate=True)
        b = c1_gpu.align(pts[:n], covs[:n], g, 5, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT, allow_degenerate=True)
        assert np.array_equal(a.normal_eq, b.normal_eq)            # every round's sums: bit for bit
        # the pose after the LAST solve comes from two separately compiled copies of the solve
        # (persistent_kernel vs close_kernel): identical up to the last bit of tiny entries
        assert np.abs(a.pose - b.pose).max() < 1e-15


def test_persistent_exchange_survives_uneven_load_and_every_exit_round(gpu_ctx, c1_inputs):
    """The in-kernel row exchange (data-as-signal, three buffers re-armed in flight, cleaned at exit) under what
    hides hand-off bugs on an idle chip: workgroups with very different amounts of work (scan sizes from one
    point to several points per thread, so most of the 256 workgroups publish at once while a few arrive late),
    launches that end after 1..7 rounds (every buffer takes its turn as the last one), converging runs, all
    back to back on the same buffers with the consumer's caches warm.  Every round's 27 sums and count must be
    the per-launch loop's, bit for bit."""
    from eskf_lio_amd import capi, synth
    vmap = c1_inputs[0]
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    big_p, big_c = synth.make_uniform_scan(300_000, vmap, seed=99)
    g = synth.default_guess()
    rng = np.random.default_rng(3)
    sizes = [1, 447, 448, 449, 5_000, 60_000, 114_688, 114_689, 300_000]
    checked = 0
    for trial in range(60):
        n = sizes[trial % len(sizes)] if trial < 27 else int(rng.integers(1, 300_000))
        rounds = 1 + trial % 7
        gpu_ctx.scan_upload(big_p[:n], big_c[:n])
        one = gpu_ctx.align_resident(g, rounds, 1e-6, 2.0, allow_degenerate=True)
        loop = gpu_ctx.align_resident(g, rounds, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT, allow_degenerate=True)
        assert one.launches == 1 and loop.launches > 1
        assert one.iterations == loop.iterations == rounds
        if n <= 114_688:                                            # same partition of points into workgroups
            assert np.array_equal(one.normal_eq, loop.normal_eq), (trial, n, rounds)
        else:                                                       # the loop uses more, smaller workgroups
            assert np.allclose(one.normal_eq, loop.normal_eq, rtol=1e-11, atol=1e-7), (trial, n, rounds)
        assert np.array_equal(one.corr_count, loop.corr_count)
        again = gpu_ctx.align_resident(g, rounds, 1e-6, 2.0, allow_degenerate=True)
        assert np.array_equal(again.normal_eq, one.normal_eq) and np.array_equal(again.pose, one.pose, equal_nan=True)
        checked += 1
    assert checked == 60 and gpu_ctx.counter(1) == 0               # no launch gave up
    # converging runs leave at a data-dependent round
    spts, scovs, _ = synth.make_structured_scan(5_000, vmap)
    for _ in range(5):
        a = gpu_ctx.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
        b = gpu_ctx.align(big_p[:70_000], big_c[:70_000], g, 6, 1e-6, 2.0)
        assert a.converged and a.iterations == 3 and a.launches == 1 and b.launches == 1


@pytest.mark.parametrize("grid", [1, 5, 16, 17, 100, 255])
def test_persistent_launch_on_fewer_workgroups(c1_inputs, monkeypatch, grid):
    """The persistent launch on a part of the device (VGICP_PERSIST_GRID: contexts of several processes sharing
    one GPU, or a partitioned one): fewer workgroups than exchange rows, fewer than folders, one more than
    folders; scans that fit the grid and scans that give every thread several points. Same counts as the
    per-launch loop, sums equal to rounding (the partition of the points differs), run to run the same bits."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    monkeypatch.setenv("VGICP_PERSIST_GRID", str(grid))
    with capi.Context(0) as ctx:
        monkeypatch.delenv("VGICP_PERSIST_GRID")
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for n in (1, 300, 5_000):
            for rounds in (1, 4, 7):
                a = ctx.align(pts[:n], covs[:n], g, rounds, 1e-6, 2.0, allow_degenerate=True)
                b = ctx.align(pts[:n], covs[:n], g, rounds, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT, allow_degenerate=True)
                again = ctx.align(pts[:n], covs[:n], g, rounds, 1e-6, 2.0, allow_degenerate=True)
                assert a.launches == 1 and b.launches > 1
                assert np.array_equal(a.corr_count, b.corr_count)
                assert np.allclose(a.normal_eq, b.normal_eq, rtol=1e-11, atol=1e-9)
                assert np.array_equal(a.normal_eq, again.normal_eq) and np.array_equal(a.pose, again.pose, equal_nan=True)
        assert ctx.counter(1) == 0


@pytest.mark.parametrize("grid,n", [(1, 9_000), (3, 20_000), (16, 131_072), (255, 131_073)])
def test_persistent_launch_with_many_points_per_thread(c1_inputs, monkeypatch, grid, n):
    """Scans much larger than the launch's grid: more points per thread than memos (12), a last pass that is partial
    in some workgroups and missing in others (the scan is dealt out in units of 64 points), one point per thread in
    the eight-wave mapping, and covariances that are NOT bitwise symmetric (parked points then hold twelve planes).
    Same counts as the per-launch loop, sums equal to rounding, the same bits run to run and with nothing parked."""
    from eskf_lio_amd import capi, synth
    vmap = c1_inputs[0]
    pts, covs = synth.make_uniform_scan(n, vmap, seed=grid)
    g = synth.default_guess()
    monkeypatch.setenv("VGICP_PERSIST_GRID", str(grid))
    with capi.Context(0) as ctx:
        monkeypatch.delenv("VGICP_PERSIST_GRID")
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for skewed in (False, True):
            c = covs.copy()
            if skewed:
                c[n // 3, 3] = np.nextafter(c[n // 3, 3], np.inf)      # c01 != c10 for one point: twelve planes are read
            ctx.scan_upload(pts, c)
            a = ctx.align_resident(g, 5, 1e-6, 2.0)
            b = ctx.align_resident(g, 5, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
            again = ctx.align_resident(g, 5, 1e-6, 2.0)
            monkeypatch.setenv("VGICP_NO_STASH", "1")
            bare = ctx.align_resident(g, 5, 1e-6, 2.0)
            monkeypatch.delenv("VGICP_NO_STASH")
            assert a.launches == 1 and b.launches > 1 and a.iterations == b.iterations == 5
            assert np.array_equal(a.corr_count, b.corr_count)
            assert np.allclose(a.normal_eq, b.normal_eq, rtol=1e-11, atol=1e-8)
            assert np.array_equal(a.normal_eq, again.normal_eq) and np.array_equal(a.pose, again.pose)
            assert np.array_equal(a.normal_eq, bare.normal_eq) and np.array_equal(a.pose, bare.pose)
        assert ctx.counter(1) == 0


def test_dense_copy_of_the_records_returns_the_same_bits(c1_inputs, oracle, monkeypatch):
    """Tables far beyond the caches' reach (2^24 slots and more: C5) get a dense copy of their FULL records that the
    several-points-per-thread launch reads a remembered voxel's payload from.  Forced here on a small table
    (VGICP_DENSE_SLOTS=1): the same bits as without it, before and after every kind of map mutation (the copy is
    rebuilt lazily), and against the oracle."""
    from eskf_lio_amd import capi, synth
    vmap = c1_inputs[0]
    pts, covs = synth.make_uniform_scan(40_000, vmap, seed=21)
    g = synth.default_guess()
    rng = np.random.default_rng(8)
    extra = (vmap.means[rng.choice(50_000, 6_000)] + 0.4, covs[rng.choice(40_000, 6_000)])

    def drive(dense_slots):
        monkeypatch.setenv("VGICP_PERSIST_GRID", "16")                 # 16 x 448 < 40 000: several points per thread
        monkeypatch.setenv("VGICP_DENSE_SLOTS", dense_slots)
        out = []
        with capi.Context(0) as ctx:
            monkeypatch.delenv("VGICP_PERSIST_GRID")
            monkeypatch.delenv("VGICP_DENSE_SLOTS")
            ctx.map_reset(vmap.voxel_size, 0)
            ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))          # copy still valid: no rebuild
            ctx.map_insert_scan(extra[0], extra[1], np.eye(4), 5)      # new voxels and changed means
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))
            ctx.map_erase(vmap.keys[:5_000])
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))
            ctx.map_evict(np.zeros(3), 4.0)
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))
            ctx.map_upsert(vmap.keys[:5_000], vmap.means[:5_000] + 0.01, vmap.covs[:5_000])
            out.append(ctx.align(pts, covs, g, 6, 1e-6, 2.0))
            assert ctx.counter(1) == 0
        return out
    plain, dense = drive("0"), drive("1")
    for a, b in zip(plain, dense):
        assert a.launches == b.launches == 1
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_eq, b.normal_eq) and np.array_equal(a.corr_count, b.corr_count)
    assert not np.array_equal(dense[0].normal_eq, dense[2].normal_eq)   # the mutations mattered
    assert not np.array_equal(dense[2].normal_eq, dense[3].normal_eq) and not np.array_equal(dense[3].normal_eq, dense[4].normal_eq)
    om = oracle.OracleMap(vmap.voxel_size, 1)
    om.insert(vmap.means, vmap.covs)
    ref = om.align(pts, covs, g, 6, 1e-6, 2.0)
    assert np.array_equal(ref.corr_count, dense[0].corr_count)
    dt, dr = pose_error(dense[0].pose, ref.pose)
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL


def test_persistent_launch_that_gives_up_falls_back_and_recovers(c1_inputs, monkeypatch):
    """A persistent launch whose in-kernel wait runs out (forced here with a poll budget of zero; in the field:
    another process holds compute units) must leave no trace: the align is re-run with one launch per
    iteration and returns the same bits, the exchange buffers are put back, the fallback is counted, the next
    aligns stay on the per-launch loop and the single launch is tried again afterwards."""
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    with capi.Context(0) as ref_ctx:
        ref_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ref_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        want = ref_ctx.align(pts, covs, g, 10, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
        good = ref_ctx.align(pts, covs, g, 10, 1e-6, 2.0)
        assert good.launches == 1 and np.array_equal(good.normal_eq, want.normal_eq)
    monkeypatch.setenv("VGICP_SPIN_LIMIT", "0")
    with capi.Context(0) as ctx:
        monkeypatch.delenv("VGICP_SPIN_LIMIT")
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        for k in range(12):
            r = ctx.align(pts, covs, g, 10, 1e-6, 2.0)
            assert r.launches > 1                                   # never the single launch: it cannot complete
            assert np.array_equal(r.normal_eq, want.normal_eq) and np.array_equal(r.pose, want.pose)
        # 12 aligns: the single launch tried at #0 and, after 8 aligns on the loop, again at #9
        assert ctx.counter(0) == 2 and ctx.counter(1) == 2


def test_upload_paths_return_the_same_bits(c1_inputs, monkeypatch):
    """vgicp_align's upload: staged by the copy threads into page-locked memory of the context and read from there by
    ONE pack launch (the default), with one or two threads copying (VGICP_UPLOAD_THREADS), handed to the runtime in
    place (VGICP_OPTION_UPLOAD_STAGE_KB = 0), and buffers page-locked by the caller (vgicp_host_register) — four ways
    to the same resident scan, hence the same bits."""
    from eskf_lio_amd import capi, synth
    vmap, _, _ = c1_inputs
    pts, covs = synth.make_uniform_scan(40_000, vmap, seed=99)          # 960 KB of points: the helper thread takes them
    g = synth.default_guess()
    results = []
    for threads in ("2", "1"):
        monkeypatch.setenv("VGICP_UPLOAD_THREADS", threads)
        with capi.Context(0) as ctx:
            ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
            # a scan goes through page-locked memory of the context by default (the copy crew): once that way ...
            p, c = pts.copy(), covs.copy()
            results.append(ctx.align(p, c, g, 6, 1e-6, 2.0))
            dp, dc = ctx.scan_download()                                  # (back through the same arena)
            assert np.array_equal(dp, pts) and np.array_equal(dc, covs)
            ctx.set_option(capi.OPTION_UPLOAD_STAGE_KB, 0)                # ... then registered in place by the runtime
            for k in range(4):
                p, c = pts.copy(), covs.copy()                            # fresh host buffers every time
                results.append(ctx.align(p, c, g, 6, 1e-6, 2.0))
            if threads == "2":
                p, c = pts.copy(), covs.copy()
                ctx.host_register(p)
                ctx.host_register(c)
                results.append(ctx.align(p, c, g, 6, 1e-6, 2.0))
                ctx.host_unregister(p)
                ctx.host_unregister(c)
                dp, dc = ctx.scan_download()
                assert np.array_equal(dp, pts) and np.array_equal(dc, covs)
    for r in results[1:]:
        assert np.array_equal(r.pose, results[0].pose) and np.array_equal(r.normal_eq, results[0].normal_eq)


def test_staged_upload_arrives_whole_for_every_size_and_thread_count(c1_inputs, monkeypatch):
    """The copy crew + pack_arena_kernel: what lands on the device (AoS copy and, through the align, the SoA planes) is
    the caller's scan bit for bit — odd counts (half a 16-byte chunk at the end of the points), one point short of /
    exactly / one beyond a unit and a block, several units, 1 / 2 / 4 copying threads — and a scan with ONE asymmetric
    covariance is reported as such (the align then reads all twelve planes: same result as the per-launch loop)."""
    from eskf_lio_amd import capi, synth
    from oracle import binding as oracle
    vmap, _, _ = c1_inputs
    omap = oracle.OracleMap(vmap.voxel_size, 1)
    omap.insert(vmap.means, vmap.covs)
    g = synth.default_guess()
    sizes = [2731, 2047, 2048, 2049, 4095, 4097, 21845, 30001, 65536]   # x 96 B: 262 KB (just staged) ... 6.3 MB
    for threads in ("1", "2", "4"):
        monkeypatch.setenv("VGICP_UPLOAD_THREADS", threads)
        with capi.Context(0) as ctx:
            ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
            ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
            for k, n in enumerate(sizes):
                pts, covs = synth.make_uniform_scan(n, vmap, seed=1000 + k)
                if k % 3 == 1:
                    covs = covs.copy()
                    covs[n // 2, 1] += 1e-9          # c10 != c01: not bitwise symmetric
                p, c = pts.copy(), covs.copy()
                got = ctx.align(p, c, g, 3, 1e-6, 2.0)
                p[:] = np.nan                        # the caller's buffers are free on return
                c[:] = np.nan
                dp, dc = ctx.scan_download()
                assert np.array_equal(dp, pts) and np.array_equal(dc, covs), (threads, n)
                want = omap.align(pts, covs, g, 3, 1e-6, 2.0)
                assert np.array_equal(got.corr_count, want.corr_count), (threads, n)
                assert np.abs(got.pose - want.pose).max() < 1e-11, (threads, n)
                again = ctx.align_resident(g, 3, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
                assert np.abs(again.pose - got.pose).max() < 1e-12 and np.array_equal(again.corr_count, got.corr_count)


def test_staged_upload_survives_copy_threads_that_are_held_up(c1_inputs):
    """A pack kernel that has stopped waiting for the host (patience of one poll, a copy thread asleep for 150 ms) leaves
    planes half filled; the upload notices how long its threads took and packs the staged scan again behind it."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, sys
        sys.path.insert(0, %r)
        from eskf_lio_amd import capi, synth
        vmap = synth.make_map(50_000)
        pts, covs = synth.make_uniform_scan(9_000, vmap, seed=5)
        g = synth.default_guess()
        with capi.Context(0) as ctx:
            ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0]); ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
            slow = ctx.align(pts.copy(), covs.copy(), g, 4, 1e-6, 2.0)
            dp, dc = ctx.scan_download()
            assert np.array_equal(dp, pts) and np.array_equal(dc, covs)
            assert ctx.counter(capi.COUNTER_UPLOAD_SLOW) == 1
        from oracle import binding as oracle
        omap = oracle.OracleMap(vmap.voxel_size, 1)
        omap.insert(vmap.means, vmap.covs)
        ref = omap.align(pts, covs, g, 4, 1e-6, 2.0)
        assert np.array_equal(ref.corr_count, slow.corr_count) and np.abs(ref.pose - slow.pose).max() < 1e-11
        # the same for a raw sweep: the preparation's first kernel gives up waiting for its units (and tells the
        # workgroups that look back at its slot), the host launches it once more when everything is staged
        st = synth.make_imu_states(48, seed=9)
        t = synth.make_point_times(9_000, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=9)
        raw = synth.make_lidar_scan(9_000, seed=41)
        ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
        with capi.Context(0) as ctx:
            kept, moved = ctx.scan_prepare(raw, t, st, ext, 0.3, 30)
            gp, gc = ctx.scan_download()
            assert ctx.counter(capi.COUNTER_UPLOAD_SLOW) == 1
        mv, _ = oracle.transform(raw, np.tile(np.eye(3).reshape(9), (len(raw), 1)), ext)
        desk, rdone = oracle.deskew(mv, t, st)
        rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
        assert moved == rdone and kept == len(rp) and np.array_equal(gp, rp) and np.array_equal(gc, rc)
        print("ok")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, VGICP_PACK_SPIN_LIMIT="1", VGICP_DEBUG_UPLOAD_DELAY_US="150000", VGICP_UPLOAD_THREADS="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_resident_scan_is_not_modified_by_align(c1_gpu, c1_inputs):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    p0, c0 = pts.copy(), covs.copy()
    c1_gpu.scan_upload(pts, covs)
    a = c1_gpu.align_resident(synth.default_guess(), 5, 1e-6, 2.0)
    b = c1_gpu.align_resident(synth.default_guess(), 5, 1e-6, 2.0)   # same resident scan again
    assert np.array_equal(a.pose, b.pose) and np.array_equal(pts, p0) and np.array_equal(covs, c0)


# ---- edge cases the reference leaves unguarded ------------------------------------------------
def test_k3_no_correspondences_returns_guess_converged(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    g = synth.default_guess()
    far = pts[:500] + 1.0e4
    ref = c1_oracle_map.align(far, covs[:500], g, 10, 1e-6, 0.9999)
    got = c1_gpu.align(far, covs[:500], g, 10, 1e-6, 0.9999)
    assert got.iterations == ref.iterations == 1 and got.converged and got.corr_count[0] == 0
    assert np.array_equal(got.pose, g)


def test_empty_scan_empty_map_and_zero_iterations(gpu_ctx, c1_inputs):
    from eskf_lio_amd import capi, synth
    vmap, pts, covs = c1_inputs
    g = synth.default_guess()
    with pytest.raises(capi.VgicpError) as e:                      # no map yet
        gpu_ctx.align(pts[:10], covs[:10], g, 5, 1e-6, 0.9999)
    assert e.value.code == capi.ERR_NOT_READY
    gpu_ctx.map_reset(vmap.voxel_size, 0)                          # empty map: zero system
    r = gpu_ctx.align(pts[:100], covs[:100], g, 5, 1e-6, 0.9999)
    assert r.iterations == 1 and r.converged and np.array_equal(r.pose, g)
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    r = gpu_ctx.align(np.zeros((0, 3)), np.zeros((0, 9)), g, 5, 1e-6, 0.9999)   # empty scan
    assert r.iterations == 1 and r.converged and np.array_equal(r.pose, g)
    r = gpu_ctx.align(pts, covs, g, 0, 1e-6, 0.9999)               # max_iteration = 0: loop never runs
    assert r.iterations == 0 and not r.converged and np.array_equal(r.pose, g)
    with pytest.raises(capi.VgicpError) as e:
        gpu_ctx.align(pts, covs, g, -1, 1e-6, 0.9999)
    assert e.value.code == capi.ERR_BAD_ARGUMENT
    with pytest.raises(capi.VgicpError):
        gpu_ctx.map_reset(-1.0, 0)


def test_degenerate_system_is_reported_not_thrown(gpu_ctx, oracle):
    """One correspondence: JTJ has rank 3. The reference is unguarded here (Registration.cpp:78): Eigen's
    pivoted LDLT divides by whatever rounding noise the last three pivots hold. The device's fallback solve
    follows the same operation order without contraction, so on the same normal equations it returns the
    oracle's se3 bit for bit; a non-finite pose comes back as a status code, never as a throw."""
    from eskf_lio_amd import capi
    gpu_ctx.map_reset(0.3, 0)
    mu = np.array([[0.95, 2.05, 3.1]])
    gpu_ctx.map_upsert(np.floor(mu / 0.3).astype(np.int32), mu, np.eye(3).reshape(1, 9))
    p = np.array([[1.0, 2.0, 3.0]])
    om = oracle.OracleMap(0.3, 1)
    om.insert(mu, np.eye(3).reshape(1, 9))
    ref = om.align(p, np.eye(3).reshape(1, 9), np.eye(4), 3, 1e-6, 0.9999)
    r = gpu_ctx.align(p, np.eye(3).reshape(1, 9), np.eye(4), 3, 1e-6, 0.9999, allow_degenerate=True)
    assert r.status in (capi.OK, capi.ERR_DEGENERATE)
    assert (r.status == capi.OK) == bool(np.isfinite(ref.pose).all())
    assert r.iterations == ref.iterations and np.array_equal(r.corr_count, ref.corr_count)
    # (the poses themselves need not agree: the last three pivots are rounding noise of sums that the device
    #  and the CPU add in different orders, and the solve amplifies that noise along the null space)
    # the first round's system through the solve hook: the fallback is taken and equals the oracle's bits
    JTJ, JTr, cnt = gpu_ctx.accumulate(p, np.eye(3).reshape(1, 9), np.eye(4))
    assert cnt == 1
    se3, step, pivoted, _ = gpu_ctx.solve_step(JTJ, JTr)
    ose3, ostep = oracle.solve_step(JTJ, JTr)
    assert pivoted and np.array_equal(se3, ose3, equal_nan=True)
    if np.isfinite(ose3).all():
        assert np.allclose(step, ostep, rtol=1e-12, atol=1e-15)


def _random_spd6(rng, cond=1e3):
    Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
    A = (Q * np.geomspace(1.0, cond, 6)) @ Q.T
    return 0.5 * (A + A.T)


def test_solve_step_matches_oracle(c1_gpu, c1_inputs, oracle):
    """The device's solve / exponential / convergence test on given normal equations against
    oracle_solve_step (reference src/Registration.cpp:37-50,78-79, src/Utils.cpp:40-63)."""
    from eskf_lio_amd import synth
    rng = np.random.default_rng(77)
    ctx = c1_gpu
    vmap, pts, covs = c1_inputs
    # (1) a real, well-conditioned system: fast path, agreement to rounding
    JTJ, JTr, _ = ctx.accumulate(pts, covs, synth.default_guess())
    se3, step, pivoted, _ = ctx.solve_step(JTJ, JTr)
    ose3, ostep = oracle.solve_step(JTJ, JTr)
    assert not pivoted
    assert np.allclose(se3, ose3, rtol=1e-10, atol=1e-16) and np.allclose(step, ostep, rtol=1e-12, atol=1e-15)
    # the same system through the pivoted solve: the oracle's bits
    se3p, _, pivoted, _ = ctx.solve_step(JTJ, JTr, force_pivoted=True)
    assert pivoted and np.array_equal(se3p, ose3)
    # (2) random SPD systems, both paths
    for _ in range(20):
        A, b = _random_spd6(rng, 10.0 ** rng.uniform(0, 8)), rng.normal(size=6)
        ose3, ostep = oracle.solve_step(A, b)
        se3, step, pivoted, _ = ctx.solve_step(A, b)
        assert not pivoted and np.allclose(se3, ose3, rtol=1e-7, atol=1e-14)
        se3p, stepp, _, _ = ctx.solve_step(A, b, force_pivoted=True)
        assert np.array_equal(se3p, ose3) and np.allclose(stepp, ostep, rtol=1e-12, atol=1e-15)
    # (3) indefinite and rank-deficient systems: never the fast path, always the oracle's bits
    for trial in range(20):
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        ev = np.abs(rng.normal(size=6)) * 10.0 ** rng.uniform(-3, 3, size=6)
        ev[rng.integers(0, 6)] *= -1.0                                # one negative eigenvalue
        A = (Q * ev) @ Q.T
        A = 0.5 * (A + A.T)
        b = rng.normal(size=6)
        se3, step, pivoted, _ = ctx.solve_step(A, b)
        ose3, ostep = oracle.solve_step(A, b)
        assert pivoted and np.array_equal(se3, ose3)
    B = rng.integers(-3, 4, size=(6, 3)).astype(np.float64)           # rank 3, noise pivots afterwards
    A = B @ B.T
    b = -(A @ rng.normal(size=6))
    se3, _, pivoted, _ = ctx.solve_step(A, b)
    ose3, _ = oracle.solve_step(A, b)
    assert pivoted and np.array_equal(se3, ose3, equal_nan=True)
    # exactly singular in exact arithmetic: zero pivots are skipped, D is pseudo-inverted
    A = np.zeros((6, 6))
    A[:2, :2] = [[4.0, 2.0], [2.0, 1.0]]
    A[2, 2], A[5, 5] = 2.0, 1.0
    b = np.array([-4.0, -2.0, -6.0, 5.0, 7.0, -3.0])
    se3, _, pivoted, _ = ctx.solve_step(A, b)
    ose3, _ = oracle.solve_step(A, b)
    assert pivoted and np.array_equal(se3, ose3) and np.isfinite(se3).all() and se3[3] == 0.0 and se3[4] == 0.0
    # (4) the all-zero system of "no correspondences" (K3): zero step, identity, converged
    se3, step, pivoted, conv = ctx.solve_step(np.zeros((6, 6)), np.zeros(6))
    assert pivoted and conv and np.array_equal(se3, np.zeros(6)) and np.array_equal(step, np.eye(4))
    # (5) se3ToSE3 edge cases (K4) through the device: JTJ = I, JTr = -xi solves to xi exactly
    for xi in ([0.1, -0.2, 0.3, 0.0, 0.0, 0.0],                       # phi = 0: R = I, t = rho
               [0.1, -0.2, 0.3, 6e-8, 0.0, 8e-8],                     # |phi| = 1e-7 < 1e-6: J_l = I
               [0.1, -0.2, 0.3, 0.0, 0.0, np.pi / 2],                 # 90 degrees about z
               [0.0, 0.0, 0.0, 1e-6, 0.0, 0.0],                       # exactly at the 1e-6 branch
               [1.0, 2.0, 3.0, 0.3, -0.4, 1.2]):
        xi = np.array(xi)
        se3, step, pivoted, _ = ctx.solve_step(np.eye(6), -xi)
        assert not pivoted and np.array_equal(se3, xi)
        ostep = oracle.se3_to_SE3(xi)
        assert np.allclose(step, ostep, rtol=0, atol=4e-16 * max(1.0, np.abs(xi[:3]).max()))
        if np.linalg.norm(xi[3:]) < 1e-6:
            assert np.array_equal(step[:3, 3], xi[:3])
    # (6) convergenceCheck (K5): equality with a threshold counts as converged, one ulp beyond does not.
    # A step along x only: |t|^2 = tx * tx whatever the device contracts.
    xi = np.array([3e-4, 0.0, 0.0, 0.0, 0.0, 0.0])
    _, step, _, _ = ctx.solve_step(np.eye(6), -xi)
    assert step[1, 3] == 0.0 and step[2, 3] == 0.0
    cos = 0.5 * (((step[0, 0] + step[1, 1]) + step[2, 2]) - 1.0)
    t2 = float(step[0, 3]) * float(step[0, 3])
    for cthr, tthr in ((cos, t2), (np.nextafter(cos, 2.0), t2), (cos, np.nextafter(t2, 0.0)),
                       (np.nextafter(cos, 0.0), np.nextafter(t2, 1.0)), (0.9999, 1e-6), (2.0, 1e-6)):
        _, _, _, conv = ctx.solve_step(np.eye(6), -xi, cosine_threshold=cthr, translation_sq_threshold=tthr)
        assert conv == oracle.convergence_check(step, cthr, tthr), (cthr, tthr)
    assert ctx.solve_step(np.eye(6), -xi, cosine_threshold=cos, translation_sq_threshold=t2)[3]
    assert not ctx.solve_step(np.eye(6), -xi, cosine_threshold=cos, translation_sq_threshold=np.nextafter(t2, 0.0))[3]
    # a rotation: equality with the cosine threshold converges, one ulp above does not
    xi = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 2e-3])
    _, step, _, _ = ctx.solve_step(np.eye(6), -xi)
    cos = 0.5 * (((step[0, 0] + step[1, 1]) + step[2, 2]) - 1.0)
    assert ctx.solve_step(np.eye(6), -xi, cosine_threshold=cos, translation_sq_threshold=1e-6)[3]
    assert not ctx.solve_step(np.eye(6), -xi, cosine_threshold=np.nextafter(cos, 2.0), translation_sq_threshold=1e-6)[3]


def test_ragged_scan_sizes(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    g = synth.default_guess()
    for n in (1, 2, 63, 64, 65, 511, 512, 513, 1025, 4999):
        ref = c1_oracle_map.align(pts[:n], covs[:n], g, 4, 1e-6, 2.0)
        got = c1_gpu.align(pts[:n], covs[:n], g, 4, 1e-6, 2.0, allow_degenerate=True)
        assert np.array_equal(got.corr_count, ref.corr_count)
        if ref.corr_count.min() >= 50:                             # well-posed: compare poses too
            dt, dr = pose_error(got.pose, ref.pose)
            assert dt <= 1e-8 and dr <= 1e-8


# ---- through the C++ host mirror (the reference's interface) ----------------------------------
def test_host_mirror_localmap_and_icp(c1_inputs, oracle):
    from eskf_lio_amd import host, synth
    vmap, pts, covs = c1_inputs
    rng = np.random.default_rng(21)
    # a map built by the reference's insertion rule: several scans, several points per voxel
    scans = [(vmap.means[rng.choice(50_000, 8_000, replace=False)] + rng.normal(scale=0.02, size=(8_000, 3)),
              covs[rng.choice(5_000, 8_000)]) for _ in range(3)]
    poses = [np.eye(4), synth.se3_to_SE3([0.2, 0.1, 0.0, 0.0, 0.0, 0.02]), synth.se3_to_SE3([0.4, 0.2, 0.0, 0.0, 0.0, 0.04])]
    lmap = host.LocalMap(0.3, 20)
    omap = oracle.OracleMap(0.3, 20)
    for (p, c), T in zip(scans, poses):
        world_p, world_c = lmap.updateLocalMap(p, c, T)            # cloud->Transform(T) then insert
        op, oc = oracle.transform(p, c, T)
        assert np.array_equal(world_p, op) and np.array_equal(world_c, oc)
        omap.insert(op, oc)
    assert len(lmap) == len(omap)
    hk, hm, hc, hn = lmap.export()
    ok, om_, oc_, on = omap.export()
    order_h, order_o = np.lexsort(hk.T), np.lexsort(ok.T)
    assert np.array_equal(hk[order_h], ok[order_o]) and np.array_equal(hn[order_h], on[order_o])
    assert np.array_equal(hm[order_h], om_[order_o]) and np.array_equal(hc[order_h], oc_[order_o])
    assert hn.max() > 1                                            # the running mean was exercised
    # correspondenceMatching: the reference's tuple, device-served
    tp, tc = oracle.transform(pts, covs, synth.default_guess())
    got = lmap.correspondenceMatching(tp, tc)
    ref = omap.match(tp, tc)
    assert all(np.array_equal(a, b) for a, b in zip(got, ref[:4]))
    # ICP::align through the C++ class
    icp = host.ICP(20, 1e-6, 2.0)
    T = icp.align(pts, covs, lmap, synth.default_guess())
    ref = omap.align(pts, covs, synth.default_guess(), 20, 1e-6, 2.0)
    assert icp.iterations == 20 and not icp.converged
    assert np.array_equal(icp.correspondence_counts, ref.corr_count)
    dt, dr = pose_error(T, ref.pose)
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL


def test_host_mirror_device_resident_map(c1_inputs, oracle, tmp_path):
    """LocalMapConfig::deviceResident: updateLocalMap inserts and evicts on the device; the voxels it
    leaves are the reference's, bit for bit, and ICP::align reads them as usual."""
    from eskf_lio_amd import host, synth
    vmap, pts, covs = c1_inputs
    rng = np.random.default_rng(33)
    cfg = dict(translation_sq_threshold=-1.0, cosine_threshold=2.0, remove_distant_points=False,
               distance_threshold=1e9, removing_period=1e9, device_resident=True)
    lmap = host.LocalMap(0.3, 20, cfg)
    omap = oracle.OracleMap(0.3, 20)
    for T in (np.eye(4), synth.se3_to_SE3([0.2, 0.1, 0.0, 0.0, 0.0, 0.02]), synth.se3_to_SE3([0.4, 0.2, 0.0, 0.0, 0.0, 0.04])):
        p = vmap.means[rng.choice(50_000, 8_000)] + rng.normal(scale=0.02, size=(8_000, 3))
        c = covs[rng.choice(5_000, 8_000)]
        wp, wc = lmap.updateLocalMap(p, c, T)
        op, oc = oracle.transform(p, c, T)
        assert np.array_equal(wp, op) and np.array_equal(wc, oc)
        omap.insert(op, oc)
        assert len(lmap) == len(omap)
    hk, hm, hc, hn = lmap.export()
    ok, om_, oc_, on = omap.export()
    oh, oo = np.lexsort(hk.T), np.lexsort(ok.T)
    assert np.array_equal(hk[oh], ok[oo]) and np.array_equal(hn[oh], on[oo])
    assert np.array_equal(hm[oh], om_[oo]) and np.array_equal(hc[oh], oc_[oo])
    icp = host.ICP(20, 1e-6, 2.0)
    T = icp.align(pts, covs, lmap, synth.default_guess())
    ref = omap.align(pts, covs, synth.default_guess(), 20, 1e-6, 2.0)
    assert np.array_equal(icp.correspondence_counts, ref.corr_count)
    dt, dr = pose_error(T, ref.pose)
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL
    # save(): with the shadow grid (keep_raw_points, the default) every stored raw point, exactly what the
    # host-authoritative map writes from the same updates (src/LocalMap.cpp:156-167) ...
    lmap.save(str(tmp_path / "m.pcd"), str(tmp_path / "t.json"))
    hmap = host.LocalMap(0.3, 20, dict(cfg, device_resident=False))
    bare = host.LocalMap(0.3, 20, dict(cfg, keep_raw_points=False))
    rng = np.random.default_rng(33)
    for T in (np.eye(4), synth.se3_to_SE3([0.2, 0.1, 0.0, 0.0, 0.0, 0.02]), synth.se3_to_SE3([0.4, 0.2, 0.0, 0.0, 0.0, 0.04])):
        p = vmap.means[rng.choice(50_000, 8_000)] + rng.normal(scale=0.02, size=(8_000, 3))
        c = covs[rng.choice(5_000, 8_000)]
        hmap.updateLocalMap(p, c, T)
        bare.updateLocalMap(p, c, T)
    hmap.save(str(tmp_path / "h.pcd"), str(tmp_path / "ht.json"))
    got, want = open(tmp_path / "m.pcd").read().splitlines(), open(tmp_path / "h.pcd").read().splitlines()
    assert got[:11] == want[:11] and sorted(got[11:]) == sorted(want[11:]) and len(got) - 11 > len(omap)
    assert open(tmp_path / "t.json").read() == open(tmp_path / "ht.json").read()
    # ... and one point per voxel (its mean) when the raw points are not kept
    bare.save(str(tmp_path / "b.pcd"), str(tmp_path / "bt.json"))
    assert f"POINTS {len(omap)}" in open(tmp_path / "b.pcd").read()


def test_host_mirror_motion_gate_and_eviction(oracle, tmp_path):
    from eskf_lio_amd import host, synth
    cfg = dict(translation_sq_threshold=1e-2, cosine_threshold=0.985, remove_distant_points=True,
               distance_threshold=5.0, removing_period=0.0, device_resident=False)
    lmap = host.LocalMap(0.3, 1000, cfg)
    rng = np.random.default_rng(2)
    near = rng.uniform(-2, 2, size=(500, 3))
    cov = np.tile(np.eye(3).reshape(1, 9), (500, 1))
    lmap.updateLocalMap(near, cov, np.eye(4))                      # first frame always inserts
    n0 = len(lmap)
    assert n0 > 100
    lmap.updateLocalMap(rng.uniform(-2, 2, size=(500, 3)), cov, synth.se3_to_SE3([0.01, 0, 0, 0, 0, 0]))
    assert len(lmap) == n0                                         # moved 1 cm: gated (LocalMap.cpp:39-42)
    T_far = synth.se3_to_SE3([20.0, 0, 0, 0, 0, 0])
    lmap.updateLocalMap(rng.uniform(-2, 2, size=(500, 3)), cov, T_far)   # inserts around x=20, evicts the old
    keys, means, _, _ = lmap.export()
    centre = (keys + 0.5) * 0.3
    assert (np.linalg.norm(centre - T_far[:3, 3], axis=1) <= 5.0).all() and len(lmap) > 100
    # the device mirror followed: old voxels are gone, new ones are there
    sp, *_ = lmap.correspondenceMatching(near, cov)
    assert len(sp) == 0
    sp, _, mp, _ = lmap.correspondenceMatching(means, np.tile(np.eye(3).reshape(1, 9), (len(means), 1)))
    assert len(sp) == len(means) and np.array_equal(mp, means)
    lmap.save(str(tmp_path / "map.pcd"), str(tmp_path / "traj.json"))
    assert "POINTS" in open(tmp_path / "map.pcd").read() and "extrinsic" in open(tmp_path / "traj.json").read()


# ---- LocalMap::updateLocalMap's insert / evict loops on the device ---------------
def _sorted_oracle_export(om):
    k, m, c, n = om.export()
    order = np.lexsort(k.T)
    return k[order], m[order], c[order], n[order]


@pytest.mark.parametrize("cap", [1, 3, 1000])
def test_device_map_insertion_is_bit_exact(gpu_ctx, c1_inputs, oracle, cap):
    """Several scans, several points per voxel, moved by different poses: the device insertion leaves
    the same voxels, means, covariances and counts as the reference's serial loop — compared with ==."""
    from eskf_lio_amd import synth
    vmap, pts, covs = c1_inputs
    rng = np.random.default_rng(100 + cap)
    poses = [np.eye(4), synth.se3_to_SE3([0.2, 0.1, 0.0, 0.0, 0.0, 0.02]), synth.se3_to_SE3([0.4, -0.2, 0.1, 0.01, 0.0, 0.04]),
             synth.se3_to_SE3([-0.3, 0.2, 0.0, 0.0, -0.02, 0.1])]
    gpu_ctx.map_reset(0.3, 0)                                      # growth + rehash on the way (keeps counts)
    om = oracle.OracleMap(0.3, cap)
    total_new = 0
    for T in poses:
        sel = rng.choice(50_000, 12_000, replace=True)             # repeats: many points per voxel
        p = vmap.means[sel] + rng.normal(scale=0.05, size=(12_000, 3))
        c = covs[rng.choice(5_000, 12_000)]
        before = gpu_ctx.map_size()[0]
        new = gpu_ctx.map_insert_scan(p, c, T, cap)
        total_new += new
        assert gpu_ctx.map_size()[0] == before + new
        wp, wc = oracle.transform(p, c, T)                         # cloud->Transform(T), then the loop
        om.insert(wp, wc)
        assert gpu_ctx.map_size()[0] == len(om)
    gk, gm, gc, gn = gpu_ctx.map_export()
    ok, om_, oc, on = _sorted_oracle_export(om)
    assert total_new == len(om) and np.array_equal(gk, ok)
    assert np.array_equal(gn, on) and on.max() == min(cap, on.max()) and (cap == 1 or on.max() > 1)
    assert np.array_equal(gm, om_) and np.array_equal(gc, oc)      # bit for bit
    # the registration reads the device-built map like a host-built one
    g = synth.default_guess()
    got = gpu_ctx.align(pts, covs, g, 10, 1e-6, 2.0)
    ref = om.align(pts, covs, g, 10, 1e-6, 2.0)
    assert_align_parity(got, ref)


def test_device_map_insert_edge_cases(gpu_ctx, oracle):
    gpu_ctx.map_reset(0.3, 0)
    assert gpu_ctx.map_insert_scan(np.zeros((0, 3)), np.zeros((0, 9)), np.eye(4), 5) == 0
    # every point in ONE voxel (longest possible segment, all claims race for one slot)
    rng = np.random.default_rng(4)
    p = 0.15 + 0.1 * rng.random((5_000, 3))
    c = np.tile(np.eye(3).reshape(1, 9), (5_000, 1)) * rng.random((5_000, 1))
    om = oracle.OracleMap(0.3, 1000)
    om.insert(p, c)
    assert gpu_ctx.map_insert_scan(p, c, np.eye(4), 1000) == 1
    gk, gm, gc, gn = gpu_ctx.map_export()
    ok, om_, oc, on = _sorted_oracle_export(om)
    assert gn[0] == on[0] == 1000 and np.array_equal(gk, ok)
    assert np.array_equal(gm, om_) and np.array_equal(gc, oc)
    # all points in distinct voxels, negative coordinates and exact cell faces included
    grid = np.stack(np.meshgrid(np.arange(-20, 20), np.arange(-20, 20), np.arange(-2, 2), indexing="ij"), -1).reshape(-1, 3)
    p = grid * 0.3                                                  # on the faces
    c = np.tile(np.eye(3).reshape(1, 9), (len(p), 1))
    gpu_ctx.map_reset(0.3, 0)
    om = oracle.OracleMap(0.3, 20)
    om.insert(p, c)
    assert gpu_ctx.map_insert_scan(p, c, np.eye(4), 20) == len(om)
    assert np.array_equal(gpu_ctx.map_export()[0], _sorted_oracle_export(om)[0])


def test_device_map_eviction_matches_the_reference_rule(gpu_ctx, c1_inputs):
    vmap, _, _ = c1_inputs
    gpu_ctx.map_reset(vmap.voxel_size, 0)
    gpu_ctx.map_insert_scan(vmap.means, vmap.covs, np.eye(4), 10)
    pos = np.array([1.3, -0.7, 0.4])
    centre = (vmap.keys.astype(np.float64) + 0.5) * vmap.voxel_size
    d = np.sqrt(((centre - pos) ** 2).sum(axis=1))
    for thr in (6.0, 3.0):
        far = d > thr
        expect_removed = int(far.sum()) - (50_000 - gpu_ctx.map_size()[0])
        assert gpu_ctx.map_evict(pos, thr) == expect_removed
        keys = gpu_ctx.map_export()[0]
        want = vmap.keys[~far]
        assert np.array_equal(keys, want[np.lexsort(want.T)])
    # evicted voxels can come back (tombstones are skipped, not reused, until the next rehash)
    back = gpu_ctx.map_insert_scan(vmap.means, vmap.covs, np.eye(4), 10)
    assert back == int((d > 3.0).sum()) and gpu_ctx.map_size()[0] == 50_000
    assert np.array_equal(gpu_ctx.map_export()[0], vmap.keys[np.lexsort(vmap.keys.T)])


# ---- BASELINE config C4's stand-in: a synthetic frame stream through register -> insert ----------
def test_frame_stream_trajectory_matches_the_cpu_loop(gpu_ctx, oracle):
    """The per-frame sequence of src/Odometry.cpp:79,86 without the filter: align the scan against the
    map with the previous pose as the guess, then insert it with the pose found.  The HILTI bag is not
    available, so a sensor is moved through a synthetic world; the GPU loop and the
    CPU loop must produce the same trajectory, the same counts and — frame after frame — the same map."""
    from eskf_lio_amd import synth
    world = synth.make_map(60_000, seed=77)                        # the world the scans see
    rng = np.random.default_rng(9)
    cap, frames, n = 20, 8, 6_000
    truth = [synth.se3_to_SE3([0.12 * f, 0.05 * f, 0.01 * f, 0.0, 0.002 * f, 0.01 * f]) for f in range(frames)]
    gpu_ctx.map_reset(0.3, 0)
    om = oracle.OracleMap(0.3, cap)
    pose_g = pose_c = truth[0]
    for f in range(frames):
        pick = rng.choice(60_000, n, replace=False)
        Tinv = synth.invert_pose(truth[f])
        scan = (world.means[pick] + rng.normal(scale=0.01, size=(n, 3))) @ Tinv[:3, :3].T + Tinv[:3, 3]
        cov = synth._conjugate(Tinv[:3, :3], world.covs[pick])
        if f > 0:
            got = gpu_ctx.align(scan, cov, pose_g, 30, 1e-6, 0.9999)
            ref = om.align(scan, cov, pose_c, 30, 1e-6, 0.9999)
            assert got.iterations == ref.iterations and got.converged == ref.converged
            assert np.array_equal(got.corr_count, ref.corr_count)
            dt, dr = pose_error(got.pose, ref.pose)
            assert dt <= 1e-9 and dr <= 1e-9
            et, er = pose_error(got.pose, truth[f])
            assert et < 5e-3 and er < 2e-3                         # it is actually tracking
            pose_g, pose_c = got.pose, ref.pose
            gpu_ctx.map_insert_resident(pose_g, cap)               # the scan align just registered
        else:
            gpu_ctx.map_insert_scan(scan, cov, pose_g, cap)
        wp, wc = oracle.transform(scan, cov, pose_c)
        om.insert(wp, wc)
        assert gpu_ctx.map_size()[0] == len(om)
    gk, gm, gc, gn = gpu_ctx.map_export()
    k, m, c, cnt = om.export()
    o = np.lexsort(k.T)
    assert np.array_equal(gk, k[o]) and np.array_equal(gn, cnt[o])
    assert np.abs(gm - m[o]).max() < 1e-9 and np.abs(gc - c[o]).max() < 1e-9   # poses agree to 1e-9, so do the maps


def test_async_frame_chain_has_one_sync_per_frame_and_the_same_bits(oracle):
    """The frame sequence of src/Odometry.cpp:73-87 without host round trips — vgicp_scan_prepare_async (tables sized
    from the raw count, kept count left on the device) -> vgicp_align_resident (reads the size from the device; its
    synchronisation is the frame's only one) -> vgicp_map_insert_resident_async (counts read at the next frame's
    synchronisation) — against the same frames through the synchronous calls on a second context: identical poses,
    counts, prepared scans and maps, bit for bit; ONE host synchronisation per frame (vgicp_get_frame_stats)."""
    from eskf_lio_amd import capi, synth
    frames, n, cap = 6, 9_000, 20
    st = synth.make_imu_states(48, seed=21)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=21, jitter=1e-3)
    ext = synth.se3_to_SE3([0.02, -0.01, 0.03, 0.01, -0.02, 0.005])
    with capi.Context(0) as a, capi.Context(0) as b:
        for c in (a, b):
            c.map_reset(0.3, 400_000)                              # no table growth (a rehash synchronises) in the loop
        pose = np.eye(4)
        a.frame_stats(reset=True)
        syncs = []
        for f in range(frames):
            raw = synth.make_lidar_scan(n, seed=500 + f, extent=25.0)
            kept_b, moved_b = b.scan_prepare(raw, t, st, ext, 0.3, 30)
            if f == 0:
                a.scan_prepare_async(raw, t, st, ext, 0.3, 30)
                pa, ca = a.scan_download()                         # (settles the pending scan)
                pb, cb = b.scan_download()
                assert np.array_equal(pa, pb) and np.array_equal(ca, cb)
                moved, _ = oracle.transform(raw, np.tile(np.eye(3).reshape(9), (n, 1)), ext)   # Open3D Transform
                rp, rc, _ = oracle.preprocess(oracle.deskew(moved, t, st)[0], 0.3, 30)
                assert np.array_equal(pa, rp) and np.array_equal(ca, rc)
                a.map_insert_resident_async(pose, cap)
                b.map_insert_resident(pose, cap)
                a.frame_stats(reset=True)
                continue
            guess = pose @ synth.se3_to_SE3([0.01, 0.0, 0.0, 0.0, 0.0, 0.002])
            rb = b.align_resident(guess, 12, 1e-6, 0.9999)
            b.map_insert_resident(rb.pose, cap)
            # context a's frame on its own (the statistics count what this host thread does, whatever the context)
            a.frame_stats(reset=True)
            a.scan_prepare_async(raw, t, st, ext, 0.3, 30)
            ra = a.align_resident(guess, 12, 1e-6, 0.9999)
            a.map_insert_resident_async(ra.pose, cap)
            fs = a.frame_stats()
            syncs.append(fs.host_syncs)
            assert fs.kernel_launches > 0
            assert ra.launches == 1 and ra.iterations == rb.iterations
            assert np.array_equal(ra.pose, rb.pose) and np.array_equal(ra.normal_eq, rb.normal_eq)
            assert a.scan_info() == (kept_b, moved_b, b.counter(4))
            pose = ra.pose
        assert syncs == [1] * (frames - 1), syncs
        assert a.map_size() == b.map_size()
        for x, y in zip(a.map_export(), b.map_export()):
            assert np.array_equal(x, y)
    # a refused scan (a point beyond the search grid) fails the align that would have used it, not the enqueue
    with capi.Context(0) as c:
        c.map_reset(0.3, 1000)
        bad = synth.make_lidar_scan(2_000, seed=3)
        bad[7, 0] = 0.3 * (2 ** 17) + 5.0
        c.scan_prepare_async(bad, None, None, None, 0.3, 30)
        with pytest.raises(capi.VgicpError) as e:
            c.align_resident(np.eye(4), 5, 1e-6, 0.9999)
        assert e.value.code == capi.ERR_BAD_ARGUMENT and "search grid" in str(e.value)
        with pytest.raises(capi.VgicpError):
            c.align_resident(np.eye(4), 5, 1e-6, 0.9999)           # no scan resident any more
        good = synth.make_lidar_scan(2_000, seed=3)
        c.scan_prepare_async(good, None, None, None, 0.3, 30)
        assert c.scan_info()[0] == len(oracle.preprocess(good, 0.3, 30)[2])


def test_sweep_staged_on_arrival_prepares_to_the_same_bits(oracle):
    """vgicp_sweep_stage + vgicp_scan_prepare_staged_async: a sweep handed over when it arrives (plain CPU copy into
    page-locked memory, from ANOTHER thread while the owner thread aligns) and prepared later from its ticket gives the
    prepared scan, pose and map of vgicp_scan_prepare_async bit for bit; a ticket works once; at most three sweeps
    are staged ahead; a sweep staged without capture times cannot be deskewed."""
    import threading
    from eskf_lio_amd import capi, synth
    st = synth.make_imu_states(48, seed=5)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    sweeps = [synth.make_lidar_scan(n, seed=40 + k) for k, n in enumerate((30_001, 8_000, 52_345))]
    times = [synth.make_point_times(len(s), st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=7 + k) for k, s in enumerate(sweeps)]

    def chain(ctx, staged):
        ctx.map_reset(0.3, 200_000)
        out = []
        pose = np.eye(4)
        for k, (sw, tt) in enumerate(zip(sweeps, times)):
            if staged:
                # the copy runs on another thread while this one is inside the library (here: a map query)
                box = {}
                th = threading.Thread(target=lambda: box.setdefault("t", ctx.sweep_stage(sw, tt)))
                th.start()
                ctx.map_size()
                th.join()
                ctx.scan_prepare_staged_async(box["t"], st, ext, 0.3, 30)
                with pytest.raises(capi.VgicpError):
                    ctx.scan_prepare_staged_async(box["t"], st, ext, 0.3, 30)      # a ticket is used once
                ctx.scan_prepare_staged_async(ctx.sweep_stage(sw, tt), st, ext, 0.3, 30)   # (the same sweep again: fine)
            else:
                ctx.scan_prepare_async(sw, tt, st, ext, 0.3, 30)
            if k:
                r = ctx.align_resident(pose, 10, 1e-6, 0.9999)
                pose = r.pose
                out.append((r.pose, r.normal_eq))
            out.append(ctx.scan_download())
            ctx.map_insert_resident_async(pose, 20)
        out.append(ctx.map_export())
        return out
    with capi.Context(0) as a, capi.Context(0) as b:
        ra, rb = chain(a, False), chain(b, True)
        for x, y in zip(ra, rb):
            for u, v in zip(x, y):
                assert np.array_equal(u, v)
        # against the oracle chain: the prepared scan of the first sweep
        moved, _ = oracle.transform(sweeps[0], np.tile(np.eye(3).reshape(9), (len(sweeps[0]), 1)), ext)
        desk, _ = oracle.deskew(moved, times[0], st)
        rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
        assert np.array_equal(rb[0][0], rp) and np.array_equal(rb[0][1], rc)
        # three sweeps ahead at most; the fourth is refused until one has been prepared
        tickets = [b.sweep_stage(sweeps[1], times[1]) for _ in range(3)]
        with pytest.raises(capi.VgicpError) as e:
            b.sweep_stage(sweeps[1], times[1])
        assert e.value.code == capi.ERR_NOT_READY
        assert "three sweeps" in b.last_error()
        # a staging call made on ANOTHER thread keeps its failure text with that thread: the owner's text is untouched
        import threading
        seen = {}
        def other_thread():
            try:
                b.sweep_stage(sweeps[1], times[1])
            except capi.VgicpError as err:
                seen["code"], seen["text"] = err.code, b.last_error()
        with pytest.raises(capi.VgicpError):
            b.scan_prepare_staged_async(10 ** 9, st, ext, 0.3, 30)                # an unknown ticket: the owner's failure
        owner_text = b.last_error()
        th = threading.Thread(target=other_thread)
        th.start()
        th.join()
        assert seen["code"] == capi.ERR_NOT_READY and "three sweeps" in seen["text"]
        assert b.last_error() == owner_text and "three sweeps" not in owner_text
        b.scan_prepare_staged_async(tickets[0], st, ext, 0.3, 30)
        assert b.scan_info()[0] == len(rb[2][0])                                  # the 8 000-point sweep's kept count
        b.sweep_stage(sweeps[1], times[1])                                        # a slot is free again (its readers are through)
        # without capture times: no deskew from that ticket, but a plain preparation
        t_plain = a.sweep_stage(sweeps[1])
        with pytest.raises(capi.VgicpError):
            a.scan_prepare_staged_async(t_plain, st, ext, 0.3, 30)
        t_plain = a.sweep_stage(sweeps[1])
        a.scan_prepare_staged_async(t_plain, None, ext, 0.3, 30)
        assert a.scan_info()[0] > 0
    # a staged sweep that is DROPPED unprepared (vgicp_sweep_unstage, ABI 6): its slot is free again at once, its ticket
    # is void; before that call three forgotten tickets took staging away from the context for good
    with capi.Context(0) as c:
        c.map_reset(0.3, 0)
        tickets = [c.sweep_stage(sweeps[1], times[1]) for _ in range(3)]
        with pytest.raises(capi.VgicpError):
            c.sweep_stage(sweeps[1], times[1])
        c.sweep_unstage(tickets[1])
        with pytest.raises(capi.VgicpError):
            c.sweep_unstage(tickets[1])                                           # used once
        with pytest.raises(capi.VgicpError):
            c.scan_prepare_staged_async(tickets[1], st, ext, 0.3, 30)             # ... and void for the preparation too
        t_new = c.sweep_stage(sweeps[0], times[0])                                # the freed slot
        c.scan_prepare_staged_async(t_new, st, ext, 0.3, 30)
        gp, gc = c.scan_download()
        assert np.array_equal(gp, rb[0][0]) and np.array_equal(gc, rb[0][1])
        for tk in (tickets[0], tickets[2]):
            c.sweep_unstage(tk)
        assert len([c.sweep_stage(sweeps[1], times[1]) for _ in range(2)]) == 2


def test_the_preparations_sort_equals_a_stable_sort():
    """eskf_lio_amd/csrc/vgicp_sort.h (tile sort in LDS + whole groups of runs merged per launch) against std::stable_sort
    over (key, index) pairs: sizes around every boundary of the plan, voxel-code-like keys with long runs of equal values,
    all-equal / sorted / reversed / random 63-bit keys, nothing written past the end (tests/native/sort_check.hip, built
    by `make sort_check`)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "eskf_lio_amd", "lib", "sort_check")
    assert os.path.exists(exe), "eskf_lio_amd/lib/sort_check is missing: run __graft_entry__.build() (make -C eskf_lio_amd/csrc sort_check)"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_scan_fetch_returns_the_prepared_scan_without_a_copy_command(oracle):
    """vgicp_scan_fetch_begin / _end (ABI 6): the host copy of an ENQUEUED preparation, written into page-locked memory by a
    kernel behind it and copied out piece by piece — equal to vgicp_scan_info + vgicp_scan_download bit for bit (and so to
    the oracle), for sizes below / at / beyond a 64 KB piece boundary, with and without the deskew, after a refused scan,
    and the context is settled afterwards (align, insertion and a second frame work as after vgicp_scan_info)."""
    from eskf_lio_amd import capi, synth
    st = synth.make_imu_states(48, seed=5)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    with capi.Context(0) as a, capi.Context(0) as b:
        for ctx in (a, b):
            ctx.map_reset(0.3, 100_000)
        for k, n in enumerate((700, 2_731, 9_000, 30_001, 61_234)):
            raw = synth.make_lidar_scan(n, seed=60 + k)
            tt = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=11 + k)
            with_states = st if k % 2 == 0 else None
            a.scan_prepare_async(raw, tt if with_states is not None else None, with_states, ext, 0.3, 30)
            fp, fc = a.scan_fetch()
            b.scan_prepare_async(raw, tt if with_states is not None else None, with_states, ext, 0.3, 30)
            kept = b.scan_info()[0]
            dp, dc = b.scan_download()
            assert kept == len(fp) == len(dp) and np.array_equal(fp, dp) and np.array_equal(fc, dc), n
            # vgicp_scan_fetch_sums: the checksums the kernel made while writing equal the ones of the delivered bytes
            sums = a.scan_fetch_sums()
            for arr, buf in enumerate((fp, fc)):
                w = np.ascontiguousarray(buf).view(np.uint64).reshape(-1)
                idx = np.arange(len(w), dtype=np.uint64)
                for lane in range(16):
                    wl = w[lane::16]
                    m = np.uint64(len(wl))
                    with np.errstate(over="ignore"):
                        assert sums[arr, 0, lane] == wl.sum(dtype=np.uint64), (n, arr, lane)
                        assert sums[arr, 1, lane] == ((m - (idx[lane::16] >> np.uint64(4))) * wl).sum(dtype=np.uint64), (n, arr, lane)
            if k == 2:   # against the oracle chain once
                moved, _ = oracle.transform(raw, np.tile(np.eye(3).reshape(9), (n, 1)), ext)
                desk, _ = oracle.deskew(moved, tt, st)
                rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
                assert np.array_equal(fp, rp) and np.array_equal(fc, rc)
            # the context is up to date: the resident chain goes on from here
            for ctx in (a, b):
                if k:
                    ctx.align_resident(np.eye(4), 5, 1e-6, 0.9999)
                ctx.map_insert_resident_async(np.eye(4), 20)
            assert a.map_size()[0] == b.map_size()[0]
        # fetch without anything pending: the two-step path's answer
        fp2, fc2 = a.scan_fetch()
        assert np.array_equal(fp2, fp) and np.array_equal(fc2, fc)
        with pytest.raises(capi.VgicpError) as e:      # ... which no kernel summed
            a.scan_fetch_sums()
        assert e.value.code == capi.ERR_NOT_READY
        # a refused scan (a point beyond the search grid): the fetch reports it like vgicp_scan_info, nothing hangs
        bad = synth.make_lidar_scan(5_000, seed=3).copy()
        bad[17, 0] = 1e9
        a.scan_prepare_async(bad, None, None, None, 0.3, 30)
        with pytest.raises(capi.VgicpError) as e:
            a.scan_fetch()
        assert e.value.code == capi.ERR_BAD_ARGUMENT
        a.scan_prepare_async(raw, None, None, ext, 0.3, 30)          # and the context recovers
        assert len(a.scan_fetch()[0]) > 0


def test_wire_format_sweep_is_widened_on_the_device(oracle):
    """vgicp_sweep_stage_cloud2: the payload of a PointCloud2 handed over as the sensor wrote it (float32 x y z and a
    float64 timestamp inside records of point_step bytes, other fields and padding around them); the device picks the
    floats out and widens them.  float -> double is exact, so the prepared scan is bit-identical to the one made from the
    cloud the reference's callback builds on the host (include/ESKF_LIO/Subscriber.hpp:89-97) -- and to the oracle's."""
    from eskf_lio_amd import capi, replay, synth
    st = synth.make_imu_states(48, seed=5)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    rng = np.random.default_rng(3)
    for n, step, offs in ((30_001, 32, (0, 4, 8, 16)), (12_345, 28, (16, 20, 24, 0)), (257, 64, (40, 4, 52, 24)), (70_000, 20, (0, 4, 8, 12))):
        pts = synth.make_lidar_scan(n, seed=60 + step).astype(np.float32).astype(np.float64)
        tt = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=step)
        raw = rng.integers(0, 256, size=(n, step), dtype=np.uint8)              # whatever else the records carry
        for c in range(3):
            raw[:, offs[c]:offs[c] + 4] = pts[:, c].astype("<f4").view(np.uint8).reshape(n, 4)
        raw[:, offs[3]:offs[3] + 8] = tt.astype("<f8").view(np.uint8).reshape(n, 8)
        with capi.Context(0) as a, capi.Context(0) as b:
            for c in (a, b):
                c.map_reset(0.3, 1000)
            a.scan_prepare_staged_async(a.sweep_stage(pts, tt), st, ext, 0.3, 30)
            b.scan_prepare_staged_async(b.sweep_stage_cloud2(raw, n, step, offs[0], offs[1], offs[2], offs[3]), st, ext, 0.3, 30)
            pa, ca = a.scan_download()
            pb, cb = b.scan_download()
            assert a.scan_info() == b.scan_info() and np.array_equal(pa, pb) and np.array_equal(ca, cb), (n, step)
            if n == 12_345:
                moved, _ = oracle.transform(pts, np.tile(np.eye(3).reshape(9), (n, 1)), ext)
                desk, _ = oracle.deskew(moved, tt, st)
                rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
                assert np.array_equal(pb, rp) and np.array_equal(cb, rc)
            # no capture times in the record: no deskew from that ticket
            t2 = b.sweep_stage_cloud2(raw, n, step, offs[0], offs[1], offs[2], None)
            with pytest.raises(capi.VgicpError):
                b.scan_prepare_staged_async(t2, st, ext, 0.3, 30)
    # a whole message, as the replay harness reads it off a bag
    pts = synth.make_lidar_scan(5_000, seed=9).astype(np.float32).astype(np.float64)
    tt = synth.make_point_times(5_000, 100.0, 100.1, seed=9)
    payload, n, step, ox, oy, oz, ot = replay.pointcloud2_payload(replay.encode_pointcloud2(pts, tt))
    host = replay.decode_pointcloud2(replay.encode_pointcloud2(pts, tt))
    with capi.Context(0) as a, capi.Context(0) as b:
        a.scan_prepare_staged_async(a.sweep_stage(host.points, host.pointTime), None, None, 0.3, 30)
        b.scan_prepare_staged_async(b.sweep_stage_cloud2(payload, n, step, ox, oy, oz, ot), None, None, 0.3, 30)
        for x, y in zip(a.scan_download(), b.scan_download()):
            assert np.array_equal(x, y)
    with capi.Context(0) as c:
        for bad in ((8, 0, 4, 8), (68, 0, 4, 8), (30, 0, 4, 8), (32, 2, 4, 8), (32, 0, 4, 30)):
            with pytest.raises(capi.VgicpError):
                c.sweep_stage_cloud2(np.zeros(64 * 100, dtype=np.uint8), 64, bad[0], bad[1], bad[2], bad[3], None)


def test_prepared_scan_of_a_sweep_larger_than_the_grid(oracle):
    """A raw sweep with more points than the persistent launch has point-carrying threads (150 000 > 256 x 448), prepared
    on the device and aligned WITHOUT waiting for the kept count: the launch plan is made from the raw count (the
    several-points-per-thread instantiation with memos and parked points), the kernel reads the real size — fewer than
    one point per thread — from the device and then runs the one-point-per-thread body, i.e. returns the bits the launch
    of a settled scan of that size returns (vgicp_scan_info, or the same scan uploaded from the host): the chain that
    does not wait and the one that does agree bit for bit whatever the raw size of the sweep."""
    from eskf_lio_amd import capi, synth
    world = synth.make_lidar_scan(260_000, seed=91)
    T = synth.se3_to_SE3(np.array([0.04, -0.02, 0.01, 0.002, -0.002, 0.008]))
    Tinv = synth.invert_pose(T)
    with capi.Context(0) as ctx:
        ctx.map_reset(0.3, 400_000)
        p0, c0, _ = ctx.preprocess(world[:110_000], 0.3, 30)
        ctx.map_insert_scan(p0, c0, np.eye(4), 100)
        raw = world[110_000:] @ Tinv[:3, :3].T + Tinv[:3, 3]           # 150 000 points seen from a moved sensor
        ctx.scan_prepare_async(raw, None, None, None, 0.3, 30)
        one = ctx.align_resident(np.eye(4), 30, 1e-6, 0.9999)           # the frame's one synchronisation
        kept = ctx.scan_info()[0]
        assert 10_000 < kept < 114_688 and one.launches == 1
        settled = ctx.align_resident(np.eye(4), 30, 1e-6, 0.9999)       # the count is known now: planned for it
        assert np.array_equal(settled.pose, one.pose) and np.array_equal(settled.normal_eq, one.normal_eq)
        ctx.scan_prepare_async(raw, None, None, None, 0.3, 30)          # the same sequence again: the same bits
        again = ctx.align_resident(np.eye(4), 30, 1e-6, 0.9999)
        assert np.array_equal(again.pose, one.pose) and np.array_equal(again.normal_eq, one.normal_eq)
        gp, gc = ctx.scan_download()
        rp, rc, _ = oracle.preprocess(raw, 0.3, 30)
        assert np.array_equal(gp, rp) and np.array_equal(gc, rc)
        up = ctx.align(gp, gc, np.eye(4), 30, 1e-6, 0.9999)             # the same scan, one point per thread
        assert up.iterations == one.iterations and np.array_equal(up.corr_count, one.corr_count)
        assert np.array_equal(up.normal_eq, one.normal_eq) and np.array_equal(up.pose, one.pose)
        et, er = pose_error(one.pose, T)
        assert et < 0.05 and er < 0.01                                   # and it recovers the motion
        assert ctx.counter(1) == 0


def test_cpp_example_tracks_a_moving_sensor():
    """examples/register_frames.cpp: the reference's frame loop written in C++ against the shim."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "register_frames")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("frame")]
    assert len(lines) == 6 and all("converged 1" in ln for ln in lines[1:])


def test_cpp_frame_chain_classes_and_resident_abi_agree():
    """examples/frame_chain.cpp: CloudPreprocessor::process -> ICP::align -> LocalMap::updateLocalMap in C++ on
    the shim, and the same frames through vgicp_scan_prepare / align_resident / map_insert_resident."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "frame_chain")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("frame")]
    assert len(lines) == 6 and "deskewed" in lines[1]
    # the third chain of the example: the calls that do not wait — same bits (the example's exit code), one host
    # synchronisation per frame
    tail = [ln for ln in out.stdout.splitlines() if ln.startswith("the same without waiting")]
    assert len(tail) == 1 and "1.0 host synchronisations per frame" in tail[0], out.stdout
    # chain A of the example: the drop-in CLASSES ride that chain (device-resident grid, deferred host copy): every
    # align found its cloud resident, and the poses are the non-waiting chain's bits (the example's exit code)
    classes = [ln for ln in out.stdout.splitlines() if ln.startswith("drop-in classes, scan handed along")]
    assert len(classes) == 1 and "5 of 5 aligns found their cloud resident" in classes[0], out.stdout


def test_bench_line_keeps_its_contract():
    """bench.py end to end (short run): ONE JSON line with the driver's keys, BASELINE.json's metric string, the
    roofline and cpu_baseline objects, and parity confirmed on the timed result."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-c5",
                          "--cpu-budget", "4"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "points/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e9 and abs(d["value"] - 100_000 * 20 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e5 and "sample" in c
    assert d["parity"]["identical_counts"] is True and d["parity"]["pose_delta_m"] < 1e-9
    assert d["config"]["persistent_fallbacks"] == 0
    # `traffic` is a quotation from profiles/, never this run's measurement: the line says where it comes from, and a
    # profile taken on another build of the kernels is not quoted at all
    src = r["traffic_source"]
    if r["traffic"] is None:
        assert "reason" in src and r["traffic_frac"] is None
    else:
        assert src["file"].startswith("profiles/") and src["tag"] and "match" in src
        assert src["library_sha256"] or src["kernel_source_sha256"]
    fc = d["frame_chain"]
    assert "error" not in fc and "error" not in fc["dropin"], fc
    assert fc["dropin"]["poses_bit_equal_to_the_abi_chain"] is True and fc["dropin_ms_per_frame"] > 0

    up = d["config"]["upload"]
    # every timed step had a fresh cloud that was freed inside the loop; nothing was repeated, no align stalled — also
    # not in the loop that gives every cloud's pages back to the kernel (munmap) right after its align
    assert up["steps"] == 5 and up["ms_per_step"] == d["ms_per_step"] and up["ms_per_step_reused"] > 0
    assert up["aligns_above_1ms"] == 0 and up["align_ms_max"] < 1.0, up
    assert up["unmapped_every_step"]["aligns_above_1ms"] == 0 and up["unmapped_every_step"]["steps"] == 5, up
    assert up["uploads_repeated_because_the_copy_threads_were_held_up"] == 0
    assert 0.2 < up["fraction_of_kernel_read_rate_54GBps"] < 1.5, up
    assert d["config"]["sharding"] == "single GPU"


def _run_bench(extra_args, env_extra, launcher=None, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable] + (launcher or []) + [os.path.join(root, "bench.py")] + extra_args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_through_the_rccl_loop_world_one():
    """BENCH_FORCE_COMM=1: the N > 1 branch of bench.py with one rank — process group, unique-id hand-off,
    vgicp_comm_init, one launch + one ncclAllReduce per iteration — and a line that says so."""
    d = _run_bench(["--steps", "4", "--warmup", "1", "--no-c5", "--no-cpu-baseline"],
                   {"BENCH_FORCE_COMM": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "RANK": "0",
                    "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    sh = d["config"]["sharding"]
    assert sh["transport"] == "rccl" and d["n_gpus"] == 1 and d["roofline"]["rounds_per_launch"] == 1
    assert d["multi_gpu_parity"]["identical_counts"] is True and d["multi_gpu_parity"]["pose_delta"] < 1e-11
    assert "replicas_aggregate" in d and "exchange_us_per_round" in sh


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_line_with_real_ranks_sharing_the_device(world):
    """Dress rehearsal of `bench.py --gpus N` (BASELINE config C3) before any multi-GPU hardware sees it: N real
    ranks launched by torch.distributed.run as N processes on THE ONE device of this box (BENCH_SHARE_DEVICE=1: gloo
    rendezvous, mailboxes wired through vgicp_peer_export / _connect, 256 / N workgroups each). The merge across
    ranks is the cross-device form of reference src/Registration.cpp:71-75; here the mailbox stores stay on one
    device instead of crossing xGMI, everything else is the code the driver's scaling run executes."""
    import sys
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                "--master-port", str(29550 + world)]
    d = _run_bench(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--no-c5"],
                   {"BENCH_SHARE_DEVICE": "1", "VGICP_SPIN_LIMIT": "400000"}, launcher=launcher, timeout=1200)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["steps"] == 6
    # (no speed expectation: N processes share one device's CUs, caches and its one PCIe link here)
    assert d["value"] > 1e6 and abs(d["value"] - 100_000 * 20 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    sh = d["config"]["sharding"]
    assert sh["transport"] == "mailbox" and "BENCH_SHARE_DEVICE" in sh["wiring"]
    assert sh["sharded_us_per_round"] > 0 and sh["shard_alone_us_per_round"] > 0
    assert abs(sh["exchange_us_per_round"] - (sh["sharded_us_per_round"] - sh["shard_alone_us_per_round"])) < 1e-9
    assert d["multi_gpu_parity"]["identical_counts"] is True and d["multi_gpu_parity"]["pose_delta"] < 1e-11
    assert d["replicas_aggregate"]["value"] > 0
    assert d["roofline"]["rounds_per_launch"] == 20 and d["config"]["persistent_fallbacks"] == 0
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only


# ---- multi-GPU code path on one device: RCCL communicator of size 1 ---------------------------
def test_rccl_path_world_size_one(c1_gpu, c1_inputs):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    g = synth.default_guess()
    single = c1_gpu.align(pts, covs, g, 20, 1e-6, 2.0)
    c1_gpu.comm_init(1, 0, c1_gpu.comm_unique_id())
    try:
        NP = 2                                                     # VGICP_FLAG_NO_PERSISTENT
        viacomm = c1_gpu.align(pts, covs, g, 20, 1e-6, 2.0, flags=NP)   # fold kernel + ncclAllReduce + prologue
        assert viacomm.world_size == 1 and viacomm.iterations == 20 and viacomm.launches == 21
        same = c1_gpu.align(pts, covs, g, 20, 1e-6, 2.0)          # one rank: nothing to exchange, the single launch
        assert same.launches == 1 and np.array_equal(same.pose, single.pose)
        assert np.array_equal(viacomm.corr_count, single.corr_count)
        dt, dr = pose_error(viacomm.pose, single.pose)
        assert dt <= 1e-12 and dr <= 1e-12
        spts, scovs, _ = synth.make_structured_scan(5_000, c1_inputs[0])
        r = c1_gpu.align(spts, scovs, np.eye(4), 100, 1e-6, 0.9999, flags=NP)
        assert r.converged and r.iterations == 3
    finally:
        c1_gpu.comm_destroy()


def test_two_gpu_sharded_align():
    """Two ranks on two GPUs through vgicp_comm_init (RCCL + the device-initiated exchange where the GPUs can
    map each other): each registers its shard, both return the single-GPU result. Needs a second device."""
    import socket
    import subprocess
    import sys
    # counted in a child: importing torch HERE, after libvgicp_hip.so, would load torch's bundled HIP runtime beside
    # /opt/rocm's (same soname, different files) and the process would abort at exit
    count = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=300)
    if count.returncode != 0 or int(count.stdout.strip().splitlines()[-1]) < 2:
        pytest.skip("needs two GPUs (the round's boxes have one); the 8-GPU scaling run is the driver's")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    worker = os.path.join(os.path.dirname(__file__), "multigpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


@pytest.mark.parametrize("world,grid", [(2, 100), (4, 60)])
def test_peer_exchange_two_processes_one_device(tmp_path, world, grid):
    """The device-initiated exchange between ranks (vgicp_peer_*: HIP IPC mailboxes written by the ranks'
    persistent kernels) with two or four PROCESSES on the one device of this box — what crosses xGMI on a multi-GPU
    node crosses the device's own memory here; protocol, mapping and ordering are the same.  Each rank
    registers its shard; every rank must obtain the same bits, equal to the whole scan on one context up to
    the grouping of the sums.  The ranks' persistent launches have to be resident together, so each uses 100
    (60 with four ranks) of the 256 compute units (VGICP_PERSIST_GRID).  Spins are bounded: a launch that gives up ends its process
    with a non-zero code.  Unmeasured on 8 GPUs (no such node was available to the build)."""
    import subprocess
    import sys
    n, rounds = 40_000, 8
    env = dict(os.environ, VGICP_PERSIST_GRID=str(grid))
    worker = os.path.join(os.path.dirname(__file__), "peer_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(tmp_path), str(n), str(rounds)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    results = [np.load(os.path.join(tmp_path, f"result{r}.npz")) for r in range(world)]
    for other in results[1:]:
        assert np.array_equal(results[0]["pose"], other["pose"])
        assert np.array_equal(results[0]["normal_eq"], other["normal_eq"])
        assert np.array_equal(results[0]["half_pose"], other["half_pose"])


@pytest.mark.parametrize("world", [2, 3])
def test_peer_exchange_give_up_is_collective(tmp_path, world):
    """A rank that gives up waiting for a late peer in the round an align ENDS in must not leave that peer returning
    success on its own (it finds the row of the rank that gave up already in its mailbox): the ranks exchange verdict
    words at the end of the launch and commit the align together or not at all. Here rank 0 has a spin limit of a few
    milliseconds and the others start a one-round align 0.5 s late; there is no RCCL communicator to fall back to, so
    EVERY rank has to report VGICP_ERR_RCCL (tests/peer_worker.py, mode "giveup")."""
    import subprocess
    import sys
    env = dict(os.environ, VGICP_PERSIST_GRID=str(200 // world))
    worker = os.path.join(os.path.dirname(__file__), "peer_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(tmp_path), "20000", "1", "giveup"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    from eskf_lio_amd import capi
    codes = [int(open(os.path.join(tmp_path, f"code{r}")).read()) for r in range(world)]
    assert codes == [capi.ERR_RCCL] * world, codes


# ---- BASELINE's full size (C2): size-independent properties ------------------------------------
def test_c2_full_size_properties(gpu_ctx, oracle):
    from eskf_lio_amd import synth
    vmap = synth.make_map(1_000_000)
    pts, covs = synth.make_uniform_scan(100_000, vmap)
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    assert gpu_ctx.map_size()[0] == 1_000_000
    g = synth.default_guess()
    a = gpu_ctx.align(pts, covs, g, 20, 1e-6, 2.0, chunk_iterations=20)
    assert a.iterations == 20 and 0.45 < a.corr_count.mean() / 1e5 < 0.55
    b = gpu_ctx.align(pts, covs, g, 20, 1e-6, 2.0, chunk_iterations=20)
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_eq, b.normal_eq)   # deterministic
    perm = np.random.default_rng(8).permutation(100_000)
    c = gpu_ctx.align(pts[perm], covs[perm], g, 20, 1e-6, 2.0, chunk_iterations=20)
    assert np.array_equal(c.corr_count, a.corr_count)              # order-free counts
    assert np.abs(c.pose - a.pose).max() < 1e-11
    # first-iteration counts equal an independent count of occupied voxels hit (integer work, exact)
    tp, _ = oracle.transform(pts, covs, g)
    keys = oracle.voxel_index(vmap.voxel_size, tp)
    def pack(k):
        k = k.astype(np.int64)
        return ((k[:, 0] + 2048) << 24) | ((k[:, 1] + 2048) << 12) | (k[:, 2] + 2048)
    assert int(np.isin(pack(keys), pack(vmap.keys)).sum()) == int(a.corr_count[0])
    # the oracle on the full size (deterministic mode, ~1 s): contractual parity at BASELINE's config
    om = oracle.OracleMap(vmap.voxel_size, 1)
    om.insert(vmap.means, vmap.covs)
    ref = om.align(pts, covs, g, 20, 1e-6, 2.0)
    assert_align_parity(a, ref)


def test_sharded_loop_on_one_device_matches_the_single_gpu_log(gpu_ctx):
    """SURVEY.md 8(e) parity expectation at C2 size without a second GPU: the scan cut by shard_bounds into
    G in {2, 4, 8} contiguous shards, every round each shard accumulated on the device at the current pose
    (vgicp_accumulate), the G rows summed on the host in rank order as the exchange does, solved on the
    device (vgicp_solve_step) and composed. Every round's 27-vector and count against the single-GPU log:
    counts identical, sums equal up to the grouping of the additions."""
    from eskf_lio_amd import synth
    from eskf_lio_amd.distributed import shard_bounds
    vmap = synth.make_map(1_000_000)
    pts, covs = synth.make_uniform_scan(100_000, vmap)
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    g = synth.default_guess()
    rounds = 8
    single = gpu_ctx.align(pts, covs, g, rounds, 1e-6, 2.0)
    assert single.iterations == rounds
    for G in (2, 4, 8):
        pose = g.copy()
        for r in range(rounds):
            JTJ, JTr, count = np.zeros((6, 6)), np.zeros(6), 0
            for rank in range(G):
                lo, hi = shard_bounds(pts.shape[0], G, rank)
                a, b, c = gpu_ctx.accumulate(pts[lo:hi], covs[lo:hi], pose)
                JTJ, JTr, count = JTJ + a, JTr + b, count + c
            assert count == int(single.corr_count[r]), (G, r)
            scale = np.abs(single.JTJ[r]).max()
            assert np.allclose(JTJ, single.JTJ[r], rtol=1e-11, atol=1e-12 * scale), (G, r)
            assert np.allclose(JTr, single.JTr[r], rtol=1e-9, atol=1e-12 * scale), (G, r)
            _, step, _, _ = gpu_ctx.solve_step(JTJ, JTr, cosine_threshold=2.0)
            pose = step @ pose
        dt, dr = pose_error(pose, single.pose)
        assert dt <= 1e-11 and dr <= 1e-11, (G, dt, dr)


def test_c5_full_size_parity(gpu_ctx, oracle):
    """BASELINE config C5 on one GPU: 1M-point scan vs 10M-voxel map (8.6 GB table, beyond the Infinity
    Cache), 3 forced iterations, against the oracle's deterministic mode at the full size; the persistent
    single launch (a thread owns ~9 points there) and the one-launch-per-round loop. Identical counts, the
    contractual 1e-4 and the tight 1e-9."""
    from eskf_lio_amd import capi, synth
    vmap = synth.make_map(10_000_000)
    pts, covs = synth.make_uniform_scan(1_000_000, vmap)
    g = synth.default_guess()
    om = oracle.OracleMap(vmap.voxel_size, 1)
    om.insert(vmap.means, vmap.covs)
    ref = om.align(pts, covs, g, 3, 1e-6, 2.0)
    del om
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    assert gpu_ctx.map_size()[0] == 10_000_000
    gpu_ctx.scan_upload(pts, covs)
    one = gpu_ctx.align_resident(g, 3, 1e-6, 2.0)
    assert one.launches == 1
    assert_align_parity(one, ref)
    loop = gpu_ctx.align_resident(g, 3, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
    assert loop.launches > 1
    assert_align_parity(loop, ref)
    scale = np.abs(ref.JTJ).max()
    for r in range(3):
        assert np.allclose(one.JTJ[r], ref.JTJ[r], rtol=1e-9, atol=1e-12 * scale)
        assert np.allclose(one.JTr[r], ref.JTr[r], rtol=1e-7, atol=1e-12 * scale)
    # 12 rounds of the persistent launch: what a thread remembers about its points between rounds is
    # exercised over many rounds and must agree with the loop that looks every point up every round
    many = gpu_ctx.align_resident(g, 12, 1e-6, 2.0)
    many_loop = gpu_ctx.align_resident(g, 12, 1e-6, 2.0, flags=capi.FLAG_NO_PERSISTENT)
    assert many.launches == 1 and np.array_equal(many.corr_count, many_loop.corr_count)
    assert np.abs(many.pose - many_loop.pose).max() < 1e-11


# ---- N2: scan preparation on the device (CloudPreprocessor.cpp:76-127) ----------------------------
def _cov_mats(c9):
    return np.asarray(c9).reshape(-1, 3, 3).transpose(0, 2, 1)


def _assert_preprocess_parity(got, ref, pts, exact=True):
    gp, gc, gi = got
    rp, rc, ri = ref
    assert np.array_equal(gi, ri)                                  # kept points: integer work, exact
    assert np.array_equal(gp, rp) and np.array_equal(gp, pts[gi.astype(np.int64)])
    G, R = _cov_mats(gc), _cov_mats(rc)
    # The device code keeps the oracle's operation order without FMA contraction, and division and
    # square root are correctly rounded on both sides: the same neighbours give the same bits. (A bound
    # in terms of a tolerance would have to follow the conditioning of the normal direction, which the
    # reference's E[xx^T] - E[x]E[x]^T makes arbitrarily bad far from the origin.)
    if exact:
        assert np.array_equal(G, R), np.abs(G - R).max()
    else:
        assert np.abs(G - R).max() < 1e-9
    # U F V^T with orthogonal U, V: singular values (1, 1, 1e-2) always; eigenvalues (1e-2, 1, 1) unless the
    # neighbourhood is degenerate (repeated / coplanar / collinear points: rounding-level eigenvalues of either sign,
    # test_preprocess_degenerate_neighbourhoods_are_the_references_class)
    assert np.allclose(np.linalg.svd(G, compute_uv=False), [1.0, 1.0, 1e-2], atol=1e-9)


@pytest.mark.parametrize("n,knn", [(3_000, 30), (20_000, 30), (20_000, 7)])
def test_preprocess_matches_oracle(gpu_ctx, oracle, n, knn):
    from eskf_lio_amd import synth
    pts = synth.make_lidar_scan(n, seed=n + knn)
    _assert_preprocess_parity(gpu_ctx.preprocess(pts, 0.3, knn), oracle.preprocess(pts, 0.3, knn), pts)


def test_preprocess_full_size_sweep(gpu_ctx, oracle):
    """A 100 000-point sweep (the size of one spinning-LiDAR revolution): dense ground near the sensor, walls,
    8 % clutter — queries whose cell pool overflows and spills are part of it. Bit-exact like the small ones."""
    from eskf_lio_amd import synth
    pts = synth.make_lidar_scan(100_000, seed=11)
    _assert_preprocess_parity(gpu_ctx.preprocess(pts, 0.3, 30), oracle.preprocess(pts, 0.3, 30), pts)


@pytest.mark.parametrize("n", [32_768, 32_769, 65_537])
def test_preprocess_either_side_of_the_sort_block_switch(gpu_ctx, oracle, n):
    """The preparation's one sort runs in 2 048-item blocks up to 32 768 points and from 65 537 on, in 4 096-item blocks
    in between (vgicp_preprocess.hip, sort_in_large_blocks); a stable sort either way, so the prepared scan is the
    oracle's bits on both sides of each switch.  (Also sizes at which the queries of sparse neighbourhoods, which start
    first, and the others split into uneven lists.)"""
    from eskf_lio_amd import synth
    pts = synth.make_lidar_scan(n, seed=n % 1000, extent=30.0)
    _assert_preprocess_parity(gpu_ctx.preprocess(pts, 0.3, 30), oracle.preprocess(pts, 0.3, 30), pts)


def test_preprocess_other_voxel_sizes_and_dense_cells(gpu_ctx, oracle):
    from eskf_lio_amd import synth
    pts = synth.make_lidar_scan(15_000, seed=5, extent=10.0)       # many points per voxel
    for h in (0.1, 0.5, 2.0):
        _assert_preprocess_parity(gpu_ctx.preprocess(pts, h, 30), oracle.preprocess(pts, h, 30), pts)


def test_preprocess_edge_cases(gpu_ctx, oracle):
    from eskf_lio_amd.capi import VgicpError
    assert len(gpu_ctx.preprocess(np.zeros((0, 3)), 0.3)[2]) == 0
    # fewer points than neighbours asked for; fewer than three -> identity before the regularisation
    two = np.array([[0.0, 0.0, 0.0], [5.0, 0.0, 0.0]])
    gp, gc, gi = gpu_ctx.preprocess(two, 0.3, 30)
    assert list(gi) == [0, 1] and np.allclose(_cov_mats(gc), np.diag([1.0, 1.0, 1e-2]), atol=1e-15)
    rng = np.random.default_rng(3)
    few = rng.normal(size=(17, 3)) * 3.0
    _assert_preprocess_parity(gpu_ctx.preprocess(few, 0.3, 30), oracle.preprocess(few, 0.3, 30), few)
    # far-apart clusters: the search has to climb to the coarsest level or look at everything
    far = np.concatenate([rng.normal(size=(20, 3)) * 0.05, rng.normal(size=(20, 3)) * 0.05 + 900.0,
                          rng.normal(size=(5, 3)) * 0.05 - 400.0])
    _assert_preprocess_parity(gpu_ctx.preprocess(far, 0.3, 30), oracle.preprocess(far, 0.3, 30), far)
    # negative coordinates and points exactly on voxel faces keep the map's floor() convention
    grid = np.stack(np.meshgrid(np.arange(-6, 6) * 0.15, np.arange(-6, 6) * 0.15, [0.0, -0.3], indexing="ij"),
                    axis=-1).reshape(-1, 3)
    grid = grid + rng.normal(size=grid.shape) * 1e-3 * np.array([0.0, 0.0, 1.0])
    _assert_preprocess_parity(gpu_ctx.preprocess(grid, 0.3, 12), oracle.preprocess(grid, 0.3, 12), grid)
    with pytest.raises(VgicpError):
        gpu_ctx.preprocess(two, 0.3, 33)                            # more neighbours than the kernel holds
    with pytest.raises(VgicpError):
        gpu_ctx.preprocess(two, 0.0, 30)


def test_preprocess_ties_duplicates_and_crowded_cells(gpu_ctx, oracle):
    """What the search's shortcuts must not change: a lattice (every distance many times over: the k-th place is
    decided by the index), exact duplicates of one point (a finest cell far more crowded than the 128 points the
    'home cell' may hold, all at distance zero), and a scan where both meet ordinary points."""
    from eskf_lio_amd import synth
    rng = np.random.default_rng(17)
    g = np.arange(-7, 8) * 0.125
    lattice = np.stack(np.meshgrid(g, g, g[:5], indexing="ij"), axis=-1).reshape(-1, 3)   # 1 125 points, exact in binary
    lattice = lattice[rng.permutation(len(lattice))]
    for k in (30, 7):
        _assert_preprocess_parity(gpu_ctx.preprocess(lattice, 0.3, k), oracle.preprocess(lattice, 0.3, k), lattice)
    pile = np.concatenate([np.tile([[0.51, -0.23, 0.07]], (700, 1)), rng.normal(size=(300, 3)) * 0.4,
                           np.tile([[3.0, 3.0, 3.0]], (40, 1))])
    pile = pile[rng.permutation(len(pile))]
    _assert_preprocess_parity(gpu_ctx.preprocess(pile, 0.3, 30), oracle.preprocess(pile, 0.3, 30), pile)
    mixed = np.concatenate([synth.make_lidar_scan(6_000, seed=3), lattice + [2.0, 1.0, 0.5], pile + [-3.0, 2.0, 0.0]])
    mixed = mixed[rng.permutation(len(mixed))]
    for h in (0.3, 0.8):
        _assert_preprocess_parity(gpu_ctx.preprocess(mixed, h, 30), oracle.preprocess(mixed, h, 30), mixed)


def test_preprocess_degenerate_neighbourhoods_are_the_references_class(gpu_ctx, oracle):
    """tests/golden/prep_degenerate.npz: an exact tilted plane, collinear and repeated points far from the origin. The
    cumulant covariance then has rounding-level eigenvalues of either sign and the reference's svd.matrixU() * F *
    svd.matrixV()^T (src/CloudPreprocessor.cpp:119-123) returns INDEFINITE matrices; the device returns the oracle's bits
    for them too and reports how many kept points were affected (VGICP_COUNTER_PREP_INDEFINITE)."""
    g = np.load(os.path.join(GOLDEN, "prep_degenerate.npz"))
    pts = g["points"]
    gp, gc, gi = gpu_ctx.preprocess(pts, float(g["voxel_size"]), int(g["knn"]))
    assert np.array_equal(gi, g["kept_index"]) and np.array_equal(gp, g["kept_points"])
    assert np.array_equal(gc, g["kept_covs"])
    assert gpu_ctx.counter(4) == int(g["indefinite"]) > 0
    rp, rc, ri, bad = oracle.preprocess_ex(pts, float(g["voxel_size"]), int(g["knn"]))
    assert np.array_equal(gc, rc) and bad == int(g["indefinite"])
    # whatever the signs, the singular values of every result are (1, 1, 1e-2): U and V are orthogonal
    sv = np.linalg.svd(_cov_mats(gc), compute_uv=False)
    assert np.allclose(sv, [1.0, 1.0, 1e-2], atol=1e-9)
    # the fused chain reports the same count, and an ordinary scan reports none
    kept, _ = gpu_ctx.scan_prepare(pts, None, None, None, float(g["voxel_size"]), int(g["knn"]))
    assert kept == len(gi) and gpu_ctx.counter(4) == int(g["indefinite"])
    from eskf_lio_amd import synth
    gpu_ctx.preprocess(synth.make_lidar_scan(3_000, seed=2), 0.3, 30)
    assert gpu_ctx.counter(4) == 0


def test_preprocess_feeds_the_registration(gpu_ctx, oracle):
    """Frame pipeline of LIOdometry (src/Odometry.cpp:153-175): prepare the scan, align it, insert it."""
    from eskf_lio_amd import synth
    world = synth.make_lidar_scan(60_000, seed=77)
    gpu_ctx.map_reset(0.3, 0)
    p0, c0, _ = gpu_ctx.preprocess(world[:30_000], 0.3, 30)
    gpu_ctx.map_insert_scan(p0, c0, np.eye(4), 100)
    T = synth.se3_to_SE3(np.array([0.05, -0.03, 0.01, 0.002, -0.003, 0.01]))
    Tinv = synth.invert_pose(T)
    moved = world[30_000:] @ Tinv[:3, :3].T + Tinv[:3, 3]
    p1, c1, i1 = gpu_ctx.preprocess(moved, 0.3, 30)
    r1 = oracle.preprocess(moved, 0.3, 30)
    assert np.array_equal(i1, r1[2])
    got = gpu_ctx.align(p1, c1, np.eye(4), 30, 1e-6, 0.9999)
    om = oracle.OracleMap(0.3, 100)
    om.insert(*oracle.transform(*oracle.preprocess(world[:30_000], 0.3, 30)[:2], np.eye(4)))
    ref = om.align(r1[0], r1[1], np.eye(4), 30, 1e-6, 0.9999)
    assert got.iterations == ref.iterations and np.array_equal(got.corr_count, ref.corr_count)
    dt, dr = pose_error(got.pose, ref.pose)
    assert dt < 1e-6 and dr < 1e-6
    et, er = pose_error(got.pose, T)
    assert et < 0.05 and er < 0.01                                 # and it recovers the motion


def test_reference_order_option_reproduces_the_references_sequence_and_chain(oracle):
    """VGICP_OPTION_REFERENCE_ORDER (round 6): the kept points in the iteration order of the reference's
    std::unordered_map (src/CloudPreprocessor.cpp:85-99) instead of ascending input index — the last documented deviation
    of a frame CHAIN from the reference's (Voxel::addPoint's running mean depends on the sequence,
    include/ESKF_LIO/LocalMap.hpp:79-87).  (1) Every preparation entry point returns oracle_preprocess_ordered(...,
    ORACLE_ORDER_REFERENCE_HASH) bit for bit: points, covariances AND their order; (2) a 12-frame chain prepare -> align
    -> insert on the device, all in that order, against the oracle's chain in that order: the same round counts and
    correspondence counts every frame, poses to rounding (1e-9; the two chains in DIFFERENT orders are 3e-5 m apart,
    tests/test_oracle.py), the same voxel set with means to rounding at the end."""
    from eskf_lio_amd import capi, synth
    world = synth.make_lidar_scan(9_000, seed=0x46524D, extent=12.0)
    with capi.Context(0) as ctx:
        ctx.set_option(capi.OPTION_REFERENCE_ORDER, 1)
        rp, rc, ri = oracle.preprocess_ordered(world, 0.3, 30, oracle.ORDER_REFERENCE_HASH)
        assert not np.array_equal(ri, np.sort(ri))                    # it IS another order
        gp, gc, gi = ctx.preprocess(world, 0.3, 30)                    # the host-returning entry point
        assert np.array_equal(gi, ri) and np.array_equal(gp, rp) and np.array_equal(gc, rc)
        ctx.map_reset(0.3, 0)
        kept, _ = ctx.scan_prepare(world, None, None, None, 0.3, 30)   # the resident one
        dp, dc = ctx.scan_download()
        assert kept == len(rp) and np.array_equal(dp, rp) and np.array_equal(dc, rc)
        ctx.scan_prepare_async(world, None, None, None, 0.3, 30)       # and the enqueued one (waited for under the option)
        fp, fc = ctx.scan_fetch()
        assert np.array_equal(fp, rp) and np.array_equal(fc, rc)
        # the chain
        frames, cap = 12, 20
        truth = [synth.se3_to_SE3([0.05 * f, 0.02 * f, 0.0, 0.0, 0.0, 0.004 * f]) for f in range(frames + 1)]
        rng = np.random.default_rng(12)
        omap = oracle.OracleMap(0.3, cap)
        ctx.map_reset(0.3, 0)
        pose_o, pose_g = np.eye(4), np.eye(4)
        for f in range(frames + 1):
            Tinv = synth.invert_pose(truth[f])
            sweep = np.ascontiguousarray((world + rng.normal(scale=0.005, size=world.shape)) @ Tinv[:3, :3].T + Tinv[:3, 3])
            p, c, _ = oracle.preprocess_ordered(sweep, 0.3, 30, oracle.ORDER_REFERENCE_HASH)
            ctx.scan_prepare_async(sweep, None, None, None, 0.3, 30)
            if f > 0:
                ro = omap.align(p, c, pose_o, 30, 1e-6, 0.9999)
                rg = ctx.align_resident(pose_g, 30, 1e-6, 0.9999)
                assert rg.iterations == ro.iterations and np.array_equal(rg.corr_count[:rg.iterations], ro.corr_count[:ro.iterations]), f
                dt, dr = pose_error(rg.pose, ro.pose)
                assert dt < 1e-9 and dr < 1e-9, (f, dt, dr)
                pose_o, pose_g = ro.pose, rg.pose
            omap.insert(*oracle.transform(p, c, pose_o))
            ctx.map_insert_resident(pose_g, cap)
        keys, means, covs, counts = ctx.map_export()
        ok, om, oc, on = omap.export()
        a = {tuple(k): i for i, k in enumerate(keys.tolist())}
        assert len(a) == len(ok) and all(tuple(k) in a for k in ok.tolist())
        order = np.array([a[tuple(k)] for k in ok.tolist()])
        assert np.array_equal(counts[order], on)
        assert np.abs(means[order] - om).max() < 1e-9 and np.abs(covs[order] - oc).max() < 1e-9


@pytest.mark.parametrize("map_voxel,scan_voxel,cap", [(0.3, 0.3, 20), (0.3, 0.1, 5), (0.3, 0.1, 100), (1.0, 0.1, 100)])
def test_resident_insertion_without_the_sort_is_bit_exact(oracle, monkeypatch, map_voxel, scan_voxel, cap):
    """LocalMap::updateLocalMap (src/LocalMap.cpp:44-72) on a scan the device down-sampled itself: a map voxel receives
    a handful of its points at most, and the insertion then keeps scan order through per-voxel lists instead of a sort
    (launch_map_insert, short_lists).  Maps three times coarser than the scan's grid give lists longer than the
    leader's register buffer (several walks); ten times coarser falls back to the sort.  Every variant must leave the
    map the serial reference loop leaves: keys, means, covariances and counts compared with ==, over three frames
    with different poses (existing voxels take addPoint, the cap freezes them)."""
    from eskf_lio_amd import capi, synth
    om = oracle.OracleMap(map_voxel, cap)
    maps = {}
    for variant in ("lists", "sort"):
        if variant == "sort":
            monkeypatch.setenv("VGICP_INSERT_SORT", "1")
        with capi.Context(0) as ctx:
            ctx.map_reset(map_voxel, 0)
            for f in range(3):
                raw = synth.make_lidar_scan(20_000, seed=40 + f)
                T = synth.se3_to_SE3(np.array([0.3 * f, -0.2 * f, 0.05 * f, 0.01 * f, -0.02 * f, 0.03 * f]))
                kept, _ = ctx.scan_prepare(raw, None, None, None, scan_voxel, 30)
                if variant == "lists":
                    gp, gc = ctx.scan_download()
                    assert kept == len(gp)
                    om.insert(*oracle.transform(gp, gc, T))
                if f == 1:
                    ctx.map_insert_resident_async(T, cap)              # the call that does not wait
                else:
                    ctx.map_insert_resident(T, cap)
            maps[variant] = ctx.map_export()
        monkeypatch.delenv("VGICP_INSERT_SORT", raising=False)
    ref = _sorted_oracle_export(om)
    for variant, got in maps.items():
        assert len(got[0]) == len(om), variant
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), variant


def test_host_mirror_cloud_preprocessor(oracle):
    """ESKF_LIO::CloudPreprocessor::voxelDownsampleAndEstimateCovariances through the C++ mirror
    (include/eskf_lio_shim/CloudPreprocessor.hpp): the cloud comes back as the oracle leaves it."""
    from eskf_lio_amd import host, synth
    pts = synth.make_lidar_scan(8_000, seed=21)
    pre = host.CloudPreprocessor(0.3)
    gp, gc = pre.voxelDownsampleAndEstimateCovariances(pts)
    rp, rc, _ = oracle.preprocess(pts, 0.3, 30)
    assert np.array_equal(gp, rp) and np.array_equal(gc, rc)
    gp, gc = pre.voxelDownsampleAndEstimateCovariances(np.zeros((0, 3)))
    assert gp.shape == (0, 3) and gc.shape == (0, 9)


# ---- N4: deskew on the device (CloudPreprocessor.cpp:25-74) -----------------------------------------
@pytest.mark.parametrize("n,states,jitter", [(5_000, 48, 0.0), (120_000, 45, 0.0), (30_000, 48, 2e-3)])
def test_deskew_matches_oracle(gpu_ctx, oracle, n, states, jitter):
    from eskf_lio_amd import synth
    st = synth.make_imu_states(states, seed=n)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 0.4 / 400.0, seed=n, jitter=jitter)
    pts = synth.make_lidar_scan(n, seed=n)
    gp, gdone = gpu_ctx.deskew(pts, t, st)
    rp, rdone = oracle.deskew(pts, t, st)
    assert gdone == rdone and 0 < gdone < n                        # segment bounds: integer work, exact
    assert np.array_equal(gp, rp)                                  # same formulas, same order, no contraction
    assert np.array_equal(gp[gdone:], pts[gdone:])


def test_deskew_edge_cases(gpu_ctx, oracle):
    from eskf_lio_amd import synth
    st = synth.make_imu_states(12, seed=4)
    t = synth.make_point_times(300, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=4)
    pts = synth.make_lidar_scan(300, seed=4)
    for states, times in ((st[:3], t), (st, t - 1.0)):            # queue does not bracket the sweep's end
        gp, done = gpu_ctx.deskew(pts, times, states)
        assert done == -1 == oracle.deskew(pts, times, states)[1] and np.array_equal(gp, pts)
    assert gpu_ctx.deskew(np.zeros((0, 3)), np.zeros(0), st)[1] == 0
    # all points before the first state: one segment, moved with the first state's pose
    early = np.full(300, st[0, 0] - 1e-3)
    early[-1] = st[5, 0] + 1e-4
    gp, done = gpu_ctx.deskew(pts, early, st)
    rp, rdone = oracle.deskew(pts, early, st)
    assert done == rdone == 299 and np.array_equal(gp, rp)
    # times equal to a state's timestamp belong to the NEXT state (strict <), and a sweep whose times
    # go backwards is cut where the reference's sequential walk cuts it
    back = t.copy()
    back[100:200] = back[100:200][::-1]
    back[50] = st[4, 0]
    gp, done = gpu_ctx.deskew(pts, back, st)
    rp, rdone = oracle.deskew(pts, back, st)
    assert done == rdone and np.array_equal(gp, rp)
    # point times in random order, with repeats and NaNs: first hits all over the scan (the parallel bounds)
    rng = np.random.default_rng(8)
    wild = rng.permutation(np.concatenate([t[:250], np.repeat(st[6, 0], 30), np.full(20, np.nan)]))
    wild[-1] = st[-3, 0] + 1e-3
    gp, done = gpu_ctx.deskew(pts, wild, st)
    rp, rdone = oracle.deskew(pts, wild, st)
    assert done == rdone and np.array_equal(gp, rp, equal_nan=True)
    # a state queue that is NOT in time order takes the reference's walk, state by state, on the device too
    mixed = st.copy()
    mixed[[3, 4]] = mixed[[4, 3]]
    for times in (t, back):
        gp, done = gpu_ctx.deskew(pts, times, mixed)
        rp, rdone = oracle.deskew(pts, times, mixed)
        assert done == rdone and np.array_equal(gp, rp)
    # a large sweep in many blocks, times with jitter across block boundaries
    n = 150_000
    big = synth.make_lidar_scan(n, seed=6)
    st2 = synth.make_imu_states(60, seed=6)
    t2 = synth.make_point_times(n, st2[1, 0] + 1e-4, st2[-3, 0] + 1e-3, seed=6)
    t2 = t2 + rng.normal(size=n) * 2e-3
    t2[-1] = st2[-3, 0] + 1e-3
    gp, done = gpu_ctx.deskew(big, t2, st2)
    rp, rdone = oracle.deskew(big, t2, st2)
    assert done == rdone and np.array_equal(gp, rp)


def test_prepare_chain_finds_the_deskew_segments_of_awkward_sweeps(oracle):
    """The scan preparation finds the deskew's segments inside its first kernel (a running maximum of per-point hit
    counts, the sweep-wide maximum from the host; reference src/CloudPreprocessor.cpp:25-74): capture times that go
    backwards, repeat a state's timestamp, are NaN, all lie before the first state, jitter across workgroup
    boundaries; sweeps of one workgroup and of hundreds; staged by the copy threads and staged on arrival — the
    prepared scan and the number of points moved equal the oracle's chain, and the stand-alone deskew's."""
    from eskf_lio_amd import capi, synth
    rng = np.random.default_rng(77)
    st = synth.make_imu_states(12, seed=4)
    cases = []
    for n in (200, 300, 5_000, 70_000):
        t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=4)
        pts = synth.make_lidar_scan(n, seed=4 + n, extent=20.0)
        back = t.copy()
        back[n // 3:2 * n // 3] = back[n // 3:2 * n // 3][::-1]
        back[n // 6] = st[4, 0]
        wild = rng.permutation(np.concatenate([t[:n - 50], np.repeat(st[6, 0], 30), np.full(20, np.nan)]))
        wild[-1] = st[-3, 0] + 1e-3
        early = np.full(n, st[0, 0] - 1e-3)
        early[-1] = st[5, 0] + 1e-4
        jitter = t + rng.normal(size=n) * 2e-3
        jitter[-1] = st[-3, 0] + 1e-3
        nan_first = t.copy()
        nan_first[0] = np.nan
        for name, times in (("plain", t), ("back", back), ("wild", wild), ("early", early), ("jitter", jitter), ("nan_first", nan_first)):
            cases.append((f"{name}/{n}", pts, times))
    ext = synth.se3_to_SE3([0.02, -0.01, 0.03, 0.01, -0.02, 0.005])
    with capi.Context(0) as ctx:
        for name, pts, times in cases:
            moved, _ = oracle.transform(pts, np.tile(np.eye(3).reshape(9), (len(pts), 1)), ext)
            desk, rdone = oracle.deskew(moved, times, st)
            rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
            kept, done = ctx.scan_prepare(pts, times, st, ext, 0.3, 30)
            gp, gc = ctx.scan_download()
            assert done == rdone, (name, done, rdone)
            assert kept == len(rp) and np.array_equal(gp, rp, equal_nan=True) and np.array_equal(gc, rc, equal_nan=True), name
            # staged on arrival: the prologue reads times and points where the ticket left them
            ticket = ctx.sweep_stage(pts, times)
            ctx.scan_prepare_staged_async(ticket, st, ext, 0.3, 30)
            gp2, gc2 = ctx.scan_download()
            assert ctx.scan_info()[:2] == (kept, done), name
            assert np.array_equal(gp2, rp, equal_nan=True) and np.array_equal(gc2, rc, equal_nan=True), name
            # the stand-alone deskew (its own kernels) agrees on the count
            assert ctx.deskew(moved, times, st)[1] == rdone, name


def test_prepare_chain_of_large_sweeps_matches_the_stand_alone_calls():
    """Sweeps too large to stage go up straight from the caller's memory: 700 000 points (the prologue still finds the
    deskew's segments itself, the capture times on the device) and 1 200 000 (more workgroups than look-back slots: the
    bounds get their own launch, as before round 5).  Both against extrinsic (numpy, Open3D's operation order) +
    vgicp_deskew + vgicp_preprocess, which run their own kernels: the same kept points, covariances and moved count."""
    from eskf_lio_amd import capi, synth
    st = synth.make_imu_states(48, seed=9)
    ext = synth.se3_to_SE3([0.01, -0.02, 0.03, 0.002, -0.001, 0.003])
    with capi.Context(0) as a, capi.Context(0) as b:
        for n in (700_000, 1_200_000):
            t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=9)
            t = t + np.random.default_rng(1).normal(size=n) * 1e-3
            t[-1] = st[-3, 0] + 1e-3
            raw = synth.make_lidar_scan(n, seed=41, extent=60.0)
            kept, moved = a.scan_prepare(raw, t, st, ext, 0.3, 30)
            gp, gc = a.scan_download()
            q = np.empty((n, 4))
            for r in range(4):
                q[:, r] = ext[r, 0] * raw[:, 0] + ext[r, 1] * raw[:, 1] + ext[r, 2] * raw[:, 2] + ext[r, 3]
            dp, done = b.deskew(q[:, :3] / q[:, 3:4], t, st)
            kp, kc, _ = b.preprocess(dp, 0.3, 30)
            assert moved == done and kept == len(kp) and np.array_equal(gp, kp) and np.array_equal(gc, kc), n


def test_prepare_chain_across_state_queue_lengths():
    """4 ... 6 000 IMU states inside one sweep: the prologue finds the segments itself up to 4 096 states (8 bytes of LDS
    each), longer queues take the serial bounds walk — the same prepared scan and moved count as vgicp_deskew +
    vgicp_preprocess either way."""
    from eskf_lio_amd import capi, synth
    n = 20_000
    raw = synth.make_lidar_scan(n, seed=41, extent=30.0)
    with capi.Context(0) as a, capi.Context(0) as b:
        for states in (4, 300, 3000, 4096, 4100, 6000):
            st = synth.make_imu_states(states, seed=states)
            t = synth.make_point_times(n, st[1, 0] + 1e-6, st[-3, 0] + 1e-6, seed=9)
            kept, moved = a.scan_prepare(raw, t, st, None, 0.3, 30)
            gp, gc = a.scan_download()
            dp, done = b.deskew(raw, t, st)
            kp, kc, _ = b.preprocess(dp, 0.3, 30)
            assert moved == done and kept == len(kp) and np.array_equal(gp, kp) and np.array_equal(gc, kc), states


def test_host_mirror_cloud_preprocessor_process(oracle):
    """CloudPreprocessor::process through the C++ mirror: LiDAR->IMU extrinsic, deskew, scan preparation
    (reference src/CloudPreprocessor.cpp:8-23) against the same chain of oracle calls."""
    from eskf_lio_amd import host, synth
    n = 20_000
    st = synth.make_imu_states(48, seed=31)
    t = synth.make_point_times(n, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=31)
    pts = synth.make_lidar_scan(n, seed=31)
    T_il = synth.se3_to_SE3(np.array([0.05, -0.02, 0.1, 0.01, -0.02, 0.03]))
    gp, gc = host.CloudPreprocessor(0.3, T_il).process(st, pts, t)
    moved, _ = oracle.transform(pts, np.tile(np.eye(3).reshape(9), (n, 1)), T_il)   # Open3D Transform
    desk, done = oracle.deskew(moved, t, st)
    rp, rc, _ = oracle.preprocess(desk, 0.3, 30)
    assert done > 0 and np.array_equal(gp, rp) and np.array_equal(gc, rc)
    with pytest.raises(RuntimeError):
        host.CloudPreprocessor(0.3).process(st[:3], pts, t)         # states end before the sweep does


def test_frame_chain_against_golden_fixture(gpu_ctx):
    """The committed frame fixture (tests/golden/frame_small.npz): deskew, scan preparation and the fused
    resident chain reproduce the stored outputs exactly."""
    g = np.load(os.path.join(GOLDEN, "frame_small.npz"))
    desk, moved = gpu_ctx.deskew(g["points"], g["point_time"], g["states"])
    assert moved == int(g["moved"]) and np.array_equal(desk, g["deskewed"])
    kp, kc, ki = gpu_ctx.preprocess(desk, float(g["voxel_size"]), int(g["knn"]))
    assert np.array_equal(ki, g["kept_index"]) and np.array_equal(kp, g["kept_points"])
    assert np.array_equal(kc, g["kept_covs"])
    kept, moved2 = gpu_ctx.scan_prepare(g["points"], g["point_time"], g["states"], None, float(g["voxel_size"]),
                                        int(g["knn"]))
    rp, rc = gpu_ctx.scan_download()
    assert kept == len(ki) and moved2 == moved and np.array_equal(rp, kp) and np.array_equal(rc, kc)


def test_scans_larger_than_the_grid_park_points_in_lds(gpu_ctx, c1_inputs, monkeypatch):
    """A scan with more points than the persistent launch has worker threads: every thread owns several
    points, the first stays in registers and up to three more are parked in LDS after round 0. Results do
    not depend on the parking and agree with the per-launch loop."""
    from eskf_lio_amd import synth
    vmap = c1_inputs[0]
    pts, covs = synth.make_uniform_scan(400_000, vmap, seed=77)   # ~3.5 points per worker thread
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    gpu_ctx.scan_upload(pts, covs)
    g = synth.default_guess()
    parked = gpu_ctx.align_resident(g, 12, 1e-6, 2.0, chunk_iterations=12)
    assert parked.launches == 1 and parked.iterations == 12
    monkeypatch.setenv("VGICP_NO_STASH", "1")
    plain = gpu_ctx.align_resident(g, 12, 1e-6, 2.0, chunk_iterations=12)
    monkeypatch.delenv("VGICP_NO_STASH")
    assert np.array_equal(parked.pose, plain.pose) and np.array_equal(parked.normal_eq, plain.normal_eq)
    loop = gpu_ctx.align_resident(g, 12, 1e-6, 2.0, chunk_iterations=4, flags=2)   # VGICP_FLAG_NO_PERSISTENT
    assert loop.launches > 1 and np.array_equal(loop.corr_count, parked.corr_count)
    # a different partition of the points into workgroups: same sums up to the order of addition
    assert np.allclose(loop.normal_eq, parked.normal_eq, rtol=1e-11, atol=1e-9)
    assert np.abs(loop.pose - parked.pose).max() < 1e-12
    # The scan above has bitwise symmetric covariances, so the launch read 9 of its 12 planes (the upload's pack kernel
    # found no exception); reading all twelve gives the same bits ...
    monkeypatch.setenv("VGICP_NO_SYM", "1")
    full = gpu_ctx.align_resident(g, 12, 1e-6, 2.0, chunk_iterations=12)
    monkeypatch.delenv("VGICP_NO_SYM")
    assert np.array_equal(full.pose, parked.pose) and np.array_equal(full.normal_eq, parked.normal_eq)
    # ... and ONE covariance that differs in its last bit above the diagonal switches the shortcut off for the scan:
    # the result is then the twelve-plane one for the new data (the loop of launches always reads everything)
    skew = covs.copy()
    skew[123_456, 3] = np.nextafter(skew[123_456, 3], np.inf)        # c01 != c10 now
    gpu_ctx.scan_upload(pts, skew)
    a = gpu_ctx.align_resident(g, 6, 1e-6, 2.0, chunk_iterations=6)
    monkeypatch.setenv("VGICP_NO_SYM", "1")
    b = gpu_ctx.align_resident(g, 6, 1e-6, 2.0, chunk_iterations=6)
    monkeypatch.delenv("VGICP_NO_SYM")
    assert a.launches == 1 and np.array_equal(a.pose, b.pose) and np.array_equal(a.normal_eq, b.normal_eq)
    sym_again = gpu_ctx.align(pts, covs, g, 6, 1e-6, 2.0)             # a later symmetric upload takes the shortcut again
    assert np.array_equal(sym_again.normal_eq, parked.normal_eq[:6])


@pytest.mark.parametrize("seed", range(6))
def test_align_randomised_configurations(gpu_ctx, oracle, seed):
    """Random small problems: voxel sizes 0.1-1.0, maps shifted up to tens of kilometres from the origin (keys
    of both signs, large magnitudes), full and sparse occupancy, ragged sizes, the reference's own thresholds.
    Identical counts and the contractual pose tolerance (and 1e-7 of slack-free agreement) against the oracle."""
    from eskf_lio_amd import synth
    rng = np.random.default_rng(1000 + seed)
    voxel = float(rng.choice([0.1, 0.3, 0.5, 1.0]))
    n_vox = int(rng.integers(500, 20_000))
    n_pts = int(rng.integers(1, 7_000))
    vmap = synth.make_map(n_vox, seed=0x4D00 + seed, voxel_size=voxel)
    shift_cells = rng.integers(-200_000, 200_000, size=3).astype(np.int64) * int(seed % 3 != 0)
    keys = (vmap.keys.astype(np.int64) + shift_cells).astype(np.int32)
    means = vmap.means + shift_cells * voxel
    pts, covs, _ = synth.make_structured_scan(min(n_pts, n_vox), vmap, seed=0x5300 + seed, noise=0.01 * voxel)
    T_shift = np.eye(4)
    T_shift[:3, 3] = shift_cells * voxel
    # the scan lives in the sensor frame; the guess carries the big translation
    guess = T_shift @ synth.se3_to_SE3(rng.normal(size=6) * np.array([0.02, 0.02, 0.02, 0.004, 0.004, 0.004]) * voxel / 0.3)
    om = oracle.OracleMap(voxel, 1)
    om.insert(means, vmap.covs)
    ref = om.align(pts, covs, guess, 100, 1e-6, 0.9999)
    gpu_ctx.map_reset(voxel, 0)
    gpu_ctx.map_upsert(keys, means, vmap.covs)
    got = gpu_ctx.align(pts, covs, guess, 100, 1e-6, 0.9999)
    assert got.iterations == ref.iterations and got.converged == ref.converged
    assert np.array_equal(got.corr_count, ref.corr_count)
    dt, dr = pose_error(got.pose, ref.pose)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD and dt < 1e-7 and dr < 1e-9


def test_preprocess_refuses_points_beyond_the_search_grid(gpu_ctx):
    """The Morton grid of the neighbour search spans +-2^17 voxel sizes; a finite coordinate beyond it would be
    clamped onto a border cell (distinct far voxels merged, search bounds void), so the scan is refused."""
    from eskf_lio_amd import capi, synth
    pts = synth.make_lidar_scan(2_000, seed=21)
    far = pts.copy()
    far[7, 0] = 0.3 * (2 ** 17) + 10.0
    with pytest.raises(capi.VgicpError) as e:
        gpu_ctx.preprocess(far, 0.3, 30)
    assert e.value.code == capi.ERR_BAD_ARGUMENT and "search grid" in str(e.value)
    edge = pts.copy()
    edge[7, 0] = 0.3 * (2 ** 17) - 1.0                              # just inside: accepted
    gp, gc, gi = gpu_ctx.preprocess(edge, 0.3, 30)
    assert len(gi) > 0


def test_preprocess_survives_non_finite_points(gpu_ctx, oracle):
    """NaN / infinite coordinates are garbage in, garbage out — but never a hang or a crash, and the finite part
    of the scan is prepared as if the bad points were not neighbours of anything."""
    from eskf_lio_amd import synth
    pts = synth.make_lidar_scan(4_000, seed=13)
    bad = pts.copy()
    bad[100] = [np.nan, 0.0, 0.0]
    bad[200] = [np.inf, 1.0, 2.0]
    bad[300] = [1.0, -np.inf, np.nan]
    gp, gc, gi = gpu_ctx.preprocess(bad, 0.3, 30)
    assert len(gi) > 0 and np.all(np.diff(gi.astype(np.int64)) > 0)
    finite = np.isfinite(gp).all(axis=1)
    ev = np.linalg.eigvalsh(gc[finite].reshape(-1, 3, 3))
    assert np.allclose(ev, [1e-2, 1.0, 1.0], atol=1e-9)
    # the clean scan without those three points gives the same covariances for the points both runs keep
    keep = np.ones(len(pts), dtype=bool)
    keep[[100, 200, 300]] = False
    rp, rc, ri = oracle.preprocess(pts[keep], 0.3, 30)
    orig = np.flatnonzero(keep)[ri.astype(np.int64)]
    common, ia, ib = np.intersect1d(gi.astype(np.int64), orig, return_indices=True)
    assert len(common) > 0.99 * len(ri) and np.array_equal(gc[ia], rc[ib])
