"""First end-to-end GPU parity checks: HIP path through the C ABI vs the CPU oracle on config C1."""
import numpy as np
import pytest

from conftest import POSE_TOL_M, POSE_TOL_RAD, TIGHT_POSE_TOL, NORMAL_EQ_RTOL, pose_error

pytestmark = pytest.mark.gpu


def test_device_is_gfx950(gpu_ctx):
    name, cus, hbm = gpu_ctx.device_info()
    assert name.startswith("gfx950")
    assert cus >= 200 and hbm > 200e9


def test_map_upsert_counts(c1_gpu, c1_inputs):
    vmap, _, _ = c1_inputs
    voxels, slots = c1_gpu.map_size()
    assert voxels == vmap.keys.shape[0]
    assert slots >= 4 * voxels and slots & (slots - 1) == 0
    # same batch again: pure overwrite, size unchanged
    c1_gpu.map_upsert(vmap.keys, vmap.means, vmap.covs)
    assert c1_gpu.map_size()[0] == voxels


def test_accumulate_matches_oracle(c1_gpu, c1_inputs, c1_oracle_map, oracle):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    guess = synth.default_guess()
    JTJ, JTr, cnt = c1_gpu.accumulate(pts, covs, guess)
    tp, tc = oracle.transform(pts, covs, guess)
    oJ, oR, oc = c1_oracle_map.accumulate(tp, tc)
    assert cnt == oc
    scale = np.abs(oJ).max()
    assert np.abs(JTJ - oJ).max() <= NORMAL_EQ_RTOL * scale
    assert np.abs(JTr - oR).max() <= NORMAL_EQ_RTOL * max(np.abs(oR).max(), 1.0)


def test_align_c1_forced_20_iterations(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    _, pts, covs = c1_inputs
    guess = synth.default_guess()
    ref = c1_oracle_map.align(pts, covs, guess, 20, 1e-6, 2.0)
    got = c1_gpu.align(pts, covs, guess, 20, 1e-6, 2.0)
    assert got.iterations == ref.iterations == 20
    assert not got.converged and not ref.converged
    assert (got.corr_count == ref.corr_count).all()          # identical correspondence counts
    dt, dr = pose_error(got.pose, ref.pose)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD            # contractual tolerance
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL      # what the HIP path actually achieves
    scale = np.abs(ref.JTJ).max(axis=(1, 2), keepdims=True)
    assert (np.abs(got.JTJ - ref.JTJ) <= 1e-8 * scale).all()


def test_align_structured_converges(c1_gpu, c1_inputs, c1_oracle_map):
    from eskf_lio_amd import synth
    vmap, _, _ = c1_inputs
    pts, covs, T_true = synth.make_structured_scan(5_000, vmap)
    ref = c1_oracle_map.align(pts, covs, np.eye(4), 100, 1e-6, 0.9999)
    got = c1_gpu.align(pts, covs, np.eye(4), 100, 1e-6, 0.9999)
    assert ref.converged and got.converged
    assert got.iterations == ref.iterations
    assert (got.corr_count == ref.corr_count).all()
    dt, dr = pose_error(got.pose, ref.pose)
    assert dt <= TIGHT_POSE_TOL and dr <= TIGHT_POSE_TOL
    dt, dr = pose_error(got.pose, T_true)
    assert dt < 2e-3 and dr < 1e-3                            # recovers the injected motion
