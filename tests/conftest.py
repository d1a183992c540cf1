"""Shared pytest plumbing.

`-m "not gpu"` : oracle vs known answers / numpy restatement / golden fixtures, host logic, and that
                 the C-ABI library loads and exports every symbol include/vgicp_hip.h declares.
`-m gpu`       : the parity tests proper — HIP path through the C ABI vs the oracle and the fixtures.
Only tests may import oracle/ (it is the checker, never the thing under test on the GPU side).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The in-process multi-device tests put two sub-contexts on the ONE device of the box; their persistent launches wait
# for each other inside the kernel and so need a hardware queue each.  The runtime hands its (default: 4) queues to
# streams as they come, and a long pytest process holds streams of many contexts: whether the two share a queue then
# depends on the test order (seen with `-k frame`: the waits give up, the align falls back — counted, right result).
# Read when the HIP runtime starts, i.e. at the first test that touches the device.  On N real devices every launch
# has its device's queues to itself and none of this applies.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

# Tolerances stated once (BASELINE.json north_star): final pose within 1e-4 m / 1e-4 rad of the CPU
# path, identical correspondence counts.  The HIP path is in fact far tighter; the tests assert the
# tight bound too so a regression shows up long before the contractual one is at risk.
POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
TIGHT_POSE_TOL = 1e-9
NORMAL_EQ_RTOL = 1e-9


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def pose_error(A, B):
    """(translation error in metres, rotation error in radians) between two 4x4 poses."""
    dt = float(np.linalg.norm(A[:3, 3] - B[:3, 3]))
    R = A[:3, :3].T @ B[:3, :3]
    c = max(-1.0, min(1.0, 0.5 * (np.trace(R) - 1.0)))
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(s, c))


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def c1_inputs():
    """BASELINE config C1: 5k-point uniform scan vs 50k-voxel map, regenerated from seeds."""
    from eskf_lio_amd import synth
    vmap = synth.make_map(50_000)
    pts, covs = synth.make_uniform_scan(5_000, vmap)
    return vmap, pts, covs


@pytest.fixture(scope="session")
def c1_oracle_map(oracle, c1_inputs):
    vmap, _, _ = c1_inputs
    om = oracle.OracleMap(vmap.voxel_size, 1)
    om.insert(vmap.means, vmap.covs)
    return om


@pytest.fixture()
def gpu_ctx():
    """A fresh vgicp context on cuda:0. No skip, no fallback: on a GPU box a missing or broken HIP
    module must fail the test."""
    from eskf_lio_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture()
def c1_gpu(gpu_ctx, c1_inputs):
    vmap, pts, covs = c1_inputs
    gpu_ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
    gpu_ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
    return gpu_ctx
