#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle (deterministic mode).

The reference ships no tests or golden vectors for this path and cannot be built or imported here
(C++ needing Eigen / Open3D / yaml-cpp / ROS 2), so these fixtures are outputs of oracle/ — itself
pinned by analytic known-answer tests and an independent numpy restatement (tests/test_oracle.py).
PARITY UNPINNED by the reference's own tests; see DESIGN.md "Oracle".

Inputs of the C1-sized cases are NOT stored: eskf_lio_amd.synth regenerates them bit-for-bit from
seeds (IEEE-exact arithmetic only). The tiny case stores its inputs too, so it is hermetic.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from eskf_lio_amd import synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def run(vmap, pts, covs, guess, max_it, tsq, cos):
    om = oracle.OracleMap(vmap.voxel_size, 1)
    om.insert(vmap.means, vmap.covs)
    r = om.align(pts, covs, guess, max_it, tsq, cos)
    return dict(pose=r.pose, iterations=np.int32(r.iterations), converged=np.bool_(r.converged),
                corr_count=r.corr_count, JTJ=r.JTJ, JTr=r.JTr, guess=guess,
                params=np.array([max_it, tsq, cos], dtype=np.float64))


def checksum(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def main():
    vmap = synth.make_map(50_000)
    pts, covs = synth.make_uniform_scan(5_000, vmap)
    out = run(vmap, pts, covs, synth.default_guess(), 20, 1e-6, 2.0)
    out["input_checksum"] = checksum(vmap.keys, vmap.means, vmap.covs, pts, covs)
    np.savez(os.path.join(HERE, "c1_uniform.npz"), **out)

    spts, scovs, T_true = synth.make_structured_scan(5_000, vmap)
    out = run(vmap, spts, scovs, np.eye(4), 100, 1e-6, 0.9999)
    out["T_true"] = T_true
    out["input_checksum"] = checksum(spts, scovs)
    np.savez(os.path.join(HERE, "c1_structured.npz"), **out)

    tmap = synth.make_map(400, seed=0x54494E59)
    tpts, tcovs = synth.make_uniform_scan(96, tmap, seed=0x54494E5A)
    out = run(tmap, tpts, tcovs, synth.default_guess(), 8, 1e-6, 2.0)
    out.update(keys=tmap.keys, means=tmap.means, covs=tmap.covs, points=tpts, point_covs=tcovs,
               voxel_size=np.float64(tmap.voxel_size))
    np.savez(os.path.join(HERE, "tiny.npz"), **out)
    # the steps either side of the path: hermetic, inputs stored
    raw = synth.make_lidar_scan(700, seed=0x5052)
    st = synth.make_imu_states(20, seed=0x5052)
    t = synth.make_point_times(700, st[1, 0] + 1e-4, st[-3, 0] + 1e-3, seed=0x5052)
    desk, moved = oracle.deskew(raw, t, st)
    kp, kc, ki = oracle.preprocess(desk, 0.3, 30)
    np.savez(os.path.join(HERE, "frame_small.npz"), points=raw, point_time=t, states=st, deskewed=desk,
             moved=np.int64(moved), kept_points=kp, kept_covs=kc, kept_index=ki, voxel_size=np.float64(0.3),
             knn=np.int32(30))
    # neighbourhoods whose cumulant covariance has rounding-level eigenvalues (exact tilted plane, collinear and
    # repeated points, far from the origin): the reference's U F V^T then returns INDEFINITE matrices
    # (src/CloudPreprocessor.cpp:119-123); inputs stored, with the count of affected points
    rng = np.random.default_rng(3)
    nrm = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    b1 = np.cross(nrm, [1.0, 0.0, 0.0])
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(nrm, b1)
    uv = rng.uniform(-1.5, 1.5, (600, 2))
    plane = np.array([50.0, -70.0, 40.0]) + uv[:, :1] * b1 + uv[:, 1:] * b2
    line = np.array([-30.0, 20.0, 60.0]) + rng.uniform(-2.0, 2.0, (200, 1)) * np.array([2.0, -1.0, 0.5]) / np.sqrt(5.25)
    same = np.repeat(np.array([[80.3, -41.7, 12.9]]), 40, axis=0)
    exact = np.repeat(np.array([[80.25, -41.5, 12.75]]), 40, axis=0)
    deg = np.concatenate([plane, line, same, exact])
    dp, dc, di, bad = oracle.preprocess_ex(deg, 0.3, 30)
    np.savez(os.path.join(HERE, "prep_degenerate.npz"), points=deg, kept_points=dp, kept_covs=dc, kept_index=di,
             indefinite=np.int64(bad), plane_normal=nrm, voxel_size=np.float64(0.3), knn=np.int32(30))
    for f in ("c1_uniform.npz", "c1_structured.npz", "tiny.npz", "frame_small.npz", "prep_degenerate.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
