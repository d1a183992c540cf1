"""Child process of tests/test_multi_device.py: an in-process multi-device context (vgicp_create_multi) with N
sub-contexts on device 0 against a single-device context, in an environment the parent chose (GPU_MAX_HW_QUEUES,
VGICP_SPIN_LIMIT).  Prints one JSON line.  usage: python multi_worker.py N"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eskf_lio_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1])
    vmap = synth.make_map(50_000)
    pts, covs = synth.make_uniform_scan(5_000, vmap)
    big_pts, big_covs = synth.make_uniform_scan(140_000, vmap, seed=77)
    g = synth.default_guess()
    with capi.Context(0) as one:
        one.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        one.map_upsert(vmap.keys, vmap.means, vmap.covs)
        ref = one.align(pts, covs, g, 20, 1e-6, 2.0)
        ref_big = one.align(big_pts, big_covs, g, 6, 1e-6, 2.0)
    out = {"world": n}
    with capi.Context([0] * n) as ctx:
        ctx.map_reset(vmap.voxel_size, vmap.keys.shape[0])
        ctx.map_upsert(vmap.keys, vmap.means, vmap.covs)
        got = [ctx.align(pts, covs, g, 20, 1e-6, 2.0) for _ in range(3)]
        big = ctx.align(big_pts, big_covs, g, 6, 1e-6, 2.0)
        out.update({
            "world_size": got[0].world_size,
            "launches": [r.launches for r in got],
            "counts_equal": bool(all(np.array_equal(r.corr_count, ref.corr_count) for r in got)),
            "pose_delta": float(max(np.abs(r.pose - ref.pose).max() for r in got)),
            "normal_eq_rel": float(max((np.abs(r.normal_eq - ref.normal_eq) / (np.abs(ref.normal_eq) + 1e-300)).max() for r in got)),
            "repeatable": bool(all(np.array_equal(r.pose, got[0].pose) for r in got)),
            "big_counts_equal": bool(np.array_equal(big.corr_count, ref_big.corr_count)),
            "big_pose_delta": float(np.abs(big.pose - ref_big.pose).max()),
            "attempts": ctx.counter(0), "fallbacks": ctx.counter(1),
        })
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
